#!/usr/bin/env python3
"""Turn the rocprofv3 output that a gpurun call left under gpurun_out/ into the small, committed
summaries under profiles/.

  python tools/summarize_profiles.py r01          # tag for the file names

Inputs (any that exist):
  gpurun_out/prof_stats/**/*_kernel_stats.csv        rocprofv3 --kernel-trace --stats
  gpurun_out/prof_fetch/**/*_counter_collection.csv  rocprofv3 --pmc FETCH_SIZE   (separate pass)
  gpurun_out/prof_write/**/*_counter_collection.csv  rocprofv3 --pmc WRITE_SIZE   (separate pass)
Outputs:
  profiles/<tag>_kernel_stats.csv   the st:: kernels' rows of the stats table
  profiles/<tag>_pmc.json           per kernel: mean FETCH_SIZE / WRITE_SIZE (KiB) and HBM bytes per launch
  profiles/pmc_latest.json          copy of the above (bench.py reads it for `roofline.traffic`)
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads (every plane access here is a 16-B-per-lane float4),
WRITE_SIZE is taken as reported (uncalibrated, stated as such).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    """'void st::k_prim_visibility<true, unsigned short>(st::KArgs)' -> 'prim_visibility<true,u16>'"""
    m = re.search(r"st::k_([a-z_0-9]+)(<[^>]*>)?", name)
    if not m:
        return name
    tpl = (m.group(2) or "").replace("unsigned short", "u16").replace("unsigned int", "u32").replace(" ", "")
    return m.group(1) + tpl


def is_ours(name: str) -> bool:
    return "st::k_" in name


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "latest"
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_stats", "**", "*_kernel_stats.csv"), recursive=True))
    if stats:
        rows = list(csv.reader(open(stats[-1])))
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(rows[0])
            for r in rows[1:]:
                if r and is_ours(r[0]):
                    w.writerow([short(r[0])] + r[1:])
    pmc = defaultdict(lambda: defaultdict(list))
    for kind, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
        files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", kind, "**", "*_counter_collection.csv"), recursive=True))
        if not files:
            continue
        for r in csv.DictReader(open(files[-1])):
            if is_ours(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                pmc[short(r["Kernel_Name"])][counter].append(float(r["Counter_Value"]))
    if pmc:
        summary = {}
        for k, c in sorted(pmc.items()):
            # skip the first 12 launches per kernel where possible (warm-up frames have cold history)
            def mean(v):
                v = v[len(v) // 2:] if len(v) > 4 else v
                return sum(v) / len(v) if v else None
            f, w = mean(c.get("FETCH_SIZE", [])), mean(c.get("WRITE_SIZE", []))
            summary[k] = {"fetch_size_kib": f, "write_size_kib": w,
                          "hbm_bytes_per_launch": None if f is None or w is None else round((2.0 * f + w) * 1024.0),
                          "launches_sampled": len(c.get("FETCH_SIZE", []))}
        # aliases under the profiler slot names bench.py uses (st_kernels.h kernel_info)
        alias = {"di_spatial_trace": "spatial_trace<u16>", "gi_spatial_trace": "spatial_trace<u16>",
                 "denoise_wavelet": "denoise_wavelet<false>", "denoise_wavelet+composition": "denoise_wavelet<true>",
                 "prim_visibility+frame_reprojection": "prim_visibility<true,u16>", "di_resolving+denoise_reproject": "di_resolving<true,u16>",
                 "gi_preview": "gi_preview<false>", "gi_preview+gi_resolving+denoise_reproject": "gi_preview<true>",
                 "di_sampling": "di_sampling<u16>", "gi_sampling_a": "gi_sampling_a<u16>", "gi_sampling_b": "gi_sampling_b<u16>"}
        for slot, kernel in alias.items():
            if kernel in summary:
                summary[slot] = summary[kernel]
        for name in (f"{tag}_pmc.json", "pmc_latest.json"):
            json.dump(summary, open(os.path.join(out_dir, name), "w"), indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(out_dir)))


if __name__ == "__main__":
    main()
