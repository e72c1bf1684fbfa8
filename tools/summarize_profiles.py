#!/usr/bin/env python3
"""Turn the rocprofv3 output that `tools/gpu_profile_round.sh` left under gpurun_out/ into the small, committed
summaries under profiles/.

  python tools/summarize_profiles.py r01                                  # tag for the file names; inputs straight under gpurun_out/
  python tools/summarize_profiles.py r04 --key dungeon_1920x1080_image   # one WORKLOAD of tools/gpu_profile_workload.sh:
      inputs gpurun_out/prof_<key>/{stats,stats_serial,fetch,write,sq,lane}/ + bench.json; outputs profiles/<tag>_<key>_*.csv / .json
      and profiles/pmc/<key>.json — the counter summary bench.py looks up BY WORKLOAD, stamped with the commit and the launch structure
      (the profiler slots of a frame) of the bench line taken in the same gpurun call; bench.py refuses a summary whose stamp names
      other launches and prints null for a workload that has none.

Inputs (any that exist):
  gpurun_out/prof_stats/**/*_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the default bench command
  gpurun_out/prof_stats_serial/**/*_kernel_stats.csv   the same with ST_NO_OVERLAP=1 --no-profile (one kernel at a time)
  gpurun_out/prof_fetch/**/*_counter_collection.csv    rocprofv3 --pmc FETCH_SIZE   (separate pass)
  gpurun_out/prof_write/**/*_counter_collection.csv    rocprofv3 --pmc WRITE_SIZE   (separate pass)
  gpurun_out/prof_sq/**/*_counter_collection.csv       rocprofv3 --pmc SQ_* (8 SQ slots, separate pass)
  gpurun_out/bench_default.json                        the bench line of the same build
Outputs:
  profiles/<tag>_kernel_stats.csv          st:: rows of the default command's stats table (two-stream region + serial region)
  profiles/<tag>_kernel_stats_serial.csv   st:: rows of the serial run + the bench slot aggregates (e.g. all 5 wavelet launches)
  profiles/<tag>_pmc.json                  per kernel: mean FETCH_SIZE / WRITE_SIZE (KiB) and HBM bytes per launch
  profiles/<tag>_sq.csv                    per kernel: VALU instructions per wave, VALU-busy share, wait shares
  profiles/pmc_latest.json                 copy of <tag>_pmc.json (bench.py reads it for `roofline.traffic`)
  profiles/<tag>_bench.json                copy of the bench line
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads (every plane access here is a 16-B-per-lane float4),
WRITE_SIZE is taken as reported (uncalibrated, stated as such).
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# bench.py profiler slot (st_kernels.h kernel_info) -> the kernel symbols it launches
SLOT_SYMBOLS = {
    "denoise_wavelet": ["denoise_wavelet_lds<1>", "denoise_wavelet_lds<2>", "denoise_wavelet_lds<4>", "denoise_wavelet_far<false>"],
    "denoise_wavelet+composition": ["denoise_wavelet_far<true>"],
    # tracing kernels: <LDS scene, [REPROJECT,] stack entry>; whichever instance the scene selected
    "prim_visibility+frame_reprojection": ["prim_visibility<true,true,u16>", "prim_visibility<false,true,u16>", "prim_visibility<false,true,u32>"],
    "gi_sampling_a+b": ["gi_sampling_ab<true,u16>", "gi_sampling_ab<false,u16>", "gi_sampling_ab<false,u32>"],
    "di_sampling+di_temporal": ["di_sampling_temporal<true,u16>", "di_sampling_temporal<false,u16>", "di_sampling_temporal<false,u32>"],
    "di_spatial_pick+trace+sample": ["di_spatial_fused<true,u16>", "di_spatial_fused<false,u16>", "di_spatial_fused<false,u32>"],
    "di_resolving+denoise_reproject": ["di_resolving<true,true,u16>", "di_resolving<false,true,u16>", "di_resolving<false,true,u32>"],
    "gi_spatial_pick+trace+sample": ["gi_spatial_fused<true,u16>", "gi_spatial_fused<false,u16>", "gi_spatial_fused<false,u32>"],
    "gi_sampling_a": ["gi_sampling_a<true,u16>", "gi_sampling_a<false,u16>", "gi_sampling_a<false,u32>"],
    "gi_sampling_b": ["gi_sampling_b<true,u16>", "gi_sampling_b<false,u16>", "gi_sampling_b<false,u32>"],
    "gi_reprojection+gi_temporal": ["gi_temporal<true>"], "gi_temporal": ["gi_temporal<false>"],
    "gi_preview": ["gi_preview<false>"], "gi_preview+gi_resolving+denoise_reproject": ["gi_preview<true>"],
    "denoise_wavelet x2 (strides 1+2)": ["denoise_wavelet_12"],
    "gi_preview x2+gi_resolving+denoise_reproject": ["gi_preview_both"], "gi_preview 2nd pass (pixels that resample)": ["gi_preview<true>"],
}


def short(name: str) -> str:
    """'void st::k_prim_visibility<true, unsigned short>(st::KArgs)' -> 'prim_visibility<true,u16>'"""
    m = re.search(r"st::(?:fast::|exact::)?k_([a-z_0-9]+)(<[^>]*>)?", name)
    if not m:
        return name
    tpl = (m.group(2) or "").replace("unsigned short", "u16").replace("unsigned int", "u32").replace(" ", "")
    return m.group(1) + tpl


def is_ours(name: str) -> bool:
    return re.search(r"st::(?:fast::|exact::)?k_", name) is not None


BASE = os.path.join(ROOT, "gpurun_out")
PASS_DIRS = {"prof_stats": "prof_stats", "prof_stats_serial": "prof_stats_serial", "prof_fetch": "prof_fetch", "prof_write": "prof_write", "prof_sq": "prof_sq", "prof_lane": "prof_lane"}


def latest(pattern):
    first, rest = pattern.split("/", 1)
    files = sorted(glob.glob(os.path.join(BASE, PASS_DIRS.get(first, first), rest), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def stats_rows(path):
    rows = list(csv.DictReader(open(path)))
    return [dict(r, Name=short(r["Name"])) for r in rows if is_ours(r["Name"])]


def main():
    global BASE
    argv = [a for a in sys.argv[1:]]
    key = None
    if "--key" in argv:
        i = argv.index("--key"); key = argv[i + 1]; del argv[i:i + 2]
    tag = argv[0] if argv else "latest"
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    bench_path = os.path.join(ROOT, "gpurun_out", "bench_default.json")
    if key:
        BASE = os.path.join(ROOT, "gpurun_out", f"prof_{key}")
        for k in list(PASS_DIRS): PASS_DIRS[k] = k[len("prof_"):]
        bench_path = os.path.join(BASE, "bench.json")
        tag = f"{tag}_{key}"
    bench_line = None
    try:
        bench_line = json.loads(open(bench_path).read().strip().splitlines()[-1])
    except Exception:
        pass

    f = latest("prof_stats/**/*_kernel_stats.csv")
    if f:
        rows = stats_rows(f)
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    f = latest("prof_stats_serial/**/*_kernel_stats.csv")
    if f:
        rows = stats_rows(f)
        by = {r["Name"]: r for r in rows}
        with open(os.path.join(out_dir, f"{tag}_kernel_stats_serial.csv"), "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
            # aggregates under the profiler slot names bench.py reports (a slot may launch several symbols)
            for slot, syms in SLOT_SYMBOLS.items():
                have = [by[s] for s in syms if s in by]
                if len(have) < 1 or (len(have) == 1 and len(syms) == 1):
                    continue
                calls = sum(int(r["Calls"]) for r in have)
                total = sum(float(r["TotalDurationNs"]) for r in have)
                w.writerow({"Name": f"[slot] {slot}", "Calls": calls, "TotalDurationNs": int(total), "AverageNs": f"{total / calls:.1f}",
                            "Percentage": f"{sum(float(r['Percentage']) for r in have):.2f}",
                            "MinNs": min(int(r["MinNs"]) for r in have), "MaxNs": max(int(r["MaxNs"]) for r in have), "StdDev": ""})

    pmc = defaultdict(lambda: defaultdict(list))
    for kind, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
        f = latest(f"{kind}/**/*_counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if is_ours(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                pmc[short(r["Kernel_Name"])][counter].append(float(r["Counter_Value"]))
    if pmc:
        def tail(v):  # second half of the launches: warm-up frames have cold history
            return v[len(v) // 2:] if len(v) > 4 else v
        summary = {}
        for k, c in sorted(pmc.items()):
            fv, wv = tail(c.get("FETCH_SIZE", [])), tail(c.get("WRITE_SIZE", []))
            fm = sum(fv) / len(fv) if fv else None
            wm = sum(wv) / len(wv) if wv else None
            summary[k] = {"fetch_size_kib": fm, "write_size_kib": wm,
                          "hbm_bytes_per_launch": None if fm is None or wm is None else round((2.0 * fm + wm) * 1024.0),
                          "launches_sampled": len(fv)}
        for slot, syms in SLOT_SYMBOLS.items():
            have = [summary[s] for s in syms if s in summary and summary[s]["hbm_bytes_per_launch"] is not None]
            if not have:
                continue
            n = sum(h["launches_sampled"] for h in have)
            summary[slot] = {"fetch_size_kib": sum(h["fetch_size_kib"] * h["launches_sampled"] for h in have) / n,
                             "write_size_kib": sum(h["write_size_kib"] * h["launches_sampled"] for h in have) / n,
                             "hbm_bytes_per_launch": round(sum(h["hbm_bytes_per_launch"] * h["launches_sampled"] for h in have) / n),
                             "launches_sampled": n, "symbols": syms}
        # the stamp: which build and which launch structure these bytes describe (bench.py refuses a mismatch)
        summary["_stamp"] = {"commit": (bench_line or {}).get("build_stamp"), "workload": key or "cornell_1920x1080_image",
                             "launch_slots": sorted((bench_line or {}).get("kernels", {}).keys()) or None,
                             "ms_per_step_of_the_same_call": (bench_line or {}).get("ms_per_step")}
        json.dump(summary, open(os.path.join(out_dir, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
        os.makedirs(os.path.join(out_dir, "pmc"), exist_ok=True)
        json.dump(summary, open(os.path.join(out_dir, "pmc", (key or "cornell_1920x1080_image") + ".json"), "w"), indent=1, sort_keys=True)

    lane = {}
    f = latest("prof_lane/**/*_counter_collection.csv")
    if f:
        ln = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            if is_ours(r["Kernel_Name"]):
                ln[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in ln.items():
            m = {n: (sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:]))) for n, v in c.items()}
            if m.get("SQ_ACTIVE_INST_VALU"):
                w = max(m.get("SQ_WAVES", 0.0), 1.0)
                lane[k] = (round(m.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * m["SQ_ACTIVE_INST_VALU"]), 3), round(m.get("SQ_INSTS_SALU", 0.0) / w), round(m.get("SQ_INSTS_LDS", 0.0) / w), round(m.get("SQ_INSTS_VMEM_RD", 0.0) / w))
    f = latest("prof_sq/**/*_counter_collection.csv")
    if f:
        sq = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            if is_ours(r["Kernel_Name"]):
                sq[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(out_dir, f"{tag}_sq.csv"), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(["kernel", "waves", "valu_insts_per_wave", "valu_cycles_share_of_wave_life", "wait_any_share", "wait_inst_share",
                        "active_any_share", "valu_floor_us_at_2.4GHz", "lane_utilisation_valu", "salu_insts_per_wave", "lds_insts_per_wave", "vmem_rd_insts_per_wave"])
            rows = []
            for k, c in sq.items():
                m = {n: (sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:]))) for n, v in c.items()}
                if not m.get("SQ_WAVES"):
                    continue
                wc = m["SQ_WAVE_CYCLES"]
                # a plain wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md): a LOWER bound, and a loose one —
                # tools/ubench/valu_rate.hip measures 2.3 cycles only for fma / mul / add / mov / and / or / xor and 4.1 for min / max / cmp / cndmask /
                # cvt / fma_mix / alignbit / perm / shifts / med3, which is what a traversal step is made of (DESIGN.md "What a VALU instruction costs"); 1024 SIMDs
                floor_us = m["SQ_INSTS_VALU"] * 2.0 / 1024.0 / 2400.0
                rows.append([k, int(m["SQ_WAVES"]), round(m["SQ_INSTS_VALU"] / m["SQ_WAVES"]), round(m["SQ_ACTIVE_INST_VALU"] / wc, 3),
                             round(m["SQ_WAIT_ANY"] / wc, 3), round(m["SQ_WAIT_INST_ANY"] / wc, 3), round(m["SQ_ACTIVE_INST_ANY"] / wc, 3), round(floor_us, 1)] + list(lane.get(k, ("", "", "", ""))))
            rows.sort(key=lambda r: -r[7])
            w.writerows(rows)

    if os.path.exists(bench_path):
        shutil.copy(bench_path, os.path.join(out_dir, f"{tag}_bench.json"))
    print("wrote", sorted(f for f in os.listdir(out_dir) if f.startswith(tag + "_")))


if __name__ == "__main__":
    main()
