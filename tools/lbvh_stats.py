#!/usr/bin/env python3
"""Readable summary of a rocprofv3 --kernel-trace --stats run of tools/spawn_cost.py: the device builder's kernels (k_lbvh.hip + the hipCUB
sort), then one build launch by launch. Reads gpurun_out/lbvh_prof (or argv[1]); printed by tools/gpu_lbvh_profile.sh into
gpurun_out/r05_lbvh_kernel_stats.txt."""
import csv
import glob
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/lbvh_prof"
print("rocprofv3 --kernel-trace --stats -- python tools/spawn_cost.py --subdivide 2   (208 k triangles; 9 spawn / despawn ticks per refresh mode: k_bvh_* belong to mode 0's host rebuilds)")
for f in sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    if not any("lbvh" in r["Name"] for r in rows):
        continue
    for r in rows:
        n = r["Name"]
        if not ("lbvh" in n or "rocprim" in n or "k_bvh" in n or "k_bake" in n):
            continue
        m = re.search(r"k_\w+", n)
        short = m.group(0) if m else n[:40]
        if "rocprim" in n:
            mm = re.search(r"(radix_sort\w*|merge_sort\w*|onesweep\w*|histogram\w*|scan\w*)", n)
            short = "rocprim " + (mm.group(1) if mm else "kernel")
        print(f"{short:40s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:7.3f} ms  max {float(r['MaxNs'])/1e3:8.1f} us")
print("one build, launch by launch (start since k_lbvh_init, duration):")
for f in sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "lbvh" in r["Kernel_Name"] or "rocprim" in r["Kernel_Name"]]
    if not any("k_lbvh_init" in r["Kernel_Name"] for r in rows):
        continue
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = max(i for i, r in enumerate(rows) if "k_lbvh_init" in r["Kernel_Name"])
    t0 = int(rows[idx]["Start_Timestamp"])
    for r in rows[idx:]:
        m = re.search(r"k_lbvh_\w+", r["Kernel_Name"])
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us  {m.group(0) if m else 'rocprim sort'}")
