#!/bin/bash
# On the GPU box: rocprofv3 --hip-trace --kernel-trace --memory-copy-trace of the driver's bench command; prints the longest HIP API calls and the
# longest gaps between consecutive kernels (where do the moving-geometry region's ~35 ms go when they appear?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/stall_trace
timeout 900 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/stall_trace -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/stall_trace.json 2> gpurun_out/stall_trace.err
tail -1 gpurun_out/stall_trace.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('geometry regions', d.get('geometry_moving', {}).get('regions_ms_per_step'), 'static', d['ms_per_step'])"
python - <<'PY'
import csv, glob
api = []
for f in glob.glob("gpurun_out/stall_trace/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"])))
api.sort(reverse=True)
t_end = max(a[2] for a in api)
print("longest HIP API calls (ms, function, seconds before the end of the trace):")
for d, f, s in api[:14]: print("  %9.3f %-40s %8.3f" % (d / 1e6, f, (t_end - s) / 1e9))
k = []
for f in glob.glob("gpurun_out/stall_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
k.sort()
gaps = []
busy_end = k[0][1]
for i in range(1, len(k)):
    if k[i][0] > busy_end: gaps.append((k[i][0] - busy_end, k[i - 1][2], k[i][2], k[i][0]))
    busy_end = max(busy_end, k[i][1])
gaps.sort(reverse=True)
print("longest idle gaps of the device (ms, kernel before, kernel after, seconds before the end):")
for g, a, b, s in gaps[:10]: print("  %9.3f  %-50s -> %-50s %8.3f" % (g / 1e6, a, b, (k[-1][1] - s) / 1e9))
longest = sorted(k, key=lambda r: r[0] - r[1])[:6]
print("longest kernels (ms):")
for s, e, n in longest: print("  %9.3f %s  %8.3f s before the end" % ((e - s) / 1e6, n, (k[-1][1] - s) / 1e9))
PY
find gpurun_out/stall_trace -name "*.csv" -delete
