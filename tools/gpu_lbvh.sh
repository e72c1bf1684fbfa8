#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fast_tolerance.py -x -q -m gpu -k "built_on_the_device" 2>&1 | tail -15
ST_TICK_TIMING=1 timeout 600 python tools/spawn_cost.py --subdivide 2 2>gpurun_out/spawn_cost_208k.err | tail -3 | tee gpurun_out/spawn_cost_208k.txt
grep "st_tick\|bake" gpurun_out/spawn_cost_208k.err | tail -12
rm -rf gpurun_out/lbvh_prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lbvh_prof -- python tools/spawn_cost.py --subdivide 2 > gpurun_out/lbvh_prof.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/lbvh_prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        n = r["Name"]
        if "lbvh" in n or "rocprim" in n or "radix" in n.lower() or "k_bvh" in n:
            print(f"{n[:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms max {float(r['MaxNs'])/1e3:9.1f} us")
PY
find gpurun_out/lbvh_prof -name "*.csv" -size +2M -delete
