#!/bin/bash
# On the GPU box: the driver's exact bench command (--steps 20 --warmup 5) and tools/stall_probe.py beside it.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_cmd_bench.json 2> gpurun_out/driver_cmd_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/driver_cmd_bench.json").read().strip().splitlines()[-1])
print("driver cmd: ms_per_step", d["ms_per_step"], "moving", d.get("ms_per_step_moving"), "geometry", d.get("ms_per_step_geometry_moving"), "present", d.get("ms_per_step_with_present"))
PY
ST_TICK_TIMING=1 timeout 300 python tools/stall_probe.py --warmup 5 --steps 20 > gpurun_out/stall_probe.txt 2> gpurun_out/stall_probe.err
cat gpurun_out/stall_probe.txt
timeout 300 python tools/stall_probe.py --warmup 5 --steps 20 --free-running 2>&1 | tail -4 | tee gpurun_out/stall_probe_free.txt
grep -c "st_tick" gpurun_out/stall_probe.err; head -40 gpurun_out/stall_probe.err
