#!/bin/bash
# On the GPU box: the driver's bench command with per-step host times (ST_BENCH_STEP_TIMES=1), then the whole GPU suite.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ST_BENCH_STEP_TIMES=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_cmd_bench3.json 2> gpurun_out/driver_cmd_bench3.err
grep "geometry regions" gpurun_out/driver_cmd_bench3.err | cut -c1-1500
tail -1 gpurun_out/driver_cmd_bench3.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('driver cmd', d['ms_per_step'], 'geometry', d.get('ms_per_step_geometry_moving'), d.get('geometry_moving', {}).get('regions_ms_per_step'))"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
