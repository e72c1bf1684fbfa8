#!/bin/bash
# Round 6, last call: the bench lines again (each now quotes the counter file of its own build), the GPU suite on the final commit, spawn cost with per-tick timing.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=r06 bash tools/gpu_bench_lines.sh 2>&1 | tail -8
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r06_gpu_suite.txt
ST_TICK_TIMING=1 timeout 600 python tools/spawn_cost.py --subdivide 2 --refresh 4 2> gpurun_out/r06_spawn_ticks.err | tail -1 | tee gpurun_out/r06_spawn_auto.txt
grep -E "device tree|tree: on the device" gpurun_out/r06_spawn_ticks.err | head -20 | tee -a gpurun_out/r06_spawn_auto.txt
