#!/bin/bash
# the device builder's GPU tests; then the all-moving tick cost in ST_BVH_BUILD_DEVICE mode (refit of the device-built tree) and with ST_NO_DEVICE_TREE_REFIT=1 (rebuild)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fast_tolerance.py -x -q -m gpu -k "built_on_the_device or device_built" 2>&1 | tail -25
{ timeout 300 python tools/tick_cost.py --device 0 --subdivide 2 --refit 3 --all --sync; ST_NO_DEVICE_TREE_REFIT=1 timeout 300 python tools/tick_cost.py --device 0 --subdivide 2 --refit 3 --all --sync; timeout 300 python tools/tick_cost.py --device 0 --subdivide 2 --refit 2 --all --sync; } 2>/dev/null | grep "^device=" | tee gpurun_out/tick_cost_mode3_v3.txt
