#!/bin/bash
# the device builder's GPU tests, then its kernel statistics (tools/gpu_lbvh_profile.sh) and the move-only tick cost
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fast_tolerance.py tests/test_c_abi.py -x -q -m gpu -k "built_on_the_device or device" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
bash tools/gpu_lbvh_profile.sh
timeout 300 python tools/spawn_cost.py --subdivide 2 2>/dev/null | tail -2 | cut -c1-420 | tee gpurun_out/spawn_cost_208k_v2.txt
timeout 300 python tools/spawn_cost.py --subdivide 0 2>/dev/null | tail -1 | cut -c1-420 | tee -a gpurun_out/spawn_cost_208k_v2.txt
