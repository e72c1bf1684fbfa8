#!/bin/bash
# Round 6, mid-way call: the whole GPU suite on the tree, the lane-refill pool against the plain launch, the ablation builds, the spawn cost again.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee gpurun_out/r6b_suite.txt
bash tools/gpu_pool.sh 2>&1 | tail -40
bash tools/gpu_abl.sh 2>&1 | tail -60
ST_TICK_TIMING=1 timeout 600 python tools/spawn_cost.py --subdivide 2 --refresh 4 3 2> gpurun_out/r6b_spawn_2.err | tail -2 | tee gpurun_out/r6b_spawn.txt
grep "device tree" gpurun_out/r6b_spawn_2.err | head -6 | tee -a gpurun_out/r6b_spawn.txt
timeout 600 python tools/spawn_cost.py --subdivide 0 --refresh 4 2>/dev/null | tail -1 | tee -a gpurun_out/r6b_spawn.txt
