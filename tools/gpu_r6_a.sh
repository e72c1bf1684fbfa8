#!/bin/bash
# Round 6, first call: the new tests first (overflow report, device-tree gates, config 4's size), then the whole GPU suite, then the tree against
# round 5's library (ab_base/base.so, tools/ab_build_base.sh) on the three tracked workloads, then the host rebuild with and without the wide
# topology and the default refresh mode's spawn cost.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q -k "overflows or device_built_tree or spawn_and_refits or no_wide_walk or (reference_mode_psnr and 3840) or built_on_the_device or refitted_while" 2>&1 | tail -15 | tee gpurun_out/r6a_new_tests.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r6a_suite.txt
bash tools/gpu_ab_w.sh cornell dungeon dungeon134k:gi_diffuse 2>&1 | tee gpurun_out/r6a_ab.txt
for sub in 2 0; do
  ST_TICK_TIMING=1 timeout 600 python tools/spawn_cost.py --subdivide $sub --refresh 0 4 2> gpurun_out/r6a_spawn_$sub.err | tail -3 | tee -a gpurun_out/r6a_spawn.txt
  grep "bvh build" gpurun_out/r6a_spawn_$sub.err | tail -4 | tee -a gpurun_out/r6a_spawn.txt
  ST_NO_WIDE_BVH=1 ST_TICK_TIMING=1 timeout 600 python tools/spawn_cost.py --subdivide $sub --refresh 0 2> gpurun_out/r6a_spawn_nowide_$sub.err | tail -1 | sed 's/^/ST_NO_WIDE_BVH=1: /' | tee -a gpurun_out/r6a_spawn.txt
  grep "bvh build" gpurun_out/r6a_spawn_nowide_$sub.err | tail -4 | tee -a gpurun_out/r6a_spawn.txt
done
