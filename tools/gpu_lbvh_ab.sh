#!/bin/bash
# On the GPU box: the device builder of this tree against ab_base/base.so (tools/ab_build_base.sh <commit>):
# the same trees? (tools/device_tree_hash.py), the builder's GPU tests, the build launch by launch, the spawn cost with either library.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
{ echo "base:"; STROLLE_HIP_LIB=$PWD/ab_base/base.so timeout 300 python tools/device_tree_hash.py 2>/dev/null | grep dungeon
  echo "this tree:"; timeout 300 python tools/device_tree_hash.py 2>/dev/null | grep dungeon; } | tee gpurun_out/r06_lbvh_sort_hashes.txt
timeout 1500 python -m pytest tests/test_gpu_fast_tolerance.py tests/test_c_abi.py tests/test_gpu_fast_steady_state.py -q -m gpu -k "device or spawn or default_mode or two_sorts" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
bash tools/gpu_lbvh_profile.sh > gpurun_out/r06_lbvh_sort_profile.txt 2>&1; cp gpurun_out/r06_lbvh_kernel_stats.txt gpurun_out/r06_lbvh_kernel_stats_one_launch.txt; tail -48 gpurun_out/r06_lbvh_sort_profile.txt
for lib in base this base this; do
  for sub in 2 0; do
    if [ $lib = base ]; then export STROLLE_HIP_LIB=$PWD/ab_base/base.so; else unset STROLLE_HIP_LIB; fi
    echo "$lib: $(timeout 300 python tools/spawn_cost.py --subdivide $sub --refresh 4 2>/dev/null | tail -1 | cut -c1-400)"
  done
done | tee gpurun_out/r06_lbvh_sort_spawn_cost.txt
