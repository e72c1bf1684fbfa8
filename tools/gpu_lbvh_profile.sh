#!/bin/bash
# rocprofv3 kernel statistics of the device builder (k_lbvh.hip + the hipCUB sort) under tools/spawn_cost.py at 208 k triangles
# -> gpurun_out/r06_lbvh_kernel_stats.txt (tools/lbvh_stats.py; copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -rf gpurun_out/lbvh_prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lbvh_prof -- python tools/spawn_cost.py --subdivide 2 --refresh 3 > gpurun_out/lbvh_prof.log 2>&1
grep "refresh mode" gpurun_out/lbvh_prof.log | cut -c1-400
python tools/lbvh_stats.py gpurun_out/lbvh_prof | tee gpurun_out/r06_lbvh_kernel_stats.txt
