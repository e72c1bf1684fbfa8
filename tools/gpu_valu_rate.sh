#!/bin/bash
# issue rate of the traversal loops' VALU instructions on the GPU box (tools/ubench/valu_rate.hip) -> gpurun_out/r06_valu_rate.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate || exit 1
timeout 45 /tmp/valu_rate | tee gpurun_out/r06_valu_rate.txt
