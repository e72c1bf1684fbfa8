#!/bin/bash
# Same-box A/B of the working tree against a commit (default HEAD): builds that commit's library here into ab_base/base.so (git-ignored, travels
# with the snapshot), then runs tools/gpu_ab_libs.sh on the GPU box — A = the commit, B = the tree as it is — interleaved, both scenes.
#   tools/ab_head.sh [commit] [bench.py arguments ...]
cd "$(dirname "$0")/.." || exit 1
REV=${1:-HEAD}; shift
rm -rf /tmp/ab_base && mkdir -p /tmp/ab_base ab_base
git archive "$REV" strolle_amd/csrc include | tar -x -C /tmp/ab_base || exit 1
make -C /tmp/ab_base/strolle_amd/csrc -j32 >/dev/null 2>&1 || { echo "base failed to build"; exit 1; }
cp /tmp/ab_base/strolle_amd/csrc/libstrolle_hip.so ab_base/base.so
make -C strolle_amd/csrc -j32 >/dev/null 2>&1 || { echo "tree failed to build"; exit 1; }
exec tools/gpurun_batch.sh 900 "bash tools/gpu_ab_libs.sh $*"
