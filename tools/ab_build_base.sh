#!/bin/bash
# Builds a commit's library (default HEAD) into ab_base/base.so (git-ignored; travels with the gpurun snapshot) for tools/gpu_ab_w.sh.
cd "$(dirname "$0")/.." || exit 1
REV=${1:-HEAD}
rm -rf /tmp/ab_base && mkdir -p /tmp/ab_base ab_base
git archive "$REV" strolle_amd/csrc include | tar -x -C /tmp/ab_base || exit 1
make -C /tmp/ab_base/strolle_amd/csrc -j8 >/dev/null 2>&1 || { echo "base failed to build"; exit 1; }
cp /tmp/ab_base/strolle_amd/csrc/libstrolle_hip.so ab_base/base.so && echo "ab_base/base.so = $(git rev-parse --short $REV)"
