#!/bin/bash
# Ablation builds of the fast library (which fast-math substitution spends config 3's tolerance budget?): the tree compiled with one
# -DST_ABL_* each into ab_base/abl_<name>.so (git-ignored; travels with the gpurun snapshot). tools/gpu_abl.sh runs the steady-state reports on them.
#   tools/abl_build.sh            # all variants
cd "$(dirname "$0")/.." || exit 1
mkdir -p ab_base
for v in "sincos:-DST_ABL_SINCOS_POLY" "hemi:-DST_ABL_HEMI_POLY" "mtdiv:-DST_ABL_MT_DIV" "sincos_mtdiv:-DST_ABL_SINCOS_POLY -DST_ABL_MT_DIV"; do
  name=${v%%:*}; flags=${v#*:}
  rm -rf /tmp/abl_$name && mkdir -p /tmp/abl_$name/strolle_amd /tmp/abl_$name/include
  cp -r strolle_amd/csrc /tmp/abl_$name/strolle_amd/ && cp include/*.h /tmp/abl_$name/include/
  (cd /tmp/abl_$name/strolle_amd/csrc && rm -f *.o *.d libstrolle_hip.so && make -s -j8 EXTRA="$flags" >/dev/null 2>&1) || { echo "$name failed to build"; exit 1; }
  cp /tmp/abl_$name/strolle_amd/csrc/libstrolle_hip.so ab_base/abl_$name.so && echo "ab_base/abl_$name.so  ($flags)"
done
