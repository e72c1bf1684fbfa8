#!/bin/bash
# VOP3-encoded selects (strolle_amd/csrc/vop3_select.py) against the build before them (ab_base/base.so): the same bits (tools/frame_hash.py), then the
# headline and the dungeon frame, interleaved, two rounds -> gpurun_out/r05_vop3_ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
echo "== frame hashes: build before (ab_base/base.so)"; STROLLE_HIP_LIB=$PWD/ab_base/base.so timeout 120 python tools/frame_hash.py 2>/dev/null
echo "== frame hashes: VOP3 selects"; timeout 120 python tools/frame_hash.py 2>/dev/null
for round in 1 2; do
  for scene in cornell dungeon; do
    for lib in base vop3; do
      if [ $lib = base ]; then export STROLLE_HIP_LIB=$PWD/ab_base/base.so; else unset STROLLE_HIP_LIB; fi
      timeout 120 python bench.py --scene $scene --steps 60 --warmup 20 --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib $scene round $round: ms_per_step', d['ms_per_step'])"
    done
  done
done
} | tee gpurun_out/r05_vop3_ab.txt
