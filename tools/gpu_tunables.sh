#!/bin/bash
# On the GPU box: the scheduling tunables once more over both scenes (the wide stream and the packets changed the kernels they were tuned on) — frame time, two rounds.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { env $2 timeout 300 python bench.py --no-cpu-baseline --no-extras --no-profile --scene $1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-44s $1: %.4f ms' % ('$2', d['ms_per_step']))"; }
for round in 1 2; do for scene in cornell dungeon; do
  for v in "ST_X=0" "ST_TILE_MAP=0" "ST_TILE_MAP=2" "ST_TILE_MAP_DENOISE=1" "ST_TILE_MAP_DENOISE=0" "ST_DI_HEAD_ON_MAIN=0" "ST_SIDE_PRIORITY=1" "ST_SIDE_PRIORITY=-1" "ST_NO_FUSE_GI_VALIDATION=1" "ST_NO_PRIMARY_PACKETS=1"; do
    run $scene "$v"
  done
done; done 2>&1 | tee gpurun_out/r6_tunables.txt
