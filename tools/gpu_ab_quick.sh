#!/bin/bash
# On the GPU box: A = ab_base/base.so, B = the tree's library; dungeon (and Cornell) frame time with the two-stream schedule and serially, 3 rounds.
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do for v in A B; do for ov in 0 1; do for scene in ${SCENES:-dungeon cornell}; do
  if [ $v = A ]; then export STROLLE_HIP_LIB=$GRAFT_REPO_ROOT/ab_base/base.so; else unset STROLLE_HIP_LIB; fi
  ST_NO_OVERLAP=$ov timeout 200 python bench.py --no-cpu-baseline --no-extras --no-profile --scene $scene "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $scene serial=$ov round $round: %.4f ms' % d['ms_per_step'])"
done; done; done; done
