#!/bin/bash
# On the GPU box: the two streams on their own CUs (ST_SIDE_CUS / ST_SIDE_CU_LAYOUT / ST_MAIN_COMPLEMENT, st_render.cpp) against the default — frame time.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-profile --scene $1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$2 $1: %.4f ms' % d['ms_per_step'])"; }
for scene in cornell dungeon; do
  unset ST_SIDE_CUS ST_SIDE_CU_LAYOUT ST_MAIN_COMPLEMENT
  run $scene "default"
  for n in 64 96 128 160 192; do for layout in 0 1; do for comp in 0 1; do
    export ST_SIDE_CUS=$n ST_SIDE_CU_LAYOUT=$layout ST_MAIN_COMPLEMENT=$comp
    run $scene "side=$n layout=$layout main_complement=$comp"
  done; done; done
  unset ST_SIDE_CUS ST_SIDE_CU_LAYOUT ST_MAIN_COMPLEMENT
  run $scene "default(again)"
done 2>&1 | tee gpurun_out/r6_cumask.txt
