#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
{ CELL_ASPECT_HEADER=1 ST_LBVH_CELL_ASPECT=0 python tools/cell_aspect.py 2>/dev/null | tail -2
  for a in 0.125 0.25 0.5 1 0 1; do ST_LBVH_CELL_ASPECT=$a python tools/cell_aspect.py 2>/dev/null | tail -1; done; } | tee gpurun_out/r06_lbvh_cell_aspect.txt
