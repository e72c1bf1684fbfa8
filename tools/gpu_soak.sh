#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
{ timeout 900 python tools/soak_builder.py --ticks 1500 --subdivide 0 2>&1 | tail -4; timeout 900 python tools/soak_builder.py --ticks 600 --subdivide 2 --seed 2 2>&1 | tail -4; timeout 900 python tools/soak_builder.py --ticks 1000 --subdivide 0 --seed 3 --observers 2>&1 | tail -4; timeout 900 python tools/soak_builder.py --ticks 400 --subdivide 2 --seed 4 --observers 2>&1 | tail -4; } | tee gpurun_out/r06_soak_builder.txt
