#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
if [ -n "$TREE_CHOICE_EXTRA" ]; then timeout 1500 python tools/tree_choice.py --rounds 2 2>/dev/null | tee gpurun_out/r06_tree_choice_extra.txt; else timeout 1500 python tools/tree_choice.py --rounds 3 2>/dev/null | tee gpurun_out/r06_tree_choice.txt; fi
