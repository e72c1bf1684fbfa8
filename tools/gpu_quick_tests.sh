#!/bin/bash
# On the GPU box: the GPU tests whose names match $1 (pytest -k), verdict and first failures.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q -x -k "$1" > gpurun_out/quick_tests.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/quick_tests.txt | tail -3
grep -E "^E  |^FAILED|Error" gpurun_out/quick_tests.txt | head -30
