#!/bin/bash
# On the GPU box: the whole GPU suite, its verdict line kept (the RCCL banner of the two-rank tests prints after pytest's summary).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite_full.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r06_gpu_suite_full.txt | tail -5 | tee gpurun_out/r06_gpu_suite.txt
