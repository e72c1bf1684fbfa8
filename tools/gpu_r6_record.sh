#!/bin/bash
# On the GPU box: the round's closing record of one build — the GPU suite's verdict, what the build renders as hashes, the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
python -c "from strolle_amd import api; print('library built from', api.library_build_commit())" | tee gpurun_out/r06_record_commit.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite_full.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r06_gpu_suite_full.txt | tail -5 | tee gpurun_out/r06_gpu_suite.txt
timeout 600 python tools/frame_hash.py 2>/dev/null | grep -E "^(cornell|dungeon)" | tee gpurun_out/r06_frame_hashes.txt
timeout 600 python bench.py 2>/dev/null | tail -1 | tee gpurun_out/r06_bench_final.json
for sub in 2 0; do timeout 300 python tools/spawn_cost.py --subdivide $sub --refresh 4 2>/dev/null | tail -1 | cut -c1-420; done | tee gpurun_out/r06_spawn_cost_auto_first.txt
