#!/bin/bash
# On the GPU box: ST_BVH_AUTO's choice of the first tree — the whole GPU suite, which tree it picks on every measured scene and whether that is the faster one
# (tools/tree_choice.py), the spawn cost in the default mode, the bench's config 3 line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite_full.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r06_gpu_suite_full.txt | tail -5 | tee gpurun_out/r06_gpu_suite.txt
grep -E "^FAILED|^ERROR|^E  " gpurun_out/r06_gpu_suite_full.txt | head -20
TREE_CHOICE_EXTRA=1 timeout 1500 python tools/tree_choice.py --rounds 2 2>/dev/null | tee gpurun_out/r06_tree_choice_auto.txt
for sub in 2 0; do timeout 300 python tools/spawn_cost.py --subdivide $sub --refresh 4 2>/dev/null | tail -1 | cut -c1-420; done | tee gpurun_out/r06_spawn_cost_auto_first.txt
timeout 600 python bench.py --scene dungeon134k --mode gi_diffuse --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r06_bench_config3_auto_first.json; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_config3_auto_first.json')); print('config 3 default mode:', d['ms_per_step'], 'ms/frame', d['value'], d['unit'], d['config'].get('bvh_tree'))"
