#!/bin/bash
# round-5 mid-way check: the new GPU tests, A/B against ab_base/base.so, tile emulation with rebalancing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fast_tolerance.py -x -q -m gpu -k "wide or axis_parallel" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_two_rank or tiles_gathered or bvh_refit_mode" 2>&1 | tail -5
bash tools/gpu_ab_w.sh dungeon134k:gi_diffuse dungeon 2>&1 | cut -c1-330
timeout 900 python bench.py --emulate-tiles 8 --scene dungeon --width 3840 --height 2160 --steps 30 --balance-rounds 3 > gpurun_out/tiles_config5_8.json 2> gpurun_out/tiles_config5_8.err
tail -1 gpurun_out/tiles_config5_8.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('full', d['full_frame_ms'])
for r in d['rounds']: print(r['per_tile_ms'], 'max/mean', r['max_over_mean'], 'speed-up', r['predicted_speedup'], r['grid']['row_edges'], r['grid']['col_edges'])"
