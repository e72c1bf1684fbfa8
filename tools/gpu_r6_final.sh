#!/bin/bash
# Round 6, final evidence in one call: the whole GPU suite, the driver's command, tile emulation (config 5 at 8 / 4 / 2 tiles, config 4), spawn cost.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build(); g.smoke(); print("smoke ok")' 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r06_gpu_suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_driver_cmd.json 2> gpurun_out/r06_driver_cmd.err
tail -1 gpurun_out/r06_driver_cmd.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('driver cmd', d['ms_per_step'], d['value'], 'moving', d.get('ms_per_step_moving'), 'geometry', d.get('ms_per_step_geometry_moving'), d.get('geometry_moving', {}).get('regions_ms_per_step'), 'present', d.get('ms_per_step_with_present'), 'frac', r['frac'], r['frac_counter'], r['lit_pixel_fraction'], r['frac_lit'])"
for n in 8 4 2; do
  timeout 900 python bench.py --emulate-tiles $n --scene dungeon --width 3840 --height 2160 --steps 30 --balance-rounds $([ $n = 8 ] && echo 3 || echo 1) > gpurun_out/tiles_config5_$n.json 2> gpurun_out/tiles_config5_$n.err
  tail -1 gpurun_out/tiles_config5_$n.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config5 tiles', d['tiles'], 'full', d['full_frame_ms'], 'equal', d['equal_split']['per_tile_ms'], d['equal_split']['max_over_mean'], d['equal_split']['predicted_speedup'], 'balanced', d['balanced']['per_tile_ms'], d['balanced']['max_over_mean'], d['balanced']['predicted_speedup'])"
done
timeout 600 python bench.py --emulate-tiles 4 --scene cornell --mode reference --width 3840 --height 2160 --steps 30 > gpurun_out/tiles_config4_4.json 2> gpurun_out/tiles_config4_4.err
tail -1 gpurun_out/tiles_config4_4.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config4 tiles', d['tiles'], 'full', d['full_frame_ms'], d['equal_split']['per_tile_ms'], d['equal_split']['max_over_mean'], d['equal_split']['predicted_speedup'])"
(timeout 600 python tools/spawn_cost.py --subdivide 2 --refresh 0 4 3; timeout 600 python tools/spawn_cost.py --subdivide 0 --refresh 0 4 3) 2>/dev/null | grep "refresh mode" | tee gpurun_out/r06_spawn_cost.txt
bash tools/gpu_lbvh_profile.sh 2>&1 | tail -45
