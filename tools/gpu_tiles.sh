#!/bin/bash
# On the GPU box: bench.py --emulate-tiles for BASELINE config 5 (dungeon 3840x2160 Image) at 8, 4 and 2 tiles, config 4 (Cornell 3840x2160
# Reference, 4 tiles), and the driver's command with the round's new roofline fields. Extra arguments go to the 8-tile run (--row-edges ...).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for n in 8 4 2; do
  timeout 600 python bench.py --emulate-tiles $n --scene dungeon --width 3840 --height 2160 --steps 30 > gpurun_out/tiles_config5_$n.json 2> gpurun_out/tiles_config5_$n.err
  tail -1 gpurun_out/tiles_config5_$n.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config5 tiles', d['tiles'], 'full', d['full_frame_ms'], 'per tile', d['per_tile_ms'], 'max/mean', d['max_over_mean'], 'predicted speed-up', d['predicted_speedup'])"
done
if [ -n "$1" ]; then
  timeout 600 python bench.py --emulate-tiles 8 --scene dungeon --width 3840 --height 2160 --steps 30 "$@" > gpurun_out/tiles_config5_8_weighted.json 2> gpurun_out/tiles_config5_8_weighted.err
  tail -1 gpurun_out/tiles_config5_8_weighted.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config5 weighted', d['row_edges'], 'per tile', d['per_tile_ms'], 'max/mean', d['max_over_mean'], 'predicted speed-up', d['predicted_speedup'])"
fi
timeout 600 python bench.py --emulate-tiles 4 --scene cornell --mode reference --width 3840 --height 2160 --steps 30 > gpurun_out/tiles_config4_4.json 2> gpurun_out/tiles_config4_4.err
tail -1 gpurun_out/tiles_config4_4.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config4 tiles', d['tiles'], 'full', d['full_frame_ms'], 'per tile', d['per_tile_ms'], 'max/mean', d['max_over_mean'], 'predicted speed-up', d['predicted_speedup'])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_cmd_bench2.json 2> gpurun_out/driver_cmd_bench2.err
tail -1 gpurun_out/driver_cmd_bench2.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('driver cmd', d['ms_per_step'], 'geometry', d.get('ms_per_step_geometry_moving'), d.get('geometry_moving', {}).get('regions_ms_per_step'), 'frac', r['frac'], 'frac_counter', r['frac_counter'], 'lit', r['lit_pixel_fraction'], 'frac_lit', r['frac_lit'])"
