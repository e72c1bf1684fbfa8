#!/bin/bash
# On the GPU box: the wide stream (StTuning::wide_bvh) against the compact binary stream (ST_NO_WIDE_BVH=1), same box, interleaved rounds;
# first the wide stream's own GPU test. Arguments: the workloads to compare ("dungeon" "dungeon134k:gi_diffuse" "dungeon:image:3840:2160").
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fast_tolerance.py -x -q -m gpu -k "wide_stream or axis_parallel" 2>&1 | tail -5
W=${@:-dungeon dungeon134k:gi_diffuse}
for round in 1 2 3; do for v in wide compact; do for w in $W; do
  IFS=: read scene mode width height <<< "$w"
  if [ $v = compact ]; then export ST_NO_WIDE_BVH=1; else unset ST_NO_WIDE_BVH; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --scene $scene --mode ${mode:-image} --width ${width:-1920} --height ${height:-1080} > gpurun_out/wide_ab_${v}_${scene}_${mode:-image}_${width:-1920}.json 2> gpurun_out/wide_ab_${v}.err
  tail -1 gpurun_out/wide_ab_${v}_${scene}_${mode:-image}_${width:-1920}.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
pick = ['prim_visibility+frame_reprojection', 'di_sampling+di_temporal', 'gi_sampling_a+b', 'gi_spatial_pick+trace+sample', 'di_resolving+denoise_reproject', 'di_spatial_pick+trace+sample']
print('$v $w round $round: %.4f ms | ' % d['ms_per_step'] + ' '.join('%s %.1f' % (n.split('+')[0], k[n].get('us_per_launch_kernel_events', k[n]['us_per_launch'])) for n in pick if n in k))"
done; done; done
unset ST_NO_WIDE_BVH
