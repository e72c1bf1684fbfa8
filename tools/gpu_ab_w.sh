#!/bin/bash
# On the GPU box: A = ab_base/base.so (a commit's library, built by tools/ab_build_base.sh), B = the tree's library; interleaved rounds over the
# given workloads ("cornell" "dungeon" "dungeon134k:gi_diffuse" "dungeon:image:3840:2160"); frame time + the tracing kernels' launch times.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
W=${@:-cornell dungeon dungeon134k:gi_diffuse}
for round in 1 2 3; do for v in A B; do for w in $W; do
  IFS=: read scene mode width height <<< "$w"
  if [ $v = A ]; then export STROLLE_HIP_LIB=$GRAFT_REPO_ROOT/ab_base/base.so; else unset STROLLE_HIP_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --scene $scene --mode ${mode:-image} --width ${width:-1920} --height ${height:-1080} > gpurun_out/abw_${v}.json 2> gpurun_out/abw_${v}.err
  tail -1 gpurun_out/abw_${v}.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('$v $w round $round: %.4f ms | ' % d['ms_per_step'] + ' '.join('%s %.1f' % (n.split('+')[0][:16], k[n].get('us_per_launch_kernel_events', k[n]['us_per_launch'])) for n in k))"
done; done; done
