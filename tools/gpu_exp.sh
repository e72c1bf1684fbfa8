#!/bin/bash
# On the GPU box: ST_EXP=<flags> against the default, interleaved, over the given workloads; prints frame time and one kernel's launch time.
#   bash tools/gpu_exp.sh <flags> <kernel slot prefix> <workloads...>
cd "$GRAFT_REPO_ROOT" || exit 1
F=$1; K=$2; shift 2
for round in 1 2 3; do for v in off on; do for w in "$@"; do
  IFS=: read scene mode width height <<< "$w"
  if [ $v = on ]; then export ST_EXP=$F; else unset ST_EXP; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --scene $scene --mode ${mode:-image} --width ${width:-1920} --height ${height:-1080} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('ST_EXP $v $w round $round: %.4f ms | ' % d['ms_per_step'] + ' '.join('%s %.1f' % (n[:24], k[n].get('us_per_launch_kernel_events', k[n]['us_per_launch'])) for n in k if n.startswith('$K')))"
done; done; done
