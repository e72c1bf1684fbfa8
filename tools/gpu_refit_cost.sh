cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q -k "refit" 2>&1 | tail -5
{ for hb in "" "--host-bake"; do timeout 300 python tools/tick_cost.py --device 0 --subdivide 2 --refit 2 --all $hb; timeout 300 python tools/tick_cost.py --device 0 --subdivide 2 --refit 2 $hb; done
  for hb in "" "--host-bake"; do timeout 300 python tools/animated_cost.py --subdivide 2 --refit 2 --all $hb; done
  timeout 300 python tools/animated_cost.py --subdivide 2 --refit 2; } 2>&1 | grep -v "warning: the BVH" | tee gpurun_out/r04_refit_cost.txt
