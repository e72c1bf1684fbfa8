#!/bin/bash
# On the GPU box: A = ab_base/base.so (STROLLE_HIP_LIB), B = the tree's library; A B A B on Cornell and the dungeon; ms per frame and the
# per-kernel table of each build's last run.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for round in 1 2; do for v in A B; do for scene in cornell dungeon; do
  if [ $v = A ]; then export STROLLE_HIP_LIB=$GRAFT_REPO_ROOT/ab_base/base.so; else unset STROLLE_HIP_LIB; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --scene $scene "$@" > gpurun_out/ab_${v}_${scene}.json 2>gpurun_out/ab_${v}_${scene}.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${v}_${scene}.json").read().strip().splitlines()[-1])
print("$v $scene round $round: %.4f ms" % d["ms_per_step"])
PY
done; done; done
python - <<PY
import json
for scene in ("cornell", "dungeon"):
    a = json.loads(open(f"gpurun_out/ab_A_{scene}.json").read().strip().splitlines()[-1])["kernels"]
    b = json.loads(open(f"gpurun_out/ab_B_{scene}.json").read().strip().splitlines()[-1])["kernels"]
    print(scene)
    for k in a: print("  %-45s A %7.1f  B %7.1f us" % (k, a[k]["us_per_launch"], b.get(k, {}).get("us_per_launch", float("nan"))))
PY
