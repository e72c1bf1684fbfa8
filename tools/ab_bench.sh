#!/bin/bash
# Same-box A/B of two builds of the kernels (run on the GPU box through gpurun):
#   tools/ab_bench.sh "<extra hipcc flags for B>" [bench.py arguments ...]
# A = the tree as it is; B = the same sources compiled with the extra flags (e.g. -DST_SOME_EXPERIMENT=1). Each build is
# benchmarked twice, interleaved (A B A B), on the default scene and on the dungeon; prints ms per frame and the per-kernel
# table of the last run of each.
cd "$GRAFT_REPO_ROOT" || exit 1
FLAGS="$1"; shift
mkdir -p gpurun_out /tmp/ab
make -C strolle_amd/csrc -j16 >/dev/null 2>&1 || exit 1
cp strolle_amd/csrc/libstrolle_hip.so /tmp/ab/A.so
make -C strolle_amd/csrc clean >/dev/null 2>&1
make -C strolle_amd/csrc -j16 EXTRA="$FLAGS" >/dev/null 2>&1 || { echo "B failed to build"; exit 1; }
cp strolle_amd/csrc/libstrolle_hip.so /tmp/ab/B.so
for round in 1 2; do for v in A B; do for scene in cornell dungeon; do
  cp /tmp/ab/$v.so strolle_amd/csrc/libstrolle_hip.so
  timeout 200 python bench.py --no-cpu-baseline --scene $scene "$@" > gpurun_out/ab_${v}_${scene}.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${v}_${scene}.json").read().strip().splitlines()[-1])
print("$v $scene round $round: %.4f ms" % d["ms_per_step"] + (" (moving %.4f)" % d["ms_per_step_moving"] if "ms_per_step_moving" in d else ""))
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
cp /tmp/ab/A.so strolle_amd/csrc/libstrolle_hip.so
