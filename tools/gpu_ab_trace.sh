#!/bin/bash
# On the GPU box: rocprofv3 kernel stats (serial schedule) of A = ab_base/base.so and B = the tree's library on one scene; per-kernel averages side by side.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
SCENE=${1:-dungeon}
for v in A B; do
  if [ $v = A ]; then export STROLLE_HIP_LIB=$GRAFT_REPO_ROOT/ab_base/base.so; else unset STROLLE_HIP_LIB; fi
  rm -rf gpurun_out/abtrace_$v
  ST_NO_OVERLAP=${SERIAL:-1} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abtrace_$v -- python bench.py --no-cpu-baseline --no-extras --no-profile --scene $SCENE > gpurun_out/abtrace_$v.log 2>&1
done
python - <<'PY'
import csv, glob, re
def load(v):
    f = glob.glob(f"gpurun_out/abtrace_{v}/**/*kernel_stats.csv", recursive=True)[0]
    return {re.sub(r"^.*?st::(fast|exact)::k_", "", r["Name"].split("(")[0]): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f)) if "st::" in r["Name"]}
a, b = load("A"), load("B")
ta = tb = 0.0
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1]):
    if k in b and a[k][1] > 10:
        print(f"{k[:50]:50s} A {a[k][0]:8.1f}  B {b[k][0]:8.1f} us  x{a[k][1]}")
        ta += a[k][0] * a[k][1]; tb += b[k][0] * b[k][1]
print(f"total A {ta / 1e3:.1f} ms  B {tb / 1e3:.1f} ms")
PY
find gpurun_out/abtrace_A gpurun_out/abtrace_B -name "*_kernel_trace.csv" -delete
