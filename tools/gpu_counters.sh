#!/bin/bash
# On the GPU box: SQ instruction + wait counters per wave of the tracing kernels for one workload (default build).  bash tools/gpu_counters.sh [scene] [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
SC=${1:-dungeon}; shift
Q="--no-cpu-baseline --no-extras --no-profile --steps 12 --warmup 12 --scene $SC $@"
for pass in a b; do
  rm -rf gpurun_out/cnt_$pass
  if [ $pass = a ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; else C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"; fi
  ST_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/cnt_$pass -- python bench.py $Q > gpurun_out/cnt_$pass.log 2>&1
done
python - <<'PY'
import csv, glob, re
from collections import defaultdict
d = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/cnt_[ab]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(d.items()):
    if not re.search(r"prim_visibility|sampling|spatial_fused|di_resolving", k): continue
    m = {c: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for c, v in cs.items()}
    w = m["SQ_WAVES"]; wc = m["SQ_WAVE_CYCLES"]
    name = re.sub(r"st::fast::k_|void |\(.*", "", k)[:38]
    g = lambda c: m.get(c, float("nan"))
    print(f"{name:38s} valu {g('SQ_INSTS_VALU') / w:6.0f} salu {g('SQ_INSTS_SALU') / w:6.0f} smem {g('SQ_INSTS_SMEM') / w:5.0f} vmem_rd {g('SQ_INSTS_VMEM_RD') / w:5.0f} lds {g('SQ_INSTS_LDS') / w:4.0f} | "
          f"cycles/wave {4 * wc / w:8.0f} lane_util {g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_ACTIVE_INST_VALU')):.2f} wait_any {g('SQ_WAIT_ANY') / wc:.2f} wait_inst {g('SQ_WAIT_INST_ANY') / wc:.2f} active_valu {g('SQ_ACTIVE_INST_VALU') / wc:.2f} active_sca {g('SQ_ACTIVE_INST_SCA') / wc:.2f}")
PY
find gpurun_out/cnt_a gpurun_out/cnt_b -name "*.csv" -delete
