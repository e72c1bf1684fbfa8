#!/bin/bash
# On the GPU box: SQ instruction counters per wave of the tracing kernels, wide stream vs compact binary stream (ST_NO_WIDE_BVH=1), dungeon 1080p.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="--no-cpu-baseline --no-extras --no-profile --steps 12 --warmup 12 --scene ${1:-dungeon}"
for v in wide compact; do
  if [ $v = compact ]; then export ST_NO_WIDE_BVH=1; else unset ST_NO_WIDE_BVH; fi
  rm -rf gpurun_out/cnt_$v
  ST_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/cnt_$v -- python bench.py $Q > gpurun_out/cnt_$v.log 2>&1
  echo "== $v"
  python - <<PY
import csv, glob, re
from collections import defaultdict
d = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/cnt_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(d.items()):
    if not re.search(r"prim_visibility|sampling|spatial_fused|di_resolving", k): continue
    m = {c: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for c, v in cs.items()}
    w = m["SQ_WAVES"]
    name = re.sub(r"st::fast::k_|void |\(.*", "", k)[:40]
    print(f"{name:40s} valu/wave {m['SQ_INSTS_VALU'] / w:7.0f} salu/wave {m['SQ_INSTS_SALU'] / w:7.0f} lds/wave {m['SQ_INSTS_LDS'] / w:6.0f} vmem_rd/wave {m['SQ_INSTS_VMEM_RD'] / w:6.0f} "
          f"wave_cycles/wave {4 * m['SQ_WAVE_CYCLES'] / w:8.0f} lane_util {m['SQ_THREAD_CYCLES_VALU'] / (64 * m['SQ_ACTIVE_INST_VALU']):.3f}")
PY
  find gpurun_out/cnt_$v -name "*.csv" -delete
done
