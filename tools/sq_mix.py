#!/usr/bin/env python3
"""Instruction mix per kernel from a rocprofv3 --pmc pass with SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD (mean over the second half of the launches).
Issue-cycle estimates per wave use tools/issue_probe.hip's figures: 2.9 cycles per VALU, 4.7 per SALU instruction and SIMD.
  python tools/sq_mix.py gpurun_out/sq_mix/**/*_counter_collection.csv"""
import csv, sys, re
from collections import defaultdict
def short(name):
    m = re.search(r"st::(?:fast::|exact::)?k_([a-z_0-9]+)(<[^>]*>)?", name)
    if not m: return name[:40]
    return m.group(1) + (m.group(2) or "").replace("unsigned short", "u16").replace("unsigned int", "u32").replace(" ", "")
data = defaultdict(lambda: defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        data[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in data.items():
    m = {c: (sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:]))) for c, v in cs.items()}
    if m.get("SQ_WAVES"): rows.append((k, m))
rows.sort(key=lambda r: -r[1]["SQ_WAVE_CYCLES"])
print(f"{'kernel':40s} {'waves':>6s} {'valu':>6s} {'salu':>6s} {'smem':>5s} {'lds':>5s} {'br':>5s} {'vmrd':>5s} {'life_cyc':>8s} {'valu_cyc':>8s} {'salu_cyc':>8s}  (per wave; x waves/SIMD resident = SIMD time)")
for k, m in rows:
    w = m["SQ_WAVES"]
    g = lambda c: m.get(c, 0) / w
    print(f"{k:40s} {w:6.0f} {g('SQ_INSTS_VALU'):6.0f} {g('SQ_INSTS_SALU'):6.0f} {g('SQ_INSTS_SMEM'):5.0f} {g('SQ_INSTS_LDS'):5.0f} {g('SQ_INSTS_BRANCH'):5.0f} {g('SQ_INSTS_VMEM_RD'):5.0f} {4*g('SQ_WAVE_CYCLES'):8.0f} {2.9*g('SQ_INSTS_VALU'):8.0f} {4.7*g('SQ_INSTS_SALU'):8.0f}")
