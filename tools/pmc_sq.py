#!/usr/bin/env python3
"""Per-kernel SQ counter summary from rocprofv3 --pmc counter_collection.csv files (mean over the second half of the launches).
  python tools/pmc_sq.py gpurun_out/pmc_sq1/**/..._counter_collection.csv [more.csv ...]"""
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
names = sorted({c for k in data.values() for c in k})
rows = []
for k, cs in data.items():
    m = {c: (sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:]))) for c, v in cs.items()}
    rows.append((k, m))
rows.sort(key=lambda r: -r[1].get("SQ_WAVE_CYCLES", r[1].get("SQ_INSTS_VMEM_RD", 0)))
for k, m in rows:
    if "st::" in k or len(k) > 39: pass
    line = f"{k:34s}"
    w = m.get("SQ_WAVES")
    if w:
        wc = m["SQ_WAVE_CYCLES"]
        line += f" waves {w:8.0f} valu/wave {m['SQ_INSTS_VALU']/w:7.0f} wave_cyc/wave {4*wc/w:8.0f} active_valu {m['SQ_ACTIVE_INST_VALU']/wc:5.2f} wait_any {m['SQ_WAIT_ANY']/wc:5.2f} wait_inst {m['SQ_WAIT_INST_ANY']/wc:5.2f} active_any {m['SQ_ACTIVE_INST_ANY']/wc:5.2f} busy_cyc {m['SQ_BUSY_CYCLES']:10.0f} gui {m.get('GRBM_GUI_ACTIVE',0):9.0f}"
    else:
        line += " " + " ".join(f"{c.replace('SQ_','')} {m[c]:.3g}" for c in names if c in m)
    print(line)
