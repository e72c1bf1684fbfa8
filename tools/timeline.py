#!/usr/bin/env python3
"""Timeline of the last frames of a rocprofv3 --kernel-trace run of bench.py (two streams): one line per dispatch — queue, start
offset, duration, the kernels in flight beside it — and how long the chip had 0 / 1 / 2+ kernels in flight. A frame starts at a
prim_visibility dispatch. Runs anywhere (plain csv)."""
import argparse, csv, re

def short(name):
    name = re.sub(r"^.*?st::(fast|exact)::k_", "", name.split("(")[0])
    return name[:44]

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("trace"); ap.add_argument("--frames", type=int, default=3)
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        if "st::" not in r["Kernel_Name"]:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[3].startswith("prim_visibility")]
    if len(starts) < a.frames + 2:
        raise SystemExit("too few frames in the trace")
    first = starts[-(a.frames + 1)]; last = starts[-1]
    sel = rows[first:last]
    t0 = sel[0][0]
    for s, e, q, n in sel:
        beside = sorted({m for (s2, e2, q2, m) in sel if q2 != q and s2 < e and e2 > s})
        print(f"q{q} {(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n:44s} | {', '.join(beside)}")
    # occupancy of the chip by kernel count
    ev = []
    for s, e, q, n in sel:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth = 0; prev = ev[0][0]; hist = {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - prev); prev = t; depth += d
    span = ev[-1][0] - ev[0][0]
    print(f"\n{a.frames} frames, {span / 1e3 / a.frames:.1f} us per frame by the trace; sum of kernel durations {sum(e - s for s, e, _, _ in sel) / 1e3 / a.frames:.1f} us per frame")
    for k in sorted(hist):
        print(f"  {k} kernel(s) in flight: {hist[k] / span * 100:5.1f} % of the time")

if __name__ == "__main__":
    main()
