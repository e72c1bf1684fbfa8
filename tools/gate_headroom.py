#!/usr/bin/env python3
"""Headroom of the fast build's tolerance gates: reads the reports the GPU tests leave (gpurun_out/fast_steady_*.json,
gpurun_out/fast_tolerance_*.json), copies them to profiles/<tag>_*, and writes profiles/<tag>_gate_headroom.json — per report and gate the
worst measured value, the limit and measured / limit (for a PSNR floor: the margin in dB). The limits themselves are pinned by
profiles/gates.json (tests/test_gate_constants.py fails when a constant in the GPU tests moves without its row there).

    python tools/gate_headroom.py r05
"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
gates = json.load(open(os.path.join(ROOT, "profiles", "gates.json")))["tests/test_gpu_fast_steady_state.py"]
out = {"tag": tag, "gates": gates, "reports": {}}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "fast_steady_*.json"))):
    name = os.path.basename(path)[:-5]
    d = json.load(open(path))
    shutil.copy(path, os.path.join(ROOT, "profiles", f"{tag}_{name}.json"))
    rows = d.get("whole_frame_rows", [])
    filtered = [r for r in rows if "psnr" in r]
    discrete = [r for r in rows if "psnr" not in r]
    rep = {}
    # the launch-by-launch loop runs only on frames the plan marks "launches": a report whose plan has none did not run it ("not run", not "0.0")
    if "launches" in (d.get("plan") or {}).values():
        worst = max([r["bad_fraction"] for r in d.get("launch_rows_with_outliers") or []] + [0.0])
        rep["BAD_FRACTION_LAUNCH"] = {"worst": worst, "limit": gates["BAD_FRACTION_LAUNCH"], "used": round(worst / gates["BAD_FRACTION_LAUNCH"], 3)}
    else:
        rep["BAD_FRACTION_LAUNCH"] = {"worst": "not run", "limit": gates["BAD_FRACTION_LAUNCH"], "used": "not run"}
    if d.get("tree_counters"):
        rep["tree"] = {"tree": d.get("tree"), **d["tree_counters"]}
    if discrete:
        w = max(discrete, key=lambda r: r["bad_fraction"])
        rep["BAD_FRACTION_FRAME_DISCRETE"] = {"worst": w["bad_fraction"], "plane": w["plane"], "frame": w["frame"], "limit": gates["BAD_FRACTION_FRAME_DISCRETE"],
                                              "used": round(w["bad_fraction"] / gates["BAD_FRACTION_FRAME_DISCRETE"], 3)}
    if filtered:
        w = max(filtered, key=lambda r: r["bad_fraction"])
        rep["BAD_FRACTION_FRAME_FILTERED"] = {"worst": w["bad_fraction"], "plane": w["plane"], "frame": w["frame"], "limit": gates["BAD_FRACTION_FRAME_FILTERED"],
                                              "used": round(w["bad_fraction"] / gates["BAD_FRACTION_FRAME_FILTERED"], 3)}
        p = min(filtered, key=lambda r: r["psnr"])
        rep["FRAME_PSNR_DB"] = {"worst": round(p["psnr"], 2), "plane": p["plane"], "frame": p["frame"], "floor": gates["FRAME_PSNR_DB"], "margin_dB": round(p["psnr"] - gates["FRAME_PSNR_DB"], 2)}
        m = max(filtered, key=lambda r: abs(r["mean_ratio"] - 1.0))
        rep["FRAME_MEAN_RTOL"] = {"worst": abs(m["mean_ratio"] - 1.0), "plane": m["plane"], "limit": gates["FRAME_MEAN_RTOL"], "used": round(abs(m["mean_ratio"] - 1.0) / gates["FRAME_MEAN_RTOL"], 3)}
    out["reports"][name] = rep
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "fast_tolerance_*.json"))):
    shutil.copy(path, os.path.join(ROOT, "profiles", f"{tag}_{os.path.basename(path)}"))
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_gate_headroom.json"), "w"), indent=1)
for name, rep in out["reports"].items():
    print(name)
    for g, v in rep.items():
        print("   %-30s %s" % (g, {k: v[k] for k in v if k not in ("plane", "frame")}), v.get("plane", ""))
