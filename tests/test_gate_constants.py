"""The fast build's tolerance gates cannot move silently: every gate constant of the GPU test modules equals its pinned value in
profiles/gates.json. Tightening a gate means editing both; LOOSENING one additionally needs a `history` line naming the profiles/ file that shows
why — which is what a reviewer then sees in the diff. (CPU test: it reads the modules' source, it does not import them.)"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _constants(path):
    src = open(os.path.join(ROOT, path)).read()
    found = {}
    for m in re.finditer(r"^([A-Z][A-Z0-9_]*(?:\s*,\s*[A-Z][A-Z0-9_]*)*)\s*=\s*([^#\n]+)", src, re.M):
        names = [n.strip() for n in m.group(1).split(",")]
        values = [v.strip() for v in m.group(2).split(",")]
        if len(names) == len(values):
            for n, v in zip(names, values):
                try:
                    found[n] = float(v)
                except ValueError:
                    pass
    return found


def test_every_gate_constant_is_the_pinned_one():
    pinned = json.load(open(os.path.join(ROOT, "profiles", "gates.json")))
    for path in ("tests/test_gpu_fast_steady_state.py", "tests/test_gpu_fast_tolerance.py"):
        have = _constants(path)
        for name, value in pinned[path].items():
            assert name in have, f"{path} no longer defines {name}: update profiles/gates.json with the reason"
            assert have[name] == value, (f"{path}: {name} = {have[name]} but profiles/gates.json pins {value}: a gate moves only together with its row there "
                                         f"(and, when it is loosened, a history line naming the profiles/ evidence)")
    assert len(pinned["history"]) >= 4 and all(isinstance(h, str) for h in pinned["history"])


def test_the_gates_in_use_are_not_looser_than_the_stated_tolerance():
    """DESIGN.md section 2.2 states the per-lane tolerance (1e-5 + 2e-3 relative) and the per-plane fractions; the pinned values are those."""
    g = json.load(open(os.path.join(ROOT, "profiles", "gates.json")))
    s, t = g["tests/test_gpu_fast_steady_state.py"], g["tests/test_gpu_fast_tolerance.py"]
    assert t["RTOL"] == 2e-3 and t["ATOL"] == 1e-5
    assert "from test_gpu_fast_tolerance import ATOL, RTOL" in open(os.path.join(ROOT, "tests", "test_gpu_fast_steady_state.py")).read()
    assert s["BAD_FRACTION_LAUNCH"] <= 5e-4 and s["BAD_FRACTION_FRAME_DISCRETE"] <= 5e-3 and s["BAD_FRACTION_FRAME_FILTERED"] <= 2e-2 and s["FRAME_PSNR_DB"] >= 65.0
    assert t["BAD_FRACTION"] <= 2e-3
