"""The build bench.py TIMES, checked against the CPU oracle at the size and in the state it is timed in.

bench.py measures the FAST arithmetic build (ST_ARITH_FAST) on Cornell 1920x1080 Image{denoise} after the renderer's
temporal state has settled (reservoir sample counts at their cap, denoiser history long). tests/test_gpu_fast_tolerance.py
compares that build with the oracle launch by launch, but at <= 256x160, on frames 2..5 (cold state: the preview passes draw
eight neighbours, the short-history variance path serves every pixel) and — because a pass mask is set — with the engine's
whole-frame switches off (gi_skip_history_copy, variance_in_reproject, skip_dead_scratch, gi_preview_both;
st_engine.cpp `whole_graph`). This module closes both holes at the benchmark's own size:

* test_fast_launches_1080p_steady_state — the oracle carries the history to frame >= 19, then the launch-by-launch loop of
  test_gpu_fast_tolerance.py runs on one frame of each of the three GI schedules (frame.rs:19-21: even tracing frame, odd
  tracing frame, validation frame) at 1920x1080: every launch of the fast build reads the oracle's state and is compared
  with the oracle's result, every plane, same per-lane tolerance.
* test_fast_whole_frame_single_step — the oracle's complete state after frame F-1 is uploaded, the product renders frame
  F UNMASKED (all whole-frame switches ON, the LEAN frame included: exactly the launch structure bench.py times), and every
  plane the frame leaves behind plus the composed frame are compared with the oracle's state after frame F. One frame each
  of the three schedules; Cornell 1080p (the headline), dungeon 1080p (config 3's scene in Image mode), dungeon 3840x2160
  (config 5). Errors accumulate over the ~15 launches of one frame but never from frame to frame (see the numbers below).
  The lean frame does not store the planes nothing reads again (include/strolle_hip.h st_debug_keep_all_planes: velocity,
  the encoded surface map, both diffuse-sample planes, the reprojected GI reservoirs of tracing frames, first-preview-pass
  results that are a plain normalisation of their input, the last a-trous pass's colours — whose content reaches the
  composed frame, which IS compared); one more frame per run is rendered with
  st_debug_keep_all_planes(1) and compared on every plane ("whole_keep").

Both draw on one oracle run per (scene, size): frames alternate between the two kinds of check.

Tolerance per 32-bit lane as in test_gpu_fast_tolerance.py (bit-equal, or floats within ATOL + RTOL * max(|a|, |b|));
per plane a stated fraction of the lanes may miss it:
  launch by launch ......... BAD_FRACTION_LAUNCH
  whole frame, reservoirs .. BAD_FRACTION_FRAME_DISCRETE: a flipped discrete choice early in the frame (a shadow ray at a
                             silhouette, `rand * w_sum < w`) hands every later pass of that pixel another sample
  whole frame, colours ..... the a-trous chain spreads one flipped sample over a 63-pixel footprint with a small weight:
                             compared by PSNR and mean instead of per lane (a per-lane relative test of a filtered image
                             measures the filter's footprint, not the arithmetic)
Measured figures are written to gpurun_out/fast_steady_<scene>_<w>x<h>.json.
"""
import json
import os

import numpy as np
import pytest

from oracle_binding import OracleEngine
from parity import psnr
from strolle_amd import Buffer, CameraMode, Engine, PassBit, scenes
from test_gpu_fast_tolerance import ATOL, RTOL, lanes_outside_tolerance

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT_ONLY = os.environ.get("ST_TOL_REPORT_ONLY") == "1"
FLOAT_BUFFERS = [b for b in Buffer if b != Buffer.DBG_USED_MEMORY]
# Measured on MI355X (round 3, gpurun_out/fast_steady_*.json): launch by launch at 1080p the worst plane of any launch has
# 3.4e-5 of its lanes outside the per-lane tolerance (an a-trous plane; the worst resampling launch 1.7e-5). After one WHOLE
# unmasked frame: Cornell 1080p <= 3.6e-4 on every plane, dungeon 1080p <= 1.7e-3 on reservoir / sample planes and
# <= 6.9e-3 on the filtered GI planes (PSNR >= 73 dB, means within 3.2e-5), dungeon 4K <= 8.9e-4 (PSNR >= 79 dB).
BAD_FRACTION_LAUNCH = 5e-4
# Whole frame: planes whose content is a discrete choice carried through the frame (reservoirs, samples before filtering)
BAD_FRACTION_FRAME_DISCRETE = 2e-3   # (round 6: 5e-3 -> 2e-3; worst measured 5.3e-4 since primary hits are exact, profiles/r06_gate_headroom.json)
# Whole frame: filtered colour planes and the composed frame (PSNR against the oracle's plane, peak = its 99.9th percentile)
BAD_FRACTION_FRAME_FILTERED = 5e-3   # (round 6: 2e-2 -> 5e-3; worst measured 1.3e-3)
FRAME_PSNR_DB = 70.0   # (round 6: 65 -> 70; worst measured 76.4)
FRAME_MEAN_RTOL = 5e-4

# planes a frame leaves behind for the next one or for the caller (everything else is scratch that later launches of
# the same frame overwrite; the fused launches of the whole-frame graph never store some of it — DESIGN.md section 4)
REF_PLANES = {Buffer.REF_HITS, Buffer.REF_RAYS, Buffer.REF_COLORS}
def lean_planes(frame):
    """planes the lean frame leaves unwritten on `frame` (st_types.h kLean*, st_engine.cpp `lean_frame`)"""
    sm = Buffer.PRIM_SURFACE_MAP_B if frame % 2 else Buffer.PRIM_SURFACE_MAP_A
    skip = {Buffer.VELOCITY_MAP, sm, Buffer.DI_DIFF_SAMPLES, Buffer.GI_DIFF_SAMPLES, Buffer.DI_DIFF_CURR_COLORS, Buffer.GI_DIFF_CURR_COLORS,
            Buffer.GI_RESERVOIRS_3}   # the first preview pass's results: stored only where that pass resampled
    if frame % 6 < 4:
        skip.add(Buffer.GI_RESERVOIRS_2)
    return skip


FILTERED = {Buffer.DI_DIFF_PREV_COLORS, Buffer.DI_DIFF_CURR_COLORS, Buffer.DI_DIFF_STASH,
            Buffer.GI_DIFF_PREV_COLORS, Buffer.GI_DIFF_CURR_COLORS, Buffer.GI_DIFF_STASH}


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product has no CPU fallback"
    return torch


def _bits(x):
    return x.view(np.uint32)


# The velocity map is a DIFFERENCE of two screen positions (prim_raster.rs:114-127: current minus previous, each ~ 10^3 pixels): under
# motion it is a sub-pixel number whose absolute error is that of its operands, so it gets an absolute tolerance in pixels (what
# consumes it, the reprojection map, holds positions and is compared with the ordinary relative tolerance).
VELOCITY_ATOL_PX = 1e-3


def _bad_fraction(got, want, chunk=1 << 23, atol=ATOL):
    """Fraction of 32-bit lanes outside the per-lane tolerance. Bit-equal stretches cost one compare; the float test runs on
    the lanes that differ, a chunk at a time (a 4K reservoir plane is 133 M lanes)."""
    gb, wb = _bits(got), _bits(want)
    bad = 0
    for i in range(0, got.size, chunk):
        ne = gb[i:i + chunk] != wb[i:i + chunk]
        if ne.any():
            bad += int(lanes_outside_tolerance(got[i:i + chunk][ne], want[i:i + chunk][ne], atol=atol).sum())
    return bad / got.size


def _diagnose(got, want, lanes_per_px, width):
    """Where a plane's outliers sit: count per lane of the pixel's record, and a few examples (pixel, lane, got, want)."""
    bad = lanes_outside_tolerance(got, want)
    idx = np.flatnonzero(bad)
    per_lane = np.bincount(idx % lanes_per_px, minlength=lanes_per_px).tolist()
    ex = [{"px": [int((i // lanes_per_px) % width), int((i // lanes_per_px) // width)], "lane": int(i % lanes_per_px), "got": float(got[i]), "want": float(want[i])} for i in idx[:: max(1, len(idx) // 6)][:6]]
    return {"outliers_per_record_lane": per_lane, "examples": ex}


def _bad_fractions(got, want, planes):
    """{plane: bad fraction} with the planes compared on a thread pool (numpy releases the GIL in these loops)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:
        futures = {b: pool.submit(_bad_fraction, got[b], want[b], 1 << 23, VELOCITY_ATOL_PX if b == Buffer.VELOCITY_MAP else ATOL) for b in planes}
        return {b: f.result() for b, f in futures.items()}


def _plane_stats(got, want):
    g = got.reshape(-1, 4)[:, :3].astype(np.float64); w = want.reshape(-1, 4)[:, :3].astype(np.float64)
    ok = np.isfinite(g).all(axis=1) & np.isfinite(w).all(axis=1)
    g, w = g[ok], w[ok]
    if not w.any() and not g.any():   # a plane the mode never writes (the DI planes under GiDiffuse): zero on both sides
        return {"psnr": 999.0, "mean_ratio": 1.0, "nonfinite_px": int((~ok).sum())}
    peak = float(max(np.percentile(w, 99.9), 1e-6))
    return {"psnr": psnr(np.clip(g, 0, peak), np.clip(w, 0, peak), peak), "mean_ratio": float(g.mean() / max(w.mean(), 1e-30)), "nonfinite_px": int((~ok).sum())}


_SCENES = {"cornell": (scenes.build_cornell, scenes.cornell_camera), "dungeon": (scenes.build_dungeon, scenes.dungeon_camera),
           # SYNTHETIC: the dungeon with every triangle split into 16 (208 k triangles) — the stand-in for BASELINE.json config 3's "~100k tris" scene
           "dungeon134k": (lambda e: scenes.build_dungeon(e, subdivide=2), scenes.dungeon_camera)}
_runs = {}


def _spawned_instance_xform(frame):
    """the fourth torus of the `spawned` runs: half the demo's scale, in front of the camera, turning and drifting a little every frame"""
    import math
    a = 1.0 + 0.07 * frame
    c, s_ = 0.25 * math.cos(a), 0.25 * math.sin(a)
    return np.array([[c, -s_, 0.0, -5.75 + 0.01 * frame], [s_, c, 0.0, 0.45], [0.0, 0.0, 0.25, -19.0]], np.float32)


def _run(scene, size, plan, moving=False, mode=CameraMode.IMAGE, tree="host"):
    """One oracle run of max(plan) frames; frame f (the ENGINE's frame number, which starts at 1 and decides the GI schedule:
    f % 6 < 4 tracing — even f samples, odd f resamples spatially —, else validation) is checked as plan[f] says ("launches" /
    "whole" / "whole_keep"); other frames only advance the oracle. Returns {"launches": [...], "whole": [...]} of report rows; cached per (scene, size)."""
    key = (scene, size, moving, int(mode), tree)
    if key in _runs:
        return _runs[key]
    torch = _torch()
    build, camera_fn = _SCENES[scene]
    prod, orac = Engine(device=0, exact=False), OracleEngine()
    assert not prod.exact
    # tree: "host" — a static scene, the host's binned-SAH tree (the reference's, and under the default ST_BVH_AUTO the first tree of every engine);
    #       "device" — the FIRST tree is k_lbvh.hip's already (st_set_bvh_refresh(ST_BVH_BUILD_DEVICE) before the scene exists; the default for large scenes);
    #       "spawned" — the default mode: host tree first, then an instance appears at frame 2 (a device BUILD) and moves at every later
    #                   frame (device REFITS of that tree, a rebuild after 15) — the oracle rebuilds its SAH tree every time.
    # Since late round 6 ST_BVH_AUTO gives a scene's FIRST tree to the device builder too when the host's tree hangs long leaf runs on large faces (st_tick.cpp
    # device_build_possible; profiles/r06_tree_choice*.txt): "dungeon134k" (every triangle split into 16) is such a scene — for it "device" IS the default mode
    # and "host" asks for the host's tree explicitly.
    big = scene == "dungeon134k"
    if tree == "device" and not big:
        prod.set_bvh_refresh(3)
    if tree == "host" and big:
        prod.set_bvh_refresh(0)
    for e in (prod, orac):
        build(e); e.set_seed(0)
    desc = camera_fn(size, mode)
    cp, co = prod.create_camera(desc), orac.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    full_mask = (1 << 64) - 1
    report = {"launches": [], "whole": []}
    held = {}  # what the product's planes hold right now (arrays it was given or that were read back from it)

    def upload(state):
        for b, data in state.items():
            if b in held and np.array_equal(_bits(held[b]), _bits(data)):
                continue
            prod.write_buffer(cp, b, data); held[b] = data

    def read_prod():
        got = {b: prod.read_buffer(cp, b) for b in FLOAT_BUFFERS}
        held.update(got)
        return got

    def read_orac():
        return {b: orac.read_buffer(co, b) for b in FLOAT_BUFFERS}

    for frame in range(1, max(plan) + 1):   # the engine numbers frames from 1 (strolle/src/lib.rs:152)
        if moving:   # what bench.py's `moving` region does: the light of cornell.rs:82-93 (1/60 s per frame), the camera on an orbit
            import math
            from strolle_amd import Light
            t = frame / 60.0
            desc = scenes.camera_for(size, (3.2 * math.sin(0.1 * t), 1.0, 3.2 * math.cos(0.1 * t)), (0.0, 1.0, 0.0), CameraMode.IMAGE)
            for e in (prod, orac):
                e.insert_light(1, Light.point((math.sin(t) / 2.0, 1.5, math.cos(t) / 2.0), 0.15, (50.0 / (4.0 * math.pi),) * 3, 20.0))
        if tree == "spawned" and frame >= 2:
            from strolle_amd import Instance
            for e in (prod, orac):
                e.insert_instance(5900, Instance(5000, 5001, _spawned_instance_xform(frame)))
        for e, c in ((prod, cp), (orac, co)):
            e.update_camera(c, desc)
        prod.tick(); orac.tick()
        assert prod.world()[1] == frame + 1 and orac.world()[1] == frame + 1, "frame numbering"
        kind = plan.get(frame)
        if kind is None:
            orac.render_camera(co, compose=False)
            continue
        before = read_orac()
        if kind == "launches":
            prod.set_pass_mask(0); prod.render_camera(cp, out.data_ptr(), stream); torch.cuda.synchronize()
            groups = prod.last_launches()
            assert groups, "no launches"
            for bits in groups:
                orac.set_pass_mask(bits)
                ref_frame = orac.render_camera(co, compose=bool(bits & PassBit.COMPOSITION))
                want = read_orac()
                upload(before)
                prod.set_pass_mask(bits)
                prod.render_camera(cp, out.data_ptr(), stream); torch.cuda.synchronize()
                got = read_prod()
                name = "+".join(p.name for p in PassBit if bits & p)
                for b, frac in _bad_fractions(got, want, FLOAT_BUFFERS).items():
                    if frac > 0:
                        report["launches"].append({"frame": frame, "launch": name, "plane": b.name, "bad_fraction": frac})
                if bits & PassBit.COMPOSITION:
                    frac = _bad_fraction(out.cpu().numpy().reshape(-1), np.ascontiguousarray(ref_frame).reshape(-1))
                    report["launches"].append({"frame": frame, "launch": name, "plane": "composed frame", "bad_fraction": frac})
                before = want
            orac.set_pass_mask(full_mask); prod.set_pass_mask(full_mask)
        else:
            ref_frame = orac.render_camera(co, compose=True)
            want = read_orac()
            upload(before)
            prod.set_pass_mask(full_mask)
            prod.keep_all_planes(kind == "whole_keep")
            prod.render_camera(cp, out.data_ptr(), stream); torch.cuda.synchronize()
            prod.keep_all_planes(False)
            assert prod.last_launches(), "no launches"
            got = read_prod()
            skip = REF_PLANES | (set() if (kind == "whole_keep" or mode != CameraMode.IMAGE) else lean_planes(frame))   # the lean frame is Image mode's
            for b, frac in _bad_fractions(got, want, [b for b in FLOAT_BUFFERS if b not in skip]).items():
                row = {"frame": frame, "kind": kind, "plane": b.name, "bad_fraction": frac}
                if frac > 1e-3 and b not in FILTERED:
                    row.update(_diagnose(got[b], want[b], got[b].size // (size[0] * size[1]), size[0]))
                if b in FILTERED:
                    row.update(_plane_stats(got[b], want[b]))
                report["whole"].append(row)
            row = {"frame": frame, "kind": kind, "plane": "composed frame", "bad_fraction": _bad_fraction(out.cpu().numpy().reshape(-1), np.ascontiguousarray(ref_frame).reshape(-1))}
            row.update(_plane_stats(out.cpu().numpy(), ref_frame))
            report["whole"].append(row)
            # reservoir sample counts: the state must be the steady one the benchmark times
            m = want[Buffer.GI_RESERVOIRS_0].reshape(-1, 16)[:, 3]
            lit = want[Buffer.PRIM_SURFACE_MAP_A if frame % 2 == 0 else Buffer.PRIM_SURFACE_MAP_B].reshape(-1, 4)[:, 2] != 0
            report.setdefault("state", []).append({"frame": frame, "gi_m_median": float(np.median(m[lit])) if lit.any() else None,
                                                  "history_median": float(np.median(want[Buffer.DI_DIFF_MOMENTS_A if frame % 2 == 0 else Buffer.DI_DIFF_MOMENTS_B].reshape(-1, 4)[:, 0][lit])) if lit.any() else None})
    report["tree"] = {"device_builds": prod.device_builds(), "device_tree_refits": prod.device_tree_refits(), "walk_overflow": list(prod.walk_overflow())}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    suffix = {"host": "", "device": "_devtree", "spawned": "_spawned"}[tree]
    with open(os.path.join(ROOT, "gpurun_out", f"fast_steady_{scene}{'_moving' if moving else ''}{'' if mode == CameraMode.IMAGE else '_' + mode.name.lower()}_{size[0]}x{size[1]}{suffix}.json"), "w") as f:
        json.dump({"scene": scene, "size": size, "tree": tree, "tree_counters": report["tree"], "plan": {str(k): v for k, v in plan.items()}, "rtol": RTOL, "atol": ATOL,
                   "launch_rows_with_outliers": sorted(report["launches"], key=lambda r: -r["bad_fraction"])[:60],
                   "whole_frame_rows": report["whole"], "state": report.get("state", [])}, f, indent=1)
    prod.close(); orac.close()
    _runs[key] = report
    return report


# frames 19 / 20 / 23: odd tracing (spatial resampling), even tracing, validation — launch by launch;
# frames 21 / 22 / 24: odd tracing, validation, even tracing as whole lean frames; 25 (odd tracing) with every plane kept
PLAN_1080P = {19: "launches", 20: "launches", 21: "whole", 22: "whole", 23: "launches", 24: "whole", 25: "whole_keep"}
PLAN_DUNGEON_1080P = {13: "whole", 14: "whole", 15: "whole_keep", 17: "whole"}   # odd tracing, even tracing, odd tracing, validation
PLAN_DUNGEON_4K = {8: "whole", 9: "whole", 10: "whole"}                           # even tracing, odd tracing, validation
PLAN_CONFIG3 = {8: "whole", 9: "whole", 10: "whole"}                             # even tracing, odd tracing, validation
PLAN_MOVING = {13: "whole", 14: "whole", 15: "launches", 16: "whole", 17: "whole_keep"}   # odd tracing, even tracing, odd tracing, validation, validation


def _check_launch_rows(rows, what):
    for r in rows:
        assert REPORT_ONLY or r["bad_fraction"] <= BAD_FRACTION_LAUNCH, f"{what} frame {r['frame']} launch {r['launch']}: plane {r['plane']}: {r['bad_fraction']:.2e} of the lanes outside rtol {RTOL} / atol {ATOL}"


def _check_whole_rows(rows, what):
    assert rows, "no whole-frame rows"
    for r in rows:
        if "psnr" in r:
            assert REPORT_ONLY or r["nonfinite_px"] == 0, f"{what} frame {r['frame']} {r['plane']}: non-finite pixels"
            assert REPORT_ONLY or r["psnr"] >= FRAME_PSNR_DB, f"{what} frame {r['frame']} {r['plane']}: PSNR {r['psnr']:.1f} dB against the oracle"
            assert REPORT_ONLY or abs(r["mean_ratio"] - 1.0) <= FRAME_MEAN_RTOL, f"{what} frame {r['frame']} {r['plane']}: mean ratio {r['mean_ratio']:.5f}"
            assert REPORT_ONLY or r["bad_fraction"] <= BAD_FRACTION_FRAME_FILTERED, f"{what} frame {r['frame']} {r['plane']}: {r['bad_fraction']:.2e} of the lanes outside tolerance after one whole frame"
        else:
            assert REPORT_ONLY or r["bad_fraction"] <= BAD_FRACTION_FRAME_DISCRETE, f"{what} frame {r['frame']} {r['plane']}: {r['bad_fraction']:.2e} of the lanes outside tolerance after one whole frame"


def test_fast_launches_1080p_steady_state():
    rep = _run("cornell", (1920, 1080), PLAN_1080P)
    assert {r["frame"] for r in rep["launches"]} <= {19, 20, 23}
    _check_launch_rows(rep["launches"], "cornell 1080p")
    for s in rep["state"]:   # the state the benchmark times: sample counts at their cap, long denoiser history
        assert REPORT_ONLY or (s["gi_m_median"] is not None and s["gi_m_median"] >= 8.0 and s["history_median"] >= 4.0), s


def test_fast_whole_frame_single_step():
    rep = _run("cornell", (1920, 1080), PLAN_1080P)
    assert {(r["frame"], r["kind"]) for r in rep["whole"]} == {(21, "whole"), (22, "whole"), (24, "whole"), (25, "whole_keep")}
    lean_frames = [r for r in rep["whole"] if r["kind"] == "whole"]
    assert not [r for r in lean_frames if r["plane"] in ("VELOCITY_MAP", "DI_DIFF_SAMPLES", "GI_DIFF_CURR_COLORS")]   # not compared: not stored
    assert [r for r in rep["whole"] if r["kind"] == "whole_keep" and r["plane"] == "GI_DIFF_CURR_COLORS"]               # the keep frame compares them
    _check_whole_rows(rep["whole"], "cornell 1080p")


def test_fast_whole_frame_single_step_dungeon_1080p():
    rep = _run("dungeon", (1920, 1080), PLAN_DUNGEON_1080P)
    _check_whole_rows(rep["whole"], "dungeon 1080p")


def test_fast_whole_frame_single_step_config3_as_written():
    """BASELINE.json config 3 AS WRITTEN (VERDICT r3 missing #3): the ~100 k-triangle dungeon — here the synthetic 208 k-triangle one,
    26 internal nodes deep, 32-bit traversal stacks — in CameraMode::GiDiffuse{denoise} at 1920x1080, the FAST build (what
    `bench.py --scene dungeon134k --mode gi_diffuse` times): whole unmasked frames of the three GI schedules from the oracle's state."""
    # (its level split x16, this scene's host tree hangs long leaf runs on large faces: the default mode's first tree is the device builder's since late round 6 — that IS what the bench times)
    rep = _run("dungeon134k", (1920, 1080), PLAN_CONFIG3, mode=CameraMode.GI_DIFFUSE, tree="device")
    assert rep["tree"]["device_builds"] >= 1, "ST_BVH_AUTO did not give this scene's first tree to the device builder"
    assert {r["frame"] for r in rep["whole"]} == set(PLAN_CONFIG3)
    _check_whole_rows(rep["whole"], "dungeon134k 1080p gi_diffuse")


def _no_walk_overflowed(rep, what):
    assert rep["tree"]["walk_overflow"][0] == 0, f"{what}: a wide walk found its stack full (st_debug_walk_overflow: {rep['tree']['walk_overflow']})"


def test_fast_whole_frame_single_step_dungeon_1080p_device_built_tree():
    """the tree k_lbvh.hip builds is held against the ORACLE, under the same profiles/gates.json limits as the host's tree — whole
    unmasked frames of every GI schedule from the oracle's own state, ReSTIR + SVGF, not only primary hits against the product's own host tree."""
    rep = _run("dungeon", (1920, 1080), PLAN_DUNGEON_1080P, tree="device")
    assert rep["tree"]["device_builds"] >= 1, "the first tree was not built on the device"
    _check_whole_rows(rep["whole"], "dungeon 1080p, device-built tree")
    _no_walk_overflowed(rep, "dungeon 1080p, device-built tree")


def test_fast_whole_frame_single_step_config3_host_tree():
    """... and BASELINE config 3 on the HOST's binned-SAH tree (st_set_bvh_refresh(ST_BVH_REBUILD): the reference's tree, what every scene got by default
    until late round 6 and what a heatmap camera or the exact build still gets)."""
    rep = _run("dungeon134k", (1920, 1080), PLAN_CONFIG3, mode=CameraMode.GI_DIFFUSE, tree="host")
    assert rep["tree"]["device_builds"] == 0
    _check_whole_rows(rep["whole"], "dungeon134k 1080p gi_diffuse, host tree")
    _no_walk_overflowed(rep, "config 3, host tree")


def test_fast_whole_frames_after_a_spawn_and_refits_in_the_default_mode():
    """The DEFAULT refresh mode (ST_BVH_AUTO): the host builds the first tree, an instance that appears at frame 2 is answered by a device build, its
    motion at every later frame by refits of that tree (15 in a row, then a rebuild) — while the oracle rebuilds the reference's SAH tree every frame.
    Whole unmasked frames 13 / 14 / 15 / 17 from the oracle's state, same gates."""
    rep = _run("dungeon", (1920, 1080), PLAN_DUNGEON_1080P, tree="spawned")
    assert rep["tree"]["device_builds"] >= 1 and rep["tree"]["device_tree_refits"] >= 10, rep["tree"]
    _check_whole_rows(rep["whole"], "dungeon 1080p, spawn + refits")
    _no_walk_overflowed(rep, "dungeon 1080p, spawn + refits")


def test_no_wide_walk_overflows_on_the_baseline_configs():
    """configs 2 / 3 / 5 never set the wide walks' overflow word (st_debug_walk_overflow) — read from the runs above."""
    for args, kw in ((("cornell", (1920, 1080), PLAN_1080P), {}), (("dungeon", (1920, 1080), PLAN_DUNGEON_1080P), {}),
                     (("dungeon134k", (1920, 1080), PLAN_CONFIG3), {"mode": CameraMode.GI_DIFFUSE, "tree": "device"}),
                     (("dungeon134k", (1920, 1080), PLAN_CONFIG3), {"mode": CameraMode.GI_DIFFUSE}), (("dungeon", (3840, 2160), PLAN_DUNGEON_4K), {})):
        _no_walk_overflowed(_run(*args, **kw), f"{args[0]} {args[1]}")


def test_fast_whole_frame_single_step_dungeon_4k():
    rep = _run("dungeon", (3840, 2160), PLAN_DUNGEON_4K)
    _check_whole_rows(rep["whole"], "dungeon 4K")


def test_fast_whole_frames_with_light_and_camera_moving():
    """The state bench.py's `moving` region times (`ms_per_step_moving`): the point light on cornell.rs's orbit and the camera
    circling the box, both updated before every tick. Reprojection follows real motion, shadow edges reset the DI history, the
    preview passes draw neighbours again and the lean frame's rebuilt first-pass records are read; whole unmasked frames of
    every GI schedule (and one launch-by-launch frame) against the oracle, Cornell 1280x720."""
    rep = _run("cornell", (1280, 720), PLAN_MOVING, moving=True)
    assert {(r["frame"], r["kind"]) for r in rep["whole"]} == {(13, "whole"), (14, "whole"), (16, "whole"), (17, "whole_keep")}
    _check_whole_rows(rep["whole"], "cornell 720p moving")
    _check_launch_rows(rep["launches"], "cornell 720p moving")


def test_report_only_is_not_set():
    """ST_TOL_REPORT_ONLY=1 turns every threshold of this module into a report (calibration runs). A run with it set must not come out
    green: this test fails then, so the switch cannot hide a regression in a gate."""
    assert not REPORT_ONLY, "ST_TOL_REPORT_ONLY=1 is set: the tolerance assertions of this module were skipped — reports only, not a passing run"
