"""Independent checks of the CPU oracle against plain float64 numpy restatements of the *definitions* (not of the
reference's code): the reference has no golden vectors for traversal or camera rays ("parity unpinned"), so
these pin the semantics instead — the BVH + traversal must return what a brute-force closest-hit search returns, and a
camera ray must be the unprojection of its pixel centre."""
import ctypes as C

import numpy as np

from oracle_binding import OracleEngine, oracle_lib
from strolle_amd import Buffer, CameraMode, scenes


def _brute_force(tris, origin, direction):
    """Möller–Trumbore over every triangle in float64; returns (t, index) of the closest front- or back-face hit."""
    p0, p1, p2 = tris[:, 0], tris[:, 1], tris[:, 2]
    e1, e2 = p1 - p0, p2 - p0
    pvec = np.cross(direction, e2)
    det = np.einsum("ij,ij->i", e1, pvec)
    ok = np.abs(det) >= np.finfo(np.float32).eps
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tvec = origin - p0
    u = np.einsum("ij,ij->i", tvec, pvec) * inv
    qvec = np.cross(tvec, e1)
    v = (qvec @ direction) * inv
    t = np.einsum("ij,ij->i", e2, qvec) * inv
    hit = ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0)
    if not hit.any():
        return np.inf, -1, np.inf
    ts = np.where(hit, t, np.inf)
    order = np.argsort(ts)
    return ts[order[0]], int(order[0]), ts[order[1]] if len(order) > 1 else np.inf


def test_bvh_traversal_equals_brute_force():
    lib = oracle_lib()
    lib.or_probe_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    e = OracleEngine()
    scenes.build_random_soup(e, 2400, seed=2)
    e.tick()
    raw = e.read_scene(1).view(np.float32).reshape(-1, 36)       # the reference's 144-B triangle: 9 float4
    tris = raw.reshape(-1, 9, 4)[:, [0, 3, 6], :3].astype(np.float64)
    rng = np.random.default_rng(7)
    checked = hits = 0
    for _ in range(400):
        origin = rng.uniform(-2.5, 2.5, 3).astype(np.float32)
        target = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
        d = (target - origin).astype(np.float64); d /= np.linalg.norm(d)
        d32 = d.astype(np.float32)
        out = np.zeros(11, np.float32)
        lib.or_probe_trace(e._h, origin.ctypes.data, d32.ctypes.data, 0.0, 0, out.ctypes.data)
        t_bf, idx, t_second = _brute_force(tris, origin.astype(np.float64), d32.astype(np.float64))
        if np.isfinite(t_bf) and (t_second - t_bf) < 1e-4 * max(1.0, t_bf):
            continue  # two surfaces within rounding of each other: either is a correct answer
        checked += 1
        if not np.isfinite(t_bf):
            assert out[0] >= 3.0e38, f"oracle hit at {out[0]} where brute force misses"
        else:
            hits += 1
            assert abs(out[0] - t_bf) <= 2e-4 * max(1.0, t_bf), f"closest hit {out[0]} vs brute force {t_bf}"
            assert out[10] > 0, "a hit must have traversed something (used_memory)"
    assert checked > 300 and hits > 100


def test_any_hit_agrees_with_brute_force_occlusion():
    lib = oracle_lib()
    lib.or_probe_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    e = OracleEngine()
    scenes.build_random_soup(e, 1500, seed=4)
    e.tick()
    tris = e.read_scene(1).view(np.float32).reshape(-1, 9, 4)[:, [0, 3, 6], :3].astype(np.float64)
    rng = np.random.default_rng(11)
    agree = total = 0
    for _ in range(300):
        a = rng.uniform(-1.5, 1.5, 3).astype(np.float32); b = rng.uniform(-1.5, 1.5, 3).astype(np.float32)
        d = (b - a).astype(np.float64); length = float(np.linalg.norm(d)); d /= length
        d32 = d.astype(np.float32)
        out = np.zeros(11, np.float32)
        lib.or_probe_trace(e._h, a.ctypes.data, d32.ctypes.data, np.float32(length), 1, out.ctypes.data)
        t_bf, _, _ = _brute_force(tris, a.astype(np.float64), d32.astype(np.float64))
        if abs(t_bf - length) < 1e-3:
            continue
        total += 1
        agree += int((out[0] < np.float32(length)) == (t_bf < length))
    assert total > 250 and agree == total, f"{total - agree} of {total} shadow rays disagree with brute force"


def test_camera_ray_is_the_unprojected_pixel_centre():
    from strolle_amd.api import StCamera
    lib = oracle_lib()
    lib.or_probe_camera_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    size = (640, 360)
    desc = scenes.camera_for(size, (0.3, 1.2, 3.0), (-0.2, 0.8, 0.0))
    c = desc.to_c()
    view = np.linalg.inv(np.array(c.transform[:], np.float64).reshape(4, 4).T)      # column-major -> world-to-view
    proj = np.array(c.projection[:], np.float64).reshape(4, 4).T
    ndc_to_world = np.linalg.inv(proj @ view)
    for (x, y) in [(0, 0), (639, 359), (320, 180), (17, 301), (600, 5)]:
        out = np.zeros(6, np.float32)
        lib.or_probe_camera_ray(C.byref(c), x, y, out.ctypes.data)
        ndc = np.array([(x + 0.5) * 2.0 / size[0] - 1.0, -((y + 0.5) * 2.0 / size[1] - 1.0)])
        def unproject(z):
            p = ndc_to_world @ np.array([ndc[0], ndc[1], z, 1.0]); return p[:3] / p[3]
        near, far = unproject(1.0), unproject(float(np.finfo(np.float32).eps))   # reverse-Z: z = 1 is the near plane
        d = far - near; d /= np.linalg.norm(d)
        assert np.allclose(out[:3], near, atol=2e-4), (out[:3], near)
        assert np.allclose(out[3:], d, atol=2e-4), (out[3:], d)
        assert np.allclose(out[:3], [0.3, 1.2, 3.0], atol=0.2)   # the ray starts on the near plane in front of the eye


def test_heatmap_counts_match_traversal_probe():
    """The heatmap plane stores exactly the used_memory of the pixel's primary ray (bvh_heatmap.rs:20-31)."""
    from strolle_amd.api import StCamera
    lib = oracle_lib()
    lib.or_probe_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    lib.or_probe_camera_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    e = OracleEngine()
    scenes.build_cornell(e)
    size = (64, 48)
    desc = scenes.cornell_camera(size, CameraMode.BVH_HEATMAP)
    cam = e.create_camera(desc)
    e.tick(); e.render_camera(cam)
    um = e.read_buffer(cam, Buffer.DBG_USED_MEMORY).reshape(size[1], size[0])
    c = desc.to_c()
    for (x, y) in [(0, 0), (32, 24), (63, 47), (10, 40)]:
        ray = np.zeros(6, np.float32); out = np.zeros(11, np.float32)
        lib.or_probe_camera_ray(C.byref(c), x, y, ray.ctypes.data)
        o, d = ray[:3].copy(), ray[3:].copy()
        lib.or_probe_trace(e._h, o.ctypes.data, d.ctypes.data, 0.0, 0, out.ctypes.data)
        assert int(out[10]) == int(um[y, x])
