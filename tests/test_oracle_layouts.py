"""Pins the oracle against the reference's own six unit tests — the only golden vectors the
reference holds for this path."""
import ctypes as C

import numpy as np
import pytest

from oracle_binding import oracle_lib
from strolle_amd import look_at_transform, perspective_infinite_reverse_rh


def test_camera_contain_known_answers():
    """strolle-gpu/src/camera.rs:152-175 — seven known-answer vectors at 1024x768."""
    lib = oracle_lib()
    lib.or_probe_camera_contain.argtypes = [C.c_float, C.c_float, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    cases = [((0, 0), (0, 0)), ((123, 456), (123, 456)), ((1023, 767), (1023, 767)), ((1024, 768), (1023, 767)),
             ((1025, 768), (1022, 767)), ((1030, 768), (1017, 767)), ((1030, 783), (1017, 752))]
    for (x, y), want in cases:
        ox, oy = C.c_uint32(), C.c_uint32()
        lib.or_probe_camera_contain(1024.0, 768.0, x, y, C.byref(ox), C.byref(oy))
        assert (ox.value, oy.value) == want


def test_gbuffer_roundtrip_tolerances():
    """strolle-gpu/src/gbuffer.rs:132-164 — pack -> unpack within eps 0.005 (alpha 0.1)."""
    lib = oracle_lib()
    src = np.array([0.1, 0.2, 0.3, 0.4, 0.26, 0.53, 0.80, 0.33, 2.0, 3.0, 4.0, 0.05, 0.25, 123.456], np.float32)
    out = np.zeros(14, np.float32); packed = np.zeros(8, np.float32)
    lib.or_probe_gbuffer_roundtrip(src.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p))
    eps = np.full(14, 0.005); eps[3] = 0.1
    n = src[4:7] / np.linalg.norm(src[4:7])  # the test's normal is nearly unit; compare against the normalised one
    want = src.copy(); want[4:7] = n
    assert np.all(np.abs(out - want) <= np.maximum(eps, 0.005 * np.abs(want)) + 1e-3)
    assert packed.view(np.uint32)[3] >> 24 == 1  # the "is some" marker byte (gbuffer.rs:71-76)


def test_di_reservoir_roundtrip_exact():
    """strolle-gpu/src/reservoir/di.rs:132-162 — ten reservoirs, exact round trip, fixes the 2x Vec4 layout."""
    lib = oracle_lib()
    lib.or_probe_di_reservoir_write.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_int]
    lib.or_probe_di_reservoir_read.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    buf = np.zeros(2 * 10 * 4, np.float32)
    for idx in range(10):
        lp = np.array([1.0, 2.0, 3.0 + idx], np.float32)
        lib.or_probe_di_reservoir_write(buf.ctypes.data, idx, 11.0, 12.0 + idx, 123.0, float(idx % 2 == 0), 3 * idx, lp.ctypes.data, int(idx % 2 == 0))
    for idx in range(10):
        out = np.zeros(9, np.float32)
        lib.or_probe_di_reservoir_read(buf.ctypes.data, idx, 10, out.ctypes.data)
        assert out[0] == 11.0 and out[1] == 12.0 + idx and out[2] == 123.0 and out[3] == float(idx % 2 == 0)
        assert out[4:5].view(np.uint32)[0] == 3 * idx
        assert tuple(out[5:8]) == (1.0, 2.0, 3.0 + idx) and out[8] == float(idx % 2 == 0)
    # layout: d0 = (m, w, pdf, bytes[occluded, confidence]), d1 = (light_point, light_id)
    assert buf[0] == 11.0 and buf[1] == 12.0 and buf[2] == 123.0 and buf[4:7].tolist() == [1.0, 2.0, 3.0]


def test_reprojection_roundtrip_exact():
    """strolle-gpu/src/reprojection.rs:81-96 — incl. the 0xcafebabe validity bit pattern."""
    lib = oracle_lib()
    lib.or_probe_reprojection_roundtrip.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    src = np.array([123.45, 234.56, 1.23], np.float32); out = np.zeros(3, np.float32); v = C.c_uint32()
    lib.or_probe_reprojection_roundtrip(src.ctypes.data, 0xCAFEBABE, out.ctypes.data, C.byref(v))
    assert np.array_equal(src, out) and v.value == 0xCAFEBABE


def test_u32_bytes_roundtrip():
    """strolle-gpu/src/utils/u32_ext.rs:31-34"""
    lib = oracle_lib()
    lib.or_probe_u32_bytes_roundtrip.restype = C.c_uint32
    assert lib.or_probe_u32_bytes_roundtrip(0xCAFEBABE) == 0xCAFEBABE


def test_white_noise_matches_pcg_reference():
    """noise/white.rs:15-45 — independent numpy restatement of the PCG-RXS-M-XS stream."""
    lib = oracle_lib()
    seed, x, y, n = 0x12345678, 17, 300, 64
    out = np.zeros(n, np.uint32)
    lib.or_probe_white_noise(C.c_uint32(seed), C.c_uint32(x), C.c_uint32(y), out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
    state = (seed ^ (48619 * x) ^ (95461 * y)) & 0xFFFFFFFF
    for i in range(n):
        state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
        word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
        assert out[i] == ((word >> 22) ^ word)


@pytest.mark.parametrize("op,fn,lo,hi,ulp", [(0, np.sin, -7.0, 7.0, 4), (1, np.cos, -7.0, 7.0, 4), (2, np.arccos, -1.0, 1.0, 4),
                                             (3, np.exp, -20.0, 20.0, 4), (6, np.log2, 1e-6, 1e6, 4), (7, np.exp2, -30.0, 30.0, 4)])
def test_deterministic_transcendentals_are_accurate(op, fn, lo, hi, ulp):
    """stm_* (or_math.h): within a few ulp of the float64 result — inside Vulkan's GLSL.std.450 envelope."""
    lib = oracle_lib()
    rng = np.random.default_rng(op)
    x = rng.uniform(lo, hi, 20000).astype(np.float32)
    out = np.zeros_like(x)
    lib.or_probe_math(op, x.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(len(x)))
    want = fn(x.astype(np.float64))
    err = np.abs(out.astype(np.float64) - want)
    tol = ulp * np.maximum(np.spacing(np.abs(want).astype(np.float32)).astype(np.float64), 1e-45)
    if op in (0, 1):
        tol = np.maximum(tol, 2.5e-7)  # absolute near the zeros of sin/cos (Cody-Waite reduction with 3 constants)
    assert np.all(err <= tol), float(np.max(err / tol))


def test_pow_accuracy_on_the_paths_domain():
    lib = oracle_lib()
    rng = np.random.default_rng(1)
    x = rng.uniform(0.0, 1.0, 20000).astype(np.float32)
    for y in (2.2, 1.0 / 2.2, 5.0, 8.0, 64.0, 2.4):
        yy = np.full_like(x, y)
        out = np.zeros_like(x)
        lib.or_probe_math(4, x.ctypes.data_as(C.c_void_p), yy.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(len(x)))
        want = np.power(x.astype(np.float64), float(np.float32(y)))
        rel = np.abs(out - want) / np.maximum(want, 1e-30)
        assert np.all((rel < 2e-5) | (want < 1e-30)), (y, float(rel.max()))


# ---------------------------------------------------------------------------------------------- glam 0.24.2 cross-check
# glam is a crates.io dependency (Cargo.lock: 0.24.2) that is not under /root/reference. Its scalar-path routines are
# restated twice, independently: in C++ (oracle/or_math.h, and again in the product's st_math.h / st_engine.cpp) and below
# in numpy float32, both from the crate's published source order. The two must agree bit for bit; a slip in the operation
# order of either (which term is subtracted first, where the reciprocal is taken) shows up here.
F = np.float32


def _glam_mat4_inverse(m):   # m[col][row]; glam-0.24.2/src/f32/scalar/mat4.rs `inverse`
    (m00, m01, m02, m03), (m10, m11, m12, m13), (m20, m21, m22, m23), (m30, m31, m32, m33) = [[F(v) for v in col] for col in m]
    c00 = m22 * m33 - m32 * m23; c02 = m12 * m33 - m32 * m13; c03 = m12 * m23 - m22 * m13
    c04 = m21 * m33 - m31 * m23; c06 = m11 * m33 - m31 * m13; c07 = m11 * m23 - m21 * m13
    c08 = m21 * m32 - m31 * m22; c10 = m11 * m32 - m31 * m12; c11 = m11 * m22 - m21 * m12
    c12 = m20 * m33 - m30 * m23; c14 = m10 * m33 - m30 * m13; c15 = m10 * m23 - m20 * m13
    c16 = m20 * m32 - m30 * m22; c18 = m10 * m32 - m30 * m12; c19 = m10 * m22 - m20 * m12
    c20 = m20 * m31 - m30 * m21; c22 = m10 * m31 - m30 * m11; c23 = m10 * m21 - m20 * m11
    v = lambda *a: np.array(a, F)
    fac0, fac1, fac2 = v(c00, c00, c02, c03), v(c04, c04, c06, c07), v(c08, c08, c10, c11)
    fac3, fac4, fac5 = v(c12, c12, c14, c15), v(c16, c16, c18, c19), v(c20, c20, c22, c23)
    vec0, vec1, vec2, vec3 = v(m10, m00, m00, m00), v(m11, m01, m01, m01), v(m12, m02, m02, m02), v(m13, m03, m03, m03)
    inv0 = (vec1 * fac0 - vec2 * fac1) + vec3 * fac2
    inv1 = (vec0 * fac0 - vec2 * fac3) + vec3 * fac4
    inv2 = (vec0 * fac1 - vec1 * fac3) + vec3 * fac5
    inv3 = (vec0 * fac2 - vec1 * fac4) + vec2 * fac5
    sign_a, sign_b = v(1, -1, 1, -1), v(-1, 1, -1, 1)
    inv = [inv0 * sign_a, inv1 * sign_b, inv2 * sign_a, inv3 * sign_b]
    col0 = v(inv[0][0], inv[1][0], inv[2][0], inv[3][0])
    dot0 = v(m00, m01, m02, m03) * col0
    dot1 = ((dot0[0] + dot0[1]) + dot0[2]) + dot0[3]
    rcp = F(1.0) / dot1
    return np.array([c * rcp for c in inv], F)


def _glam_any_orthonormal_pair(n):   # glam-0.24.2/src/f32/vec3.rs `any_orthonormal_pair`
    x, y, z = [F(c) for c in n]
    sign = F(np.copysign(F(1.0), z))
    a = F(-1.0) / (sign + z)
    b = x * y * a
    return np.array([F(1.0) + sign * x * x * a, sign * b, -sign * x, b, sign + y * y * a, -y], F)


def _glam_project_point3(m, p):   # glam-0.24.2/src/f32/scalar/mat4.rs `project_point3`
    cols = [np.array(c, F) for c in m]
    res = cols[0] * F(p[0])
    res = cols[1] * F(p[1]) + res
    res = cols[2] * F(p[2]) + res
    res = cols[3] + res
    res = res * (F(1.0) / res[3])
    return res[:3]


def _cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], F)


def _dot3(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _glam_affine_inverse(a12):   # Mat3A::inverse + Affine3A::inverse (glam-0.24.2/src/f32/affine3a.rs, scalar Mat3A)
    x, y, z, t = [np.array(a12[3 * i:3 * i + 3], F) for i in range(4)]
    tmp0, tmp1, tmp2 = _cross(y, z), _cross(z, x), _cross(x, y)
    inv_det = F(1.0) / _dot3(z, tmp2)
    c0, c1, c2 = tmp0 * inv_det, tmp1 * inv_det, tmp2 * inv_det
    rx, ry, rz = np.array([c0[0], c1[0], c2[0]], F), np.array([c0[1], c1[1], c2[1]], F), np.array([c0[2], c1[2], c2[2]], F)
    m = rx * t[0]; m = m + ry * t[1]; m = m + rz * t[2]
    return np.concatenate([rx, ry, rz, -m]).astype(F)


def _probe_glam(op, values, n_out):
    lib = oracle_lib()
    lib.or_probe_glam.restype = None
    lib.or_probe_glam.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    src = np.ascontiguousarray(values, F); out = np.zeros(n_out, F)
    lib.or_probe_glam(op, src.ctypes.data, out.ctypes.data)
    return out


def test_glam_restatements_agree_bitwise_with_a_numpy_float32_restatement():
    rng = np.random.default_rng(3)
    with np.errstate(all="ignore"):
        for trial in range(300):
            m = rng.normal(size=(4, 4)).astype(F)
            if trial % 3 == 0:   # camera-like matrices: rigid transform, and an infinite-reverse perspective
                m = look_at_transform(rng.uniform(-3, 3, 3), rng.uniform(-1, 1, 3)).T.copy() if trial % 2 else perspective_infinite_reverse_rh(0.7 + trial * 1e-3, 16 / 9, 0.1).T.copy()
            got = _probe_glam(0, m.reshape(-1), 16).reshape(4, 4)
            assert np.array_equal(got.view(np.uint32), _glam_mat4_inverse(m).view(np.uint32)), ("Mat4::inverse", trial)
            n = rng.normal(size=3); n = (n / np.linalg.norm(n)).astype(F)
            if trial % 50 == 0:
                n = np.array([0.0, 0.0, -1.0 if trial % 100 else 1.0], F)
            assert np.array_equal(_probe_glam(1, n, 6).view(np.uint32), _glam_any_orthonormal_pair(n).view(np.uint32)), ("any_orthonormal_pair", trial)
            p = rng.normal(size=3).astype(F)
            assert np.array_equal(_probe_glam(2, np.concatenate([m.reshape(-1), p]), 3).view(np.uint32), _glam_project_point3(m, p).view(np.uint32)), ("project_point3", trial)
            a = rng.normal(size=12).astype(F)
            assert np.array_equal(_probe_glam(3, a, 12).view(np.uint32), _glam_affine_inverse(a).view(np.uint32)), ("Affine3A::inverse", trial)
