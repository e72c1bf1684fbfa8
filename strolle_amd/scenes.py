"""Benchmark scenes of the reference's examples, rebuilt through the Engine API.

cornell: bevy-strolle/examples/cornell.rs (scene.gltf + one point light, sun below the horizon).
dungeon: bevy-strolle/examples/demo.rs (level.glb + six point lights; materials forced to
reflectance 0 / perceptual_roughness 1 by `adjust_materials`, demo.rs:247-260).

Inputs pinned here because Bevy is not available (stated in DESIGN.md): glTF roughnessFactor
default 1.0 -> perceptual_roughness, StandardMaterial reflectance default 0.5, PointLight
range default 20.0, PerspectiveProjection fov pi/4 near 0.1 (infinite reverse-Z).
"""
import math
import os

import numpy as np

from .api import Camera, CameraMode, Instance, Light, Material, Mesh, Sun, look_at_transform, perspective_infinite_reverse_rh

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def load_blue_noise() -> np.ndarray:
    return np.load(os.path.join(ASSETS, "blue_noise.npy"))


def _subdivide(pos, nrm, uv, levels: int):
    """Midpoint subdivision, 4 triangles per level: the surface, normals and texture mapping are unchanged, only the
    triangle count (and with it the BVH) grows. float32 midpoints, deterministic."""
    for _ in range(levels):
        out = []
        for a in (pos, nrm, uv):
            v0, v1, v2 = a[:, 0], a[:, 1], a[:, 2]
            m01, m12, m20 = (v0 + v1) * np.float32(0.5), (v1 + v2) * np.float32(0.5), (v2 + v0) * np.float32(0.5)
            out.append(np.concatenate([np.stack([v0, m01, m20], 1), np.stack([m01, v1, m12], 1), np.stack([m20, m12, v2], 1), np.stack([m01, m12, m20], 1)], 0))
        pos, nrm, uv = out
    return pos, nrm, uv


def _insert_gltf(engine, npz, material_overrides=None, first_handle=1, subdivide: int = 0, subdivide_meshes=None):
    n_mat = len(npz["material_metallic"])
    n_img = int(npz["n_images"]) if "n_images" in npz else 0
    for i in range(n_img):
        engine.insert_image(1000 + i, npz[f"image_{i}"], srgb=True)
    for i in range(n_mat):
        img = int(npz["material_image"][i])
        kw = dict(
            base_color=npz["material_base_color"][i].tolist(),
            emissive=npz["material_emissive"][i].tolist() + [1.0],
            perceptual_roughness=float(npz["material_perceptual_roughness"][i]),
            metallic=float(npz["material_metallic"][i]),
            reflectance=0.5,
            alpha_mode=int(npz["material_alpha_mode"][i]),
            base_color_texture=(1000 + img) if img >= 0 else None,
        )
        if material_overrides:
            kw.update(material_overrides)
        engine.insert_material(first_handle + i, Material(**kw))
    n = int(npz["n_meshes"])
    for i in range(n):
        pos, nrm, uv = npz[f"positions_{i}"], npz[f"normals_{i}"], npz[f"uvs_{i}"]
        if subdivide and (subdivide_meshes is None or i in subdivide_meshes):
            n_tri = len(np.asarray(pos).reshape(-1, 3, 3))
            pos, nrm, uv = _subdivide(np.asarray(pos, np.float32).reshape(n_tri, 3, 3), np.asarray(nrm, np.float32).reshape(n_tri, 3, 3),
                                      np.asarray(uv, np.float32).reshape(n_tri, 3, 2), subdivide)
        engine.insert_mesh(first_handle + i, Mesh(pos, nrm, uv))
        x = npz[f"xform_{i}"].reshape(4, 3).T  # [x_axis y_axis z_axis t] columns -> 3x4
        engine.insert_instance(first_handle + i, Instance(first_handle + i, first_handle + int(npz[f"material_{i}"]), x))
    return n


def camera_for(size, eye, target, mode=CameraMode.IMAGE, denoise=True, depth=0) -> Camera:
    w, h = size
    return Camera(mode=mode, denoise=denoise, depth=depth, size=(w, h),
                  transform=look_at_transform(eye, target),
                  projection=perspective_infinite_reverse_rh(math.pi / 4.0, w / h, 0.1))


def build_cornell(engine, t: float = 0.0):
    """cornell.rs:38-93 at time t (the point light orbits; the benchmark fixes t = 0)."""
    npz = np.load(os.path.join(ASSETS, "cornell.npz"))
    engine.set_blue_noise(load_blue_noise())
    _insert_gltf(engine, npz)
    intensity = 50.0 / (4.0 * math.pi)  # extract.rs:285-297
    engine.insert_light(1, Light.point((math.sin(t) / 2.0, 1.5, math.cos(t) / 2.0), 0.15, (intensity,) * 3, 20.0))
    engine.update_sun(Sun(azimuth=0.0, altitude=-1.0))  # cornell.rs:87


def cornell_camera(size, mode=CameraMode.IMAGE, denoise=True, depth=0) -> Camera:
    return camera_for(size, (0.0, 1.0, 3.2), (0.0, 1.0, 0.0), mode, denoise, depth)  # cornell.rs:76-78


def bevy_torus(radius: float = 1.0, ring_radius: float = 0.5, subdivisions_segments: int = 32, subdivisions_sides: int = 24):
    """Bevy 0.12 `Mesh::from(shape::Torus)` (bevy_render/src/mesh/shape/torus.rs; the crate is not under /root/reference, the
    tessellation is restated from its published source): a (segments+1) x (sides+1) vertex grid — position
    (cos t (R + r cos p), r sin p, sin t (R + r cos p)), normal = normalize(position - ring centre), uv = (segment / segments,
    side / sides) — and two triangles per quad: (lt, rt, lb), (rt, rb, lb). Defaults: R 1, r 0.5, 32 x 24 = 1,536 triangles.
    Returns (positions, normals, uvs) as [n, 3, *] float32 arrays in the order bevy-strolle's mesh stage builds them
    (prepare.rs:85-113: one MeshTriangle per index triple)."""
    f32 = np.float32
    seg_stride = f32(2.0) * f32(math.pi) / f32(subdivisions_segments)
    side_stride = f32(2.0) * f32(math.pi) / f32(subdivisions_sides)
    pos, nrm, uv = [], [], []
    for segment in range(subdivisions_segments + 1):
        theta = f32(seg_stride * f32(segment))
        ct, st = f32(np.cos(theta, dtype=f32)), f32(np.sin(theta, dtype=f32))
        for side in range(subdivisions_sides + 1):
            phi = f32(side_stride * f32(side))
            cp, sp = f32(np.cos(phi, dtype=f32)), f32(np.sin(phi, dtype=f32))
            ring = f32(f32(radius) + f32(f32(ring_radius) * cp))
            p = np.array([ct * ring, f32(ring_radius) * sp, st * ring], f32)
            c = np.array([f32(radius) * ct, 0.0, f32(radius) * st], f32)
            d = (p - c).astype(f32)
            n = (d * (f32(1.0) / f32(np.sqrt(np.dot(d, d).astype(f32))))).astype(f32)   # glam Vec3::normalize = v * (1 / length)
            pos.append(p); nrm.append(n)
            uv.append([f32(segment) / f32(subdivisions_segments), f32(side) / f32(subdivisions_sides)])
    pos, nrm, uv = np.array(pos, f32), np.array(nrm, f32), np.array(uv, f32)
    per_row = subdivisions_sides + 1
    idx = []
    for segment in range(subdivisions_segments):
        for side in range(subdivisions_sides):
            lt = side + segment * per_row; rt = side + 1 + segment * per_row
            lb = side + (segment + 1) * per_row; rb = side + 1 + (segment + 1) * per_row
            idx += [[lt, rt, lb], [rt, rb, lb]]
    idx = np.array(idx)
    return pos[idx], nrm[idx], uv[idx]


def _srgb_to_linear(c: float) -> float:
    """Bevy Color::as_linear_rgba_f32 per channel (bevy_render color: x <= 0.04045 ? x / 12.92 : ((x + 0.055) / 1.055)^2.4)."""
    return c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4


DUNGEON_TORI = [(-0.5, 0.33, -5.5), (-11.0, 0.33, 28.0), (-11.5, 0.33, 13.5)]   # demo.rs:195-199
DUNGEON_DESCRIPTION = "dungeon: level.glb (8,393 triangles, 45 textured materials) + the demo's three emissive Bevy tori (3 x 1,536 triangles) = 13,001 triangles, 7 light slots"


def build_dungeon(engine, subdivide: int = 0, tori: bool = True, tori_subdivide=None, subdivide_meshes=None, copies: int = 1):
    """demo.rs:155-218: level.glb, six point lights, the three emissive tori (`shape::Torus::default()`, scale 0.5, rotated
    1 rad about Z; material base colour sRGB (0.9, 0.6, 0.3), emissive 10 x that, then — like every material of the scene —
    reflectance 0 and perceptual roughness 1, demo.rs:254-258), sun below the horizon. The spot light has intensity 0 and
    is dropped by the extract stage (extract.rs:307-311).
    subdivide = k splits every triangle into 4^k (SYNTHETIC: k = 2 gives the ~208 k-triangle variant that stands in for
    BASELINE.json's "~100k tris" dungeon; the demo scene itself has 13,001)."""
    npz = np.load(os.path.join(ASSETS, "dungeon.npz"))
    engine.set_blue_noise(load_blue_noise())
    # (the subdivided variants are 25 / 26 internal nodes deep: since round 5 the launches that walk the contract stream take a stack as deep as
    # the tree needs, up to 32 entries — StTuning::allow_deep_bvh is no longer set here, no push is dropped)
    n = _insert_gltf(engine, npz, material_overrides=dict(reflectance=0.0, perceptual_roughness=1.0), subdivide=subdivide, subdivide_meshes=subdivide_meshes)   # (subdivide_meshes: measurement scenes in which only some of the level's 45 meshes are split)
    for c in range(1, copies):   # (measurement scenes: the level instanced again on a 4-wide grid of 50 x 100 m cells — a LARGE scene that is not a subdivided small one)
        for i in range(n):
            x = npz[f"xform_{i}"].reshape(4, 3).T.copy()
            x[0, 3] += 50.0 * (c % 4); x[2, 3] += 100.0 * (c // 4)
            engine.insert_instance(20000 + 100 * c + i, Instance(1 + i, 1 + int(npz[f"material_{i}"]), x))
    if tori:
        pos, nrm, uv = bevy_torus()
        if subdivide if tori_subdivide is None else tori_subdivide:   # (tori_subdivide: measurement scenes whose tori are split more or less often than the level)
            pos, nrm, uv = _subdivide(pos, nrm, uv, subdivide if tori_subdivide is None else tori_subdivide)
        mesh_handle, first = 5000, 5001
        engine.insert_mesh(mesh_handle, Mesh(pos, nrm, uv))
        srgb = (0.9, 0.6, 0.3)
        base = [_srgb_to_linear(c) for c in srgb] + [1.0]
        emissive = [_srgb_to_linear(c * 10.0) for c in srgb] + [1.0]
        c1, s1 = math.cos(1.0), math.sin(1.0)
        for i, t in enumerate(DUNGEON_TORI):
            engine.insert_material(first + i, Material(base_color=base, emissive=emissive, perceptual_roughness=1.0, metallic=0.0, reflectance=0.0))
            # Transform::from_translation(t).with_rotation(Quat::from_rotation_z(1.0)).with_scale(0.5): columns of R_z(1) * 0.5, then t
            x = np.array([[0.5 * c1, -0.5 * s1, 0.0, t[0]], [0.5 * s1, 0.5 * c1, 0.0, t[1]], [0.0, 0.0, 0.5, t[2]]], np.float32)
            engine.insert_instance(first + i, Instance(mesh_handle, first + i, x))
    intensity = 5000.0 / (4.0 * math.pi)
    lights = [(-3.0, 0.75, -23.0), (-23.5, 0.75, -31.0), (1.25, 0.75, -10.5), (-3.15, 0.75, 1.25), (-3.25, 0.75, 20.25), (13.25, 0.75, -28.25)]
    for i, p in enumerate(lights):
        engine.insert_light(1 + i, Light.point(p, 0.15, (intensity,) * 3, 35.0))
    engine.update_sun(Sun(azimuth=0.0, altitude=-1.0))


def dungeon_camera(size, mode=CameraMode.IMAGE, denoise=True, depth=0) -> Camera:
    return camera_for(size, (-5.75, 0.5, -16.8), (-5.75, 0.5, -17.0), mode, denoise, depth)  # demo.rs:150-151


def build_random_soup(engine, n_triangles: int, seed: int = 0, n_lights: int = 3, blend_fraction: float = 0.0):
    """Synthetic stress scene for parity tests: a random triangle soup in [-1,1]^3 with a few materials/lights.
    blend_fraction > 0 turns that share of the materials into AlphaMode::Blend ones with an RGBA texture whose alpha is
    0, 0.5 or 1 per texel (traversal's alpha test, ray.rs:184-214); n_lights > 16 exercises the 16-pick RIS over a longer
    light table (reservoir/ephemeral.rs:14-55)."""
    rng = np.random.default_rng(seed)
    engine.set_blue_noise(load_blue_noise())
    n_mat = 4
    n_blend = int(round(blend_fraction * n_mat))
    if n_blend:
        tex = rng.integers(0, 256, (32, 32, 4), dtype=np.uint8)
        tex[..., 3] = rng.choice(np.array([0, 128, 255], np.uint8), (32, 32))
        engine.insert_image(900, tex, srgb=True)
        engine.insert_image(901, rng.integers(0, 256, (16, 24, 4), dtype=np.uint8), srgb=False)   # metallic-roughness map
        engine.insert_image(902, rng.integers(0, 256, (8, 8, 4), dtype=np.uint8), srgb=True)      # emissive map
    for i in range(n_mat):
        blend = i < n_blend
        engine.insert_material(1 + i, Material(base_color=rng.uniform(0.2, 0.9, 3).tolist() + [0.9 if blend else 1.0], perceptual_roughness=float(rng.uniform(0.2, 1.0)),
                                               metallic=float(rng.uniform(0.0, 0.8)) if i % 2 else 0.0,
                                               emissive=(rng.uniform(0, 0.5, 3).tolist() + [1.0]) if i == 3 else (0, 0, 0, 0),
                                               alpha_mode=1 if blend else 0, base_color_texture=900 if blend else None,
                                               metallic_roughness_texture=901 if (n_blend and i == 2) else None,
                                               emissive_texture=902 if (n_blend and i == 3) else None))
    per = max(1, n_triangles // n_mat)
    for i in range(n_mat):
        c = rng.uniform(-1, 1, (per, 1, 3)).astype(np.float32)
        pos = c + rng.uniform(-0.15, 0.15, (per, 3, 3)).astype(np.float32)
        e1 = pos[:, 1] - pos[:, 0]; e2 = pos[:, 2] - pos[:, 0]
        nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
        nrm = np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32)
        uv = rng.uniform(0, 1, (per, 3, 2)).astype(np.float32)
        engine.insert_mesh(1 + i, Mesh(pos, nrm, uv))
        ang = float(rng.uniform(0, 1))
        rot = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]], np.float32)
        x = np.concatenate([rot * np.float32(1.0 + 0.1 * i), rng.uniform(-0.1, 0.1, (3, 1)).astype(np.float32)], axis=1)
        engine.insert_instance(1 + i, Instance(1 + i, 1 + i, x))
    for i in range(n_lights):
        engine.insert_light(1 + i, Light.point(rng.uniform(-1.5, 1.5, 3).tolist(), 0.1, rng.uniform(0.5, 2.0, 3).tolist(), 20.0))
    engine.update_sun(Sun(azimuth=0.0, altitude=-1.0))


def build_sliver_bundle(engine, n_triangles: int = 20000, seed: int = 5):
    """ADVERSARIAL for a traversal stack (tests of the wide walk's overflow report): n long slivers, each from one corner of a thin box
    (2 x 0.04 x 0.04) to the opposite one, so that every triangle's bounding box is nearly the whole bundle. A ray along the bundle's axis meets
    (almost) every node's every child box and hardly any triangle: the 4-wide walk has to keep three siblings pending per level —
    27 entries at 20,000 slivers (host model: tests/test_wide_bvh.py), more than the reference's 24 (strolle-gpu/src/lib.rs:76) — while the
    binary contract tree stays 26 internal nodes deep (within the 32 its walks can hold)."""
    rng = np.random.default_rng(seed)
    engine.set_blue_noise(load_blue_noise())
    engine.insert_material(1, Material(base_color=(0.8, 0.8, 0.8, 1.0)))
    n = n_triangles
    j = lambda s: rng.uniform(-s, s, n)
    pos = np.zeros((n, 3, 3), np.float32)
    pos[:, 0] = np.stack([-1.0 + j(0.3), -0.02 + j(0.004), -0.02 + j(0.004)], 1)
    pos[:, 1] = np.stack([1.0 + j(0.3), 0.02 + j(0.004), 0.02 + j(0.004)], 1)
    pos[:, 2] = pos[:, 0] + np.stack([j(0.05), j(0.0005), j(0.0005)], 1)
    nrm = np.zeros_like(pos); nrm[..., 2] = 1
    engine.insert_mesh(1, Mesh(pos, nrm))
    engine.insert_instance(1, Instance(1, 1, np.eye(4, dtype=np.float32)[:3]))
    engine.insert_light(1, Light.point((-3.0, 0.5, 0.5), 0.1, (5.0, 5.0, 5.0), 20.0))
    engine.update_sun(Sun(azimuth=0.0, altitude=-1.0))


def sliver_bundle_camera(size, mode=CameraMode.REFERENCE, depth=0) -> Camera:
    """looks down the bundle's axis from outside it (narrow view: the bundle fills the frame's centre)"""
    w, h = size
    return Camera(mode=mode, denoise=True, depth=depth, size=(w, h), transform=look_at_transform((-3.0, 0.012, -0.012), (0.0, 0.012, -0.012)),
                  projection=perspective_infinite_reverse_rh(math.pi / 60.0, w / h, 0.1))
