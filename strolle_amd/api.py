"""Python host-side mirror of `strolle::Engine` over the C ABI of libstrolle_hip.so.

Names, argument meaning and error behaviour follow the reference's public API
(strolle/src/lib.rs:105-409, camera.rs, light.rs, material.rs, instance.rs,
mesh_triangle.rs, sun.rs) so tests read like reference usage. This module is a
thin ctypes binding: all engine logic (scene stores, BVH build, pass graph) is
C++ inside the shared library, all per-pixel work is HIP. There is no CPU
fallback: if the library or a GPU is missing, calls raise.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import enum
import math
import os
import sys
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STROLLE_HIP_LIB") or os.path.join(_HERE, "csrc", "libstrolle_hip.so")   # STROLLE_HIP_LIB: another build of the library (same-box A/B of two builds)
if os.environ.get("STROLLE_HIP_LIB"):  # experiments: an alternative build of the same sources (never a different implementation)
    LIB_PATH = os.environ["STROLLE_HIP_LIB"]


# ----------------------------------------------------------------------------- C structs (include/strolle_hip.h)
class StMeshTriangle(C.Structure):
    _fields_ = [("positions", C.c_float * 9), ("normals", C.c_float * 9), ("uvs", C.c_float * 6), ("tangents", C.c_float * 12)]


class StMaterial(C.Structure):
    _fields_ = [
        ("base_color", C.c_float * 4), ("emissive", C.c_float * 4),
        ("perceptual_roughness", C.c_float), ("metallic", C.c_float), ("reflectance", C.c_float), ("ior", C.c_float),
        ("base_color_texture", C.c_uint64), ("emissive_texture", C.c_uint64),
        ("metallic_roughness_texture", C.c_uint64), ("normal_map_texture", C.c_uint64),
        ("alpha_mode", C.c_uint32), ("_pad", C.c_uint32),
    ]


class StLight(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("position", C.c_float * 3), ("radius", C.c_float), ("color", C.c_float * 3),
                ("range", C.c_float), ("direction", C.c_float * 3), ("angle", C.c_float)]


class StCamera(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("denoise", C.c_uint32), ("depth", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("pos_x", C.c_uint32), ("pos_y", C.c_uint32), ("_pad", C.c_uint32), ("transform", C.c_float * 16), ("projection", C.c_float * 16)]


class StDistRect(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]

    def as_tuple(self):
        return (self.x0, self.y0, self.x1, self.y1)


class StDistGrid(C.Structure):
    """include/strolle_hip.h StDistGrid: a (cost-weighted) tile grid every rank holds identically."""
    _fields_ = [("cols", C.c_uint32), ("rows", C.c_uint32), ("row_edge", C.c_uint32 * 17), ("col_edge", (C.c_uint32 * 17) * 16)]

    def tiles(self):
        return [dist_grid_tile(self, r) for r in range(self.cols * self.rows)]

    def describe(self):
        return {"cols": self.cols, "rows": self.rows, "row_edges": list(self.row_edge[:self.rows + 1]),
                "col_edges": [list(self.col_edge[k][:self.cols + 1]) for k in range(self.rows)]}


class StDistUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class StTuning(C.Structure):
    """include/strolle_hip.h StTuning: scheduling / tuning switches of one engine."""
    _fields_ = [(n, C.c_uint32) for n in ("struct_size", "overlap", "fuse", "fuse_di_head", "fuse_spatial", "fuse_gi_sampling", "fuse_gi_validation",
                                          "fuse_gi_reprojection", "fuse_wavelet", "fuse_compose", "preview_both", "variance_in_reproject",
                                          "lean_frame", "skip_scratch_stores", "di_head_on_main", "alias_gi_history", "tile_map", "tile_map_denoise")] + \
               [("side_priority", C.c_int32)] + \
               [(n, C.c_uint32) for n in ("staging", "double_buffer", "packed_base", "tick_timing", "anyhit_fast", "compact_bvh",
                                          "allow_deep_bvh", "device_bake", "wide_bvh", "wide_stack_entries", "primary_packets")]


class StKernelProfile(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint32), ("total_ms", C.c_float), ("algorithmic_bytes", C.c_double), ("traversal_bytes", C.c_double)]


class StGltfOptions(C.Structure):
    _fields_ = [("first_handle", C.c_uint64), ("first_image_handle", C.c_uint64), ("override_mask", C.c_uint32), ("reflectance", C.c_float),
                ("perceptual_roughness", C.c_float), ("subdivide", C.c_uint32), ("light_radius", C.c_float), ("_pad", C.c_uint32)]


class StGltfSummary(C.Structure):
    _fields_ = [("meshes", C.c_uint32), ("triangles", C.c_uint32), ("materials", C.c_uint32), ("images", C.c_uint32), ("images_dropped", C.c_uint32),
                ("primitives_skipped", C.c_uint32), ("lights", C.c_uint32), ("lights_skipped", C.c_uint32)]


assert C.sizeof(StMeshTriangle) == 144 and C.sizeof(StMaterial) == 88 and C.sizeof(StLight) == 52 and C.sizeof(StCamera) == 160


class StrolleError(RuntimeError):
    pass


class OutputFormat(enum.IntEnum):
    """StOutputFormat"""
    RGBA32F = 0
    RGBA16F = 1
    RGBA8_UNORM_SRGB = 2
    BGRA8_UNORM_SRGB = 3


class Buffer(enum.IntEnum):
    """Per-camera buffers (strolle/src/camera_controller/buffers.rs:7-51); values == StBufferId."""
    PRIM_GBUFFER_D0_A = 0; PRIM_GBUFFER_D0_B = 1; PRIM_GBUFFER_D1_A = 2; PRIM_GBUFFER_D1_B = 3
    PRIM_SURFACE_MAP_A = 4; PRIM_SURFACE_MAP_B = 5; REPROJECTION_MAP = 6; VELOCITY_MAP = 7
    DI_RESERVOIRS_0 = 8; DI_RESERVOIRS_1 = 9; DI_RESERVOIRS_2 = 10
    DI_DIFF_SAMPLES = 11; DI_DIFF_PREV_COLORS = 12; DI_DIFF_CURR_COLORS = 13
    DI_DIFF_MOMENTS_A = 14; DI_DIFF_MOMENTS_B = 15; DI_DIFF_STASH = 16; DI_SPEC_SAMPLES = 17
    GI_D0 = 18; GI_D1 = 19; GI_D2 = 20
    GI_RESERVOIRS_0 = 21; GI_RESERVOIRS_1 = 22; GI_RESERVOIRS_2 = 23; GI_RESERVOIRS_3 = 24
    GI_DIFF_SAMPLES = 25; GI_DIFF_PREV_COLORS = 26; GI_DIFF_CURR_COLORS = 27
    GI_DIFF_MOMENTS_A = 28; GI_DIFF_MOMENTS_B = 29; GI_DIFF_STASH = 30; GI_SPEC_SAMPLES = 31
    REF_HITS = 32; REF_RAYS = 33; REF_COLORS = 34; DBG_USED_MEMORY = 35


class PassBit(enum.IntFlag):
    """StPassBit: one bit per reference pass (camera_controller.rs:87-174 order)."""
    PRIM_VISIBILITY = 1 << 0; FRAME_REPROJECTION = 1 << 1
    DI_SAMPLING = 1 << 2; DI_TEMPORAL = 1 << 3; DI_SPATIAL_PICK = 1 << 4; DI_SPATIAL_TRACE = 1 << 5; DI_SPATIAL_SAMPLE = 1 << 6; DI_RESOLVING = 1 << 7
    GI_REPROJECTION = 1 << 8; GI_SAMPLING_A = 1 << 9; GI_SAMPLING_B = 1 << 10; GI_TEMPORAL = 1 << 11
    GI_SPATIAL_PICK = 1 << 12; GI_SPATIAL_TRACE = 1 << 13; GI_SPATIAL_SAMPLE = 1 << 14
    GI_PREVIEW_0 = 1 << 15; GI_PREVIEW_1 = 1 << 16; GI_RESOLVING = 1 << 17
    DENOISE_REPROJECT_DI = 1 << 18; DENOISE_REPROJECT_GI = 1 << 19; DENOISE_VARIANCE = 1 << 20
    DENOISE_WAVELET_0 = 1 << 21; DENOISE_WAVELET_1 = 1 << 22; DENOISE_WAVELET_2 = 1 << 23; DENOISE_WAVELET_3 = 1 << 24; DENOISE_WAVELET_4 = 1 << 25
    COMPOSITION = 1 << 26; BVH_HEATMAP = 1 << 27; REF_TRACING = 1 << 28; REF_SHADING = 1 << 29


# ----------------------------------------------------------------------------- value types mirroring the reference
class CameraMode(enum.IntEnum):
    """strolle/src/camera.rs:83-105"""
    IMAGE = 0; DI_DIFFUSE = 1; DI_SPECULAR = 2; GI_DIFFUSE = 3; GI_SPECULAR = 4; BVH_HEATMAP = 5; REFERENCE = 6


@dataclasses.dataclass
class Camera:
    """strolle/src/camera.rs:8-14. `transform`/`projection` are 4x4 numpy arrays in maths layout
    (m[row, col]); they are handed over column-major like glam's Mat4."""
    mode: CameraMode = CameraMode.IMAGE
    denoise: bool = True
    depth: int = 0
    size: tuple = (512, 512)
    position: tuple = (0, 0)
    transform: np.ndarray = dataclasses.field(default_factory=lambda: np.eye(4, dtype=np.float32))
    projection: np.ndarray = dataclasses.field(default_factory=lambda: np.eye(4, dtype=np.float32))

    def to_c(self) -> StCamera:
        c = StCamera()
        c.mode = int(self.mode); c.denoise = 1 if self.denoise else 0; c.depth = int(self.depth)
        c.width, c.height = int(self.size[0]), int(self.size[1])
        c.pos_x, c.pos_y = int(self.position[0]), int(self.position[1])
        c.transform[:] = np.asarray(self.transform, dtype=np.float32).T.reshape(-1).tolist()
        c.projection[:] = np.asarray(self.projection, dtype=np.float32).T.reshape(-1).tolist()
        return c


@dataclasses.dataclass
class Material:
    """strolle/src/material.rs:8-70 (defaults :53-70)."""
    base_color: Sequence[float] = (1.0, 1.0, 1.0, 1.0)
    base_color_texture: Optional[int] = None
    emissive: Sequence[float] = (0.0, 0.0, 0.0, 0.0)
    emissive_texture: Optional[int] = None
    perceptual_roughness: float = 0.5
    metallic: float = 0.0
    metallic_roughness_texture: Optional[int] = None
    reflectance: float = 0.5
    ior: float = 1.0
    normal_map_texture: Optional[int] = None
    alpha_mode: int = 0  # 0 Opaque, 1 Blend

    def to_c(self) -> StMaterial:
        m = StMaterial()
        m.base_color[:] = [float(x) for x in self.base_color]
        m.emissive[:] = [float(x) for x in self.emissive]
        m.perceptual_roughness = self.perceptual_roughness; m.metallic = self.metallic
        m.reflectance = self.reflectance; m.ior = self.ior
        m.base_color_texture = self.base_color_texture or 0
        m.emissive_texture = self.emissive_texture or 0
        m.metallic_roughness_texture = self.metallic_roughness_texture or 0
        m.normal_map_texture = self.normal_map_texture or 0
        m.alpha_mode = self.alpha_mode
        return m


@dataclasses.dataclass
class Light:
    """strolle/src/light.rs:6-22"""
    kind: int
    position: Sequence[float]
    radius: float
    color: Sequence[float]
    range: float
    direction: Sequence[float] = (0.0, 0.0, 0.0)
    angle: float = 0.0

    @staticmethod
    def point(position, radius, color, range) -> "Light":
        return Light(0, position, radius, color, range)

    @staticmethod
    def spot(position, radius, color, range, direction, angle) -> "Light":
        return Light(1, position, radius, color, range, direction, angle)

    def to_c(self) -> StLight:
        l = StLight()
        l.kind = self.kind; l.position[:] = [float(x) for x in self.position]; l.radius = self.radius
        l.color[:] = [float(x) for x in self.color]; l.range = self.range
        l.direction[:] = [float(x) for x in self.direction]; l.angle = self.angle
        return l


@dataclasses.dataclass
class Sun:
    """strolle/src/sun.rs:1-14"""
    azimuth: float = 0.0
    altitude: float = 0.35


class Mesh:
    """strolle/src/mesh.rs + mesh_triangle.rs: object-space triangles.
    positions/normals: [n,3,3] float32, uvs: [n,3,2], tangents: [n,3,4] (optional)."""

    def __init__(self, positions, normals, uvs=None, tangents=None):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3, 3)
        n = len(self.positions)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(n, 3, 3)
        self.uvs = np.zeros((n, 3, 2), np.float32) if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(n, 3, 2)
        self.tangents = np.zeros((n, 3, 4), np.float32) if tangents is None else np.ascontiguousarray(tangents, dtype=np.float32).reshape(n, 3, 4)

    def pack(self) -> np.ndarray:
        n = len(self.positions)
        out = np.concatenate([self.positions.reshape(n, 9), self.normals.reshape(n, 9), self.uvs.reshape(n, 6), self.tangents.reshape(n, 12)], axis=1)
        return np.ascontiguousarray(out, dtype=np.float32)


@dataclasses.dataclass
class Instance:
    """strolle/src/instance.rs:16-31. transform: glam Affine3A as a 3x4 numpy array [R | t]."""
    mesh_handle: int
    material_handle: int
    transform: np.ndarray

    def xform12(self):
        t = np.asarray(self.transform, dtype=np.float32).reshape(3, 4)
        return np.concatenate([t[:, 0], t[:, 1], t[:, 2], t[:, 3]]).astype(np.float32)


# ----------------------------------------------------------------------------- camera matrices (what Bevy hands to the engine)
def perspective_infinite_reverse_rh(fov_y: float, aspect: float, z_near: float) -> np.ndarray:
    """glam Mat4::perspective_infinite_reverse_rh — Bevy's PerspectiveProjection (default fov pi/4, near 0.1)."""
    f = np.float32(1.0 / math.tan(0.5 * fov_y))
    m = np.zeros((4, 4), np.float32)
    m[0, 0] = np.float32(f / np.float32(aspect)); m[1, 1] = f; m[3, 2] = -1.0; m[2, 3] = np.float32(z_near)
    return m


def look_at_transform(eye, target, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """World transform of a camera at `eye` looking at `target` (Bevy Transform::looking_at), 4x4."""
    eye = np.asarray(eye, np.float64); target = np.asarray(target, np.float64); up = np.asarray(up, np.float64)
    back = eye - target; back /= np.linalg.norm(back)
    right = np.cross(up, back); right /= np.linalg.norm(right)
    upv = np.cross(back, right)
    m = np.eye(4)
    m[:3, 0] = right; m[:3, 1] = upv; m[:3, 2] = back; m[:3, 3] = eye
    return m.astype(np.float32)


# ----------------------------------------------------------------------------- binding
class _Binding:
    """Function table over a C library exporting `<prefix>engine_create` etc."""

    def __init__(self, lib: C.CDLL, prefix: str, has_device: bool):
        self.lib, self.prefix, self.has_device = lib, prefix, has_device
        u64, vp, i32, u32, sz = C.c_uint64, C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
        P = C.POINTER

        def fn(name, args):
            f = getattr(lib, prefix + name)
            f.restype = i32
            f.argtypes = args
            return f

        self.engine_create = fn("engine_create", [i32, P(vp)] if has_device else [P(vp)])
        self.engine_destroy = getattr(lib, prefix + "engine_destroy"); self.engine_destroy.restype = None; self.engine_destroy.argtypes = [vp]
        self.mesh_insert = fn("mesh_insert", [vp, u64, vp, sz]); self.mesh_remove = fn("mesh_remove", [vp, u64])
        self.material_insert = fn("material_insert", [vp, u64, P(StMaterial)]); self.material_has = fn("material_has", [vp, u64])
        self.material_remove = fn("material_remove", [vp, u64])
        self.instance_insert = fn("instance_insert", [vp, u64, u64, u64, P(C.c_float)]); self.instance_remove = fn("instance_remove", [vp, u64])
        self.light_insert = fn("light_insert", [vp, u64, P(StLight)]); self.light_remove = fn("light_remove", [vp, u64])
        self.sun_update = fn("sun_update", [vp, C.c_float, C.c_float])
        self.camera_create = fn("camera_create", [vp, P(StCamera), P(u64)]); self.camera_update = fn("camera_update", [vp, u64, P(StCamera)])
        self.camera_delete = fn("camera_delete", [vp, u64])
        self.tick = fn("tick", [vp, vp] if has_device else [vp])
        self.render_camera = fn("render_camera", [vp, u64, vp, vp] if has_device else [vp, u64, vp])
        self.set_seed = fn("set_seed", [vp, u64]); self.set_blue_noise = fn("set_blue_noise", [vp, vp, sz])
        self.debug_read_lut = fn("debug_read_lut", [vp, i32, vp, sz, P(sz)])
        self.camera_read_buffer = fn("camera_read_buffer", [vp, u64, i32, vp, sz, P(sz)])
        self.camera_ray_count = fn("camera_ray_count", [vp, u64, P(u64), i32])
        self.debug_read_scene = fn("debug_read_scene", [vp, i32, vp, sz, P(sz)])
        self.debug_world = fn("debug_world", [vp, P(u32), P(u32)])
        self.debug_image_rect = fn("debug_image_rect", [vp, u64, P(u32)])
        self.set_bvh_refresh = fn("set_bvh_refresh", [vp, i32]); self.debug_bvh_refits = fn("debug_bvh_refits", [vp, P(u64), P(u64)])
        if has_device:
            self.debug_bvh_depth = fn("debug_bvh_depth", [vp, P(u32), P(u32)])
            if hasattr(lib, prefix + "debug_walk_overflow"):   # (round 6; a round-5 library loaded for a same-box A/B has none)
                self.debug_walk_overflow = fn("debug_walk_overflow", [vp, P(u64), P(u32), P(u32)])
            self.debug_bvh_device_refits = fn("debug_bvh_device_refits", [vp, P(u64)])
            self.debug_device_bakes = fn("debug_device_bakes", [vp, P(u64), P(u64)])
            if hasattr(lib, prefix + "debug_auto_tree"):   # (round 6, late)
                self.debug_auto_tree = fn("debug_auto_tree", [vp, P(C.c_float), P(u32)])
            self.debug_device_builds = fn("debug_device_builds", [vp, P(u64)])
            self.debug_device_tree_refits = fn("debug_device_tree_refits", [vp, P(u64)])
            self.engine_get_tuning = fn("engine_get_tuning", [vp, P(StTuning)]); self.engine_set_tuning = fn("engine_set_tuning", [vp, P(StTuning)])
            self.debug_copy_bandwidth = fn("debug_copy_bandwidth", [vp, sz, i32, P(C.c_double)])
            self.debug_variance_flags = fn("debug_variance_flags", [vp, u64, vp, sz, P(sz)])
            self.camera_set_window = fn("camera_set_window", [vp, u64, u32, u32, u32, u32])
            self.dist_init = fn("dist_init", [vp, i32, i32, P(StDistUniqueId)]); self.dist_init_local = fn("dist_init_local", [vp, i32, i32, u64])
            self.dist_shutdown = fn("dist_shutdown", [vp]); self.dist_rank = fn("dist_rank", [vp, P(i32), P(i32)])
            self.dist_set_partition = fn("dist_set_partition", [vp, u64, u32, u32, P(StDistRect), P(StDistRect)])
            self.dist_set_grid = fn("dist_set_grid", [vp, u64, P(StDistGrid), u32, P(StDistRect), P(StDistRect)])
            self.dist_gather = fn("dist_gather", [vp, u64, vp, vp, vp]); self.dist_wait = fn("dist_wait", [vp, u64, vp, vp, i32])
            self.dist_gather_ms = fn("dist_gather_ms", [vp, u64, P(C.c_float)])
        self.image_insert_rgba8 = fn("image_insert_rgba8", [vp, u64, u32, u32, vp, i32]); self.image_remove = fn("image_remove", [vp, u64])
        if has_device:
            self.camera_set_rows = fn("camera_set_rows", [vp, u64, u32, u32])
            self.camera_set_output_format = fn("camera_set_output_format", [vp, u64, i32])
            self.image_insert_device_rgba8 = fn("image_insert_device_rgba8", [vp, u64, u32, u32, vp, sz, i32])
            self.debug_bvh_refresh = fn("debug_bvh_refresh", [vp, P(u64), P(u64)])
            self.scene_load_gltf = fn("scene_load_gltf", [vp, C.c_char_p, P(StGltfOptions), P(StGltfSummary)])
            self.scene_load_gltf_memory = fn("scene_load_gltf_memory", [vp, vp, sz, C.c_char_p, P(StGltfOptions), P(StGltfSummary)])
            self.decode_png = fn("decode_png", [vp, sz, vp, sz, P(u32), P(u32)])
            self.engine_set_arithmetic = fn("engine_set_arithmetic", [vp, i32]); self.engine_get_arithmetic = fn("engine_get_arithmetic", [vp, P(i32)])
            self.camera_write_buffer = fn("camera_write_buffer", [vp, u64, i32, vp, sz])
            self.debug_set_pass_mask = fn("debug_set_pass_mask", [vp, u64]); self.debug_set_launch_filter = fn("debug_set_launch_filter", [vp, u64]); self.debug_last_launches = fn("debug_last_launches", [vp, P(u64), sz, P(sz)])
            self.debug_keep_all_planes = fn("debug_keep_all_planes", [vp, i32])
            self.camera_buffer_stale = fn("camera_buffer_stale", [vp, u64, i32, P(i32)])
            self.camera_present_copy = fn("camera_present_copy", [vp, u64, vp, vp, sz, vp])
            self.camera_present_ready = fn("camera_present_ready", [vp, u64, vp, i32, P(i32)])
            self.profile_enable = fn("profile_enable", [vp, i32])
            self.profile_read = fn("profile_read", [vp, P(StKernelProfile), sz, P(sz), i32])
            self.last_error = getattr(lib, prefix + "last_error"); self.last_error.restype = C.c_char_p; self.last_error.argtypes = []


_STATUS = {1: "invalid argument", 2: "no HIP device (host-only engine or HIP unavailable)", 3: "camera does not exist",
           4: "mesh contains no triangles", 5: "HIP runtime error", 6: "no more space in the atlas", 7: "file could not be read",
           8: "malformed scene file", 9: "unsupported scene file feature",
           10: "the BVH is deeper than the kernels' traversal stack (the scene was uploaded; pushes beyond 24 pending entries are dropped)",
           11: "collective transport error"}
ST_ERR_BVH_TOO_DEEP = 10


class EngineBase:
    """Shared call sequence; `Engine` binds it to libstrolle_hip.so."""

    def __init__(self, binding: _Binding, device: int = 0):
        self._b = binding
        h = C.c_void_p()
        self._check(binding.engine_create(device, C.byref(h)) if binding.has_device else binding.engine_create(C.byref(h)))
        self._h = h
        self._keep = []

    def _check(self, status: int):
        if status != 0:
            detail = ""
            if self._b.has_device:
                msg = self._b.last_error()
                detail = f": {msg.decode(errors='replace')}" if msg else ""
            raise StrolleError(f"{_STATUS.get(status, 'error')} (status {status}){detail}")

    def close(self):
        if getattr(self, "_h", None):
            self._b.engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- scene (lib.rs:161-246)
    def insert_mesh(self, handle: int, mesh: Mesh):
        data = mesh.pack()
        self._check(self._b.mesh_insert(self._h, handle, data.ctypes.data, len(data)))

    def remove_mesh(self, handle: int):
        self._check(self._b.mesh_remove(self._h, handle))

    def insert_material(self, handle: int, material: Material):
        m = material.to_c()
        self._check(self._b.material_insert(self._h, handle, C.byref(m)))

    def has_material(self, handle: int) -> bool:
        return bool(self._b.material_has(self._h, handle))

    def remove_material(self, handle: int):
        self._check(self._b.material_remove(self._h, handle))

    def insert_image(self, handle: int, rgba: np.ndarray, srgb: bool = True):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert rgba.ndim == 3 and rgba.shape[2] == 4
        self._check(self._b.image_insert_rgba8(self._h, handle, rgba.shape[1], rgba.shape[0], rgba.ctypes.data, 1 if srgb else 0))

    def remove_image(self, handle: int):
        self._check(self._b.image_remove(self._h, handle))

    def insert_instance(self, handle: int, instance: Instance):
        x = instance.xform12()
        self._check(self._b.instance_insert(self._h, handle, instance.mesh_handle, instance.material_handle, x.ctypes.data_as(C.POINTER(C.c_float))))

    def remove_instance(self, handle: int):
        self._check(self._b.instance_remove(self._h, handle))

    def insert_light(self, handle: int, light: Light):
        l = light.to_c()
        self._check(self._b.light_insert(self._h, handle, C.byref(l)))

    def remove_light(self, handle: int):
        self._check(self._b.light_remove(self._h, handle))

    def update_sun(self, sun: Sun):
        self._check(self._b.sun_update(self._h, sun.azimuth, sun.altitude))

    # --- cameras (lib.rs:252-297)
    def create_camera(self, camera: Camera) -> int:
        c = camera.to_c(); out = C.c_uint64()
        self._check(self._b.camera_create(self._h, C.byref(c), C.byref(out)))
        return out.value

    def update_camera(self, handle: int, camera: Camera):
        c = camera.to_c()
        self._check(self._b.camera_update(self._h, handle, C.byref(c)))

    def delete_camera(self, handle: int):
        self._check(self._b.camera_delete(self._h, handle))

    # --- seams
    def set_seed(self, seed: int):
        self._check(self._b.set_seed(self._h, seed))

    def set_blue_noise(self, rgba: np.ndarray):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        self._check(self._b.set_blue_noise(self._h, rgba.ctypes.data, rgba.nbytes))

    # --- read-back
    def buffer_stale(self, camera: int, buffer: Buffer) -> bool:
        """st_camera_buffer_stale: the camera's last frame (a lean frame) left this plane unwritten."""
        if not hasattr(self._b, "camera_buffer_stale"):
            return False
        out = C.c_int()
        self._check(self._b.camera_buffer_stale(self._h, camera, int(buffer), C.byref(out)))
        return out.value != 0

    def read_buffer(self, camera: int, buffer: Buffer, strict: bool = False) -> np.ndarray:
        """st_camera_read_buffer. strict=True refuses a plane the last frame did not write (the lean frame of the fast build,
        include/strolle_hip.h st_debug_keep_all_planes) instead of returning an earlier launch's content."""
        if strict and self.buffer_stale(camera, buffer):
            raise StrolleError(f"{Buffer(int(buffer)).name} was not written by the last frame (lean frame): keep_all_planes(True) stores every plane")
        n = C.c_size_t()
        self._check(self._b.camera_read_buffer(self._h, camera, int(buffer), None, 0, C.byref(n)))
        dtype = np.uint32 if int(buffer) == int(Buffer.DBG_USED_MEMORY) else np.float32
        out = np.empty(n.value // 4, dtype=dtype)
        self._check(self._b.camera_read_buffer(self._h, camera, int(buffer), out.ctypes.data, out.nbytes, C.byref(n)))
        return out

    def ray_count(self, camera: int, reset: bool = False) -> int:
        out = C.c_uint64()
        self._check(self._b.camera_ray_count(self._h, camera, C.byref(out), 1 if reset else 0))
        return out.value

    def read_scene(self, what: int) -> np.ndarray:
        n = C.c_size_t()
        self._check(self._b.debug_read_scene(self._h, what, None, 0, C.byref(n)))
        out = np.empty(n.value // 4, dtype=np.float32)
        if n.value:
            self._check(self._b.debug_read_scene(self._h, what, out.ctypes.data, out.nbytes, C.byref(n)))
        return out

    def read_lut(self, what: int) -> np.ndarray:
        """Atmosphere LUT (0 transmittance 256x64, 1 scattering 32x32, 2 sky 256x256) as [H, W, 4] float32."""
        n = C.c_size_t()
        self._check(self._b.debug_read_lut(self._h, what, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        self._check(self._b.debug_read_lut(self._h, what, out.ctypes.data, out.size, C.byref(n)))
        w = (256, 32, 256)[what]
        return out.reshape(-1, w, 4)

    def world(self):
        lc, fr = C.c_uint32(), C.c_uint32()
        self._check(self._b.debug_world(self._h, C.byref(lc), C.byref(fr)))
        return lc.value, fr.value

    def set_bvh_refresh(self, refit):
        """st_set_bvh_refresh: False / 0 = rebuild on every change (the reference's behaviour), True / 1 = refit while instances only
        move, 2 = the same with the boxes recomputed on the device (ST_BVH_REFIT_DEVICE; libstrolle_hip.so only), 3 = the tree BUILT on the device
        straight into the wide stream while nothing observes the contract stream (ST_BVH_BUILD_DEVICE; libstrolle_hip.so only), 4 = the library's default
        (ST_BVH_AUTO: the first tree on the host — unless it hangs long leaf runs on large faces (auto_tree()), then the device builder's —, every later change as mode 3)."""
        refit = int(refit)
        if refit not in (0, 1, 2, 3, 4):
            raise StrolleError(f"unknown BVH refresh mode {refit}")
        if refit >= 2 and not hasattr(self._b, "debug_bvh_device_refits"):
            raise StrolleError("ST_BVH_REFIT_DEVICE is a mode of libstrolle_hip.so; this engine's library does not have it")
        self._check(self._b.set_bvh_refresh(self._h, refit))

    def bvh_device_refits(self) -> int:
        if not hasattr(self._b, "debug_bvh_device_refits"):
            raise StrolleError("bvh_device_refits() is a seam of libstrolle_hip.so (st_debug_bvh_device_refits); this engine's library does not export it")
        out = C.c_uint64()
        self._check(self._b.debug_bvh_device_refits(self._h, C.byref(out)))
        return out.value

    def device_builds(self) -> int:
        """ticks whose tree was built on the device so far (ST_BVH_BUILD_DEVICE)."""
        n = C.c_uint64()
        self._check(self._b.debug_device_builds(self._h, C.byref(n)))
        return int(n.value)

    def device_tree_refits(self) -> int:
        """ticks of ST_BVH_BUILD_DEVICE in which instances only moved and the device-built tree was refitted instead of rebuilt."""
        n = C.c_uint64()
        self._check(self._b.debug_device_tree_refits(self._h, C.byref(n)))
        return int(n.value)

    def device_bakes(self):
        """(launches of the device bake, triangles they baked) so far — StTuning::device_bake."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._b.debug_device_bakes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def bvh_refits(self):
        """(rebuilds, refits) so far."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._b.debug_bvh_refits(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def bvh_depth(self):
        """(longest chain of internal nodes in the uploaded BVH, traversal stack entries per ray)."""
        if not hasattr(self._b, "debug_bvh_depth"):
            raise StrolleError("bvh_depth() is a seam of libstrolle_hip.so (st_debug_bvh_depth); this engine's library does not export it")
        a, b = C.c_uint32(), C.c_uint32()
        self._check(self._b.debug_bvh_depth(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def walk_overflow(self):
        """(ticks that found a wide walk's overflow word set, entries the wide walks' stack holds now, 1 once the packet walk overflowed) — st_debug_walk_overflow."""
        n, entries, off = C.c_uint64(), C.c_uint32(), C.c_uint32()
        self._check(self._b.debug_walk_overflow(self._h, C.byref(n), C.byref(entries), C.byref(off)))
        return n.value, entries.value, off.value

    def auto_tree(self):
        """(surface-area-weighted mean leaf-run length of the host's last tree, True when ST_BVH_AUTO gave this scene's first tree to the device builder) — st_debug_auto_tree."""
        w, d = C.c_float(), C.c_uint32()
        self._check(self._b.debug_auto_tree(self._h, C.byref(w), C.byref(d)))
        return float(w.value), bool(d.value)

    def image_rect(self, handle: int):
        """(x, y, w, h) of an image in the atlas, in texels."""
        r = (C.c_uint32 * 4)()
        self._check(self._b.debug_image_rect(self._h, handle, r))
        return tuple(r)

    def bvh_refresh(self):
        """(primitives in the tree, primitives that arrived inside subtrees reused from the previous tree)."""
        n, r = C.c_uint64(), C.c_uint64()
        self._check(self._b.debug_bvh_refresh(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value


_lib_cache = {}


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if path not in _lib_cache:
        if not os.path.exists(path):
            raise StrolleError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        # One HIP runtime per process: PyTorch ships its own libamdhip64. If this library were loaded first it would bring
        # in /opt/rocm's copy, a later `import torch` would load its bundled one beside it, and device enumeration fails in
        # whichever comes second. So: a process that uses torch together with this package imports torch FIRST (tests and
        # bench.py do; then both share torch's runtime — this library only needs the HIP API). Nothing is imported behind
        # the caller's back unless STROLLE_HIP_PRELOAD_TORCH=1 asks for it.
        if os.environ.get("STROLLE_HIP_PRELOAD_TORCH") == "1" and "torch" not in sys.modules:
            import torch  # noqa: F401
        _lib_cache[path] = C.CDLL(path)
    return _lib_cache[path]


def library_build_commit(path: str = LIB_PATH) -> Optional[str]:
    """st_build_commit(): the commit compiled into the library that is loaded (None: a library older than round 6)."""
    lib = load_library(path)
    if not hasattr(lib, "st_build_commit"):
        return None
    lib.st_build_commit.restype = C.c_char_p
    return lib.st_build_commit().decode(errors="replace")


class Engine(EngineBase):
    """MI355X engine. device >= 0: HIP ordinal; device = -1: host-only (scene + BVH logic, no rendering)."""

    def __init__(self, device: int = 0, exact: Optional[bool] = None):
        """exact=None: the library default (fast arithmetic unless ST_EXACT=1 is set); True / False: st_engine_set_arithmetic."""
        super().__init__(_Binding(load_library(), "st_", True), device)
        if exact is not None:
            self.set_exact(exact)

    def set_exact(self, exact: bool):
        """st_engine_set_arithmetic: True = the bit-exact build of the kernels, False = the fast (default) build."""
        self._check(self._b.engine_set_arithmetic(self._h, 1 if exact else 0))

    @property
    def exact(self) -> bool:
        out = C.c_int()
        self._check(self._b.engine_get_arithmetic(self._h, C.byref(out)))
        return out.value == 1

    def write_buffer(self, camera: int, buffer: "Buffer", data: np.ndarray):
        """st_camera_write_buffer: the inverse of read_buffer (parity tests hand a launch its reference input planes)."""
        data = np.ascontiguousarray(data)
        self._check(self._b.camera_write_buffer(self._h, camera, int(buffer), data.ctypes.data, data.nbytes))

    def keep_all_planes(self, keep: bool):
        """st_debug_keep_all_planes: True = every frame stores every plane the reference does; False (default in the fast build) =
        the lean frame, which leaves planes nothing reads again unwritten (include/strolle_hip.h lists them)."""
        self._check(self._b.debug_keep_all_planes(self._h, 1 if keep else 0))

    def set_pass_mask(self, mask: int):
        """st_debug_set_pass_mask: bit set = that reference pass runs (PassBit)."""
        self._check(self._b.debug_set_pass_mask(self._h, mask & 0xFFFFFFFFFFFFFFFF))

    def set_launch_filter(self, mask: int):
        """st_debug_set_launch_filter: bit i set = the i-th launch of the frame's serial order is enqueued (measurement only)."""
        self._check(self._b.debug_set_launch_filter(self._h, mask & 0xFFFFFFFFFFFFFFFF))

    def last_launches(self):
        """Pass bits of every launch the last render_camera considered, in launch order."""
        arr = (C.c_uint64 * 64)(); n = C.c_size_t()
        self._check(self._b.debug_last_launches(self._h, arr, 64, C.byref(n)))
        return [arr[i] for i in range(min(n.value, 64))]

    def tick(self, stream: int = 0):
        self._check(self._b.tick(self._h, stream))

    def render_camera(self, handle: int, out_device_ptr: int = 0, stream: int = 0):
        """Enqueue CameraController::render; `out_device_ptr` = device address of a W*H RGBA32F buffer (0 = skip composition)."""
        self._check(self._b.render_camera(self._h, handle, out_device_ptr, stream))

    def set_output_format(self, handle: int, fmt: "OutputFormat"):
        """viewport.format (camera.rs:170-175): what st_render_camera writes into its output buffer."""
        self._check(self._b.camera_set_output_format(self._h, handle, int(fmt)))

    def present_copy(self, handle: int, src_device_ptr: int, dst_host_ptr: int, nbytes: int, stream: int = 0):
        """st_camera_present_copy: the frame just composed into `src_device_ptr` on `stream` -> page-locked host memory,
        asynchronously on the camera's copy stream (the facade's present path; the next frame overlaps the copy)."""
        self._check(self._b.camera_present_copy(self._h, handle, src_device_ptr, dst_host_ptr, nbytes, stream))

    def present_ready(self, handle: int, dst_host_ptr: int, wait: bool = False) -> bool:
        """st_camera_present_ready: has the copy into `dst_host_ptr` landed? wait=True blocks until it has."""
        out = C.c_int()
        self._check(self._b.camera_present_ready(self._h, handle, dst_host_ptr, 1 if wait else 0, C.byref(out)))
        return out.value == 1

    def set_camera_rows(self, handle: int, y0: int, y1: int):
        self._check(self._b.camera_set_rows(self._h, handle, y0, y1))

    def variance_flags(self, cam: int) -> np.ndarray:
        """st_debug_variance_flags: one uint64 per 8x8 tile, bit = pixel whose variance estimate takes the short-history branch."""
        n = C.c_size_t()
        self._check(self._b.debug_variance_flags(self._h, cam, None, 0, C.byref(n)))
        mask = np.zeros(n.value, np.uint64)
        self._check(self._b.debug_variance_flags(self._h, cam, mask.ctypes.data, mask.size, C.byref(n)))
        return mask

    def set_camera_window(self, handle: int, x0: int, y0: int, x1: int, y1: int):
        """st_camera_set_window: restrict the camera's launches to [x0, x1) x [y0, y1) of the viewport (0, 0 = everything on that axis)."""
        self._check(self._b.camera_set_window(self._h, handle, x0, y0, x1, y1))

    # ---- multi-GPU behind the boundary (include/strolle_hip.h st_dist_*)
    def dist_init(self, rank: int, world: int, unique_id: bytes):
        """RCCL transport: `unique_id` = dist_unique_id() of rank 0, handed to every rank by the caller's own means."""
        if not isinstance(unique_id, (bytes, bytearray)) or len(unique_id) != 128:   # a truncated id would hang every rank in ncclCommInitRank
            raise StrolleError(f"dist_init: the RCCL unique id has 128 bytes, got {len(unique_id) if hasattr(unique_id, '__len__') else type(unique_id).__name__}")
        uid = StDistUniqueId(); C.memmove(C.byref(uid), bytes(unique_id), 128)
        self._check(self._b.dist_init(self._h, rank, world, C.byref(uid)))

    def dist_init_local(self, rank: int, world: int, group: int = 1):
        """In-process transport: the engines of this process that share `group` (tests, single-GPU boxes)."""
        self._check(self._b.dist_init_local(self._h, rank, world, group))

    def dist_shutdown(self):
        self._check(self._b.dist_shutdown(self._h))

    def dist_set_partition(self, cam: int, cols: int = 0, apron: int = 0):
        """This rank's tile of the camera's frame (+ apron) becomes the camera's window. Returns (owned, window) as (x0, y0, x1, y1)."""
        o, w = StDistRect(), StDistRect()
        self._check(self._b.dist_set_partition(self._h, cam, cols, apron, C.byref(o), C.byref(w)))
        return o.as_tuple(), w.as_tuple()

    def dist_set_grid(self, cam: int, grid: "StDistGrid", apron: int = 0):
        """st_dist_set_grid: like dist_set_partition with the tiles of `grid` (dist_grid / dist_grid_rebalance); every rank sets the same grid."""
        o, w = StDistRect(), StDistRect()
        self._check(self._b.dist_set_grid(self._h, cam, C.byref(grid), apron, C.byref(o), C.byref(w)))
        return o.as_tuple(), w.as_tuple()

    def dist_gather(self, cam: int, frame_ptr: int, full_ptr: int = 0, stream: int = 0):
        """st_dist_gather: this rank's tile of `frame_ptr` travels to rank 0, which assembles the frame at `full_ptr`."""
        self._check(self._b.dist_gather(self._h, cam, frame_ptr, full_ptr or None, stream))

    def dist_wait(self, cam: int, frame_ptr: int = 0, stream: int = 0, host: bool = True):
        """st_dist_wait: behind the gather that read `frame_ptr` (0: every gather in flight); host=True blocks the caller."""
        self._check(self._b.dist_wait(self._h, cam, frame_ptr or None, stream, 1 if host else 0))

    def dist_gather_ms(self, cam: int) -> float:
        out = C.c_float()
        self._check(self._b.dist_gather_ms(self._h, cam, C.byref(out)))
        return out.value

    def insert_device_image(self, handle: int, device_ptr: int, width: int, height: int, row_pitch_bytes: int = 0, dynamic: bool = False):
        """ImageData::Texture: RGBA8 pixels in device memory (e.g. `tensor.data_ptr()` of a [h, w, 4] uint8 CUDA tensor)."""
        self._check(self._b.image_insert_device_rgba8(self._h, handle, width, height, device_ptr, row_pitch_bytes or width * 4, 1 if dynamic else 0))

    def load_gltf(self, source, base_dir: Optional[str] = None, first_handle: int = 1, first_image_handle: int = 1000,
                  reflectance: Optional[float] = None, perceptual_roughness: Optional[float] = None, subdivide: int = 0, light_radius: float = 0.0) -> dict:
        """st_scene_load_gltf: `source` is a path to a .gltf / .glb file, or the file's bytes (external buffers and images
        are then read relative to `base_dir`). Returns the loader's summary."""
        opt = StGltfOptions(first_handle, first_image_handle, 0, 0.0, 0.0, subdivide, light_radius, 0)
        if reflectance is not None:
            opt.override_mask |= 1; opt.reflectance = reflectance
        if perceptual_roughness is not None:
            opt.override_mask |= 2; opt.perceptual_roughness = perceptual_roughness
        out = StGltfSummary()
        if isinstance(source, (bytes, bytearray, memoryview)):
            data = bytes(source)
            self._check(self._b.scene_load_gltf_memory(self._h, data, len(data), base_dir.encode() if base_dir else None, C.byref(opt), C.byref(out)))
        else:
            self._check(self._b.scene_load_gltf(self._h, os.fsencode(source), C.byref(opt), C.byref(out)))
        return {name: getattr(out, name) for name, _ in StGltfSummary._fields_}

    def tuning(self) -> "StTuning":
        """st_engine_get_tuning: the engine's scheduling / tuning switches (a copy; change fields and hand it to set_tuning)."""
        t = StTuning()
        self._check(self._b.engine_get_tuning(self._h, C.byref(t)))
        return t

    def set_tuning(self, tuning=None, **fields):
        """st_engine_set_tuning. Either a whole StTuning, or keyword fields applied on top of the current one: set_tuning(fuse=0)."""
        t = tuning if tuning is not None else self.tuning()
        for k, v in fields.items():
            if not hasattr(t, k):
                raise StrolleError(f"StTuning has no field {k!r}")
            setattr(t, k, int(v))
        self._check(self._b.engine_set_tuning(self._h, C.byref(t)))

    def copy_bandwidth(self, nbytes: int = 1 << 30, iters: int = 6) -> float:
        """st_debug_copy_bandwidth: GB/s (read + write) of the library's own grid-stride float4 copy on this device."""
        out = C.c_double()
        self._check(self._b.debug_copy_bandwidth(self._h, nbytes, iters, C.byref(out)))
        return out.value

    def profile_enable(self, flags):
        """st_profile_enable: bit 0 = per-kernel event timing (serial execution), bit 1 = traversal-byte counters; True = 1."""
        self._check(self._b.profile_enable(self._h, int(flags)))

    def profile_read(self, reset: bool = True):
        arr = (StKernelProfile * 48)(); n = C.c_size_t()
        self._check(self._b.profile_read(self._h, arr, 48, C.byref(n), 1 if reset else 0))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms, algorithmic_bytes=arr[i].algorithmic_bytes, traversal_bytes=arr[i].traversal_bytes)
                for i in range(n.value)]


def dist_partition(width: int, height: int, world: int, rank: int, cols: int = 0):
    """st_dist_partition: the tile (x0, y0, x1, y1) rank `rank` of `world` owns (cols = 0: the default grid)."""
    lib = load_library()
    r = StDistRect()
    lib.st_dist_partition.restype = C.c_int
    lib.st_dist_partition.argtypes = [C.c_uint32] * 5 + [C.POINTER(StDistRect)]
    if lib.st_dist_partition(width, height, world, cols, rank, C.byref(r)) != 0:
        lib.st_last_error.restype = C.c_char_p
        raise StrolleError(lib.st_last_error().decode(errors="replace"))
    return r.as_tuple()


def _dist_call(name, argtypes, *args):
    lib = load_library()
    f = getattr(lib, name)
    f.restype = C.c_int; f.argtypes = argtypes
    if f(*args) != 0:
        lib.st_last_error.restype = C.c_char_p
        raise StrolleError(lib.st_last_error().decode(errors="replace"))


def dist_grid(width: int, height: int, world: int, cols: int = 0) -> StDistGrid:
    """st_dist_grid: the equal split as a grid (its tiles are st_dist_partition's)."""
    g = StDistGrid()
    _dist_call("st_dist_grid", [C.c_uint32] * 4 + [C.POINTER(StDistGrid)], width, height, world, cols, C.byref(g))
    return g


def dist_grid_tile(grid: StDistGrid, rank: int):
    r = StDistRect()
    _dist_call("st_dist_grid_tile", [C.POINTER(StDistGrid), C.c_uint32, C.POINTER(StDistRect)], C.byref(grid), rank, C.byref(r))
    return r.as_tuple()


def dist_grid_rebalance(width: int, height: int, grid: StDistGrid, tile_cost, max_step: int = 0) -> StDistGrid:
    """st_dist_grid_rebalance: the grid whose rows, then each row's tiles, would cost the same, from one cost per tile in rank order."""
    cost = (C.c_float * (grid.cols * grid.rows))(*[float(v) for v in tile_cost])
    out = StDistGrid()
    _dist_call("st_dist_grid_rebalance", [C.c_uint32, C.c_uint32, C.POINTER(StDistGrid), C.POINTER(C.c_float), C.c_uint32, C.POINTER(StDistGrid)],
               width, height, C.byref(grid), cost, max_step, C.byref(out))
    return out


def dist_window(width: int, height: int, owned, apron: int):
    """st_dist_window: `owned` widened by `apron` pixels towards its neighbours, on the 16 x 8 pixel grid."""
    lib = load_library()
    o, w = StDistRect(*owned), StDistRect()
    lib.st_dist_window.restype = C.c_int
    lib.st_dist_window.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(StDistRect), C.c_uint32, C.POINTER(StDistRect)]
    if lib.st_dist_window(width, height, C.byref(o), apron, C.byref(w)) != 0:
        lib.st_last_error.restype = C.c_char_p
        raise StrolleError(lib.st_last_error().decode(errors="replace"))
    return w.as_tuple()


def dist_unique_id() -> bytes:
    """st_dist_unique_id (ncclGetUniqueId): rank 0 makes it, every rank passes it to Engine.dist_init."""
    lib = load_library()
    uid = StDistUniqueId()
    lib.st_dist_unique_id.restype = C.c_int
    lib.st_dist_unique_id.argtypes = [C.POINTER(StDistUniqueId)]
    if lib.st_dist_unique_id(C.byref(uid)) != 0:
        lib.st_last_error.restype = C.c_char_p
        raise StrolleError(lib.st_last_error().decode(errors="replace"))
    return bytes(C.string_at(C.byref(uid), 128))


def decode_png(data: bytes) -> np.ndarray:
    """st_decode_png: PNG bytes -> [h, w, 4] uint8 (the decoder st_scene_load_gltf uses for textures)."""
    return _decode(data, "st_decode_png")


def decode_image(data: bytes) -> np.ndarray:
    """st_decode_image: PNG or JPEG bytes -> [h, w, 4] uint8."""
    return _decode(data, "st_decode_image")


def _decode(data: bytes, symbol: str) -> np.ndarray:
    lib = load_library()
    fn = getattr(lib, symbol)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    w, h = C.c_uint32(), C.c_uint32()
    data = bytes(data)
    status = fn(data, len(data), None, 0, C.byref(w), C.byref(h))
    if status == 0:
        out = np.empty((h.value, w.value, 4), np.uint8)
        status = fn(data, len(data), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h))
    if status != 0:
        lib.st_last_error.restype = C.c_char_p
        raise StrolleError(f"{_STATUS.get(status, 'error')} (status {status}): {lib.st_last_error().decode(errors='replace')}")
    return out
