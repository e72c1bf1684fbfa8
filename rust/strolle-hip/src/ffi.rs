//! Raw bindings of include/strolle_hip.h (the C ABI of libstrolle_hip.so) and of the HIP runtime calls the present
//! step needs (allocation only: the copy itself is the library's, `st_camera_present_copy`). One `extern "C"` item per entry point; the comment names the `strolle::Engine` method it stands behind
//! (reference: strolle/src/lib.rs:132-395).
#![allow(non_camel_case_types, dead_code)]
use std::ffi::c_void;
use std::os::raw::c_char;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct StMeshTriangle {
    pub positions: [[f32; 3]; 3],
    pub normals: [[f32; 3]; 3],
    pub uvs: [[f32; 2]; 3],
    pub tangents: [[f32; 4]; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct StMaterial {
    pub base_color: [f32; 4],
    pub emissive: [f32; 4],
    pub perceptual_roughness: f32,
    pub metallic: f32,
    pub reflectance: f32,
    pub ior: f32,
    pub base_color_texture: u64, // 0 = None
    pub emissive_texture: u64,
    pub metallic_roughness_texture: u64,
    pub normal_map_texture: u64,
    pub alpha_mode: u32, // 0 Opaque, 1 Blend
    pub _pad: u32,
}

pub const ST_LIGHT_POINT: u32 = 0;
pub const ST_LIGHT_SPOT: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct StLight {
    pub kind: u32,
    pub position: [f32; 3],
    pub radius: f32,
    pub color: [f32; 3],
    pub range: f32,
    pub direction: [f32; 3],
    pub angle: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct StCamera {
    pub mode: u32,
    pub denoise: u32,
    pub depth: u32,
    pub width: u32,
    pub height: u32,
    pub pos_x: u32,
    pub pos_y: u32,
    pub _pad: u32,
    pub transform: [f32; 16],  // glam Mat4::to_cols_array
    pub projection: [f32; 16],
}

pub const ST_OK: i32 = 0;
pub const ST_FORMAT_RGBA32F: i32 = 0;
pub const ST_FORMAT_RGBA16F: i32 = 1;
pub const ST_FORMAT_RGBA8_UNORM_SRGB: i32 = 2;
pub const ST_FORMAT_BGRA8_UNORM_SRGB: i32 = 3;

pub const ST_ERR_BVH_TOO_DEEP: i32 = 10; // st_tick: the tree is deeper than the kernels' traversal stack (uploaded all the same)
pub const ST_ERR_DIST: i32 = 11;

/// include/strolle_hip.h StTuning: scheduling / tuning switches of one engine (get, change, set).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct StTuning {
    pub struct_size: u32,
    pub overlap: u32,
    pub fuse: u32,
    pub fuse_di_head: u32,
    pub fuse_spatial: u32,
    pub fuse_gi_sampling: u32,
    pub fuse_gi_validation: u32,
    pub fuse_gi_reprojection: u32,
    pub fuse_wavelet: u32,
    pub fuse_compose: u32,
    pub preview_both: u32,
    pub variance_in_reproject: u32,
    pub lean_frame: u32,
    pub skip_scratch_stores: u32,
    pub di_head_on_main: u32,
    pub alias_gi_history: u32,
    pub tile_map: u32,
    pub tile_map_denoise: u32,
    pub side_priority: i32,
    pub staging: u32,
    pub double_buffer: u32,
    pub packed_base: u32,
    pub tick_timing: u32,
    pub anyhit_fast: u32,
    pub compact_bvh: u32,
    pub allow_deep_bvh: u32,
    pub device_bake: u32,
    pub wide_bvh: u32,
    pub wide_stack_entries: u32,
    pub primary_packets: u32,
}

/// [x0, x1) x [y0, y1) in pixels (st_dist_partition / st_dist_window)
/// A (cost-weighted) tile grid every rank holds identically (st_dist_grid / st_dist_grid_rebalance / st_dist_set_grid)
#[repr(C)]
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub struct StDistGrid {
    pub cols: u32,
    pub rows: u32,
    pub row_edge: [u32; 17],
    pub col_edge: [[u32; 17]; 16],
}

#[repr(C)]
#[derive(Clone, Copy, Default, PartialEq, Eq, Debug)]
pub struct StDistRect {
    pub x0: u32,
    pub y0: u32,
    pub x1: u32,
    pub y1: u32,
}

/// ncclUniqueId: rank 0 makes one (st_dist_unique_id) and hands it to the other processes by the host's own means
#[repr(C)]
#[derive(Clone, Copy)]
pub struct StDistUniqueId {
    pub internal: [c_char; 128],
}

pub enum StEngine {}

extern "C" {
    pub fn st_engine_create(device_ordinal: i32, out: *mut *mut StEngine) -> i32; // Engine::new
    pub fn st_engine_destroy(e: *mut StEngine); // Drop
    pub fn st_last_error() -> *const c_char;
    pub fn st_build_commit() -> *const c_char;
    pub fn st_mesh_insert(e: *mut StEngine, id: u64, triangles: *const StMeshTriangle, count: usize) -> i32; // insert_mesh
    pub fn st_mesh_remove(e: *mut StEngine, id: u64) -> i32; // remove_mesh
    pub fn st_material_insert(e: *mut StEngine, id: u64, material: *const StMaterial) -> i32; // insert_material
    pub fn st_material_has(e: *mut StEngine, id: u64) -> i32; // has_material
    pub fn st_material_remove(e: *mut StEngine, id: u64) -> i32; // remove_material
    pub fn st_image_insert_rgba8(e: *mut StEngine, id: u64, w: u32, h: u32, rgba: *const u8, srgb: i32) -> i32; // insert_image
    pub fn st_image_insert_device_rgba8(e: *mut StEngine, id: u64, w: u32, h: u32, device_rgba: *const c_void, row_pitch: usize, is_dynamic: i32) -> i32;
    pub fn st_image_remove(e: *mut StEngine, id: u64) -> i32; // remove_image
    pub fn st_instance_insert(e: *mut StEngine, id: u64, mesh: u64, material: u64, xform12: *const f32) -> i32; // insert_instance
    pub fn st_instance_remove(e: *mut StEngine, id: u64) -> i32; // remove_instance
    pub fn st_light_insert(e: *mut StEngine, id: u64, light: *const StLight) -> i32; // insert_light
    pub fn st_light_remove(e: *mut StEngine, id: u64) -> i32; // remove_light
    pub fn st_sun_update(e: *mut StEngine, azimuth: f32, altitude: f32) -> i32; // update_sun
    pub fn st_camera_create(e: *mut StEngine, camera: *const StCamera, out: *mut u64) -> i32; // create_camera
    pub fn st_camera_update(e: *mut StEngine, camera: u64, desc: *const StCamera) -> i32; // update_camera
    pub fn st_camera_delete(e: *mut StEngine, camera: u64) -> i32; // delete_camera
    pub fn st_camera_set_output_format(e: *mut StEngine, camera: u64, format: i32) -> i32; // Camera::viewport.format
    pub fn st_tick(e: *mut StEngine, hip_stream: *mut c_void) -> i32; // tick
    pub fn st_render_camera(e: *mut StEngine, camera: u64, out_device: *mut c_void, hip_stream: *mut c_void) -> i32; // render_camera
    // the present hand-over (present.rs): frame N leaves for host memory behind its composition while frame N+1 renders
    pub fn st_camera_present_copy(e: *mut StEngine, camera: u64, src_device: *const c_void, dst_host: *mut c_void, bytes: usize, hip_stream: *mut c_void) -> i32;
    pub fn st_camera_present_ready(e: *mut StEngine, camera: u64, dst_host: *const c_void, wait: i32, ready: *mut i32) -> i32;
    pub fn st_set_seed(e: *mut StEngine, seed: u64) -> i32;
    pub fn st_set_blue_noise(e: *mut StEngine, rgba: *const u8, bytes: usize) -> i32; // Noise::new (noise.rs:40-50)
    pub fn st_engine_set_arithmetic(e: *mut StEngine, arithmetic: i32) -> i32;
    pub fn st_set_bvh_refresh(e: *mut StEngine, mode: i32) -> i32; // 0 rebuild, 1 refit, 2 refit on the device, 3 build on the device (ST_BVH_BUILD_DEVICE), 4 the default (ST_BVH_AUTO: first tree on the host unless its leaf runs are long, changes on the device)
    pub fn st_debug_walk_overflow(e: *mut StEngine, overflows: *mut u64, wide_stack_entries: *mut u32, packets_off: *mut u32) -> i32;
    pub fn st_debug_auto_tree(e: *mut StEngine, leaf_run_weight: *mut f32, first_tree_on_device: *mut u32) -> i32;
    pub fn st_debug_device_builds(e: *mut StEngine, ticks: *mut u64) -> i32;
    pub fn st_debug_device_tree_refits(e: *mut StEngine, ticks: *mut u64) -> i32;
    pub fn st_engine_get_tuning(e: *mut StEngine, out: *mut StTuning) -> i32;
    pub fn st_engine_set_tuning(e: *mut StEngine, tuning: *const StTuning) -> i32;
    // multi-GPU behind the boundary: one process per GPU, tiles + ONE gather to rank 0 over RCCL (dist.rs)
    pub fn st_camera_set_window(e: *mut StEngine, camera: u64, x0: u32, y0: u32, x1: u32, y1: u32) -> i32;
    pub fn st_dist_partition(width: u32, height: u32, world: u32, cols: u32, rank: u32, owned: *mut StDistRect) -> i32;
    pub fn st_dist_window(width: u32, height: u32, owned: *const StDistRect, apron: u32, window: *mut StDistRect) -> i32;
    pub fn st_dist_unique_id(out: *mut StDistUniqueId) -> i32;
    pub fn st_dist_init(e: *mut StEngine, rank: i32, world: i32, id: *const StDistUniqueId) -> i32;
    pub fn st_dist_shutdown(e: *mut StEngine) -> i32;
    pub fn st_dist_set_partition(e: *mut StEngine, camera: u64, cols: u32, apron: u32, owned: *mut StDistRect, window: *mut StDistRect) -> i32;
    pub fn st_dist_grid(width: u32, height: u32, world: u32, cols: u32, out: *mut StDistGrid) -> i32;
    pub fn st_dist_grid_tile(grid: *const StDistGrid, rank: u32, owned: *mut StDistRect) -> i32;
    pub fn st_dist_grid_rebalance(width: u32, height: u32, current: *const StDistGrid, tile_cost: *const f32, max_step: u32, out: *mut StDistGrid) -> i32;
    pub fn st_dist_set_grid(e: *mut StEngine, camera: u64, grid: *const StDistGrid, apron: u32, owned: *mut StDistRect, window: *mut StDistRect) -> i32;
    pub fn st_dist_gather(e: *mut StEngine, camera: u64, frame: *const c_void, full_on_root: *mut c_void, hip_stream: *mut c_void) -> i32;
    pub fn st_dist_wait(e: *mut StEngine, camera: u64, frame: *const c_void, hip_stream: *mut c_void, host_wait: i32) -> i32;
}

// ---- the HIP runtime, as far as the staging-copy present needs it (libamdhip64)
pub type hipStream_t = *mut c_void;
extern "C" {
    pub fn hipMalloc(ptr: *mut *mut c_void, bytes: usize) -> i32;
    pub fn hipFree(ptr: *mut c_void) -> i32;
    pub fn hipHostMalloc(ptr: *mut *mut c_void, bytes: usize, flags: u32) -> i32;
    pub fn hipHostFree(ptr: *mut c_void) -> i32;
    pub fn hipStreamCreate(stream: *mut hipStream_t) -> i32;
    pub fn hipStreamDestroy(stream: hipStream_t) -> i32;
}
