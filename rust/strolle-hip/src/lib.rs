//! `strolle` with its per-pixel hot path — BVH traversal, ReSTIR DI / GI, the SVGF denoiser, composition — running as
//! hand-written HIP kernels on an AMD Instinct MI355X (libstrolle_hip.so, C ABI in include/strolle_hip.h) instead of
//! rust-gpu shaders on wgpu.
//!
//! The public surface is the reference's (`strolle/src/lib.rs:105-409`): the same `Engine<P>` methods with the same
//! signatures, the same value types, the same `Params` trait, so `bevy-strolle` and the `cornell` / `demo` examples build
//! against this crate by changing one line of `Cargo.toml`. wgpu stays what the *host application* renders with; this
//! crate touches it in two places only — reading `ImageData::Texture` images back, and the present pass (present.rs).
//!
//! Status: written against the C ABI and the reference's call sites; the build container has no Rust toolchain, so this
//! crate has not been compiled there. The C ABI underneath is exercised by the repository's test suites.
pub mod dist;
pub mod ffi;
mod handles;
mod present;
mod types;

use std::collections::HashMap;
use std::ffi::CStr;
use std::fmt::Debug;
use std::hash::Hash;
use std::ops::Deref;

pub use glam;
use log::{info, warn};

use self::handles::Interner;
use self::present::Presenter;
pub use self::types::*;

/// Handle types of the embedding application (lib.rs:402-409 of the reference, unchanged).
pub trait Params {
    type ImageHandle: Clone + Copy + Debug + Eq + Hash;
    type ImageTexture: Debug + Deref<Target = wgpu::Texture>;
    type InstanceHandle: Clone + Copy + Debug + Eq + Hash;
    type LightHandle: Clone + Copy + Debug + Eq + Hash;
    type MaterialHandle: Clone + Copy + Debug + Eq + Hash;
    type MeshHandle: Clone + Copy + Debug + Eq + Hash;
}

#[derive(Debug)]
struct CameraSlot {
    raw: u64, // StHandle of the camera inside the library
    camera: Camera,
    presenter: Presenter,
}

/// An image whose pixels live in a wgpu texture and must be (re-)read by `tick` (`ImageData::Texture`).
#[derive(Debug)]
struct TextureImage<P: Params> {
    texture: P::ImageTexture,
    size: (u32, u32),
    is_dynamic: bool,
    pending: bool,
}

#[derive(Debug)]
pub struct Engine<P>
where
    P: Params,
{
    raw: *mut ffi::StEngine,
    stream: ffi::hipStream_t,
    meshes: Interner<P::MeshHandle>,
    materials: Interner<P::MaterialHandle>,
    images: Interner<P::ImageHandle>,
    instances: Interner<P::InstanceHandle>,
    lights: Interner<P::LightHandle>,
    texture_images: HashMap<u64, TextureImage<P>>,
    cameras: HashMap<CameraHandle, CameraSlot>,
    next_camera: usize,
    warned_deep_bvh: bool,
}

// one owner at a time, like the reference's `&mut self` API; the raw pointers are only used through it
unsafe impl<P: Params> Send for Engine<P> where P::ImageTexture: Send {}
unsafe impl<P: Params> Sync for Engine<P> where P::ImageTexture: Sync {}

fn check(status: i32) {
    if status != ffi::ST_OK {
        let message = unsafe { CStr::from_ptr(ffi::st_last_error()) }.to_string_lossy().into_owned();
        // the reference panics where the library reports: unknown camera (camera_controllers.rs:21-34), empty mesh
        // (triangles.rs:50-53); a full atlas is only a warning there (images.rs:71-79)
        if status == 6 {
            warn!("strolle-hip: {message}");
        } else {
            panic!("strolle-hip: {message} (status {status})");
        }
    }
}

impl<P> Engine<P>
where
    P: Params,
{
    /// `device` is the host application's wgpu device; the MI355X is opened through HIP (ordinal from `STROLLE_HIP_DEVICE`,
    /// default 0).
    pub fn new(_device: &wgpu::Device) -> Self {
        info!("Initializing (HIP back end)");
        let ordinal = std::env::var("STROLLE_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::st_engine_create(ordinal, &mut raw) });
        let mut stream = std::ptr::null_mut();
        assert_eq!(unsafe { ffi::hipStreamCreate(&mut stream) }, 0, "hipStreamCreate");
        // Noise::new (noise.rs:40-50) embeds strolle/assets/blue-noise.png; the library takes texels, and the repository
        // carries that texture decoded (assets/blue_noise.npy: a 128-byte NumPy header, then 256 x 256 RGBA8 — written by
        // tools/convert_assets.py, sha256 of the texels in SURVEY.md section 8c)
        let noise: &[u8] = &include_bytes!("../../../assets/blue_noise.npy")[128..];
        assert_eq!(noise.len(), 256 * 256 * 4, "assets/blue_noise.npy");
        check(unsafe { ffi::st_set_blue_noise(raw, noise.as_ptr(), noise.len()) });
        // rand::thread_rng() seeds in the reference (camera_controller.rs:189-194); the library derives per-pass seeds
        // from one base seed
        let seed = std::time::SystemTime::now().duration_since(std::time::UNIX_EPOCH).map(|d| d.as_nanos() as u64).unwrap_or(0);
        check(unsafe { ffi::st_set_seed(raw, seed) });
        Self {
            raw,
            stream,
            meshes: Default::default(),
            materials: Default::default(),
            images: Default::default(),
            instances: Default::default(),
            lights: Default::default(),
            texture_images: Default::default(),
            cameras: Default::default(),
            next_camera: 0,
            warned_deep_bvh: false,
        }
    }

    /// Creates or updates a mesh.
    pub fn insert_mesh(&mut self, handle: P::MeshHandle, item: Mesh) {
        let triangles: Vec<ffi::StMeshTriangle> = item.triangles().iter().map(MeshTriangle::to_ffi).collect();
        check(unsafe { ffi::st_mesh_insert(self.raw, self.meshes.id(handle), triangles.as_ptr(), triangles.len()) });
    }

    /// Removes a mesh (instances that refer to it stay, as in the reference).
    pub fn remove_mesh(&mut self, handle: P::MeshHandle) {
        if let Some(id) = self.meshes.forget(handle) {
            check(unsafe { ffi::st_mesh_remove(self.raw, id) });
        }
    }

    /// Creates or updates a material.
    pub fn insert_material(&mut self, handle: P::MaterialHandle, item: Material<P>) {
        let mut image = |h: Option<P::ImageHandle>| h.map(|h| self.images.id(h)).unwrap_or(0);
        let m = ffi::StMaterial {
            base_color: item.base_color.to_array(),
            emissive: item.emissive.to_array(),
            perceptual_roughness: item.perceptual_roughness,
            metallic: item.metallic,
            reflectance: item.reflectance,
            ior: item.ior,
            base_color_texture: image(item.base_color_texture),
            emissive_texture: image(item.emissive_texture),
            metallic_roughness_texture: image(item.metallic_roughness_texture),
            normal_map_texture: image(item.normal_map_texture),
            alpha_mode: match item.alpha_mode {
                AlphaMode::Opaque => 0,
                AlphaMode::Blend => 1,
            },
            _pad: 0,
        };
        check(unsafe { ffi::st_material_insert(self.raw, self.materials.id(handle), &m) });
    }

    /// Returns whether given material exists.
    pub fn has_material(&self, handle: P::MaterialHandle) -> bool {
        self.materials.get(handle).map(|id| unsafe { ffi::st_material_has(self.raw, id) } != 0).unwrap_or(false)
    }

    /// Removes a material.
    pub fn remove_material(&mut self, handle: P::MaterialHandle) {
        if let Some(id) = self.materials.forget(handle) {
            check(unsafe { ffi::st_material_remove(self.raw, id) });
        }
    }

    /// Creates or updates an image.
    pub fn insert_image(&mut self, image_handle: P::ImageHandle, image: Image<P>) {
        let id = self.images.id(image_handle);
        let size = image.texture_descriptor.size;
        match image.data {
            ImageData::Raw { data } => {
                self.texture_images.remove(&id);
                let srgb = image.texture_descriptor.format.is_srgb() as i32;
                assert_eq!(data.len(), (size.width * size.height * 4) as usize, "strolle-hip takes RGBA8 images");
                check(unsafe { ffi::st_image_insert_rgba8(self.raw, id, size.width, size.height, data.as_ptr(), srgb) });
            }
            // read back in tick(), where the queue is available
            ImageData::Texture { texture, is_dynamic } => {
                self.texture_images.insert(id, TextureImage { texture, size: (size.width, size.height), is_dynamic, pending: true });
            }
        }
    }

    /// Removes an image.
    pub fn remove_image(&mut self, handle: P::ImageHandle) {
        if let Some(id) = self.images.get(handle) {
            // the id stays interned: materials that still refer to the handle keep pointing at "no such image"
            self.texture_images.remove(&id);
            check(unsafe { ffi::st_image_remove(self.raw, id) });
        }
    }

    /// Creates or updates an instance.
    pub fn insert_instance(&mut self, instance_handle: P::InstanceHandle, instance: Instance<P>) {
        let xform = instance.xform12();
        check(unsafe {
            ffi::st_instance_insert(self.raw, self.instances.id(instance_handle), self.meshes.id(instance.mesh_handle), self.materials.id(instance.material_handle), xform.as_ptr())
        });
    }

    /// Removes an instance.
    pub fn remove_instance(&mut self, handle: P::InstanceHandle) {
        if let Some(id) = self.instances.forget(handle) {
            check(unsafe { ffi::st_instance_remove(self.raw, id) });
        }
    }

    /// Creates or updates a light.
    pub fn insert_light(&mut self, handle: P::LightHandle, item: Light) {
        let light = item.to_ffi();
        check(unsafe { ffi::st_light_insert(self.raw, self.lights.id(handle), &light) });
    }

    /// Removes a light.
    pub fn remove_light(&mut self, handle: P::LightHandle) {
        if let Some(id) = self.lights.forget(handle) {
            check(unsafe { ffi::st_light_remove(self.raw, id) });
        }
    }

    /// Updates sun's parameters.
    pub fn update_sun(&mut self, sun: Sun) {
        check(unsafe { ffi::st_sun_update(self.raw, sun.azimuth, sun.altitude) });
    }

    /// Creates a new camera (allocates every per-camera buffer of camera_controller/buffers.rs on the MI355X: about
    /// 900 B per pixel).
    pub fn create_camera(&mut self, device: &wgpu::Device, camera: Camera) -> CameraHandle {
        let desc = camera.to_ffi();
        let mut raw = 0u64;
        check(unsafe { ffi::st_camera_create(self.raw, &desc, &mut raw) });
        let (format, _) = camera.output_format().expect("viewport.format");
        check(unsafe { ffi::st_camera_set_output_format(self.raw, raw, format) });
        let handle = CameraHandle(self.next_camera);
        self.next_camera += 1;
        let presenter = Presenter::new(device, &camera);
        self.cameras.insert(handle, CameraSlot { raw, camera, presenter });
        handle
    }

    /// Updates camera, changing its mode, position, size etc.
    pub fn update_camera(&mut self, device: &wgpu::Device, handle: CameraHandle, camera: Camera) {
        let slot = self.cameras.get_mut(&handle).unwrap_or_else(|| panic!("camera does not exist: {:?}", handle));
        let desc = camera.to_ffi();
        check(unsafe { ffi::st_camera_update(self.raw, slot.raw, &desc) });
        let resized = camera.viewport.size != slot.camera.viewport.size || camera.viewport.format != slot.camera.viewport.format;
        if resized {
            let (format, _) = camera.output_format().expect("viewport.format");
            check(unsafe { ffi::st_camera_set_output_format(self.raw, slot.raw, format) });
            slot.presenter.finish(self.raw, slot.raw); // its buffers are about to be freed
            slot.presenter = Presenter::new(device, &camera);
        } else {
            slot.presenter.set_position(&camera);
        }
        slot.camera = camera;
    }

    /// Renders camera to texture: records the pass that puts the frame `tick` rendered into `view` (present.rs).
    pub fn render_camera(&self, handle: CameraHandle, encoder: &mut wgpu::CommandEncoder, view: &wgpu::TextureView) {
        let slot = self.cameras.get(&handle).unwrap_or_else(|| panic!("camera does not exist: {:?}", handle));
        slot.presenter.record(encoder, view);
    }

    /// Deletes a camera.
    pub fn delete_camera(&mut self, handle: CameraHandle) {
        if let Some(slot) = self.cameras.remove(&handle) {
            check(unsafe { ffi::st_camera_delete(self.raw, slot.raw) });
        }
    }

    /// Sends all changes to the GPU, renders every camera's frame on the MI355X and hands the frames to wgpu.
    pub fn tick(&mut self, device: &wgpu::Device, queue: &wgpu::Queue) {
        self.flush_texture_images(device, queue);
        // a tree deeper than the kernels' traversal stack is uploaded and rendered all the same; the reference indexes past its
        // stack there (strolle-gpu/src/lib.rs:76), this facade says so once and goes on
        let status = unsafe { ffi::st_tick(self.raw, self.stream) };
        if status == ffi::ST_ERR_BVH_TOO_DEEP {
            if !self.warned_deep_bvh {
                self.warned_deep_bvh = true;
                let msg = unsafe { CStr::from_ptr(ffi::st_last_error()) }.to_string_lossy().into_owned();
                warn!("strolle-hip: {msg}");
            }
        } else {
            check(status);
        }
        for slot in self.cameras.values() {
            check(unsafe { ffi::st_render_camera(self.raw, slot.raw, slot.presenter.device_frame(), self.stream) });
            slot.presenter.upload(self.raw, slot.raw, queue, self.stream);
        }
    }

    /// `ImageData::Texture` images (images.rs:187-213 copies them texture-to-texture into the atlas): here the texels have
    /// to leave wgpu — copy to a mappable buffer, wait, hand the bytes to the library. Static images once, dynamic ones at
    /// every tick (that is a device -> host -> device round trip per frame; use `ImageData::Raw` where that matters).
    fn flush_texture_images(&mut self, device: &wgpu::Device, queue: &wgpu::Queue) {
        for (&id, img) in self.texture_images.iter_mut() {
            if !img.pending && !img.is_dynamic {
                continue;
            }
            img.pending = false;
            let (w, h) = img.size;
            let row = (w * 4 + 255) / 256 * 256; // COPY_BYTES_PER_ROW_ALIGNMENT
            let buffer = device.create_buffer(&wgpu::BufferDescriptor {
                label: Some("strolle_hip_image_readback"),
                size: (row * h) as u64,
                usage: wgpu::BufferUsages::COPY_DST | wgpu::BufferUsages::MAP_READ,
                mapped_at_creation: false,
            });
            let mut encoder = device.create_command_encoder(&Default::default());
            encoder.copy_texture_to_buffer(
                wgpu::ImageCopyTexture { texture: &img.texture, mip_level: 0, origin: wgpu::Origin3d::ZERO, aspect: wgpu::TextureAspect::All },
                wgpu::ImageCopyBuffer { buffer: &buffer, layout: wgpu::ImageDataLayout { offset: 0, bytes_per_row: Some(row), rows_per_image: Some(h) } },
                wgpu::Extent3d { width: w, height: h, depth_or_array_layers: 1 },
            );
            queue.submit([encoder.finish()]);
            let slice = buffer.slice(..);
            slice.map_async(wgpu::MapMode::Read, |r| r.expect("map image read-back"));
            device.poll(wgpu::Maintain::Wait);
            let mapped = slice.get_mapped_range();
            let mut texels = Vec::with_capacity((w * h * 4) as usize);
            for y in 0..h as usize {
                texels.extend_from_slice(&mapped[y * row as usize..y * row as usize + (w * 4) as usize]);
            }
            drop(mapped);
            buffer.unmap();
            let srgb = img.texture.format().is_srgb() as i32;
            check(unsafe { ffi::st_image_insert_rgba8(self.raw, id, w, h, texels.as_ptr(), srgb) });
        }
    }

    /// The library's engine handle, for the multi-GPU layer (`dist::Node::join`); valid while this `Engine` lives.
    pub fn raw_handle(&mut self) -> *mut ffi::StEngine {
        self.raw
    }

    /// Not part of the reference's API: selects the bit-exact build of the kernels (for parity work).
    pub fn set_exact_arithmetic(&mut self, exact: bool) {
        check(unsafe { ffi::st_engine_set_arithmetic(self.raw, exact as i32) });
    }

    /// Not part of the reference's API: refit the BVH instead of rebuilding it while instances only move — on the device
    /// (ST_BVH_REFIT_DEVICE = 2: st_tick sends the moved triangles only; same bits as the host refit). `false` goes back to the library's
    /// default (ST_BVH_AUTO = 4: the first tree on the host unless it hangs long leaf runs on large faces (then the device builder's), every later change answered on the device while nothing observes the contract stream).
    pub fn set_bvh_refit(&mut self, refit: bool) {
        check(unsafe { ffi::st_set_bvh_refresh(self.raw, if refit { 2 } else { 4 }) });
    }

    /// Not part of the reference's API: build EVERY tree on the device, the first one too (ST_BVH_BUILD_DEVICE = 3: a spawn costs about half a
    /// millisecond instead of the host rebuild) while nothing observes the contract stream — no BvhHeatmap camera, fast arithmetic. `false`: the default.
    pub fn set_bvh_build_on_device(&mut self, on_device: bool) {
        check(unsafe { ffi::st_set_bvh_refresh(self.raw, if on_device { 3 } else { 4 }) });
    }

    /// Not part of the reference's API: the reference's own behaviour — every scene change rebuilds the binned-SAH tree on the host
    /// (ST_BVH_REBUILD = 0; tens of milliseconds per spawn at 200 k triangles). The library's default is ST_BVH_AUTO.
    pub fn set_bvh_rebuild_on_host(&mut self) {
        check(unsafe { ffi::st_set_bvh_refresh(self.raw, 0) });
    }
}

impl<P> Drop for Engine<P>
where
    P: Params,
{
    fn drop(&mut self) {
        for slot in self.cameras.values() {
            slot.presenter.finish(self.raw, slot.raw);
        }
        self.cameras.clear(); // presenters free their HIP buffers before the engine goes
        unsafe {
            ffi::st_engine_destroy(self.raw);
            ffi::hipStreamDestroy(self.stream);
        }
    }
}
