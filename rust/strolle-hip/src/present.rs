//! Getting a HIP-rendered frame into the `wgpu::TextureView` that `Engine::render_camera` is handed
//! (reference: strolle/src/camera_controller/passes/frame_composition.rs draws a full-screen triangle into the view).
//!
//! The composition KERNEL already produced the pixels — in the viewport's own format (`st_camera_set_output_format`) — in
//! device memory owned by HIP. This module moves them across the API boundary with a staging copy that never joins a
//! stream, so the engine's cross-frame pipelining (st_engine.cpp `render`: frame N+1's primary rays and reservoir passes
//! overlap frame N's denoiser) survives the facade:
//!
//!   tick() of frame N:  st_render_camera -> device frame [N % 2]
//!                       st_camera_present_copy -> page-locked host frame [N % 2]   (asynchronous, on the camera's copy
//!                                                                                    stream, behind frame N only)
//!                       host frame [(N - 1) % 2] — enqueued a whole frame ago — -> queue.write_texture -> `frame` texture
//!   render_camera():    one render pass that copies `frame` into the target view (full-screen triangle, textureLoad)
//!
//! What the view shows at frame N is therefore frame N - 1: ONE FRAME OF LATENCY, the price of not waiting for the GPU
//! inside `tick` (the first tick presents a cleared texture). `st_camera_present_ready` is polled; if frame N - 1's copy has
//! not landed yet (the application ticks faster than the MI355X renders) the call blocks on that copy alone.
//!
//! Rendering moves into `tick` because that is the only call that has the `wgpu::Queue`; bevy-strolle calls
//! `update_camera`, then `tick`, then `render_camera` every frame (stages/prepare.rs:300-340, rendering_node.rs:14-36).
//! The copy costs 8 MB per 1080p frame in an 8-bit swap-chain format (33 MB in Rgba32Float); measured from C with the same
//! call order (`bench.py --present rgba8`). The zero-copy alternative — allocate `frame` as Vulkan
//! external memory, import it with hipImportExternalMemory and let the kernel write into it — needs wgpu-hal escape hatches
//! and is left out.
use std::ffi::c_void;

use std::cell::Cell;

use crate::{ffi, Camera};

const BLIT_WGSL: &str = r#"
@group(0) @binding(0) var frame: texture_2d<f32>;

@vertex
fn vs(@builtin(vertex_index) i: u32) -> @builtin(position) vec4<f32> {
    // one triangle that covers the viewport
    let p = vec2<f32>(f32((i << 1u) & 2u), f32(i & 2u));
    return vec4<f32>(p * 2.0 - 1.0, 0.0, 1.0);
}

struct Origin { x: f32, y: f32 };
var<push_constant> origin: Origin;

@fragment
fn fs(@builtin(position) pos: vec4<f32>) -> @location(0) vec4<f32> {
    return textureLoad(frame, vec2<i32>(i32(pos.x - origin.x), i32(pos.y - origin.y)), 0);
}
"#;

#[derive(Debug)]
pub(crate) struct Presenter {
    bytes_per_pixel: u32,
    size: (u32, u32),
    position: (u32, u32),
    device_frame: [*mut c_void; 2], // HIP allocations the composition kernel writes, alternating by frame
    host_frame: [*mut c_void; 2],   // page-locked staging copies, alternating with them
    ticks: Cell<u64>,               // frames handed to `upload` so far
    frame: wgpu::Texture,
    bind_group: wgpu::BindGroup,
    pipeline: wgpu::RenderPipeline,
}

// the raw pointers are HIP allocations owned by this struct
unsafe impl Send for Presenter {}
unsafe impl Sync for Presenter {}

impl Presenter {
    pub fn new(device: &wgpu::Device, camera: &Camera) -> Self {
        let (_, bytes_per_pixel) = camera.output_format().unwrap_or_else(|| {
            panic!("strolle-hip cannot render into {:?}; supported: Rgba32Float, Rgba16Float, Rgba8UnormSrgb, Bgra8UnormSrgb", camera.viewport.format)
        });
        let (w, h) = (camera.viewport.size.x.max(1), camera.viewport.size.y.max(1));
        let bytes = (w * h * bytes_per_pixel) as usize;
        let (mut device_frame, mut host_frame) = ([std::ptr::null_mut(); 2], [std::ptr::null_mut(); 2]);
        for k in 0..2 {
            unsafe {
                assert_eq!(ffi::hipMalloc(&mut device_frame[k], bytes), 0, "hipMalloc");
                assert_eq!(ffi::hipHostMalloc(&mut host_frame[k], bytes, 0), 0, "hipHostMalloc");
            }
        }
        let frame = device.create_texture(&wgpu::TextureDescriptor {
            label: Some("strolle_hip_frame"),
            size: wgpu::Extent3d { width: w, height: h, depth_or_array_layers: 1 },
            mip_level_count: 1,
            sample_count: 1,
            dimension: wgpu::TextureDimension::D2,
            format: camera.viewport.format,
            usage: wgpu::TextureUsages::COPY_DST | wgpu::TextureUsages::TEXTURE_BINDING,
            view_formats: &[],
        });
        let layout = device.create_bind_group_layout(&wgpu::BindGroupLayoutDescriptor {
            label: Some("strolle_hip_present"),
            entries: &[wgpu::BindGroupLayoutEntry {
                binding: 0,
                visibility: wgpu::ShaderStages::FRAGMENT,
                ty: wgpu::BindingType::Texture { sample_type: wgpu::TextureSampleType::Float { filterable: false }, view_dimension: wgpu::TextureViewDimension::D2, multisampled: false },
                count: None,
            }],
        });
        let bind_group = device.create_bind_group(&wgpu::BindGroupDescriptor {
            label: Some("strolle_hip_present"),
            layout: &layout,
            entries: &[wgpu::BindGroupEntry { binding: 0, resource: wgpu::BindingResource::TextureView(&frame.create_view(&Default::default())) }],
        });
        let shader = device.create_shader_module(wgpu::ShaderModuleDescriptor { label: Some("strolle_hip_present"), source: wgpu::ShaderSource::Wgsl(BLIT_WGSL.into()) });
        let pipeline_layout = device.create_pipeline_layout(&wgpu::PipelineLayoutDescriptor {
            label: Some("strolle_hip_present"),
            bind_group_layouts: &[&layout],
            push_constant_ranges: &[wgpu::PushConstantRange { stages: wgpu::ShaderStages::FRAGMENT, range: 0..8 }],
        });
        let pipeline = device.create_render_pipeline(&wgpu::RenderPipelineDescriptor {
            label: Some("strolle_hip_present"),
            layout: Some(&pipeline_layout),
            vertex: wgpu::VertexState { module: &shader, entry_point: "vs", buffers: &[] },
            primitive: wgpu::PrimitiveState::default(),
            depth_stencil: None,
            multisample: wgpu::MultisampleState::default(),
            fragment: Some(wgpu::FragmentState {
                module: &shader,
                entry_point: "fs",
                targets: &[Some(wgpu::ColorTargetState { format: camera.viewport.format, blend: None, write_mask: wgpu::ColorWrites::ALL })],
            }),
            multiview: None,
        });
        Self {
            bytes_per_pixel,
            size: (w, h),
            position: (camera.viewport.position.x, camera.viewport.position.y),
            device_frame,
            host_frame,
            ticks: Cell::new(0),
            frame,
            bind_group,
            pipeline,
        }
    }

    /// Where `st_render_camera` writes this tick's composed frame.
    pub fn device_frame(&self) -> *mut c_void {
        self.device_frame[(self.ticks.get() & 1) as usize]
    }

    pub fn set_position(&mut self, camera: &Camera) {
        self.position = (camera.viewport.position.x, camera.viewport.position.y);
    }

    /// Call after this tick's `st_render_camera` was enqueued on `stream`: sends the new frame on its way to the host and
    /// hands the PREVIOUS frame — whose copy has had a whole frame to land — to wgpu. No stream is joined.
    pub fn upload(&self, engine: *mut ffi::StEngine, camera: u64, queue: &wgpu::Queue, stream: ffi::hipStream_t) {
        let (w, h) = self.size;
        let bytes = (w * h * self.bytes_per_pixel) as usize;
        let n = self.ticks.get();
        let (cur, prev) = ((n & 1) as usize, ((n + 1) & 1) as usize);
        unsafe {
            assert_eq!(ffi::st_camera_present_copy(engine, camera, self.device_frame[cur], self.host_frame[cur], bytes, stream), ffi::ST_OK, "st_camera_present_copy");
        }
        self.ticks.set(n + 1);
        // The first frame after creation or a resize has no predecessor to show: it is presented SYNCHRONOUSLY (wait for its own copy),
        // so the view never shows a cleared texture; from the second frame on the previous frame is presented while this one renders
        // (one frame of latency, stated in INTEGRATION.md).
        let prev = if n == 0 { cur } else { prev };
        let mut ready = 0;
        unsafe {
            if n == 0 {
                assert_eq!(ffi::st_camera_present_ready(engine, camera, self.host_frame[prev], 1, &mut ready), ffi::ST_OK, "st_camera_present_ready");
            }
            assert_eq!(ffi::st_camera_present_ready(engine, camera, self.host_frame[prev], 0, &mut ready), ffi::ST_OK, "st_camera_present_ready");
            if ready == 0 {
                // the application is ahead of the GPU: wait for that one copy (not for the stream)
                assert_eq!(ffi::st_camera_present_ready(engine, camera, self.host_frame[prev], 1, &mut ready), ffi::ST_OK, "st_camera_present_ready");
            }
        }
        let pixels = unsafe { std::slice::from_raw_parts(self.host_frame[prev] as *const u8, bytes) };
        queue.write_texture(
            wgpu::ImageCopyTexture { texture: &self.frame, mip_level: 0, origin: wgpu::Origin3d::ZERO, aspect: wgpu::TextureAspect::All },
            pixels,
            wgpu::ImageDataLayout { offset: 0, bytes_per_row: Some(w * self.bytes_per_pixel), rows_per_image: Some(h) },
            wgpu::Extent3d { width: w, height: h, depth_or_array_layers: 1 },
        );
    }

    /// Blocks until no copy into this presenter's host frames is in flight (before its buffers are freed).
    pub fn finish(&self, engine: *mut ffi::StEngine, camera: u64) {
        for k in 0..2 {
            let mut ready = 0;
            unsafe {
                assert_eq!(ffi::st_camera_present_ready(engine, camera, self.host_frame[k], 1, &mut ready), ffi::ST_OK, "st_camera_present_ready");
            }
        }
    }

    /// The stand-in for the reference's frame-composition render pass: same target, same viewport rectangle.
    pub fn record(&self, encoder: &mut wgpu::CommandEncoder, view: &wgpu::TextureView) {
        let mut pass = encoder.begin_render_pass(&wgpu::RenderPassDescriptor {
            label: Some("strolle_hip_present"),
            color_attachments: &[Some(wgpu::RenderPassColorAttachment { view, resolve_target: None, ops: wgpu::Operations { load: wgpu::LoadOp::Load, store: true } })],
            depth_stencil_attachment: None,
        });
        let (x, y) = (self.position.0 as f32, self.position.1 as f32);
        pass.set_viewport(x, y, self.size.0 as f32, self.size.1 as f32, 0.0, 1.0);
        pass.set_scissor_rect(self.position.0, self.position.1, self.size.0, self.size.1);
        pass.set_pipeline(&self.pipeline);
        pass.set_bind_group(0, &self.bind_group, &[]);
        pass.set_push_constants(wgpu::ShaderStages::FRAGMENT, 0, &[x.to_ne_bytes(), y.to_ne_bytes()].concat());
        pass.draw(0..3, 0..1);
    }
}

impl Drop for Presenter {
    fn drop(&mut self) {
        // callers run `finish` (or delete the camera, which joins its copy stream) first: no copy is in flight here
        for k in 0..2 {
            unsafe {
                ffi::hipFree(self.device_frame[k]);
                ffi::hipHostFree(self.host_frame[k]);
            }
        }
    }
}
