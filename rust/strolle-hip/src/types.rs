//! The public value types of the `strolle` crate (reference: strolle/src/{camera,light,material,mesh,mesh_triangle,
//! instance,sun,image}.rs), with the same names, fields and constructors, so that `bevy-strolle` and the examples compile
//! against this crate unchanged. What differs is where they go: each type converts into the POD struct of the C ABI
//! (`ffi::St*`) instead of into a `strolle_gpu` uniform.
use std::fmt;

use glam::{Affine3A, Mat4, UVec2, Vec2, Vec3, Vec4};

use crate::{ffi, Params};

// ------------------------------------------------------------------------------------------------ camera (camera.rs)
#[derive(Clone, Debug, Default)]
pub struct Camera {
    pub mode: CameraMode,
    pub viewport: CameraViewport,
    pub transform: Mat4,
    pub projection: Mat4,
}

impl Camera {
    pub(crate) fn to_ffi(&self) -> ffi::StCamera {
        let (mode, denoise, depth) = match self.mode {
            CameraMode::Image { denoise } => (0, denoise, 0),
            CameraMode::DiDiffuse { denoise } => (1, denoise, 0),
            CameraMode::DiSpecular { denoise } => (2, denoise, 0),
            CameraMode::GiDiffuse { denoise } => (3, denoise, 0),
            CameraMode::GiSpecular { denoise } => (4, denoise, 0),
            CameraMode::BvhHeatmap => (5, false, 0),
            CameraMode::Reference { depth } => (6, false, depth as u32),
        };
        ffi::StCamera {
            mode,
            denoise: denoise as u32,
            depth,
            width: self.viewport.size.x,
            height: self.viewport.size.y,
            pos_x: self.viewport.position.x,
            pos_y: self.viewport.position.y,
            _pad: 0,
            transform: self.transform.to_cols_array(),
            projection: self.projection.to_cols_array(),
        }
    }

    /// What `viewport.format` means for the composition kernel (`StOutputFormat`), or `None` for a format this library
    /// cannot write (the reference accepts whatever wgpu can render to).
    pub(crate) fn output_format(&self) -> Option<(i32, u32)> {
        use wgpu::TextureFormat as F;
        match self.viewport.format {
            F::Rgba32Float => Some((ffi::ST_FORMAT_RGBA32F, 16)),
            F::Rgba16Float => Some((ffi::ST_FORMAT_RGBA16F, 8)),
            F::Rgba8UnormSrgb => Some((ffi::ST_FORMAT_RGBA8_UNORM_SRGB, 4)),
            F::Bgra8UnormSrgb => Some((ffi::ST_FORMAT_BGRA8_UNORM_SRGB, 4)),
            _ => None,
        }
    }
}

impl fmt::Display for Camera {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        let v = &self.viewport;
        write!(f, "pos={}x{}, size={}x{}, format={:?}", v.position.x, v.position.y, v.size.x, v.size.y, v.format)
    }
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum CameraMode {
    /// The final composed image (default)
    Image { denoise: bool },
    /// Direct diffuse lighting only
    DiDiffuse { denoise: bool },
    /// Direct specular lighting only
    DiSpecular { denoise: bool },
    /// Indirect diffuse lighting only
    GiDiffuse { denoise: bool },
    /// Indirect specular lighting only
    GiSpecular { denoise: bool },
    /// Heatmap of the BVH traversal cost
    BvhHeatmap,
    /// Brute-force path-traced reference; slow
    Reference { depth: u8 },
}

impl Default for CameraMode {
    fn default() -> Self {
        Self::Image { denoise: true }
    }
}

#[derive(Clone, Debug)]
pub struct CameraViewport {
    pub format: wgpu::TextureFormat,
    pub size: UVec2,
    pub position: UVec2,
}

impl Default for CameraViewport {
    fn default() -> Self {
        Self { format: wgpu::TextureFormat::Rgba8UnormSrgb, size: UVec2::ZERO, position: UVec2::ZERO }
    }
}

/// Handle of a camera created with `Engine::create_camera` (camera_controllers.rs).
#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
pub struct CameraHandle(pub(crate) usize);

// ------------------------------------------------------------------------------------------------ light (light.rs)
#[derive(Clone, Debug)]
pub enum Light {
    Point { position: Vec3, radius: f32, color: Vec3, range: f32 },
    Spot { position: Vec3, radius: f32, color: Vec3, range: f32, direction: Vec3, angle: f32 },
}

impl Light {
    pub(crate) fn to_ffi(&self) -> ffi::StLight {
        match *self {
            Light::Point { position, radius, color, range } => ffi::StLight {
                kind: ffi::ST_LIGHT_POINT,
                position: position.to_array(),
                radius,
                color: color.to_array(),
                range,
                direction: [0.0; 3],
                angle: 0.0,
            },
            Light::Spot { position, radius, color, range, direction, angle } => ffi::StLight {
                kind: ffi::ST_LIGHT_SPOT,
                position: position.to_array(),
                radius,
                color: color.to_array(),
                range,
                direction: direction.to_array(),
                angle,
            },
        }
    }
}

// ------------------------------------------------------------------------------------------------ material (material.rs)
#[derive(Clone, Debug)]
pub struct Material<P: Params> {
    pub base_color: Vec4,
    pub base_color_texture: Option<P::ImageHandle>,
    pub emissive: Vec4,
    pub emissive_texture: Option<P::ImageHandle>,
    pub perceptual_roughness: f32,
    pub metallic: f32,
    pub metallic_roughness_texture: Option<P::ImageHandle>,
    pub reflectance: f32,
    pub ior: f32,
    pub normal_map_texture: Option<P::ImageHandle>,
    pub alpha_mode: AlphaMode,
}

impl<P: Params> Default for Material<P> {
    fn default() -> Self {
        Self {
            base_color: Vec4::ONE,
            base_color_texture: None,
            emissive: Vec4::ZERO,
            emissive_texture: None,
            perceptual_roughness: 0.5,
            metallic: 0.0,
            metallic_roughness_texture: None,
            reflectance: 0.5,
            ior: 1.0,
            normal_map_texture: None,
            alpha_mode: AlphaMode::default(),
        }
    }
}

/// Whether a material may be (partially) transparent.
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub enum AlphaMode {
    #[default]
    Opaque,
    Blend,
}

// ------------------------------------------------------------------------------------------------ mesh (mesh.rs, mesh_triangle.rs)
#[derive(Clone, Debug)]
pub struct Mesh {
    triangles: Vec<MeshTriangle>,
}

impl Mesh {
    pub fn new(triangles: Vec<MeshTriangle>) -> Self {
        Self { triangles }
    }

    pub(crate) fn triangles(&self) -> &[MeshTriangle] {
        &self.triangles
    }
}

#[derive(Clone, Debug, Default)]
pub struct MeshTriangle {
    positions: [Vec3; 3],
    normals: [Vec3; 3],
    uvs: [Vec2; 3],
    tangents: [Vec4; 3],
}

impl MeshTriangle {
    pub fn with_positions(mut self, positions: [impl Into<Vec3>; 3]) -> Self {
        self.positions = positions.map(Into::into);
        self
    }
    pub fn with_normals(mut self, normals: [impl Into<Vec3>; 3]) -> Self {
        self.normals = normals.map(Into::into);
        self
    }
    pub fn with_uvs(mut self, uvs: [impl Into<Vec2>; 3]) -> Self {
        self.uvs = uvs.map(Into::into);
        self
    }
    pub fn with_tangents(mut self, tangents: [impl Into<Vec4>; 3]) -> Self {
        self.tangents = tangents.map(Into::into);
        self
    }
    pub fn positions(&self) -> [Vec3; 3] {
        self.positions
    }
    pub fn normals(&self) -> [Vec3; 3] {
        self.normals
    }
    pub fn uvs(&self) -> [Vec2; 3] {
        self.uvs
    }

    /// Object-space triangle as the C ABI takes it; world-space baking (instances.rs:100-139) happens inside the library.
    pub(crate) fn to_ffi(&self) -> ffi::StMeshTriangle {
        ffi::StMeshTriangle {
            positions: self.positions.map(|v| v.to_array()),
            normals: self.normals.map(|v| v.to_array()),
            uvs: self.uvs.map(|v| v.to_array()),
            tangents: self.tangents.map(|v| v.to_array()),
        }
    }
}

// ------------------------------------------------------------------------------------------------ instance (instance.rs)
#[derive(Debug)]
pub struct Instance<P: Params> {
    pub(crate) mesh_handle: P::MeshHandle,
    pub(crate) material_handle: P::MaterialHandle,
    pub(crate) transform: Affine3A,
}

impl<P: Params> Instance<P> {
    pub fn new(mesh_handle: P::MeshHandle, material_handle: P::MaterialHandle, transform: Affine3A) -> Self {
        Self { mesh_handle, material_handle, transform }
    }

    /// glam `Affine3A` as the 12 floats `st_instance_insert` takes: x, y, z axes, then the translation.
    pub(crate) fn xform12(&self) -> [f32; 12] {
        let m = self.transform.matrix3;
        let t = self.transform.translation;
        [m.x_axis.x, m.x_axis.y, m.x_axis.z, m.y_axis.x, m.y_axis.y, m.y_axis.z, m.z_axis.x, m.z_axis.y, m.z_axis.z, t.x, t.y, t.z]
    }
}

// ------------------------------------------------------------------------------------------------ sun (sun.rs)
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct Sun {
    pub azimuth: f32,
    pub altitude: f32,
}

impl Default for Sun {
    fn default() -> Self {
        Self { azimuth: 0.0, altitude: 0.35 }
    }
}

// ------------------------------------------------------------------------------------------------ image (image.rs)
#[derive(Debug)]
pub struct Image<P: Params> {
    pub(crate) data: ImageData<P>,
    pub(crate) texture_descriptor: wgpu::TextureDescriptor<'static>,
    pub(crate) _sampler_descriptor: wgpu::SamplerDescriptor<'static>,
}

impl<P: Params> Image<P> {
    pub fn new(data: ImageData<P>, texture_descriptor: wgpu::TextureDescriptor<'static>, sampler_descriptor: wgpu::SamplerDescriptor<'static>) -> Self {
        assert_eq!(texture_descriptor.dimension, wgpu::TextureDimension::D2);
        Self { data, texture_descriptor, _sampler_descriptor: sampler_descriptor }
    }
}

#[derive(Debug)]
pub enum ImageData<P: Params> {
    /// RGBA8 texels in host memory
    Raw { data: Vec<u8> },
    /// A texture that lives on the wgpu device; `is_dynamic` = its contents change and must be re-read every tick
    Texture { texture: P::ImageTexture, is_dynamic: bool },
}
