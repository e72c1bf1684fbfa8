#!/usr/bin/env python3
"""Generates tools/reference_dump/strolle_deterministic_dump.patch against a checkout of Patryk27/strolle.

  python tools/reference_dump/make_patch.py [/path/to/strolle-checkout]        (default /root/reference)

The patch makes the reference renderer reproducible and observable, so that its buffers can be compared with this
repository's CPU oracle (tests/test_reference_dump.py):

  1. Seeds. `rand::thread_rng().gen()` (strolle/src/camera_controller.rs:189-194, passes/ref_tracing.rs:49-53,
     passes/ref_shading.rs:55-59) becomes pass_seed(base, frame, pass_id) — the PCG hash chain of oracle/or_host.h and
     strolle_amd/csrc/st_engine.cpp, with this repository's pass ids; base = $STROLLE_SEED (default 0).
  2. Read-back. Every per-camera texture / storage buffer also gets COPY_SRC usage, and `Engine::dump_camera_buffers`
     (new file strolle/src/dump.rs) copies them into host memory under the names of include/strolle_hip.h's StBufferId.
  3. The example strolle/examples/dump_cornell.rs (copied from tools/reference_dump/dump_cornell.rs) renders the scene that
     export_scene.py wrote and stores the dumps.

The edits are scripted (not a hand-written diff) so that the patch is regenerated, and checked to apply, against the exact
revision under /root/reference: `make_patch.py` copies the touched files, edits the copies, diffs, and dry-runs `patch`.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

# pass ids: strolle_amd/csrc/st_engine.cpp `PassSeedId` (passes that draw no random numbers get 0)
SEED_IDS = {
    "strolle/src/camera_controller/passes/di_sampling.rs": [1],
    "strolle/src/camera_controller/passes/di_temporal_resampling.rs": [2],
    "strolle/src/camera_controller/passes/di_spatial_resampling.rs": [3, 4, 5],        # pick, trace, sample
    "strolle/src/camera_controller/passes/di_resolving.rs": [0],
    "strolle/src/camera_controller/passes/gi_reprojection.rs": [0],
    "strolle/src/camera_controller/passes/gi_sampling.rs": [8, 9],                       # a, b
    "strolle/src/camera_controller/passes/gi_temporal_resampling.rs": [10],
    "strolle/src/camera_controller/passes/gi_spatial_resampling.rs": [11, 12, 13],      # pick, trace, sample
    "strolle/src/camera_controller/passes/gi_preview_resampling.rs": [14],               # one seed for both preview passes
    "strolle/src/camera_controller/passes/frame_denoising.rs": [0, 0, 0],
}

PASS_SEED_RS = '''
/// Deterministic per-pass seeds (tools/reference_dump): the same PCG hash chain as the HIP port and its CPU oracle.
fn seed_hash(v: u32) -> u32 {
    let v = v.wrapping_mul(747796405).wrapping_add(2891336453);
    let w = ((v >> ((v >> 28) + 4)) ^ v).wrapping_mul(277803737);
    (w >> 22) ^ w
}

pub(crate) fn pass_seed(frame: u32, pass_id: u32) -> u32 {
    let base: u64 = std::env::var("STROLLE_SEED").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
    seed_hash((base as u32) ^ seed_hash(((base >> 32) as u32) ^ seed_hash(frame ^ seed_hash(pass_id))))
}
'''

DUMP_RS = '''//! tools/reference_dump: read-back of every per-camera buffer (camera_controller/buffers.rs) under the buffer names of the
//! HIP port's C ABI (include/strolle_hip.h StBufferId), as raw little-endian bytes in the buffers' own layout
//! (Rgba32Float texels row by row; storage buffers as they are).
use crate::{CameraHandle, Engine, Params, StorageBuffer, Texture};

fn read_buffer(device: &wgpu::Device, queue: &wgpu::Queue, src: &wgpu::Buffer, bytes: u64) -> Vec<u8> {
    let staging = device.create_buffer(&wgpu::BufferDescriptor {
        label: Some("dump_staging"),
        size: bytes,
        usage: wgpu::BufferUsages::COPY_DST | wgpu::BufferUsages::MAP_READ,
        mapped_at_creation: false,
    });
    let mut encoder = device.create_command_encoder(&Default::default());
    encoder.copy_buffer_to_buffer(src, 0, &staging, 0, bytes);
    queue.submit([encoder.finish()]);
    let slice = staging.slice(..);
    slice.map_async(wgpu::MapMode::Read, |r| r.expect("map"));
    device.poll(wgpu::Maintain::Wait);
    let out = slice.get_mapped_range().to_vec();
    staging.unmap();
    out
}

fn read_texture(device: &wgpu::Device, queue: &wgpu::Queue, tex: &Texture, w: u32, h: u32) -> Vec<u8> {
    let row = (w * 16 + 255) / 256 * 256; // Rgba32Float rows, padded to COPY_BYTES_PER_ROW_ALIGNMENT
    let staging = device.create_buffer(&wgpu::BufferDescriptor {
        label: Some("dump_staging"),
        size: (row * h) as u64,
        usage: wgpu::BufferUsages::COPY_DST | wgpu::BufferUsages::MAP_READ,
        mapped_at_creation: false,
    });
    let mut encoder = device.create_command_encoder(&Default::default());
    encoder.copy_texture_to_buffer(
        wgpu::ImageCopyTexture { texture: tex.tex(), mip_level: 0, origin: wgpu::Origin3d::ZERO, aspect: wgpu::TextureAspect::All },
        wgpu::ImageCopyBuffer { buffer: &staging, layout: wgpu::ImageDataLayout { offset: 0, bytes_per_row: Some(row), rows_per_image: Some(h) } },
        wgpu::Extent3d { width: w, height: h, depth_or_array_layers: 1 },
    );
    queue.submit([encoder.finish()]);
    let slice = staging.slice(..);
    slice.map_async(wgpu::MapMode::Read, |r| r.expect("map"));
    device.poll(wgpu::Maintain::Wait);
    let mapped = slice.get_mapped_range();
    let mut out = Vec::with_capacity((w * h * 16) as usize);
    for y in 0..h as usize {
        out.extend_from_slice(&mapped[y * row as usize..y * row as usize + (w * 16) as usize]);
    }
    drop(mapped);
    staging.unmap();
    out
}

impl<P> Engine<P>
where
    P: Params,
{
    /// (StBufferId name, bytes) for every buffer of the camera. Double-buffered textures are reported as `_A` / `_B` in
    /// allocation order (`DoubleBuffered::new` creates A, then B; `curr()` is A on even frames).
    pub fn dump_camera_buffers(&self, device: &wgpu::Device, queue: &wgpu::Queue, handle: CameraHandle) -> Vec<(&'static str, Vec<u8>)> {
        let camera = self.cameras.get(handle);
        let b = camera.buffers();
        let size = camera.viewport_size();
        let (w, h) = (size.x, size.y);
        let px = (w * h) as u64;
        let tex = |t: &Texture| read_texture(device, queue, t, w, h);
        let buf = |s: &StorageBuffer, bytes_per_pixel: u64| read_buffer(device, queue, s.buffer(), px * bytes_per_pixel);
        vec![
            ("PRIM_GBUFFER_D0_A", tex(b.prim_gbuffer_d0.get(false))), ("PRIM_GBUFFER_D0_B", tex(b.prim_gbuffer_d0.get(true))),
            ("PRIM_GBUFFER_D1_A", tex(b.prim_gbuffer_d1.get(false))), ("PRIM_GBUFFER_D1_B", tex(b.prim_gbuffer_d1.get(true))),
            ("PRIM_SURFACE_MAP_A", tex(b.prim_surface_map.get(false))), ("PRIM_SURFACE_MAP_B", tex(b.prim_surface_map.get(true))),
            ("REPROJECTION_MAP", tex(&b.reprojection_map)), ("VELOCITY_MAP", tex(&b.velocity_map)),
            ("DI_RESERVOIRS_0", buf(&b.di_reservoirs[0], 32)), ("DI_RESERVOIRS_1", buf(&b.di_reservoirs[1], 32)), ("DI_RESERVOIRS_2", buf(&b.di_reservoirs[2], 32)),
            ("DI_DIFF_SAMPLES", tex(&b.di_diff_samples)), ("DI_DIFF_PREV_COLORS", tex(&b.di_diff_prev_colors)), ("DI_DIFF_CURR_COLORS", tex(&b.di_diff_curr_colors)),
            ("DI_DIFF_MOMENTS_A", tex(b.di_diff_moments.get(false))), ("DI_DIFF_MOMENTS_B", tex(b.di_diff_moments.get(true))),
            ("DI_DIFF_STASH", tex(&b.di_diff_stash)), ("DI_SPEC_SAMPLES", tex(&b.di_spec_samples)),
            ("GI_D0", tex(&b.gi_d0)), ("GI_D1", tex(&b.gi_d1)), ("GI_D2", tex(&b.gi_d2)),
            ("GI_RESERVOIRS_0", buf(&b.gi_reservoirs[0], 64)), ("GI_RESERVOIRS_1", buf(&b.gi_reservoirs[1], 64)),
            ("GI_RESERVOIRS_2", buf(&b.gi_reservoirs[2], 64)), ("GI_RESERVOIRS_3", buf(&b.gi_reservoirs[3], 64)),
            ("GI_DIFF_SAMPLES", tex(&b.gi_diff_samples)), ("GI_DIFF_PREV_COLORS", tex(&b.gi_diff_prev_colors)), ("GI_DIFF_CURR_COLORS", tex(&b.gi_diff_curr_colors)),
            ("GI_DIFF_MOMENTS_A", tex(b.gi_diff_moments.get(false))), ("GI_DIFF_MOMENTS_B", tex(b.gi_diff_moments.get(true))),
            ("GI_DIFF_STASH", tex(&b.gi_diff_stash)), ("GI_SPEC_SAMPLES", tex(&b.gi_spec_samples)),
            ("REF_HITS", buf(&b.ref_hits, 32)), ("REF_RAYS", buf(&b.ref_rays, 48)), ("REF_COLORS", tex(&b.ref_colors)),
        ]
    }
}
'''


def edit(path, fn):
    text = open(path).read()
    new = fn(text)
    assert new != text, f"no change in {path}"
    open(path, "w").write(new)


def sub_once(text, old, new):
    assert text.count(old) == 1, (text.count(old), old)
    return text.replace(old, new)


def main():
    work = tempfile.mkdtemp(prefix="refpatch_")
    a, b = os.path.join(work, "a"), os.path.join(work, "b")
    touched = list(SEED_IDS) + ["strolle/src/camera_controller.rs", "strolle/src/camera_controller/passes/ref_tracing.rs",
                                "strolle/src/camera_controller/passes/ref_shading.rs", "strolle/src/buffers/texture.rs",
                                "strolle/src/buffers/storage_buffer.rs", "strolle/src/buffers/double_buffered.rs", "strolle/src/lib.rs"]
    for rel in touched:
        for root in (a, b):
            os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
            shutil.copy(os.path.join(REF, rel), os.path.join(root, rel))

    # 1. seeds
    for rel, ids in SEED_IDS.items():
        it = iter(ids)
        def repl(text, it=it, n=len(ids), rel=rel):
            assert text.count("camera.pass_params()") == n, (rel, text.count("camera.pass_params()"), n)
            return re.sub(r"camera\.pass_params\(\)", lambda m: f"camera.pass_params({next(it)})", text)
        edit(os.path.join(b, rel), repl)

    def controller(text):
        text = sub_once(text, "use rand::Rng;\n", "")
        text = sub_once(text, """    fn pass_params(&self) -> gpu::PassParams {
        gpu::PassParams {
            seed: rand::thread_rng().gen(),
            frame: self.frame,
        }
    }""", """    fn pass_params(&self, pass_id: u32) -> gpu::PassParams {
        gpu::PassParams {
            seed: pass_seed(self.frame.get(), pass_id),
            frame: self.frame,
        }
    }

    pub(crate) fn buffers(&self) -> &CameraBuffers {
        &self.buffers
    }

    pub(crate) fn viewport_size(&self) -> spirv_std::glam::UVec2 {
        self.camera.viewport.size
    }""")
        return sub_once(text, "#[derive(Debug)]\npub struct CameraController {", PASS_SEED_RS.lstrip("\n") + "\n#[derive(Debug)]\npub struct CameraController {")
    edit(os.path.join(b, "strolle/src/camera_controller.rs"), controller)

    for rel, what in (("strolle/src/camera_controller/passes/ref_tracing.rs", "tracing"), ("strolle/src/camera_controller/passes/ref_shading.rs", "shading")):
        def ref(text):
            text = sub_once(text, "use rand::Rng;\n", "")
            # the HIP port seeds reference shading with 200 + depth (255 = the accumulation pass); tracing draws nothing
            return sub_once(text, "            seed: rand::thread_rng().gen(),", "            seed: crate::camera_controller::pass_seed(camera.frame.get(), 200 + depth as u32),")
        edit(os.path.join(b, rel), ref)

    # 2. read-back
    edit(os.path.join(b, "strolle/src/buffers/texture.rs"), lambda t: sub_once(
        t, '        let usage = usage.expect("Missing property: usage");', '        let usage = usage.expect("Missing property: usage") | wgpu::TextureUsages::COPY_SRC; // tools/reference_dump'))
    def storage(text):
        text = sub_once(text, "            usage: wgpu::BufferUsages::STORAGE,", "            usage: wgpu::BufferUsages::STORAGE | wgpu::BufferUsages::COPY_SRC, // tools/reference_dump")
        return sub_once(text, "    /// Creates an immutable storage-buffer binding:", "    pub fn buffer(&self) -> &wgpu::Buffer {\n        &self.buffer\n    }\n\n    /// Creates an immutable storage-buffer binding:")
    edit(os.path.join(b, "strolle/src/buffers/storage_buffer.rs"), storage)
    db = open(os.path.join(b, "strolle/src/buffers/double_buffered.rs")).read()
    if "pub fn get(" not in db:
        m = re.search(r"impl<T> DoubleBuffered<T>[^{]*\{", db)
        assert m, "DoubleBuffered impl not found"
        fields = re.search(r"pub struct DoubleBuffered<T>\s*\{(.*?)\}", db, re.S).group(1)
        names = re.findall(r"(\w+):\s*T", fields)
        assert len(names) == 2, names
        db = db[:m.end()] + f"\n    /// tools/reference_dump: the first (`alternate == false`) or the second allocation\n    pub fn get(&self, alternate: bool) -> &T {{\n        if alternate {{\n            &self.{names[1]}\n        }} else {{\n            &self.{names[0]}\n        }}\n    }}\n" + db[m.end():]
        open(os.path.join(b, "strolle/src/buffers/double_buffered.rs"), "w").write(db)
    edit(os.path.join(b, "strolle/src/lib.rs"), lambda t: sub_once(t, "mod camera_controllers;\n", "mod camera_controllers;\nmod dump;\n"))
    open(os.path.join(b, "strolle/src/dump.rs"), "w").write(DUMP_RS)
    os.makedirs(os.path.join(b, "strolle/examples"), exist_ok=True)
    shutil.copy(os.path.join(HERE, "dump_cornell.rs"), os.path.join(b, "strolle/examples/dump_cornell.rs"))

    diff = subprocess.run(["diff", "-ruN", "a", "b"], cwd=work, capture_output=True, text=True).stdout
    out = os.path.join(HERE, "strolle_deterministic_dump.patch")
    open(out, "w").write(diff)
    # does it apply to a pristine copy?
    check = os.path.join(work, "check")
    shutil.copytree(a, check)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-d", check, "-i", out], capture_output=True, text=True)
    print(r.stdout[-600:], r.stderr[-300:])
    assert r.returncode == 0, "patch does not apply"
    print(f"wrote {out}: {len(diff.splitlines())} lines, applies cleanly to {REF}")
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
