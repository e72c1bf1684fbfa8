"""GPU parity tests: every buffer the HIP path writes must equal the CPU oracle's bit for bit
(NaN == NaN), frame after frame, through the C ABI. Sizes are chosen so the oracle runs in seconds."""
import os

import numpy as np
import pytest

from oracle_binding import OracleEngine
from parity import assert_bits_equal, bits_equal_mask
from strolle_amd import Buffer, CameraMode, Engine, OutputFormat, StrolleError, scenes

pytestmark = pytest.mark.gpu

ALL_FLOAT_BUFFERS = [b for b in Buffer if b != Buffer.DBG_USED_MEMORY]


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product has no CPU fallback"
    return torch


def _pair(build, size, mode, depth=0, denoise=True, seed=11, camera_fn=scenes.cornell_camera):
    prod, orac = Engine(device=0, exact=True), OracleEngine()
    for e in (prod, orac):
        build(e)
        e.set_seed(seed)
    desc = camera_fn(size, mode, denoise=denoise, depth=depth)
    return prod, orac, desc, prod.create_camera(desc), orac.create_camera(desc)


def _step(torch, prod, orac, desc, cp, co, out):
    prod.update_camera(cp, desc); orac.update_camera(co, desc)
    prod.tick(); orac.tick()
    prod.render_camera(cp, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    ref = orac.render_camera(co)
    torch.cuda.synchronize()
    return out.cpu().numpy(), ref


def _compare_all(prod, orac, cp, co, frame, buffers=ALL_FLOAT_BUFFERS):
    for b in buffers:
        assert_bits_equal(prod.read_buffer(cp, b), orac.read_buffer(co, b), f"frame {frame} buffer {b.name}")


@pytest.mark.parametrize("scene,size", [("cornell", (256, 256)), ("cornell", (333, 200)), ("soup", (200, 120)), ("dungeon", (320, 180)), ("dungeon134k", (320, 180))])
def test_bvh_heatmap_used_memory_bit_exact(scene, size):
    torch = _torch()
    build = {"cornell": scenes.build_cornell, "soup": lambda e: scenes.build_random_soup(e, 3000, seed=5), "dungeon": scenes.build_dungeon,
             "dungeon134k": lambda e: scenes.build_dungeon(e, subdivide=2)}[scene]  # synthetic ~100k-triangle stand-in (BASELINE.json config 3)
    cam = scenes.dungeon_camera if scene.startswith("dungeon") else scenes.cornell_camera
    prod, orac, desc, cp, co = _pair(build, size, CameraMode.BVH_HEATMAP, camera_fn=cam)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    img, ref = _step(torch, prod, orac, desc, cp, co, out)
    got = prod.read_buffer(cp, Buffer.DBG_USED_MEMORY); want = orac.read_buffer(co, Buffer.DBG_USED_MEMORY)
    assert np.array_equal(got, want), f"{(got != want).sum()} heatmap integers differ"
    assert_bits_equal(img, ref, "heatmap colours")
    assert prod.ray_count(cp) == orac.ray_count(co) == size[0] * size[1]


@pytest.mark.parametrize("scene", ["cornell", "soup"])
def test_reference_mode_bit_exact(scene):
    torch = _torch()
    build = scenes.build_cornell if scene == "cornell" else (lambda e: scenes.build_random_soup(e, 2000, seed=9))
    size = (192, 128)
    prod, orac, desc, cp, co = _pair(build, size, CameraMode.REFERENCE, depth=2)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(4):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame, [Buffer.REF_HITS, Buffer.REF_RAYS, Buffer.REF_COLORS])
        assert_bits_equal(img, ref, f"reference frame {frame}")
    assert prod.ray_count(cp) == orac.ray_count(co)


@pytest.mark.parametrize("scene,size,frames", [("cornell", (160, 96), 14), ("cornell", (203, 77), 7), ("soup", (128, 96), 8)])
def test_image_mode_every_buffer_bit_exact(scene, size, frames):
    """ReSTIR DI + GI + SVGF over more than two 6-frame GI cycles: all 35 planes after every frame."""
    torch = _torch()
    build = scenes.build_cornell if scene == "cornell" else (lambda e: scenes.build_random_soup(e, 1500, seed=2, n_lights=5))
    prod, orac, desc, cp, co = _pair(build, size, CameraMode.IMAGE)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(frames):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"image frame {frame}")
    assert prod.ray_count(cp) == orac.ray_count(co)


@pytest.mark.parametrize("mode", [CameraMode.DI_DIFFUSE, CameraMode.DI_SPECULAR, CameraMode.GI_DIFFUSE, CameraMode.GI_SPECULAR])
@pytest.mark.parametrize("denoise", [True, False])
def test_partial_modes_bit_exact(mode, denoise):
    torch = _torch()
    size = (96, 64)
    prod, orac, desc, cp, co = _pair(lambda e: scenes.build_random_soup(e, 800, seed=4), size, mode, denoise=denoise)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(7):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        assert_bits_equal(img, ref, f"{mode.name} denoise={denoise} frame {frame}")


def test_moving_camera_and_light_bit_exact():
    """Reprojection, velocity and the prev-light path (lights.rs commit/rollback) under motion."""
    torch = _torch()
    size = (160, 96)
    prod, orac = Engine(device=0, exact=True), OracleEngine()
    for e in (prod, orac):
        scenes.build_cornell(e); e.set_seed(3)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    cp, co = prod.create_camera(desc), orac.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    from strolle_amd import Light
    import math
    for frame in range(9):
        t = 0.05 * frame
        desc = scenes.camera_for(size, (0.3 * math.sin(t), 1.0 + 0.1 * t, 3.2 - 0.2 * t), (0.0, 1.0, 0.0))
        light = Light.point((math.sin(t) / 2.0, 1.5, math.cos(t) / 2.0), 0.15, (50.0 / (4 * math.pi),) * 3, 20.0)
        for e in (prod, orac):
            e.insert_light(1, light)
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"moving frame {frame}")


def _full_size_run(torch, build, camera_fn, size, mode, frames, all_planes_at, seed=5):
    """Exact build vs oracle at a BASELINE.json configuration's stated size: the composed frame after every frame, every
    per-camera plane at the frames listed in `all_planes_at` (the last one included: it carries the whole temporal state —
    reservoirs, history colours, moments — so a divergence in any earlier frame would show there), and the ray counts."""
    prod, orac, desc, cp, co = _pair(build, size, mode, seed=seed, camera_fn=camera_fn)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(frames):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        assert_bits_equal(img, ref, f"{size[0]}x{size[1]} {mode.name} frame {frame}")
        if frame in all_planes_at:
            _compare_all(prod, orac, cp, co, frame, list(Buffer))
    assert prod.ray_count(cp) == orac.ray_count(co)
    prod.close(); orac.close()


def test_config2_cornell_1080p_image_14_frames_bit_exact():
    """BASELINE.json config 2 at its stated size — Cornell 1920x1080, Image{denoise:true} (camera_controller.rs:113-172, the
    headline configuration) — over 14 frames (more than two 6-frame GI cycles, frame.rs:19-21)."""
    _full_size_run(_torch(), scenes.build_cornell, scenes.cornell_camera, (1920, 1080), CameraMode.IMAGE, 14, {5, 13})


def test_config3_dungeon_1080p_gi_diffuse_8_frames_bit_exact():
    """BASELINE.json config 3 at its stated size — dungeon 1920x1080, GiDiffuse{denoise:true}; its BVH-heatmap check is
    test_dungeon_heatmap_1080p_bit_exact."""
    _full_size_run(_torch(), scenes.build_dungeon, scenes.dungeon_camera, (1920, 1080), CameraMode.GI_DIFFUSE, 8, {7})


def test_config5_dungeon_4k_image_4_frames_bit_exact():
    """BASELINE.json config 5's per-frame work at its stated size — dungeon 3840x2160, Image{denoise:true} — on one GPU
    (the 8-way split of the same frame is test_row_bands_*)."""
    _full_size_run(_torch(), scenes.build_dungeon, scenes.dungeon_camera, (3840, 2160), CameraMode.IMAGE, 4, {3})


def test_full_size_ray_budget_and_determinism_1080p():
    """Size-independent properties at the headline size, in the DEFAULT (fast) build: two engines with the same seed give
    identical bits, every frame is finite, and the ray budget is N <= rays <= 5N per frame."""
    torch = _torch()
    size = (1920, 1080)
    n = size[0] * size[1]
    outs = []
    for rep in range(2):
        prod = Engine(device=0)
        scenes.build_cornell(prod); prod.set_seed(5)
        desc = scenes.cornell_camera(size, CameraMode.IMAGE)
        cp = prod.create_camera(desc)
        out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
        for frame in range(8):
            prod.update_camera(cp, desc); prod.tick()
            prod.ray_count(cp, reset=True)
            prod.render_camera(cp, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            rays = prod.ray_count(cp)
            assert n <= rays <= 5 * n, (frame, rays, n)
        img = out.cpu().numpy()
        assert np.isfinite(img).all()
        outs.append(img)
        prod.close()
    assert_bits_equal(outs[0], outs[1], "two identical runs")
    prod, orac, desc, cp, co = _pair(scenes.build_cornell, size, CameraMode.BVH_HEATMAP)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    _step(torch, prod, orac, desc, cp, co, out)
    assert np.array_equal(prod.read_buffer(cp, Buffer.DBG_USED_MEMORY), orac.read_buffer(co, Buffer.DBG_USED_MEMORY))


def test_errors_are_loud():
    from strolle_amd import StrolleError
    prod = Engine(device=0, exact=True)
    with pytest.raises(StrolleError):
        prod.render_camera(12345, 0, 0)       # unknown camera: the reference panics (camera_controllers.rs:21-34)
    host_only = Engine(device=-1)
    scenes.build_cornell(host_only)
    cam = host_only.create_camera(scenes.cornell_camera((64, 64)))
    host_only.tick()
    with pytest.raises(StrolleError):
        host_only.render_camera(cam, 0, 0)    # no device => loud failure, never a CPU path


def test_dungeon_image_mode_bit_exact():
    """BASELINE.json config 3's scene (level.glb + three tori: 13,001 triangles, 48 materials): atlas sampling,
    multi-triangle leaves, six lights through the 16-sample RIS."""
    torch = _torch()
    size = (160, 96)
    prod, orac, desc, cp, co = _pair(scenes.build_dungeon, size, CameraMode.IMAGE, camera_fn=scenes.dungeon_camera)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(8):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"dungeon frame {frame}")
    assert float(ref[..., :3].mean()) > 0.0


def test_dungeon_heatmap_1080p_bit_exact():
    """Config 3's check at its full size: BVH-heatmap integer counts, bit-exact."""
    torch = _torch()
    size = (1920, 1080)
    prod, orac, desc, cp, co = _pair(scenes.build_dungeon, size, CameraMode.BVH_HEATMAP, camera_fn=scenes.dungeon_camera)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    _step(torch, prod, orac, desc, cp, co, out)
    assert np.array_equal(prod.read_buffer(cp, Buffer.DBG_USED_MEMORY), orac.read_buffer(co, Buffer.DBG_USED_MEMORY))


def _render_bands(torch, build, size, mode, depth, frames, n_bands, apron, camera_fn=scenes.cornell_camera, exact=None):
    """Emulates the multi-GPU partition on one GPU: one engine per band, each restricted to its row window."""
    from strolle_amd.distributed import assemble_bands_numpy, band_for_rank, render_window
    w, h = size
    full = []
    for r in range(n_bands):
        e = Engine(device=0, exact=(mode != CameraMode.IMAGE) if exact is None else exact)
        build(e); e.set_seed(21)
        desc = camera_fn(size, mode, depth=depth)
        cam = e.create_camera(desc)
        y0, y1 = render_window(h, band_for_rank(h, n_bands, r), apron)
        e.set_camera_rows(cam, y0, y1)
        out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        for _ in range(frames):
            e.update_camera(cam, desc); e.tick()
            e.render_camera(cam, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        full.append(out.cpu().numpy())
        e.close()
    return assemble_bands_numpy(full, h, w)


@pytest.mark.parametrize("mode,depth", [(CameraMode.BVH_HEATMAP, 0), (CameraMode.REFERENCE, 1)])
def test_row_bands_reproduce_the_single_gpu_frame_exactly(mode, depth):
    """Reference/Heatmap touch only their own pixel and key the RNG on absolute coordinates: a tiled frame is bit-identical."""
    torch = _torch()
    size = (256, 200)
    single = _render_bands(torch, scenes.build_cornell, size, mode, depth, 3, 1, 0)
    tiled = _render_bands(torch, scenes.build_cornell, size, mode, depth, 3, 4, 0)
    assert_bits_equal(tiled, single, "4 row bands vs 1")


@pytest.mark.parametrize("bands", [4, 8])
def test_row_bands_image_mode_seam_psnr(bands):
    """Image mode has cross-band taps (spatial resampling, a-trous, reprojection). With the SHIPPED partition — `bands` row
    bands, a 16-row apron (bench.py --apron default) — the assembled frame must stay within BASELINE.json's 40 dB of the
    single-GPU frame after 14 frames (both in the default fast build; 1080 rows so that a band of 8 is 135 rows as on the node)."""
    from parity import psnr
    torch = _torch()
    size = (480, 1080)
    single = _render_bands(torch, scenes.build_cornell, size, CameraMode.IMAGE, 0, 14, 1, 0)
    tiled = _render_bands(torch, scenes.build_cornell, size, CameraMode.IMAGE, 0, 14, bands, 16)
    value = psnr(np.clip(tiled[..., :3], 0, 1), np.clip(single[..., :3], 0, 1))
    print(f"seam PSNR ({bands} bands, apron 16) =", value)
    assert value >= 40.0, value


def _tiled_ranks(torch, build, size, mode, depth, world, cols, apron, group, camera_fn=scenes.cornell_camera, exact=True, seed=21):
    """`world` engines on ONE device joined through the in-process transport of st_dist_* (same partition, pack / unpack and stream
    ordering as the RCCL path): returns [(engine, camera, frame tensor, desc)]."""
    ranks = []
    for r in range(world):
        e = Engine(device=0, exact=exact)
        build(e); e.set_seed(seed)
        desc = camera_fn(size, mode, depth=depth)
        cam = e.create_camera(desc)
        e.dist_init_local(r, world, group)
        e.dist_set_partition(cam, cols=cols, apron=apron)
        ranks.append((e, cam, torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0"), desc))
    return ranks


@pytest.mark.parametrize("world,cols", [(4, 0), (8, 0), (2, 0), (4, 4)])
def test_tiles_gathered_through_the_c_abi_reproduce_the_single_gpu_frame_exactly(world, cols):
    """BASELINE.json config 4's shape — Reference mode, 2 x 2 tiles (also 4 x 2, bands and column strips) — through st_dist_set_partition
    + st_dist_gather on device engines: the frame rank 0 assembles equals the single-engine frame bit for bit, over 3 accumulated frames,
    with the gather of frame N in flight while frame N+1 renders into the other buffer."""
    torch = _torch()
    size = (272, 200)   # tile edges that are not multiples of the frame's halves: st_dist_partition rounds them onto its 16 x 8 grid
    single = _render_bands(torch, scenes.build_cornell, size, CameraMode.REFERENCE, 1, 3, 1, 0)
    ranks = _tiled_ranks(torch, scenes.build_cornell, size, CameraMode.REFERENCE, 1, world, cols, 0, group=4000 + world * 8 + cols)
    alt = [torch.zeros_like(ranks[0][2]) for _ in ranks]
    full = [torch.zeros_like(ranks[0][2]), torch.zeros_like(ranks[0][2])]
    stream = torch.cuda.current_stream().cuda_stream
    for f in range(3):
        for r in range(world - 1, -1, -1):     # in-process transport: rank 0 last
            e, cam, out, desc = ranks[r]
            target = out if f % 2 == 0 else alt[r]
            e.update_camera(cam, desc); e.tick(stream)
            e.render_camera(cam, target.data_ptr(), stream)
            e.dist_gather(cam, target.data_ptr(), full[f % 2].data_ptr() if r == 0 else 0, stream)
    ranks[0][0].dist_wait(ranks[0][1], host=True)
    torch.cuda.synchronize()
    assert_bits_equal(full[0].cpu().numpy(), single, f"{world} tiles ({cols or 'default'} columns) vs one engine")
    assert ranks[0][0].dist_gather_ms(ranks[0][1]) >= 0.0
    for e, *_ in ranks:
        e.close()


def test_lean_frame_planes_are_flagged_stale_instead_of_read_silently():
    """the fast build's lean frame leaves seven planes unwritten; st_camera_buffer_stale says which, and a strict read-back
    refuses them instead of returning an earlier launch's content. With every plane kept (or in the exact build) nothing is stale."""
    torch = _torch()
    size = (128, 96)
    e = Engine(device=0)
    scenes.build_cornell(e); e.set_seed(5)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    cam = e.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    lean = (Buffer.VELOCITY_MAP, Buffer.PRIM_SURFACE_MAP_A, Buffer.DI_DIFF_SAMPLES, Buffer.GI_DIFF_SAMPLES, Buffer.DI_DIFF_CURR_COLORS, Buffer.GI_DIFF_CURR_COLORS)
    for _ in range(3):
        e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert all(e.buffer_stale(cam, b) for b in lean) and not e.buffer_stale(cam, Buffer.PRIM_GBUFFER_D0_A) and not e.buffer_stale(cam, Buffer.DI_RESERVOIRS_0)
    with pytest.raises(StrolleError, match="lean frame"):
        e.read_buffer(cam, Buffer.VELOCITY_MAP, strict=True)
    e.read_buffer(cam, Buffer.VELOCITY_MAP)             # the default read still returns what is there
    e.keep_all_planes(True)
    e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert not any(e.buffer_stale(cam, Buffer(i)) for i in range(35))
    e.read_buffer(cam, Buffer.VELOCITY_MAP, strict=True)
    e.close()


def test_rccl_transport_binds_and_runs_with_one_rank():
    """The RCCL transport of st_dist_* on a real device as far as ONE GPU allows: librccl is found (the copy torch already loaded is
    shared), ncclGetUniqueId / ncclCommInitRank take the 128-byte id by value through the dlsym'd prototypes, a one-rank communicator
    comes up on the engine's device, and st_dist_gather assembles the (single) tile on the engine's communication stream. Two ranks
    need two GPUs: that is the driver's scaling run (bench.py --gpus N calls exactly these entry points)."""
    from strolle_amd.api import dist_unique_id
    torch = _torch()
    uid = dist_unique_id()
    assert len(uid) == 128 and any(uid)
    e = Engine(device=0, exact=True)
    scenes.build_cornell(e); e.set_seed(3)
    size = (160, 96)
    desc = scenes.cornell_camera(size, CameraMode.REFERENCE, depth=1)
    cam = e.create_camera(desc)
    e.dist_init(0, 1, uid)
    owned, window = e.dist_set_partition(cam, apron=16)
    assert owned == window == (0, 0, 160, 96)
    out = torch.zeros((96, 160, 4), dtype=torch.float32, device="cuda:0"); full = torch.zeros_like(out)
    stream = torch.cuda.current_stream().cuda_stream
    e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream)
    e.dist_gather(cam, out.data_ptr(), full.data_ptr(), stream)
    e.dist_wait(cam, host=True)
    torch.cuda.synchronize()
    assert bool((full == out).all()) and float(out[..., :3].abs().sum()) > 0.0
    e.dist_shutdown(); e.close()


def test_image_mode_tiles_under_motion_hold_40_db_for_120_frames():
    """VERDICT r3 item 3 / SURVEY 8(e): seams "drift through temporal history". Config 5's partition AT CONFIG 5's SIZE — 3840 x 2160
    in 8 tiles (4 x 2) of 960 x 1080, apron 16 — against the single-engine frame over 130 frames with the light orbiting as
    cornell.rs animates it and the camera orbiting the box: every one of the LAST 30 frames must stay within BASELINE.json's 40 dB
    (fast build on both sides, as shipped). Measured (profiles/r04_seam_motion.txt, tools/seam_motion_sweep.py): 45.0 dB worst here,
    59.7 dB on the moving dungeon; the apron is a property of the TILE size — the same 8 tiles on a 1280 x 720 frame (320 x 360 each)
    give 38.3 dB with apron 16 and need 32 for 40.2 dB, which is why bench.py --apron is a flag."""
    import math
    from parity import psnr
    from strolle_amd import Light
    from strolle_amd.api import dist_partition
    torch = _torch()
    size, world, frames = (3840, 2160), 8, 130
    ranks = _tiled_ranks(torch, scenes.build_cornell, size, CameraMode.IMAGE, 0, world, 0, 16, group=4100, exact=False)
    ref = Engine(device=0)
    scenes.build_cornell(ref); ref.set_seed(21)
    rdesc = scenes.cornell_camera(size, CameraMode.IMAGE)
    rcam = ref.create_camera(rdesc)
    rout = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    full = torch.zeros_like(rout)
    stream = torch.cuda.current_stream().cuda_stream
    worst = 1e9
    for f in range(frames):
        t = f / 60.0
        light = Light.point((math.sin(t) / 2.0, 1.5, math.cos(t) / 2.0), 0.15, (50.0 / (4.0 * math.pi),) * 3, 20.0)
        a = 0.1 * t
        desc = scenes.camera_for(size, (3.2 * math.sin(a), 1.0, 3.2 * math.cos(a)), (0.0, 1.0, 0.0), CameraMode.IMAGE, True, 0)
        for e, cam, out in [(ref, rcam, rout)] + [(e, cam, out) for e, cam, out, _ in reversed(ranks)]:
            e.insert_light(1, light); e.update_camera(cam, desc); e.tick(stream)
            e.render_camera(cam, out.data_ptr(), stream)
        if f >= frames - 30:
            for r in range(world - 1, -1, -1):
                e, cam, out, _ = ranks[r]
                e.dist_gather(cam, out.data_ptr(), full.data_ptr() if r == 0 else 0, stream)
            ranks[0][0].dist_wait(ranks[0][1], host=True)
            torch.cuda.synchronize()
            value = psnr(np.clip(full.cpu().numpy()[..., :3], 0, 1), np.clip(rout.cpu().numpy()[..., :3], 0, 1))
            worst = min(worst, value)
    print(f"worst seam PSNR over the last 30 of {frames} frames (8 tiles, apron 16, light + camera moving) = {worst:.2f} dB")
    assert worst >= 40.0, worst
    ref.close()
    for e, *_ in ranks:
        e.close()


def test_atmosphere_luts_and_daylight_bit_exact():
    """the three LUT-generation kernels (transmittance, multi-scattering, sky view) against the
    oracle, texel for texel (f16-rounded), then a daylight dungeon frame sequence that samples them."""
    torch = _torch()
    from strolle_amd import Sun
    size = (160, 96)

    def build(e):
        scenes.build_dungeon(e)
        e.update_sun(Sun(azimuth=0.6, altitude=0.5))
    prod, orac, desc, cp, co = _pair(build, size, CameraMode.IMAGE, camera_fn=scenes.dungeon_camera)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(7):
        if frame == 4:  # the sky LUT is regenerated when the sun's altitude changes (passes/atmosphere.rs:98-109)
            for e in (prod, orac):
                e.update_sun(Sun(azimuth=0.7, altitude=0.2))
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        if frame in (0, 4):
            for what, name in enumerate(("transmittance", "scattering", "sky")):
                lut = orac.read_lut(what)
                assert np.isfinite(lut).all() and lut[..., :3].max() > 0
                assert_bits_equal(prod.read_lut(what), lut, f"{name} LUT at frame {frame}")
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"daylight frame {frame}")


def test_cornell_4k_reference_4spp_bit_exact():
    """BASELINE.json config 4 on one GPU: Cornell 3840x2160, Reference{depth:1}, 4 frames accumulated with a static camera
    (ref_shading.rs:53-67) — every pixel of the accumulated frame equals the oracle's."""
    torch = _torch()
    size = (3840, 2160)
    prod, orac, desc, cp, co = _pair(scenes.build_cornell, size, CameraMode.REFERENCE, depth=1)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(4):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
    assert_bits_equal(img, ref, "4K reference, 4 spp")
    assert np.all(prod.read_buffer(cp, Buffer.REF_COLORS).reshape(size[1], size[0], 4)[..., 3] == 4.0)
    assert prod.ray_count(cp) == orac.ray_count(co)


@pytest.mark.parametrize("scene", ["cornell", "dungeon"])
def test_back_to_back_frames_without_sync_bit_exact(scene):
    """The engine software-pipelines consecutive frames over two HIP streams (GI chain of frame N+1 under the denoiser of
    frame N). Enqueue many frames with no host synchronisation in between, then compare EVERY plane and the composed frame
    with the oracle: any missing cross-frame dependency shows up as a mismatch."""
    torch = _torch()
    size = (960, 540)
    frames = 20
    build = scenes.build_cornell if scene == "cornell" else scenes.build_dungeon
    cam_fn = scenes.cornell_camera if scene == "cornell" else scenes.dungeon_camera
    finals = []
    for rep in range(2):
        prod = Engine(device=0, exact=True)
        build(prod); prod.set_seed(31)
        desc = cam_fn(size, CameraMode.IMAGE)
        cp = prod.create_camera(desc)
        out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(frames):
            prod.update_camera(cp, desc); prod.tick(stream)
            prod.render_camera(cp, out.data_ptr(), stream)
        torch.cuda.synchronize()
        finals.append((prod, cp, out.cpu().numpy()))
    assert_bits_equal(finals[0][2], finals[1][2], "two pipelined runs")
    orac = OracleEngine()
    build(orac); orac.set_seed(31)
    co = orac.create_camera(desc)
    for _ in range(frames):
        orac.update_camera(co, desc); orac.tick(); ref = orac.render_camera(co)
    prod, cp, img = finals[0]
    for b in ALL_FLOAT_BUFFERS:
        assert_bits_equal(prod.read_buffer(cp, b), orac.read_buffer(co, b), f"buffer {b.name} after {frames} unsynchronised frames")
    assert_bits_equal(img, ref, "composed frame")
    assert prod.ray_count(cp) == orac.ray_count(co)


def test_large_bvh_uses_32bit_stack_entries_bit_exact():
    """A BVH stream of more than 65,535 entries switches the traversal kernels to 32-bit LDS stack entries (k_common.h
    ST_LAUNCH_TRACE): heatmap integers, Reference and Image planes must still equal the oracle."""
    torch = _torch()
    size = (160, 96)
    build = lambda e: scenes.build_random_soup(e, 40000, seed=9)
    prod, orac, desc, cp, co = _pair(build, size, CameraMode.BVH_HEATMAP)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    img, ref = _step(torch, prod, orac, desc, cp, co, out)
    w = prod.read_scene(0).reshape(-1, 4).view(np.uint32)[:, 3]
    entries, p = 0, 0
    while p < len(w):  # an internal node is four float4 of the serializer's stream, a leaf entry one
        p += 4 if w[p] == 0 else 1
        entries += 1
    assert entries > 65536, f"scene too small to leave the 16-bit path: {entries} stream entries"
    assert np.array_equal(prod.read_buffer(cp, Buffer.DBG_USED_MEMORY), orac.read_buffer(co, Buffer.DBG_USED_MEMORY))
    assert_bits_equal(img, ref, "heatmap colours (u32 stack)")
    for mode, frames in ((CameraMode.REFERENCE, 2), (CameraMode.IMAGE, 5)):
        desc = scenes.cornell_camera(size, mode, depth=1)
        for frame in range(frames):
            img, ref = _step(torch, prod, orac, desc, cp, co, out)
            _compare_all(prod, orac, cp, co, frame)
            assert_bits_equal(img, ref, f"{mode.name} frame {frame} (u32 stack)")


def test_alpha_blend_materials_and_many_lights_bit_exact():
    """Blend materials with a textured alpha channel (traversal's alpha test and its +128 B of used_memory) and a light
    table longer than the 16 RIS picks."""
    torch = _torch()
    size = (128, 80)
    build = lambda e: scenes.build_random_soup(e, 1500, seed=21, n_lights=24, blend_fraction=0.5)
    prod, orac, desc, cp, co = _pair(build, size, CameraMode.BVH_HEATMAP)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    img, ref = _step(torch, prod, orac, desc, cp, co, out)
    assert np.array_equal(prod.read_buffer(cp, Buffer.DBG_USED_MEMORY), orac.read_buffer(co, Buffer.DBG_USED_MEMORY))
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    for frame in range(7):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"blend+lights frame {frame}")


def test_empty_world_bit_exact():
    """No instances: the BVH is empty, primary visibility clears its targets, the DI/GI chains are skipped
    (camera_controller.rs:118-143) and composition shows the sky."""
    torch = _torch()
    size = (96, 64)
    def build(e):
        e.set_blue_noise(scenes.load_blue_noise())
        from strolle_amd import Sun
        e.update_sun(Sun(azimuth=0.3, altitude=0.6))
    for mode in (CameraMode.IMAGE, CameraMode.REFERENCE, CameraMode.BVH_HEATMAP):
        prod, orac, desc, cp, co = _pair(build, size, mode, depth=1)
        out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
        for frame in range(3):
            img, ref = _step(torch, prod, orac, desc, cp, co, out)
            _compare_all(prod, orac, cp, co, frame)
            assert_bits_equal(img, ref, f"empty world {mode.name} frame {frame}")


def test_scene_edits_between_frames_bit_exact():
    """The host path under change (Engine::tick, lib.rs:301-395): a light removed (its slot is killed for one frame —
    the 0xcafebabe marker di_temporal looks for), a light added, an instance removed, a material replaced, the camera
    resized (buffers are rebuilt, camera.rs:17-48) and its mode switched, all between rendered frames."""
    torch = _torch()
    from strolle_amd import Light, Material
    size = (128, 80)
    prod, orac = Engine(device=0, exact=True), OracleEngine()
    for e in (prod, orac):
        scenes.build_random_soup(e, 1200, seed=13, n_lights=4); e.set_seed(5)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    cp, co = prod.create_camera(desc), orac.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(14):
        for e in (prod, orac):
            if frame == 3: e.remove_light(2)
            if frame == 5: e.insert_light(9, Light.point((0.2, 0.8, 0.4), 0.1, (1.5, 1.2, 0.9), 20.0))
            if frame == 6: e.remove_instance(3)
            if frame == 8: e.insert_material(1, Material(base_color=(0.9, 0.2, 0.2, 1.0), perceptual_roughness=0.4, metallic=0.3))
            if frame == 9: e.remove_light(1)
        if frame == 10:
            size = (96, 72)
            desc = scenes.cornell_camera(size, CameraMode.IMAGE)
            out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
        if frame == 12:
            desc = scenes.cornell_camera(size, CameraMode.GI_DIFFUSE)
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"edited scene frame {frame}")


def test_bench_two_rank_control_flow_on_one_gpu(tmp_path):
    """bench.py's N > 1 path (row bands, aprons, double-buffered targets, per-frame gather on a side stream, ray
    accounting, JSON) run as two processes that share cuda:0 and gather through gloo (ST_BENCH_DEBUG_SHARED_GPU=1):
    a functional check of everything except RCCL itself. In Reference mode the gathered frame must equal a
    single-process render of the whole frame bit for bit."""
    import json, os, subprocess, sys
    torch = _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = tmp_path / "frame.npy"
    env = dict(os.environ, ST_BENCH_DEBUG_SHARED_GPU="1", MASTER_ADDR="127.0.0.1", ST_EXACT="1")  # exact build: the bit-compare below needs it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--width", "128", "--height", "64", "--mode", "reference",
           "--preroll", "0", "--no-cpu-baseline", "--no-profile", "--dump-frame", str(dump), "--extras-size", "160", "96"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["width"] == 128 and out["config"]["height"] == 128 and out["config"]["frame_finite"]
    assert out["config"]["rays_per_frame"] > 0 and out["value"] > 0
    # the line the driver's N > 1 runs will produce also carries BASELINE.json's config 5 as written (here shrunk to 160x96):
    # one dungeon Image frame split into the ranks' row bands + apron, gathered to rank 0
    mg = out["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and mg["backend"] == "gloo"   # shared-GPU debug mode gathers through gloo
    c5 = mg["strong_config5"]
    # (round 5: the extras rebalance the tiles from the ranks' own frame times before the timed region — st_dist_grid_rebalance, four rounds,
    # an edge moves by at most the apron per round and a band keeps at least 32 rows: rank 0's band ends somewhere in [32, 64] on the 8-row grid)
    bal = c5["balance"]
    assert len(bal["rounds"]) == 4 and bal["rounds"][0]["grid"]["row_edges"] == [0, 48, 96] and all(len(r["per_rank_ms"]) == 2 for r in bal["rounds"])
    edge = bal["final_grid"]["row_edges"][1]
    assert bal["final_grid"]["row_edges"] == [0, edge, 96] and edge % 8 == 0 and 32 <= edge <= 64
    assert all(abs(a["grid"]["row_edges"][1] - b["grid"]["row_edges"][1]) <= 16 for a, b in zip(bal["rounds"], bal["rounds"][1:]))
    assert c5["band_rows"] == edge and c5["apron_rows"] == 16 and c5["frame_finite"] and c5["ms_per_step"] > 0 and c5["Mray_per_s"] > 0
    assert len(c5["per_rank_ms"]) == 2 and c5["gather_ms"] is not None and c5["gathered_bytes_per_frame"] == (96 - edge) * 160 * 16
    assert abs(c5["apron_overhead_frac"]["max"] - (64 / 48 - 1)) < 1e-3 and "n1_ms_reference" in c5   # (of the equal split) 96 rows in 2 bands of 48, apron 16
    assert "strong_config4" not in mg   # N = 4 only
    got = np.load(dump)
    # the same 5 frames in one process
    prod = Engine(device=0, exact=True)
    scenes.build_cornell(prod); prod.set_seed(0)
    desc = scenes.cornell_camera((128, 128), CameraMode.REFERENCE, depth=1)
    cam = prod.create_camera(desc)
    frame = torch.zeros((128, 128, 4), dtype=torch.float32, device="cuda:0")
    for _ in range(5):
        prod.update_camera(cam, desc); prod.tick(); prod.render_camera(cam, frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert_bits_equal(got, frame.cpu().numpy(), "gathered two-band frame vs single-process frame")


def test_bench_n_gt_1_line_is_config_5_strong(tmp_path):
    """with the driver's N > 1 command (no --scaling, the default workload) the ONE JSON line IS BASELINE.json's config 5 as
    written — dungeon Image, one frame in N cost-balanced tiles, strong scaling: value / ms_per_step / config.workload / scaling / speedup_vs_n1 —
    and the weak region (N x 1080p Cornell) rides along under multi_gpu.weak_scaling_extra. Two processes sharing cuda:0 (gloo gather), config 5
    shrunk to 160x96 through --extras-size."""
    import json, os, subprocess, sys
    _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ST_BENCH_DEBUG_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--preroll", "0", "--extras-size", "160", "96"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    mg = out["multi_gpu"]; c5 = mg["strong_config5"]
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3 and c5["steps"] == 3
    assert out["value"] == c5["Mray_per_s"] > 0 and out["ms_per_step"] == c5["ms_per_step"] > 0
    assert "config 5" in out["config"]["workload"] and "dungeon 160x96" in out["config"]["workload"] and (out["config"]["width"], out["config"]["height"]) == (160, 96)
    assert out["speedup_vs_n1"] is None and "profiles/n1_reference.json" in out["speedup_vs_n1_note"]   # (not the config's 3840x2160: no N = 1 figure applies)
    assert "cpu_baseline" not in out and "_profile" not in c5
    weak = mg["weak_scaling_extra"]
    assert (weak["width"], weak["height"]) == (1920, 2160) and weak["value_Mray_per_s"] > 0 and len(weak["per_rank_ms_per_step"]) == 2 and "Cornell" in weak["workload"]
    assert mg["rccl_ranks"] == 2 and "strong_config5" in mg["main_region"]
    # the roofline object is this config's own dominant kernel on rank 0's tile
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["traffic"] is None and out["roofline"]["achieved"] > 0
    assert "kernels" in out and out["config"]["frame_finite"]


def test_bench_strong_scaling_two_ranks_on_one_gpu(tmp_path):
    """bench.py --scaling strong (BASELINE.json configs 4 and 5 as written: the frame keeps its size, ranks split it): the
    gathered Reference frame equals the single-process frame of the same size bit for bit, and the JSON carries the per-rank
    and gather timings."""
    import json, os, subprocess, sys
    torch = _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = tmp_path / "frame.npy"
    env = dict(os.environ, ST_BENCH_DEBUG_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--width", "160", "--height", "96", "--mode", "reference", "--scaling", "strong",
           "--exact", "--preroll", "0", "--no-cpu-baseline", "--no-profile", "--no-extras", "--dump-frame", str(dump)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and (out["config"]["width"], out["config"]["height"]) == (160, 96)
    assert out["config"]["per_gpu_rows"] == 48 and out["config"]["apron_rows"] == 0
    assert len(out["multi_gpu"]["per_rank_ms_per_step"]) == 2 and out["multi_gpu"]["gathered_bytes_per_frame"] == 48 * 160 * 16
    prod = Engine(device=0, exact=True)
    scenes.build_cornell(prod); prod.set_seed(0)
    desc = scenes.cornell_camera((160, 96), CameraMode.REFERENCE, depth=1)
    cam = prod.create_camera(desc)
    frame = torch.zeros((96, 160, 4), dtype=torch.float32, device="cuda:0")
    for _ in range(4):
        prod.update_camera(cam, desc); prod.tick(); prod.render_camera(cam, frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert_bits_equal(np.load(dump), frame.cpu().numpy(), "strong-scaling two-band frame vs single-process frame")


def test_two_cameras_on_two_streams_bit_exact():
    """Two cameras of one engine, rendered back to back on DIFFERENT caller streams without host synchronisation in between:
    each camera's frame pipeline (side stream + cross-frame events) is its own, so neither waits on nor races with the other's
    frames, and both stay bit-identical to the oracle's two cameras."""
    torch = _torch()
    prod, orac = Engine(device=0, exact=True), OracleEngine()
    for e in (prod, orac):
        scenes.build_cornell(e); e.set_seed(31)
    sizes = [(160, 96), (136, 120)]
    descs = [scenes.cornell_camera(sizes[0], CameraMode.IMAGE), scenes.camera_for(sizes[1], (0.4, 1.2, 3.0), (0.0, 0.9, 0.0))]
    cps = [prod.create_camera(d) for d in descs]; cos = [orac.create_camera(d) for d in descs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.zeros((s[1], s[0], 4), dtype=torch.float32, device="cuda:0") for s in sizes]
    refs = [None, None]
    torch.cuda.synchronize()
    for frame in range(9):
        for k in range(2):
            prod.update_camera(cps[k], descs[k]); orac.update_camera(cos[k], descs[k])
        prod.tick(streams[0].cuda_stream); orac.tick()
        for k in range(2):
            prod.render_camera(cps[k], outs[k].data_ptr(), streams[k].cuda_stream)
        for k in range(2):
            refs[k] = orac.render_camera(cos[k])
        if frame in (3, 8):
            torch.cuda.synchronize()
            for k in range(2):
                _compare_all(prod, orac, cps[k], cos[k], frame)
                assert_bits_equal(outs[k].cpu().numpy(), refs[k], f"camera {k} frame {frame}")
    prod.close(); orac.close()


def test_moving_instances_velocity_bit_exact():
    """An instance re-inserted with a new transform every frame: primary visibility derives the surface point's previous
    position from the owning instance's transforms (prev_xform * curr_xform_inv * point, prim_raster.rs:21-27), the BVH is
    rebuilt every tick, and reprojection follows the resulting velocity map."""
    torch = _torch()
    import math
    from strolle_amd import Instance
    size = (144, 96)
    prod, orac = Engine(device=0, exact=True), OracleEngine()
    for e in (prod, orac):
        scenes.build_random_soup(e, 1600, seed=17, n_lights=3); e.set_seed(9)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    cp, co = prod.create_camera(desc), orac.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    moved = 0
    for frame in range(9):
        ang = 0.04 * frame
        rot = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]], np.float32)
        x = np.concatenate([rot * np.float32(1.1), np.array([[0.02 * frame], [0.01 * frame], [0.0]], np.float32)], axis=1)
        for e in (prod, orac):
            if frame >= 2 and frame != 6:   # frame 6: not re-inserted, so its prev transform stays one step behind (as in the reference)
                e.insert_instance(2, Instance(2, 2, x))
            if frame == 7:
                e.remove_instance(4)        # frees a transform slot ...
            if frame == 8:
                e.insert_instance(9, Instance(1, 3, np.concatenate([np.eye(3, dtype=np.float32) * 0.5, np.array([[0.3], [0.2], [0.1]], np.float32)], axis=1)))  # ... that a new instance reuses
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"moving instance frame {frame}")
        moved += int(np.count_nonzero(orac.read_buffer(co, Buffer.VELOCITY_MAP)))
    assert moved > 0, "no pixel ever had a velocity: the moving instance was not seen"


def test_spot_lights_metallic_surfaces_bit_exact():
    """Spot lights (angle falloff through angle_between + acos_approx, light.rs:143-160) over a scene with metallic
    materials (GGX specular lobes in DI and GI), plus a point light with infinite range."""
    torch = _torch()
    from strolle_amd import Light
    size = (128, 80)
    def build(e):
        scenes.build_random_soup(e, 1200, seed=31, n_lights=1)
        e.insert_light(5, Light.spot((0.2, 1.4, 0.9), 0.1, (6.0, 5.0, 4.0), 25.0, (-0.1, -0.9, -0.4), 0.6))
        e.insert_light(6, Light.spot((-0.8, 0.3, 1.2), 0.05, (2.0, 3.0, 5.0), 15.0, (0.5, -0.2, -0.8), 0.25))
        e.insert_light(7, Light.point((0.0, 2.5, 0.0), 0.2, (0.8, 0.8, 0.8), float("inf")))
    prod, orac, desc, cp, co = _pair(build, size, CameraMode.IMAGE)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(8):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"spot lights frame {frame}")
    desc = scenes.cornell_camera(size, CameraMode.REFERENCE, depth=2)
    for frame in range(2):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        assert_bits_equal(img, ref, f"spot lights reference frame {frame}")
    assert float(np.nanmean(ref[..., :3])) > 0.0


@pytest.mark.parametrize("seed", [101, 202])
def test_random_edit_history_renders_bit_exact(seed):
    """Random scene edits between frames (instances moved / removed / re-added, lights added / removed, materials
    replaced) with every plane compared after every frame: the temporal passes meet killed and remapped light slots, fresh
    and vanished geometry, reused triangle and transform slots."""
    torch = _torch()
    from strolle_amd import Instance, Light, Material
    rng = np.random.default_rng(seed)
    size = (112, 72)
    prod, orac = Engine(device=0, exact=True), OracleEngine()
    for e in (prod, orac):
        scenes.build_random_soup(e, 900, seed=seed, n_lights=3); e.set_seed(seed)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    cp, co = prod.create_camera(desc), orac.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    lights = {1, 2, 3}; present = {1, 2, 3, 4}

    def xform():
        a = float(rng.uniform(0, 1.0))
        r = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32) * np.float32(rng.uniform(0.8, 1.2))
        return np.concatenate([r, rng.uniform(-0.2, 0.2, (3, 1)).astype(np.float32)], axis=1)

    for frame in range(16):
        for _ in range(int(rng.integers(0, 3))):
            op = int(rng.integers(0, 5))
            if op == 0:
                h = int(rng.integers(1, 5)); inst = Instance(h, int(rng.integers(1, 5)), xform()); present.add(h)
                for e in (prod, orac): e.insert_instance(h, inst)
            elif op == 1 and len(present) > 1:
                h = int(rng.choice(sorted(present))); present.discard(h)
                for e in (prod, orac): e.remove_instance(h)
            elif op == 2:
                h = int(rng.integers(1, 7)); lights.add(h)
                l = Light.point(rng.uniform(-1.5, 1.5, 3).tolist(), 0.1, rng.uniform(0.3, 2.0, 3).tolist(), 20.0)
                for e in (prod, orac): e.insert_light(h, l)
            elif op == 3 and len(lights) > 1:
                h = int(rng.choice(sorted(lights))); lights.discard(h)
                for e in (prod, orac): e.remove_light(h)
            elif op == 4:
                h = int(rng.integers(1, 5))
                m = Material(base_color=rng.uniform(0.1, 0.9, 3).tolist() + [1.0], perceptual_roughness=float(rng.uniform(0.2, 1.0)), metallic=float(rng.uniform(0, 0.9)))
                for e in (prod, orac): e.insert_material(h, m)
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"seed {seed} frame {frame}")


@pytest.mark.gpu
def test_device_resident_and_dynamic_images_bit_exact():
    """ImageData::Texture (image.rs:46-59, images.rs:160-213): pixels that already live in device memory — copied into the
    atlas once (static) or at every tick (dynamic) — must render exactly as the same pixels handed over as raw bytes,
    which is all the oracle knows. The dynamic texture changes every frame; frames are enqueued without host syncs in
    between, so the per-tick copy has to be ordered against both streams of the frame graph."""
    from strolle_amd import StrolleError
    torch = _torch()
    size = (128, 80)
    build = lambda e: scenes.build_random_soup(e, 1200, seed=33, n_lights=5, blend_fraction=0.5)
    prod, orac, desc, cp, co = _pair(build, size, CameraMode.IMAGE)
    rng = np.random.default_rng(5)
    # the soup's textures again, this time from device memory: 900 (alpha-tested base colour) becomes dynamic, 902 (emissive)
    # static with padded rows; 901 stays raw. Same sizes, so the rectangles stay where they are in both engines.
    base = rng.integers(0, 256, (32, 32, 4), dtype=np.uint8); base[..., 3] = rng.choice(np.array([0, 128, 255], np.uint8), (32, 32))
    emissive = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    d_base = torch.from_numpy(base).cuda()
    d_emissive = torch.zeros((8, 16, 4), dtype=torch.uint8, device="cuda:0")   # pitch 64 B, rows of 32 B
    d_emissive[:, :8] = torch.from_numpy(emissive).cuda()
    torch.cuda.synchronize()
    prod.insert_device_image(900, d_base.data_ptr(), 32, 32, dynamic=True)
    prod.insert_device_image(902, d_emissive.data_ptr(), 8, 8, row_pitch_bytes=64, dynamic=False)
    orac.insert_image(900, base); orac.insert_image(902, emissive)
    assert prod.image_rect(900) == orac.image_rect(900) and prod.image_rect(902) == orac.image_rect(902)
    stream = torch.cuda.current_stream().cuda_stream
    outs = [torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0") for _ in range(6)]
    refs = []
    for frame in range(6):
        if frame:
            base = base.copy(); base[..., :3] = (base[..., :3].astype(np.int32) + 37 * frame) % 256
            if frame == 3:
                base[..., 3] = 255 - base[..., 3]     # the alpha test flips: traversal itself changes
            d_base.copy_(torch.from_numpy(base), non_blocking=False)   # same stream as the tick: ordered before its copy
            orac.insert_image(900, base)
        prod.update_camera(cp, desc); orac.update_camera(co, desc)
        prod.tick(stream); orac.tick()
        prod.render_camera(cp, outs[frame].data_ptr(), stream)
        refs.append(orac.render_camera(co))
    torch.cuda.synchronize()
    for frame in range(6):
        assert_bits_equal(outs[frame].cpu().numpy(), refs[frame], f"device images frame {frame}")
    _compare_all(prod, orac, cp, co, 5)
    # a later raw re-upload of the whole atlas must not lose the static device image (its host mirror is complete)
    extra = rng.integers(0, 256, (4, 4, 4), dtype=np.uint8)
    prod.insert_image(950, extra); orac.insert_image(950, extra)
    out = outs[0]
    img, ref = _step(torch, prod, orac, desc, cp, co, out)
    assert_bits_equal(img, ref, "frame after a full atlas re-upload")
    # host-only engines say so instead of pretending
    with pytest.raises(StrolleError, match="device"):
        Engine(device=-1).insert_device_image(1, d_base.data_ptr(), 32, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [1, 2, 3])
def test_output_formats_bit_exact(fmt):
    """camera.rs:170-175 viewport.format: Rgba16Float and the two 8-bit sRGB swap-chain formats, written by the composition
    kernel, against the oracle's RGBA32F frame pushed through or_encode_output."""
    from oracle_binding import encode_output
    from strolle_amd import OutputFormat
    torch = _torch()
    size = (160, 96)
    prod, orac, desc, cp, co = _pair(scenes.build_cornell, size, CameraMode.IMAGE)
    prod.set_output_format(cp, OutputFormat(fmt))
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float16 if fmt == 1 else torch.uint8, device="cuda:0")
    for frame in range(4):
        prod.update_camera(cp, desc); orac.update_camera(co, desc)
        prod.tick(); orac.tick()
        prod.render_camera(cp, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        ref = orac.render_camera(co)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        got = got.view(np.uint16) if fmt == 1 else got
        assert np.array_equal(got, encode_output(ref, fmt)), f"format {fmt} frame {frame}"
    assert len(np.unique(got[..., :3])) > 50, "a frame with this little variety would not test the encoders"
    # the default goes back to RGBA32F
    prod.set_output_format(cp, OutputFormat.RGBA32F)
    out32 = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    img, ref = _step(torch, prod, orac, desc, cp, co, out32)
    assert_bits_equal(img, ref, "RGBA32F after switching back")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2], ids=["host", "device"])
def test_bvh_refit_mode_renders_bit_exact(mode):
    """ST_BVH_REFIT with instances moving every frame: the tree of frame 0 is refitted, never rebuilt, and only the moved
    triangles travel to the device — heatmap integers, every Image-mode plane and the frame must still equal the oracle's,
    which refits its own tree. A removal then forces a rebuild, after which refitting resumes.
    mode 2 = ST_BVH_REFIT_DEVICE: the boxes are recomputed by k_bvh.hip from the moved triangles' bounds (the host sends 80 B per
    moved triangle instead of the whole stream); after every frame the stream read back FROM THE DEVICE must equal, bit for bit,
    the device form of the host's own refit — which equals the oracle's."""
    torch = _torch()
    import math
    from strolle_amd import Instance
    size = (144, 96)
    prod, orac = Engine(device=0, exact=True), OracleEngine()
    for e in (prod, orac):
        scenes.build_random_soup(e, 2400, seed=23, n_lights=3); e.set_seed(4)
    prod.set_bvh_refresh(mode); orac.set_bvh_refresh(True)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    heat = scenes.cornell_camera(size, CameraMode.BVH_HEATMAP)
    cp, co = prod.create_camera(desc), orac.create_camera(desc)
    hp, ho = prod.create_camera(heat), orac.create_camera(heat)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    out_h = torch.zeros_like(out)
    stream = torch.cuda.current_stream().cuda_stream
    for frame in range(8):
        ang = 0.05 * frame
        rot = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]], np.float32)
        for e in (prod, orac):
            if frame >= 1:
                e.insert_instance(2, Instance(2, 2, np.concatenate([rot * np.float32(1.1), np.array([[0.03 * frame], [0.0], [0.01 * frame]], np.float32)], axis=1)))
                e.insert_instance(3, Instance(3, 3, np.concatenate([rot.T * np.float32(1.2), np.array([[0.0], [0.02 * frame], [0.0]], np.float32)], axis=1)))
            if frame == 5:
                e.remove_instance(4)
        prod.update_camera(cp, desc); orac.update_camera(co, desc)
        prod.tick(stream); orac.tick()
        prod.render_camera(cp, out.data_ptr(), stream); prod.render_camera(hp, out_h.data_ptr(), stream)
        ref = orac.render_camera(co); orac.render_camera(ho)
        torch.cuda.synchronize()
        assert np.array_equal(prod.read_buffer(hp, Buffer.DBG_USED_MEMORY), orac.read_buffer(ho, Buffer.DBG_USED_MEMORY)), f"refit frame {frame}: used_memory"
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(out.cpu().numpy(), ref, f"refit frame {frame}")
        if mode == 2:
            assert_bits_equal(prod.read_scene(6), prod.read_scene(4), f"refit frame {frame}: the stream on the device vs the host's refit")
            assert_bits_equal(prod.read_scene(0), orac.read_scene(0), f"refit frame {frame}: the host's (lazily refitted) stream vs the oracle's")
    assert prod.bvh_refits() == orac.bvh_refits() == (2, 6)
    assert prod.bvh_device_refits() == (4 if mode == 2 else 0)   # refit ticks whose target copy already held the tree (the first refit after a build goes to the other copy in full)
    if mode == 2:   # StTuning::device_bake (default on): those same ticks baked the moved instances ON THE DEVICE from their object-space meshes — the
        # comparisons above (stream read back from the device, every plane of every frame) are therefore device bake vs host bake, bit for bit
        launches, triangles = prod.device_bakes()
        assert launches == 4 and triangles > 0


@pytest.mark.gpu
def test_device_refit_of_the_dungeon_equals_the_host_refit():
    """ST_BVH_REFIT_DEVICE on a tree deep and large enough for several launches of k_bvh_refit (52 k triangles, 26 levels; the
    work list is cut into 512-leaf tasks): with every instance of the dungeon moving each tick, the stream on the device must stay,
    bit for bit, the device form of the host's own backward sweep over the same tree."""
    from strolle_amd import Instance
    e = Engine(device=0)
    scenes.build_dungeon(e, subdivide=1)
    e.set_bvh_refresh(2)
    npz = np.load(os.path.join(scenes.ASSETS, "dungeon.npz"))
    e.tick()
    for tick in range(5):
        for k in range(int(npz["n_meshes"])):
            if tick == 3 and k % 3: continue          # one tick moves only a third of the instances
            x = npz[f"xform_{k}"].reshape(4, 3).T.copy(); x[:3, 3] += 0.01 * (tick + 1) * np.array([1.0, -0.5, 0.25]) * (1 + k % 5)
            e.insert_instance(1 + k, Instance(1 + k, 1 + int(npz[f"material_{k}"]), x))
        e.tick()
        on_device, on_host = e.read_scene(6), e.read_scene(4)
        assert on_device.size > 4 * 60_000
        assert_bits_equal(on_device, on_host, f"tick {tick}: the stream on the device vs the host's refit")
    assert e.bvh_refits() == (1, 5) and e.bvh_device_refits() == 4
    launches, triangles = e.device_bakes()
    assert launches == 4 and triangles >= 3 * 33_000, (launches, triangles)   # every instance of the 52 k-triangle level on three of the four device ticks
    # ... and with the device bake switched off the same sequence gives the same stream (host bake + 80 B per triangle over PCIe)
    h = Engine(device=0)
    scenes.build_dungeon(h, subdivide=1)
    h.set_tuning(device_bake=0)
    h.set_bvh_refresh(2)
    h.tick()
    for tick in range(5):
        for k in range(int(npz["n_meshes"])):
            if tick == 3 and k % 3: continue
            x = npz[f"xform_{k}"].reshape(4, 3).T.copy(); x[:3, 3] += 0.01 * (tick + 1) * np.array([1.0, -0.5, 0.25]) * (1 + k % 5)
            h.insert_instance(1 + k, Instance(1 + k, 1 + int(npz[f"material_{k}"]), x))
        h.tick()
    assert h.device_bakes() == (0, 0)
    assert_bits_equal(h.read_scene(6), e.read_scene(6), "device bake vs host bake: the stream on the device")
    assert_bits_equal(h.read_scene(1), e.read_scene(1), "device bake vs host bake: the host's triangles once they caught up")


def _cornell_glb() -> bytes:
    """assets/cornell.npz written back out as a GLB (the original file does not travel to the GPU box)."""
    import json, struct
    npz = np.load(os.path.join(scenes.ASSETS, "cornell.npz"))
    blob, views, accessors, meshes, nodes = bytearray(), [], [], [], []
    def add(array, kind):
        data = np.ascontiguousarray(array, np.float32).tobytes()
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}); blob.extend(data)
        accessors.append({"bufferView": len(views) - 1, "componentType": 5126, "count": len(array), "type": kind})
        return len(accessors) - 1
    for i in range(int(npz["n_meshes"])):
        pos, nrm = npz[f"positions_{i}"].reshape(-1, 3), npz[f"normals_{i}"].reshape(-1, 3)
        meshes.append({"primitives": [{"attributes": {"POSITION": add(pos, "VEC3"), "NORMAL": add(nrm, "VEC3")}, "material": int(npz[f"material_{i}"])}]})
        x = npz[f"xform_{i}"].reshape(4, 3)
        nodes.append({"mesh": i, "matrix": [*x[0], 0.0, *x[1], 0.0, *x[2], 0.0, *x[3], 1.0]})
    mats = [{"pbrMetallicRoughness": {"baseColorFactor": [float(v) for v in npz["material_base_color"][j]], "metallicFactor": float(npz["material_metallic"][j]),
                                      "roughnessFactor": float(npz["material_perceptual_roughness"][j])}} for j in range(len(npz["material_metallic"]))]
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": [{k: ([float(v) for v in val] if k == "matrix" else val) for k, val in n.items()} for n in nodes],
           "meshes": meshes, "materials": mats, "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob)}]}
    js = json.dumps(doc).encode(); js += b" " * ((-len(js)) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + bytes(blob)
    return struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body


@pytest.mark.gpu
def test_c_example_renders_the_cornell_box(tmp_path):
    """examples/render_gltf.c — plain C over the ABI, no Python in the loop: loads the Cornell box from a GLB, renders 24
    Image-mode frames into an 8-bit sRGB target and writes a PPM. The scene it loads is checked against the .npz route,
    the picture for being the lit box (not black, not flat, brighter in the middle than in the corners)."""
    import subprocess
    from test_c_abi import _compile_example
    glb = tmp_path / "cornell.glb"
    glb.write_bytes(_cornell_glb())
    a, b = Engine(device=-1), Engine(device=-1)
    a.load_gltf(str(glb)); scenes._insert_gltf(b, np.load(os.path.join(scenes.ASSETS, "cornell.npz")))
    a.tick(); b.tick()
    for what in range(4):
        assert_bits_equal(a.read_scene(what), b.read_scene(what), f"cornell.glb vs cornell.npz: scene buffer {what}")
    exe = str(tmp_path / "render_gltf")
    _compile_example(exe)
    out = tmp_path / "cornell.ppm"
    run = subprocess.run([exe, str(glb), str(out), "320", "240", "24"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    assert "32 triangles" in run.stderr and " rays" in run.stderr
    assert "24 frames (24 presented" in run.stderr, run.stderr   # every frame left through st_camera_present_copy
    data = out.read_bytes()
    assert data.startswith(b"P6\n320 240\n255\n")
    img = np.frombuffer(data[len(b"P6\n320 240\n255\n"):], np.uint8).reshape(240, 320, 3).astype(np.float32)
    assert img.mean() > 20 and img.std() > 20, (img.mean(), img.std())
    assert img[60:180, 80:240].mean() > img[:20, :20].mean()
    # left wall red, right wall green (the Cornell box): the walls' colour channels dominate on their side
    left, right = img[100:140, 10:40].mean((0, 1)), img[100:140, 280:310].mean((0, 1))
    assert left[0] > left[1] and right[1] > right[0], (left, right)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [4, 8])
def test_c_example_gathers_tiles_through_the_c_abi(tmp_path, world):
    """examples/dist_tiles.c — the multi-GPU host loop (st_dist_init_local / st_dist_set_partition / st_dist_gather / st_dist_wait) from
    plain C: `world` engines render their tiles of a Reference frame, rank 0 assembles them, and the program itself compares the result
    with the same frame from one engine, byte for byte (exit status 0 only if none differs and the frame is lit)."""
    import subprocess
    from test_c_abi import _compile_example
    glb = tmp_path / "cornell.glb"
    glb.write_bytes(_cornell_glb())
    exe = str(tmp_path / "dist_tiles")
    _compile_example(exe, "dist_tiles.c")
    run = subprocess.run([exe, str(glb), str(world), "272", "200", "3"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    assert f"{world} ranks, 3 frames" in run.stderr and " 0 of 870400 bytes differ" in run.stderr, run.stderr


def test_profile_flags_timing_and_traversal_bytes():
    """st_profile_enable bit 0 = per-kernel event timing, bit 1 = the tracing kernels also sum the reference's `used_memory`
    over their rays (off by default: it costs a cross-lane reduction per ray). Rays are counted either way, identically."""
    torch = _torch()
    size = (160, 96)
    e = Engine(device=0)
    scenes.build_cornell(e); e.set_seed(5)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    cam = e.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")

    def frames(n):
        for _ in range(n):
            e.update_camera(cam, desc); e.tick(); e.render_camera(cam, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    frames(6)
    rays = {}
    for flags in (0, 1, 2, 3):
        e.profile_enable(flags); e.profile_read(reset=True); e.ray_count(cam, reset=True)
        frames(6)
        prof = e.profile_read(reset=True)
        rays[flags] = e.ray_count(cam)
        timed = sum(p["total_ms"] for p in prof) > 0.0
        traversal = sum(p["traversal_bytes"] for p in prof)
        assert timed == bool(flags & 1), (flags, prof)
        assert (traversal > 0) == bool(flags & 2), (flags, traversal)
        if flags & 2:   # every traced ray reads at least the root: 16 B + one internal node's 48 B
            assert traversal >= 64 * rays[flags]
    e.profile_enable(0)
    assert rays[0] > 0 and rays[0] >= size[0] * size[1] * 6
    e.close()


@pytest.mark.parametrize("switches", [
    {"ST_NO_FUSE": "1"},
    {"ST_NO_OVERLAP": "1", "ST_NO_FUSE_WAVELET": "1"},
    {"ST_TILE_MAP": "0"},
    {"ST_NO_PREVIEW_BOTH": "1", "ST_NO_VARIANCE_IN_REPROJECT": "1", "ST_KEEP_SCRATCH": "1", "ST_DI_HEAD_ON_MAIN": "0", "ST_NO_FUSE_GI_VALIDATION": "1"},
    {"ST_NO_FUSE_SPATIAL": "1", "ST_NO_FUSE_DI_HEAD": "1", "ST_NO_FUSE_GI_REPROJECTION": "1", "ST_NO_FUSE_GI_SAMPLING": "1"},
], ids=lambda s: "+".join(sorted(s)))
def test_every_scheduling_variant_produces_the_same_bits(switches):
    """The engine's fusion / overlap / mapping switches (read from the environment when an engine is created) select other
    launch structures for the same pass graph: each must leave every plane bit-identical to the oracle's, frame after frame
    (exact build; Cornell 160x96 Image{denoise}, 8 frames = every GI schedule)."""
    torch = _torch()
    size = (160, 96)
    saved = {k: os.environ.get(k) for k in switches}
    os.environ.update(switches)
    try:
        prod = Engine(device=0, exact=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    orac = OracleEngine()
    for e in (prod, orac):
        scenes.build_cornell(e); e.set_seed(17)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    cp, co = prod.create_camera(desc), orac.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for frame in range(8):
        img, ref = _step(torch, prod, orac, desc, cp, co, out)
        _compare_all(prod, orac, cp, co, frame)
        assert_bits_equal(img, ref, f"composed frame {frame}")
    prod.close(); orac.close()


@pytest.mark.gpu
def test_present_copy_hands_over_every_frame_without_a_stream_sync():
    """st_camera_present_copy / st_camera_present_ready (the facade's half of Engine::render_camera, lib.rs:279-286): frame N
    is copied to page-locked host memory behind its composition while frame N+1 renders; what arrives is what a
    synchronous read-back of the same device buffer gives, for every frame, with two buffers alternating — and a render
    into a buffer whose copy is still pending is ordered behind that copy."""
    torch = _torch()
    size = (320, 200)
    e = Engine(device=0, exact=True)
    scenes.build_cornell(e); e.set_seed(3)
    desc = scenes.cornell_camera(size, CameraMode.IMAGE)
    cam = e.create_camera(desc)
    e.set_output_format(cam, OutputFormat.RGBA8_UNORM_SRGB)
    stream = torch.cuda.current_stream().cuda_stream
    dev = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    host = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8).pin_memory() for _ in range(2)]
    twin = Engine(device=0, exact=True)   # the same frames, read back synchronously
    scenes.build_cornell(twin); twin.set_seed(3)
    tcam = twin.create_camera(desc); twin.set_output_format(tcam, OutputFormat.RGBA8_UNORM_SRGB)
    tout = torch.zeros_like(dev[0])
    want = []
    for i in range(7):
        twin.update_camera(tcam, desc); twin.tick(stream); twin.render_camera(tcam, tout.data_ptr(), stream)
        torch.cuda.synchronize(); want.append(tout.cpu().numpy().copy())
    for i in range(7):
        k = i & 1
        e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, dev[k].data_ptr(), stream)
        e.present_copy(cam, dev[k].data_ptr(), host[k].data_ptr(), dev[k].numel(), stream)
        if i > 0:
            e.present_ready(cam, host[k ^ 1].data_ptr(), wait=True)
            assert np.array_equal(host[k ^ 1].numpy(), want[i - 1]), f"frame {i - 1} through the present path differs from its read-back"
    assert e.present_ready(cam, host[0].data_ptr(), wait=True)
    assert np.array_equal(host[0].numpy(), want[6])
    # one buffer only: the next composition must wait for the pending copy (no tearing)
    for i in range(7, 10):
        twin.update_camera(tcam, desc); twin.tick(stream); twin.render_camera(tcam, tout.data_ptr(), stream)
        torch.cuda.synchronize(); want.append(tout.cpu().numpy().copy())
    for i in range(7, 10):
        e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, dev[0].data_ptr(), stream)
        e.present_copy(cam, dev[0].data_ptr(), host[i & 1].data_ptr(), dev[0].numel(), stream)
    for i in (8, 9):
        e.present_ready(cam, host[i & 1].data_ptr(), wait=True)
        assert np.array_equal(host[i & 1].numpy(), want[i]), f"frame {i}: a later frame's composition overtook the copy"
    e.close(); twin.close()
