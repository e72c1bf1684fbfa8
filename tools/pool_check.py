#!/usr/bin/env python3
"""Lane refill for the GI bounce rays (st_device.h closest_hit_wide_pool, ST_EXP bits 0x100 / 0x200): what the pooled launch leaves in the GI planes
against the plain launch, frame by frame from the same seeds — the walk per ray is the same, so the planes must be bit-identical.
    python tools/pool_check.py [--scene dungeon] [--subdivide 0] [--frames 7]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--subdivide", type=int, default=0)
ap.add_argument("--frames", type=int, default=7)
ap.add_argument("--size", type=int, nargs=2, default=(960, 544))
args = ap.parse_args()
import torch
from strolle_amd import Buffer, CameraMode, Engine, scenes

PLANES = (Buffer.GI_D0, Buffer.GI_D1, Buffer.GI_D2, Buffer.GI_RESERVOIRS_1, Buffer.GI_RESERVOIRS_0, Buffer.PRIM_GBUFFER_D0_A)
W, H = args.size


def run(exp, extra_env):
    for k in ("ST_EXP", "ST_NO_FUSE_GI_VALIDATION"):
        os.environ.pop(k, None)
    if exp:
        os.environ["ST_EXP"] = hex(exp)
    os.environ.update(extra_env)
    e = Engine(device=0)
    e.keep_all_planes(True)
    scenes.build_dungeon(e, subdivide=args.subdivide); e.set_seed(3)
    desc = scenes.dungeon_camera((W, H), CameraMode.IMAGE)
    cam = e.create_camera(desc)
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    frames = []
    for _ in range(args.frames):
        e.update_camera(cam, desc); e.tick(); e.render_camera(cam, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        frames.append([e.read_buffer(cam, b).copy() for b in PLANES] + [out.cpu().numpy().copy()])
    rays = e.ray_count(cam)
    lit = float(np.count_nonzero(e.read_buffer(cam, Buffer.PRIM_GBUFFER_D0_A).reshape(-1, 4)[:, 0]) / (W * H))
    assert lit > 0.9, f"only {lit:.2f} of the primary rays hit the dungeon: the scene on the device is broken"
    e.close()
    return frames, rays


base, rays0 = run(0, {})
base_nv, rays0nv = run(0, {"ST_NO_FUSE_GI_VALIDATION": "1"})
for name, exp, env, ref, rays_ref in (("fused pool (0x100)", 0x100, {}, base, rays0), ("fused pool, refill at 8 (0x2100)", 0x2100, {}, base, rays0),
                                      ("split pool (0x200, gi validation unfused)", 0x200, {"ST_NO_FUSE_GI_VALIDATION": "1"}, base_nv, rays0nv)):
    got, rays = run(exp, env)
    worst = []
    for f, (a, b) in enumerate(zip(got, ref)):
        for plane, x, y in zip([p.name for p in PLANES] + ["composed frame"], a, b):
            ne = int((x.view(np.uint32) != y.view(np.uint32)).sum())
            if ne:
                worst.append((f + 1, plane, ne, x.size))
    print(f"{name}: rays {rays} (plain {rays_ref}); " + ("every plane of every frame bit-identical to the plain launch" if not worst else f"{len(worst)} (frame, plane) pairs differ, first: {worst[:6]}"))
