#!/usr/bin/env python3
"""Seam error of the tiled Image-mode frame UNDER MOTION (light orbiting as cornell.rs animates it, camera orbiting the box) against
the single-engine frame: worst PSNR over the last 30 of N frames, per apron width and frame size. One process, one GPU: the ranks
are engines joined by the in-process transport of st_dist_* (tests/test_gpu_parity.py::test_image_mode_tiles_under_motion_...).
  python tools/seam_motion_sweep.py [--size W H] [--world 8] [--aprons 0 16 32 64] [--frames 130] [--scene cornell|dungeon]"""
import argparse, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from parity import psnr
from strolle_amd import CameraMode, Engine, Light, scenes
from strolle_amd.distributed import tile_overhead

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, nargs=2, default=(1280, 720))
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--cols", type=int, default=0)
ap.add_argument("--aprons", type=int, nargs="+", default=[0, 16, 32, 64])
ap.add_argument("--frames", type=int, default=130)
ap.add_argument("--scene", default="cornell")
args = ap.parse_args()
size = tuple(args.size)
build = scenes.build_cornell if args.scene == "cornell" else scenes.build_dungeon
stream = torch.cuda.current_stream().cuda_stream

def pose(f):
    t = f / 60.0
    if args.scene == "cornell":
        light = (1, Light.point((math.sin(t) / 2.0, 1.5, math.cos(t) / 2.0), 0.15, (50.0 / (4.0 * math.pi),) * 3, 20.0))
        a = 0.1 * t
        desc = scenes.camera_for(size, (3.2 * math.sin(a), 1.0, 3.2 * math.cos(a)), (0.0, 1.0, 0.0), CameraMode.IMAGE, True, 0)
    else:   # the demo's camera walking down the corridor and turning, the nearest light swaying
        light = (1, Light.point((-3.0 + 0.3 * math.sin(2 * t), 0.75, -23.0), 0.15, (5000.0 / (4.0 * math.pi),) * 3, 35.0))
        eye = (-5.75 + 0.2 * math.sin(t), 0.5, -16.8 - 0.5 * t)
        desc = scenes.camera_for(size, eye, (eye[0] + 0.3 * math.sin(0.5 * t), 0.5, eye[2] - 0.2), CameraMode.IMAGE, True, 0)
    return light, desc

for apron in args.aprons:
    ranks = []
    for r in range(args.world):
        e = Engine(device=0); build(e); e.set_seed(21)
        cam = e.create_camera(pose(0)[1])
        e.dist_init_local(r, args.world, 9000 + apron); e.dist_set_partition(cam, cols=args.cols, apron=apron)
        ranks.append((e, cam, torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")))
    ref = Engine(device=0); build(ref); ref.set_seed(21)
    rcam = ref.create_camera(pose(0)[1]); rout = torch.zeros_like(ranks[0][2]); full = torch.zeros_like(rout)
    worst, mean = 1e9, []
    for f in range(args.frames):
        (lid, light), desc = pose(f)
        for e, cam, out in [(ref, rcam, rout)] + list(reversed(ranks)):
            e.insert_light(lid, light); e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream)
        if f >= args.frames - 30:
            for r in range(args.world - 1, -1, -1):
                e, cam, out = ranks[r]
                e.dist_gather(cam, out.data_ptr(), full.data_ptr() if r == 0 else 0, stream)
            ranks[0][0].dist_wait(ranks[0][1], host=True); torch.cuda.synchronize()
            v = psnr(np.clip(full.cpu().numpy()[..., :3], 0, 1), np.clip(rout.cpu().numpy()[..., :3], 0, 1))
            worst = min(worst, v); mean.append(v)
    mx, mn = tile_overhead(size[0], size[1], args.world, apron, args.cols)
    print(f"{args.scene} {size[0]}x{size[1]} {args.world} tiles apron {apron:3d}: worst PSNR {worst:6.2f} dB, mean {sum(mean) / len(mean):6.2f} dB over the last 30 of {args.frames} moving frames; redundant pixels max {mx * 100:.1f} % mean {mn * 100:.1f} %", flush=True)
    ref.close()
    for e, *_ in ranks: e.close()
    del ranks, rout, full
    torch.cuda.empty_cache()
