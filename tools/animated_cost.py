#!/usr/bin/env python3
"""Frame time with a scene that changes every frame (stress-bvh.rs style): one dungeon instance is re-inserted with a nudged
transform before every tick, so each frame pays the host refresh (bake, BVH rebuild, flatten, upload) + the render.
  python tools/animated_cost.py [--subdivide K] [--frames N] [--refit]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strolle_amd import CameraMode, Engine, Instance, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--subdivide", type=int, default=0)
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--light", action="store_true", help="move a light every frame instead of an instance (cornell.rs animates its light)")
ap.add_argument("--cornell", action="store_true")
ap.add_argument("--refit", nargs="?", const=1, default=0, type=int, help="1 = ST_BVH_REFIT (boxes refitted on the host instead of a rebuild), 2 = ST_BVH_REFIT_DEVICE (refitted by k_bvh.hip)")
ap.add_argument("--all", action="store_true", help="move EVERY instance every frame (stress-bvh.rs: many bodies under physics)")
ap.add_argument("--only", choices=["static", "animated"], default=None, help="run one of the two phases only (for a profiler)")
ap.add_argument("--host-bake", action="store_true", help="StTuning::device_bake = 0")
args = ap.parse_args()
e = Engine(device=0)
scenes.build_cornell(e) if args.cornell else scenes.build_dungeon(e, subdivide=args.subdivide)
e.set_bvh_refresh(args.refit)
if args.host_bake:
    e.set_tuning(device_bake=0)
size = (1920, 1080)
desc = (scenes.cornell_camera if args.cornell else scenes.dungeon_camera)(size, CameraMode.IMAGE)
cam = e.create_camera(desc)
out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
npz = np.load(os.path.join(scenes.ASSETS, "dungeon.npz"))
base = npz["xform_0"].reshape(4, 3).T.copy(); mat = 1 + int(npz["material_0"])
in_tick = [0.0]
def frame(i, animate):
    if animate and args.light:
        import math
        from strolle_amd import Light
        t = 0.05 * i
        e.insert_light(1, Light.point((math.sin(t) / 2, 1.5, math.cos(t) / 2) if args.cornell else (-3.0 + 0.2 * math.sin(t), 0.75, -23.0), 0.15, (50.0 / (4 * math.pi),) * 3 if args.cornell else (5000.0 / (4 * math.pi),) * 3, 20.0 if args.cornell else 35.0))
    elif animate and args.all:
        for k in range(int(npz["n_meshes"])):
            x = npz[f"xform_{k}"].reshape(4, 3).T.copy(); x[0, 3] += 0.0005 * ((i % 20) - 10)
            e.insert_instance(1 + k, Instance(1 + k, 1 + int(npz[f"material_{k}"]), x))
    elif animate:
        x = base.copy(); x[0, 3] += 0.0005 * ((i % 20) - 10)
        e.insert_instance(1, Instance(1, mat, x))
    e.update_camera(cam, desc)
    t = time.perf_counter(); e.tick(stream); in_tick[0] += time.perf_counter() - t
    e.render_camera(cam, out.data_ptr(), stream)
for animate in ((False, True) if args.only is None else ((args.only == 'animated'),)):
    for i in range(12): frame(i, animate)
    torch.cuda.synchronize(); t = time.perf_counter(); in_tick[0] = 0.0
    for i in range(args.frames): frame(i, animate)
    torch.cuda.synchronize()
    print(f"{'cornell' if args.cornell else 'dungeon'} subdivide={args.subdivide} refit={args.refit} animate={('light' if args.light else ('every instance' if args.all else 'instance')) if animate else False} device_bakes={e.device_bakes()[0]}: {(time.perf_counter() - t) / args.frames * 1e3:.3f} ms/frame, of which the host spends {in_tick[0] / args.frames * 1e3:.3f} ms inside st_tick")
