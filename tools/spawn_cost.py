#!/usr/bin/env python3
"""What does spawning an instance cost?  (bevy-strolle/examples/stress-bvh.rs:111-167)

The dungeon (--subdivide 2: 208 k triangles) renders; every few frames a small mesh is spawned or removed. Per refresh mode — 0 the host's
binned-SAH rebuild (the reference's tree), 3 the device build (ST_BVH_BUILD_DEVICE, k_lbvh.hip) — prints the host time inside st_tick, the
time until the device is idle again, and the steady frame time between the changes (the device-built tree is another tree: what it costs
the rays shows there).

    python tools/spawn_cost.py [--subdivide 2] [--size 1920 1080] [--spawns 8]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--subdivide", type=int, default=2)
ap.add_argument("--size", type=int, nargs=2, default=(1920, 1080))
ap.add_argument("--spawns", type=int, default=8)
ap.add_argument("--mode", default="image")
ap.add_argument("--refresh", type=int, nargs="*", default=[0, 3, 4], help="refresh modes to run: 0 host rebuild, 3 device build, 4 the default (ST_BVH_AUTO: first tree on the host unless its leaf runs are long, changes on the device)")
args = ap.parse_args()
import torch
from strolle_amd import CameraMode, Engine, Instance, Mesh, scenes

rng = np.random.default_rng(2)
pos = (rng.uniform(-0.3, 0.3, (200, 1, 3)) + rng.uniform(-0.05, 0.05, (200, 3, 3))).astype(np.float32)
nrm = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
blob = Mesh(pos, np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32))
W, H = args.size
for mode in args.refresh:
    e = Engine(device=0)
    e.set_bvh_refresh(mode)
    scenes.build_dungeon(e, subdivide=args.subdivide); e.set_seed(1); e.insert_mesh(7777, blob)
    desc = scenes.dungeon_camera((W, H), CameraMode.IMAGE if args.mode == "image" else CameraMode.GI_DIFFUSE, depth=1)
    cam = e.create_camera(desc)
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream

    def frames(n):
        for _ in range(n):
            e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream)
    frames(30); torch.cuda.synchronize()
    t0 = time.perf_counter(); frames(30); torch.cuda.synchronize()
    steady = (time.perf_counter() - t0) / 30 * 1e3
    ticks, idles, first = [], [], []
    for k in range(args.spawns):
        place = np.eye(4, dtype=np.float32)[:3].copy(); place[:, 3] = (-5.75 + 0.2 * k, 0.6, -18.2)
        if k % 2 == 0: e.insert_instance(7000 + k, Instance(7777, 2, place))
        else: e.remove_instance(7000 + k - 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e.update_camera(cam, desc); e.tick(stream); t1 = time.perf_counter()
        torch.cuda.synchronize(); t1b = time.perf_counter()      # the tick's own device work (uploads, the device build) is through
        e.render_camera(cam, out.data_ptr(), stream); torch.cuda.synchronize(); t2 = time.perf_counter()
        ticks.append((t1 - t0) * 1e3); idles.append((t1b - t0) * 1e3); first.append((t2 - t1b) * 1e3)
        frames(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); frames(30); torch.cuda.synchronize()
    after = (time.perf_counter() - t0) / 30 * 1e3
    print(f"refresh mode {mode} ({ {0: 'host rebuild', 3: 'device build', 4: 'auto: changes on the device, the first tree too when the leaf runs of the host tree are long'}.get(mode, '?') }), subdivide {args.subdivide}: steady frame {steady:.3f} ms before / {after:.3f} ms after the changes | "
          f"spawn / despawn: st_tick {np.median(ticks):.2f} ms on the host (max {max(ticks):.2f}), tick until its device work is through {np.median(idles):.2f} ms (max {max(idles):.2f}), the first frame after it {np.median(first):.2f} ms | "
          f"host rebuilds {e.bvh_refits()[0]}, device builds {e.device_builds()} (+ {e.device_tree_refits()} refits), finite {bool(torch.isfinite(out).all())}")
    e.close()
