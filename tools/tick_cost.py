#!/usr/bin/env python3
"""Host cost of Engine::tick when the scene changes every frame (the stress-bvh.rs situation): one instance of the
dungeon is nudged per tick, which re-bakes it and rebuilds + re-flattens + re-uploads the BVH.
  python tools/tick_cost.py [--device 0|-1] [--subdivide K] [--refit]   (--refit: ST_BVH_REFIT, boxes refitted instead of a rebuild)"""
import argparse, os, sys, time
import numpy as np
if "--sync" in sys.argv:
    import torch   # torch's bundled HIP runtime has to be the first one the process loads
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strolle_amd import Engine, Instance, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--device", type=int, default=-1)
ap.add_argument("--subdivide", type=int, default=0)
ap.add_argument("--ticks", type=int, default=20)
ap.add_argument("--refit", nargs="?", const=1, default=0, type=int, help="1 = ST_BVH_REFIT (boxes refitted on the host instead of a rebuild), 2 = ST_BVH_REFIT_DEVICE (refitted by k_bvh.hip), 3 = ST_BVH_BUILD_DEVICE (rebuilt on the device by k_lbvh.hip)")
ap.add_argument("--sync", action="store_true", help="also time until the tick's device work is through (needs torch)")
ap.add_argument("--host-bake", action="store_true", help="StTuning::device_bake = 0: moved instances are baked on the host and sent (80 + 64 B per triangle)")
ap.add_argument("--all", action="store_true", help="move every instance per tick, not just one (stress-bvh.rs: many bodies under physics)")
args = ap.parse_args()
e = Engine(device=args.device)
scenes.build_dungeon(e, subdivide=args.subdivide)
e.set_bvh_refresh(args.refit)
if args.host_bake:
    e.set_tuning(device_bake=0)
e.tick()
npz = np.load(os.path.join(scenes.ASSETS, "dungeon.npz"))
base = npz["xform_0"].reshape(4, 3).T.copy()
mat = 1 + int(npz["material_0"])
ts, ds = [], []
for i in range(args.ticks):
    for k in (range(int(npz["n_meshes"])) if args.all else (0,)):
        x = npz[f"xform_{k}"].reshape(4, 3).T.copy(); x[0, 3] += 0.001 * (i + 1)
        e.insert_instance(1 + k, Instance(1 + k, 1 + int(npz[f"material_{k}"]), x))
    if args.sync: torch.cuda.synchronize()
    t = time.perf_counter(); e.tick(); ts.append(time.perf_counter() - t)
    if args.sync: torch.cuda.synchronize(); ds.append(time.perf_counter() - t)
bakes = e.device_bakes() if args.device >= 0 else (0, 0)
tris = e.read_scene(1).nbytes // 144
print(f"device={args.device} triangles={tris} refit={args.refit} (rebuilds, refits)={e.bvh_refits()}: tick with {'every' if args.all else 'one'} instance moved: median {np.median(ts)*1e3:.2f} ms, min {min(ts)*1e3:.2f} ms; device bakes {bakes[0]} ({bakes[1]} triangles)" + (f"; until the device is through: median {np.median(ds)*1e3:.2f} ms" if ds else "") + (f"; device builds {e.device_builds()}, refits of the device-built tree {e.device_tree_refits()}" if args.device >= 0 else ""))
