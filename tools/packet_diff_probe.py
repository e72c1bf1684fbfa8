#!/usr/bin/env python3
"""How do the packet walk's primary hits differ from the per-lane walk's? (depth by relative size, packed material / base-colour bytes)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strolle_amd import Buffer, CameraMode, Engine, scenes
size = (1920, 1080)
runs = []
for packets in (1, 0):
    e = Engine(device=0, exact=False); e.set_tuning(primary_packets=packets); e.keep_all_planes(True)
    scenes.build_dungeon(e); e.set_seed(5)
    desc = scenes.dungeon_camera(size, CameraMode.IMAGE, depth=1); cam = e.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for _ in range(2):
        e.update_camera(cam, desc); e.tick(); e.render_camera(cam, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    runs.append({b: e.read_buffer(cam, b).reshape(size[1], size[0], 4) for b in (Buffer.PRIM_GBUFFER_D0_A, Buffer.PRIM_GBUFFER_D1_A)})
    e.close()
a, b = runs[0][Buffer.PRIM_GBUFFER_D0_A], runs[1][Buffer.PRIM_GBUFFER_D0_A]
da, db = a[..., 0].astype(np.float64), b[..., 0].astype(np.float64)
rel = np.abs(da - db) / np.maximum(np.abs(db), 1e-30)
print("pixels", da.size, "hit", (db != 0).mean())
for th in (0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
    print(f"  depth differs by more than {th:g} (relative): {(rel > th).mean():.3e}")
print("  packed material bytes (d0.w) differ:", (a[..., 3].view(np.uint32) != b[..., 3].view(np.uint32)).mean())
print("  packed base colour (d1.w) differ:", (runs[0][Buffer.PRIM_GBUFFER_D1_A][..., 3].view(np.uint32) != runs[1][Buffer.PRIM_GBUFFER_D1_A][..., 3].view(np.uint32)).mean())
print("  normal (d0.yz) differs by more than 1e-4:", (np.abs(a[..., 1:3] - b[..., 1:3]) > 1e-4).any(-1).mean())
big = rel > 1e-3
ys, xs = np.nonzero(big)
print("  pixels whose depth differs by > 1e-3:", int(big.sum()), "first few:", list(zip(xs[:8].tolist(), ys[:8].tolist())), [ (float(da[y,x]), float(db[y,x])) for x,y in zip(xs[:4],ys[:4])])
