#!/usr/bin/env python3
"""The device builder's Morton cells: steady frame over the device-built tree (refresh mode 3) for ST_LBVH_CELL_ASPECT in the environment (1 = cubic cells, 0 = every axis
its own 1,024 cells, between: an axis is quantised by max(its extent, aspect x the largest extent)), one process per value (tools/gpu_cell_aspect.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strolle_amd import CameraMode, Engine, scenes
cases = [("13 k", dict(subdivide=0), CameraMode.IMAGE), ("52 k", dict(subdivide=1), CameraMode.IMAGE), ("85 k (meshes 0-16 x16)", dict(subdivide=2, tori_subdivide=0, subdivide_meshes=frozenset(range(17))), CameraMode.IMAGE),
         ("134 k (no tori)", dict(subdivide=2, tori=False), CameraMode.GI_DIFFUSE), ("208 k", dict(subdivide=2), CameraMode.IMAGE), ("208 k", dict(subdivide=2), CameraMode.GI_DIFFUSE),
         ("139 k (16 copies)", dict(copies=16), CameraMode.IMAGE), ("139 k (16 copies)", dict(copies=16), CameraMode.GI_DIFFUSE), ("537 k (16 copies x4)", dict(subdivide=1, copies=16), CameraMode.IMAGE)]
size = (1920, 1080)
out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
row = []
for name, kw, mode in cases:
    e = Engine(device=0); e.set_bvh_refresh(3); scenes.build_dungeon(e, **kw); e.set_seed(1)
    desc = scenes.dungeon_camera(size, mode, depth=1); cam = e.create_camera(desc)
    def frames(n):
        for _ in range(n): e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream)
    frames(36); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); frames(30); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
    assert e.device_builds() == 1
    row.append(f"{best:.4f}"); e.close()
print(f"aspect {os.environ.get('ST_LBVH_CELL_ASPECT', 'default'):8s} " + " | ".join(row), flush=True)
if os.environ.get("CELL_ASPECT_HEADER"): print("scenes:          " + " | ".join(f"{n} {m.name.lower()}" for n, _, m in cases))
