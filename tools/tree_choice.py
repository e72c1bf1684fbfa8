#!/usr/bin/env python3
"""Which first tree should ST_BVH_AUTO pick? The steady frame of the fast build over the host's binned-SAH tree (refresh mode 0: the reference's tree,
leaf runs) against the device builder's (mode 3: LBVH, single-triangle leaves), same box, alternating: the dungeon at 13 k / 52 k / 208 k triangles
(subdivide 0 / 1 / 2, tori included), 134 k without the tori, Image and GiDiffuse at 1920x1080, the 208 k scene at 3840x2160 too.

    python tools/tree_choice.py [--rounds 3]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--rounds", type=int, default=3); args = ap.parse_args()
import torch
from strolle_amd import CameraMode, Engine, scenes

def steady(subdivide, tori, mode, size, refresh):
    e = Engine(device=0); e.set_bvh_refresh(refresh)
    if isinstance(tori, int) and not isinstance(tori, bool): scenes.build_dungeon(e, subdivide=subdivide, tori=True, tori_subdivide=tori)   # (the tori split `tori` times)
    else: scenes.build_dungeon(e, subdivide=subdivide, tori=tori)
    e.set_seed(1)
    desc = scenes.dungeon_camera(size, mode, depth=1)
    cam = e.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    t_load = time.perf_counter()
    e.update_camera(cam, desc); e.tick(stream); torch.cuda.synchronize()
    t_load = (time.perf_counter() - t_load) * 1e3
    def frames(n):
        for _ in range(n):
            e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream)
    frames(36); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); frames(30); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
    builds = e.device_builds()
    e.close()
    return best, t_load, builds

extra = [("107 k (level x4, tori x16)", 1, 2, CameraMode.IMAGE, (1920, 1080)), ("139 k (level x16, tori x1)", 2, 0, CameraMode.IMAGE, (1920, 1080))]   # TREE_CHOICE_EXTRA=1
cases = [("13 k", 0, True, CameraMode.IMAGE, (1920, 1080)), ("52 k", 1, True, CameraMode.IMAGE, (1920, 1080)), ("134 k (no tori)", 2, False, CameraMode.IMAGE, (1920, 1080)),
         ("134 k (no tori)", 2, False, CameraMode.GI_DIFFUSE, (1920, 1080)), ("208 k", 2, True, CameraMode.IMAGE, (1920, 1080)), ("208 k", 2, True, CameraMode.GI_DIFFUSE, (1920, 1080)),
         ("208 k", 2, True, CameraMode.IMAGE, (3840, 2160))]

for name, sub, tori, mode, size in (extra if os.environ.get('TREE_CHOICE_EXTRA') else cases):
    rows = {0: [], 3: []}; loads = {0: [], 3: []}
    for _ in range(args.rounds):
        for refresh in (0, 3):
            ms, load, builds = steady(sub, tori, mode, size, refresh)
            assert builds == (1 if refresh == 3 else 0)
            rows[refresh].append(ms); loads[refresh].append(load)
    h, d = min(rows[0]), min(rows[3])
    print(f"dungeon {name:26s} {mode.name.lower():10s} {size[0]}x{size[1]}: host tree {h:.4f} ms ({' '.join(f'{x:.3f}' for x in rows[0])}) | device tree {d:.4f} ms ({' '.join(f'{x:.3f}' for x in rows[3])}) | "
          f"device / host {d / h:.3f} | first tick (upload + tree) {min(loads[0]):.1f} ms host, {min(loads[3]):.1f} ms device", flush=True)
