#!/usr/bin/env python3
"""Which first tree should ST_BVH_AUTO pick — and does it? The steady frame of the fast build over the host's binned-SAH tree (refresh mode 0: the reference's tree,
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
    if isinstance(tori, str): scenes.build_dungeon(e, subdivide=subdivide, copies=int(tori))   # (the level instanced that many times)
    elif isinstance(tori, (set, frozenset)): scenes.build_dungeon(e, subdivide=subdivide, tori=True, tori_subdivide=0, subdivide_meshes=tori)   # (only those of the level's 45 meshes split)
    elif isinstance(tori, int) and not isinstance(tori, bool): scenes.build_dungeon(e, subdivide=subdivide, tori=True, tori_subdivide=tori)   # (the tori split `tori` times)
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
    weight = e.auto_tree()[0] if refresh in (0, 4) else None   # (of the host's tree: modes 0 and 4 build it)
    e.close()
    return best, t_load, builds, weight

extra = [("134 k (16 copies of the level, nothing split)", 0, "16", CameraMode.IMAGE, (1920, 1080)), ("134 k (16 copies of the level, nothing split)", 0, "16", CameraMode.GI_DIFFUSE, (1920, 1080)),
         ("537 k (16 copies, x4)", 1, "16", CameraMode.IMAGE, (1920, 1080)),
         ("107 k (level x4, tori x16)", 1, 2, CameraMode.IMAGE, (1920, 1080)), ("139 k (level x16, tori x1)", 2, 0, CameraMode.IMAGE, (1920, 1080)),   # TREE_CHOICE_EXTRA=1
         ("85 k (meshes 0-16 x16)", 2, frozenset(range(17)), CameraMode.IMAGE, (1920, 1080)), ("47 k (meshes 0-15 x16)", 2, frozenset(range(16)), CameraMode.IMAGE, (1920, 1080)),
         ("51 k (mesh 16 x16)", 2, frozenset({16}), CameraMode.IMAGE, (1920, 1080)), ("67 k (meshes 17-44 x16)", 2, frozenset(range(17, 45)), CameraMode.IMAGE, (1920, 1080)),
         ("67 k (meshes 17-44 x16)", 2, frozenset(range(17, 45)), CameraMode.GI_DIFFUSE, (1920, 1080))]
cases = [("13 k", 0, True, CameraMode.IMAGE, (1920, 1080)), ("52 k", 1, True, CameraMode.IMAGE, (1920, 1080)), ("134 k (no tori)", 2, False, CameraMode.IMAGE, (1920, 1080)),
         ("134 k (no tori)", 2, False, CameraMode.GI_DIFFUSE, (1920, 1080)), ("208 k", 2, True, CameraMode.IMAGE, (1920, 1080)), ("208 k", 2, True, CameraMode.GI_DIFFUSE, (1920, 1080)),
         ("208 k", 2, True, CameraMode.IMAGE, (3840, 2160))]

for name, sub, tori, mode, size in (cases + extra if os.environ.get('TREE_CHOICE_EXTRA') else cases):
    rows = {0: [], 3: [], 4: []}; loads = {0: [], 3: [], 4: []}; weight = None; auto_builds = None
    for _ in range(args.rounds):
        for refresh in (0, 3, 4):   # 4 = ST_BVH_AUTO, the default: which tree does it pick, and is it the faster one?
            ms, load, builds, w = steady(sub, tori, mode, size, refresh)
            if refresh != 4: assert builds == (1 if refresh == 3 else 0)
            else: auto_builds = builds
            if w is not None: weight = w
            rows[refresh].append(ms); loads[refresh].append(load)
    h, d, a = min(rows[0]), min(rows[3]), min(rows[4])
    picked = "device" if auto_builds else "host"
    right = (picked == "device") == (d < h) or abs(d / h - 1.0) < 0.01
    print(f"dungeon {name:26s} {mode.name.lower():10s} {size[0]}x{size[1]}: host tree {h:.4f} ms ({' '.join(f'{x:.3f}' for x in rows[0])}) | device tree {d:.4f} ms ({' '.join(f'{x:.3f}' for x in rows[3])}) | "
          f"device / host {d / h:.3f} | leaf-run weight of the host's tree {weight:.2f} -> ST_BVH_AUTO picks the {picked} tree: {a:.4f} ms ({'the faster one' if right else 'NOT the faster one'}) | "
          f"first tick (upload + tree) {min(loads[0]):.1f} ms host, {min(loads[3]):.1f} ms device, {min(loads[4]):.1f} ms auto", flush=True)
