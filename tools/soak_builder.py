#!/usr/bin/env python3
"""Soak of the default refresh mode (ST_BVH_AUTO) under stress-bvh.rs-style churn: the dungeon renders while, tick after tick, instances of a small mesh
appear, disappear and move at random (device builds and refits of the tree, both scene copies alternating). Every 50 ticks the device's wide tree is read
back and walked from the root — every live triangle exactly once, no node twice — the frame must be finite and no wide walk may have overflowed; at the
end the primary hits are compared with an engine that builds the same final scene on the host.

    python tools/soak_builder.py [--ticks 1500] [--subdivide 0]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--ticks", type=int, default=1500); ap.add_argument("--subdivide", type=int, default=0); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--observers", action="store_true", help="every ~100 ticks a BvhHeatmap camera appears for two ticks (the host's tree comes back, then the device's again)")
args = ap.parse_args()
import torch
from strolle_amd import Buffer, CameraMode, Engine, Instance, Material, Mesh, scenes

rng = np.random.default_rng(args.seed)
pos = (rng.uniform(-0.3, 0.3, (200, 1, 3)) + rng.uniform(-0.05, 0.05, (200, 3, 3))).astype(np.float32)
nrm = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
blob = Mesh(pos, np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32))
size = (640, 360)

def place_at(p):
    m = np.eye(4, dtype=np.float32)[:3].copy(); m[:, 3] = p; return m

def reachable(e):
    nodes = e.read_scene(16).view(np.uint32).reshape(-1, 16); leaves = e.read_scene(17).reshape(-1, 3, 4)
    live = len(leaves)
    lw = nodes[:, 12:16]
    links = np.stack([lw[:, 0] & 0xffff, lw[:, 0] >> 16, lw[:, 1] & 0xffff, lw[:, 1] >> 16], 1) if live < 32768 else lw
    node_seen, leaf_seen = np.zeros(len(nodes), np.int64), np.zeros(live, np.int64)
    frontier = np.array([0], np.int64); node_seen[0] = 1
    while len(frontier):
        l = links[frontier]; used = l != 0
        idx = (l >> 1).astype(np.int64)
        np.add.at(leaf_seen, idx[used & ((l & 1) == 1)], 1)
        kids = idx[used & ((l & 1) == 0)]
        np.add.at(node_seen, kids, 1); frontier = kids
    return live, int((leaf_seen != 1).sum()), int(node_seen.max())

e = Engine(device=0)
scenes.build_dungeon(e, subdivide=args.subdivide); e.set_seed(1); e.insert_mesh(7777, blob)
e.insert_material(9000, Material(base_color=[0.8, 0.3, 0.2, 1.0]))   # the blobs' own material: edited at random below (a materials-only tick rebuilds on the device too)
desc = scenes.dungeon_camera(size, CameraMode.IMAGE, depth=1)
cam = e.create_camera(desc)
out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
alive = {}
next_handle = 7000
base = None
observers = 0
edits = 0
for tick in range(args.ticks):
    r = rng.random()
    if r < 0.25 and len(alive) < 40:
        p = (-5.75 + rng.uniform(-2, 2), rng.uniform(0.2, 1.5), -18.2 + rng.uniform(-3, 1)); alive[next_handle] = p
        e.insert_instance(next_handle, Instance(7777, 9000, place_at(p))); next_handle += 1
    elif r < 0.45 and alive:
        h = list(alive)[rng.integers(len(alive))]; e.remove_instance(h); del alive[h]
    elif r < 0.9 and alive:
        for h in list(alive)[: rng.integers(1, len(alive) + 1)]:
            p = tuple(np.add(alive[h], rng.uniform(-0.05, 0.05, 3))); alive[h] = p
            e.insert_instance(h, Instance(7777, 9000, place_at(p)))
    elif r < 0.95:
        e.insert_material(9000, Material(base_color=rng.uniform(0.1, 0.9, 3).tolist() + [1.0], perceptual_roughness=float(rng.uniform(0.2, 1.0)))); edits += 1
    if args.observers and tick % 97 == 60:
        hdesc = scenes.dungeon_camera((160, 96), CameraMode.BVH_HEATMAP)
        hcam = e.create_camera(hdesc); hout = torch.zeros((96, 160, 4), dtype=torch.float32, device="cuda:0")
        before = e.bvh_refits()[0]
        for _ in range(2):
            e.update_camera(hcam, hdesc); e.update_camera(cam, desc); e.tick(stream)
            e.render_camera(hcam, hout.data_ptr(), stream); e.render_camera(cam, out.data_ptr(), stream)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(hout).all()) and float(hout.abs().sum()) > 0.0, f"tick {tick}: the heatmap is empty"
        assert e.bvh_refits()[0] >= before
        e.delete_camera(hcam); observers += 1
    e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream)
    if tick % 50 == 49 or tick == args.ticks - 1:
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all()), f"tick {tick}: the frame is not finite"
        assert e.walk_overflow()[0] == 0, f"tick {tick}: a wide walk overflowed {e.walk_overflow()}"
        if e.device_builds() > 0:
            live, bad, twice = reachable(e)
            if base is None: base = live - 200 * len(alive)
            assert live == base + 200 * len(alive), f"tick {tick}: {live} leaf records for {len(alive)} instances"
            assert bad == 0 and twice == 1, f"tick {tick}: {bad} triangles not reached exactly once, a node linked {twice} times"
print(f"{args.ticks} ticks{f' ({observers} heatmap observers came and went)' if args.observers else ''}: {e.device_builds()} device builds, {e.device_tree_refits()} refits, {e.bvh_refits()[0]} host rebuilds, {len(alive)} instances alive, {edits} material edits, {e.walk_overflow()[0]} overflows", flush=True)
# the final scene on the host's tree: the same primary hits
ref_desc = scenes.dungeon_camera(size, CameraMode.REFERENCE, depth=0)
hits = []
for eng, fresh in ((e, False), (Engine(device=0), True)):
    if fresh:
        eng.set_bvh_refresh(0); scenes.build_dungeon(eng, subdivide=args.subdivide); eng.set_seed(1); eng.insert_mesh(7777, blob); eng.insert_material(9000, Material(base_color=[0.8, 0.3, 0.2, 1.0]))
        for h, p in alive.items(): eng.insert_instance(h, Instance(7777, 9000, place_at(p)))
    c = eng.create_camera(ref_desc)
    eng.update_camera(c, ref_desc); eng.tick(stream); eng.render_camera(c, out.data_ptr(), stream); torch.cuda.synchronize()
    hits.append(eng.read_buffer(c, Buffer.REF_HITS).reshape(size[1], size[0], -1).copy())
a, b = hits
fin = np.isfinite(a) & np.isfinite(b)
differ = (np.abs(np.where(fin, a, 0) - np.where(fin, b, 0)) > 1e-4 * np.maximum(1.0, np.abs(np.where(fin, b, 0)))) | (np.isfinite(a) != np.isfinite(b))
frac = float(differ.any(-1).mean())
print(f"primary hits of the churned engine against a host-built engine of the same final scene: {frac:.2e} of the pixels differ", flush=True)
assert frac <= 2e-3
print("soak ok")
