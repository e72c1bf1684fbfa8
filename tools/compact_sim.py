#!/usr/bin/env python3
"""What would compacting a workgroup's rays buy?  A host model (no GPU) of the step VERDICT r4 item 1b asks for before building: the GI bounce rays
(closest hit) and their shadow rays (any hit) of 256-ray workgroups — 4 waves = 4 horizontally adjacent 8 x 8 tiles, as the kernels launch them —
walked over the wide tree with the product's rules; each ray's sequence of node / leaf steps is recorded, then replayed under two schedules:

  shipped    every wave runs until its longest ray is through; an iteration costs the node body if any lane is at a node + the leaf body if any is
             at a leaf record (89 / 41 VALU instructions)
  compacted  every K iterations the workgroup meets at a barrier, counts its live rays (ballot + popcount per wave, prefix sum over the waves) and, when
             they fit fewer waves than hold them, repacks them densely into the first waves through LDS; emptied waves sleep at the barrier. A repack
             costs every live ray its state through LDS (C_SWAP VALU-equivalents per wave that takes part) and the barrier pair

Also printed: for the PRIMARY rays of each 8 x 8 tile, node steps per ray, of the tile's longest ray, and in the union of the tile's paths — what a
wave-wide packet visits. Then issue work per workgroup (VALU-equivalents summed over the waves that are awake), the lane utilisation of both, and the ratio.

  python tools/compact_sim.py [--scene dungeon] [--size 128 64] [--k 8]
"""
import argparse, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strolle_amd import Engine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dungeon")
ap.add_argument("--size", type=int, nargs=2, default=(128, 64))
ap.add_argument("--subdivide", type=int, default=0)
ap.add_argument("--k", type=int, nargs="*", default=[4, 8, 16])
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
W, H = args.size
assert W % 32 == 0 and H % 8 == 0
NODE, LEAF, C_SWAP, C_VOTE = 89.0, 41.0, 60.0, 12.0   # VALU-equivalents: node body, leaf body, a repack per participating wave, the vote every K iterations

e = Engine(device=-1)
scenes.build_dungeon(e, subdivide=args.subdivide) if args.scene != "cornell" else scenes.build_cornell(e)
eye, target = ((-5.75, 0.5, -16.8), (-5.75, 0.5, -17.0)) if args.scene != "cornell" else ((0.0, 1.0, 3.2), (0.0, 1.0, 0.0))
e.tick()
S = e.read_scene(4).reshape(-1, 4, 4).astype(np.float32); SU = S.view(np.uint32)
topo = e.read_scene(14).view(np.uint32)[1:].reshape(-1, 8); leaf_entry = e.read_scene(15).view(np.uint32)
e.close()
F = 3.4e38

def walk(o, d, limit=F, any_hit=False, ids=None):
    inv = 1.0 / d
    kinds = []; stack = []; best = limit; found = None; cur = 0
    while True:
        if ids is not None: ids.append(cur)
        if not cur & 1:
            kinds.append(0); hits = []
            for s, l in zip(topo[cur >> 1, :4], topo[cur >> 1, 4:]):
                if s == 0xffffffff: continue
                at = S[int(s) >> 1, 2 * (int(s) & 1):2 * (int(s) & 1) + 2, :3]
                t1, t2 = (at[0] - o) * inv, (at[1] - o) * inv
                a, b = max(0.0, float(np.minimum(t1, t2).max())), float(np.maximum(t1, t2).min())
                if a <= b and a < best: hits.append((a, int(l)))
            hits.sort()
            if hits:
                stack += [h[1] for h in reversed(hits[1:])]; cur = hits[0][1]; continue
        else:
            kinds.append(1)
            k = int(leaf_entry[cur >> 1]); p0, e1, e2 = S[k, 1, :3], S[k, 2, :3], S[k, 3, :3]
            pv = np.cross(d, e2); det = float(e1 @ pv)
            if abs(det) >= 1.19e-7:
                tv = o - p0; u = float(tv @ pv) / det; qv = np.cross(tv, e1); v = float(d @ qv) / det; t = float(e2 @ qv) / det
                if not (u < 0 or u > 1 or v < 0 or u + v > 1 or t <= 0 or t >= best):
                    best = t; found = t
                    if any_hit: return found, kinds
            if SU[k, 0, 0] & 1: cur += 2; continue
        if not stack: break
        cur = stack.pop()
    return found, kinds

eye = np.array(eye); fwd = np.array(target) - eye; fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right); up = np.cross(right, fwd); tan = math.tan(math.pi / 8)
rng = np.random.default_rng(args.seed)
def fix(v): v = v.copy(); v[np.abs(v) < 1e-9] = 1e-9; return v
groups = {"GI bounce (closest)": [], "shadow from the bounce hit (any)": []}
union = {"node_mean": 0.0, "node_max": 0, "node_union": 0, "leaf_max": 0, "leaf_union": 0, "tiles": 0}   # primary rays per 8 x 8 tile: what a wave-wide packet would visit
for ty in range(0, H, 8):
    for tx in range(0, W, 32):
        rays = {k: [] for k in groups}
        for w in range(4):
            tile_ids, per_node, per_leaf = set(), [], []
            for y in range(ty, ty + 8):
                for x in range(tx + 8 * w, tx + 8 * w + 8):
                    px = ((x + .5) / W * 2 - 1) * tan * (W / H); py = (1 - (y + .5) / H * 2) * tan
                    d = fix((fwd + px * right + py * up) / np.linalg.norm(fwd + px * right + py * up))
                    ids = []
                    t, _ = walk(eye, d, ids=ids)
                    tile_ids |= set(ids); per_node.append(sum(1 for i in ids if not i & 1)); per_leaf.append(sum(1 for i in ids if i & 1))
                    k1 = k2 = []
                    if t is not None:
                        p = eye + d * t; n = -d; r = rng.normal(size=3); r /= np.linalg.norm(r)
                        if r @ n < 0: r = -r
                        r = fix(r); o2 = p + n * 1e-3
                        t2, k1 = walk(o2, r)
                        if t2 is not None:
                            p2 = o2 + r * t2 - r * 1e-3; l = p + np.array((0.0, 0.4, 0.0)); dl = l - p2; dist = np.linalg.norm(dl)
                            if dist > 1e-4: _, k2 = walk(p2, fix(dl / dist), limit=dist, any_hit=True)
                    rays["GI bounce (closest)"].append(k1); rays["shadow from the bounce hit (any)"].append(k2)
            union["node_mean"] += float(np.mean(per_node)); union["node_max"] += max(per_node); union["node_union"] += sum(1 for i in tile_ids if not i & 1)
            union["leaf_max"] += max(per_leaf); union["leaf_union"] += sum(1 for i in tile_ids if i & 1); union["tiles"] += 1
        for k in groups: groups[k].append(rays[k])

def replay(block, K):
    """block: 256 step sequences in lane order. Returns (issue work, useful work)"""
    pos = [0] * 256
    slots = [list(range(64 * w, 64 * w + 64)) for w in range(4)]   # which rays each wave holds (None-free lists; finished rays stay until a repack)
    work = 0.0; useful = sum(NODE * k.count(0) + LEAF * k.count(1) for k in block)
    it = 0
    while True:
        live = [[r for r in s if pos[r] < len(block[r])] for s in slots]
        if not any(live): break
        if K and it and it % K == 0:
            awake = sum(1 for l in live if l)
            work += C_VOTE * awake
            total = sum(len(l) for l in live)
            need = (total + 63) // 64
            if need < awake:
                flat = [r for l in live for r in l]
                slots = [flat[64 * w:64 * w + 64] for w in range(4)]
                work += C_SWAP * awake
                live = slots
        for l in live:
            if not l: continue
            kinds = [block[r][pos[r]] for r in l]
            work += (NODE if 0 in kinds else 0.0) + (LEAF if 1 in kinds else 0.0)
            for r in l: pos[r] += 1
        it += 1
    return work, useful / 64.0

n = union["tiles"]
print(f"primary rays per 8 x 8 tile (what a wave-wide packet walks, st_device.h closest_hit_packet): node steps {union['node_mean'] / n:.1f} per ray, {union['node_max'] / n:.1f} for the tile's longest ray, "
      f"{union['node_union'] / n:.1f} in the UNION of the tile's paths; leaf records {union['leaf_max'] / n:.1f} (longest ray) / {union['leaf_union'] / n:.1f} (union)")
for name, blocks in groups.items():
    base = [replay(b, 0) for b in blocks]
    w0 = sum(x[0] for x in base); u0 = sum(x[1] for x in base)
    print(f"{name}: shipped schedule {w0 / len(blocks):8.0f} VALU-equivalents per workgroup, lane utilisation {u0 / w0:.3f}")
    for K in args.k:
        c = [replay(b, K) for b in blocks]
        w1 = sum(x[0] for x in c)
        print(f"    compacted every {K:2d} iterations: {w1 / len(blocks):8.0f} ({w1 / w0:.3f} of the shipped schedule's issue work), lane utilisation {u0 / w1:.3f}")
