#!/usr/bin/env python3
"""What would a 4-wide compact BVH buy the fast build's rays?  A host model (no GPU) over the engine's own device stream.

VERDICT r4 item 1: the traversal loop is bound by the 64-B lines / 16-B texels it fetches per step and by its round trips; the
shipped compact binary node spends 32 B (2 texels; a quarter of them straddle two lines) on TWO child boxes. A 4-wide node with four
conservative f16 child boxes (4 x 3 axis words = 48 B) + four links (16 B) is exactly ONE aligned 64-B line for FOUR boxes.

The binary tree (the contract stream's, st_bvh.h) is collapsed top-down: a node's children are replaced by their own children —
largest surface area first — until it has four or only leaf runs are left. Rays (primary, one uniform-hemisphere bounce per hit
pixel = GI sampling's closest-hit ray, and a shadow ray from the bounce's hit to a light = its any-hit ray) are walked through both
trees with the product's rules (nearest child first, the others pushed, leaf runs triangle by triangle). Reported per ray kind:
steps per ray by kind, texels and lines fetched, and — for 64-ray waves of 8x8 pixel tiles — loop iterations per wave (= dependent
round trips) and the VALU issue model of DESIGN.md section 4 (a wave pays a body if any lane needs it).

  python tools/bvh4_sim.py [--scene dungeon|cornell] [--size W H] [--subdivide K] [--width 4]
"""
import argparse, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strolle_amd import Engine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dungeon")
ap.add_argument("--size", type=int, nargs=2, default=(160, 96))
ap.add_argument("--subdivide", type=int, default=0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--width", type=int, default=4)
args = ap.parse_args()
W, H = args.size
assert W % 8 == 0 and H % 8 == 0

e = Engine(device=-1)
if args.scene == "cornell":
    scenes.build_cornell(e); eye, target = (0.0, 1.0, 3.2), (0.0, 1.0, 0.0); light = np.array((0.0, 1.5, 0.5), np.float32)
else:
    scenes.build_dungeon(e, subdivide=args.subdivide); eye, target = (-5.75, 0.5, -16.8), (-5.75, 0.5, -17.0); light = None
e.tick()
S = e.read_scene(4).reshape(-1, 4, 4).astype(np.float32)
SU = S.view(np.uint32)
n_entries = len(S)
is_internal = SU[:, 0, 3] == 0
F32MAX = np.float32(3.4028235e38)


def area(lo, hi):
    d = np.maximum(hi - lo, 0.0)
    return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0])


def children2(k):
    """the two (lo, hi, entry) of binary internal entry k"""
    far = int(SU[k, 1, 3] // 64)
    return [(S[k, 0, :3], S[k, 1, :3], k + 1), (S[k, 2, :3], S[k, 3, :3], far)]


def collapse(width):
    """wide[k] = list of (lo, hi, entry) for every binary internal entry k that heads a wide node; returns (wide, root)"""
    wide = {}
    todo = [0] if is_internal[0] else []
    while todo:
        k = todo.pop()
        ch = children2(k)
        while len(ch) < width:
            cand = [(area(c[0], c[1]), i) for i, c in enumerate(ch) if is_internal[c[2]]]
            if not cand: break
            _, i = max(cand)
            c = ch.pop(i)
            ch[i:i] = children2(c[2])
        wide[k] = ch
        todo.extend(c[2] for c in ch if is_internal[c[2]])
    return wide


def slab(lo, hi, o, inv):
    t1 = (lo - o) * inv; t2 = (hi - o) * inv
    tmin = max(0.0, float(np.minimum(t1, t2).max())); tmax = float(np.maximum(t1, t2).min())
    return tmin if tmin <= tmax else float(F32MAX)


def tri(k, o, d, limit):
    p0, e1, e2 = S[k, 1, :3], S[k, 2, :3], S[k, 3, :3]
    pvec = np.cross(d, e2); det = float(e1 @ pvec)
    if abs(det) < 1.1920929e-07: return None
    inv = 1.0 / det
    tvec = o - p0; u = float(tvec @ pvec) * inv
    qvec = np.cross(tvec, e1); v = float(d @ qvec) * inv; t = float(e2 @ qvec) * inv
    if u < 0 or u > 1 or v < 0 or u + v > 1 or t <= 0 or t >= limit: return None
    return t


def walk(wide, o, d, limit=float(F32MAX), any_hit=False):
    """returns (t or None, [kinds]: 0 node step, 1 leaf step, deepest stack)"""
    inv = 1.0 / d
    kinds = []; stack = []; best = limit; found = None; deepest = 0
    cur = 0
    while True:
        if is_internal[cur]:
            kinds.append(0)
            hits = sorted(((slab(lo, hi, o, inv), c) for lo, hi, c in wide[cur]), key=lambda x: x[0])
            hits = [h for h in hits if h[0] < best]
            if hits:
                for h in reversed(hits[1:]): stack.append(h[1])
                deepest = max(deepest, len(stack))
                cur = hits[0][1]
                continue
        else:
            kinds.append(1)
            t = tri(cur, o, d, best)
            if t is not None:
                best = t; found = t
                if any_hit: return found, kinds, deepest
            if SU[cur, 0, 0] & 1:
                cur += 1; continue
        if not stack: break
        cur = stack.pop()
    return found, kinds, deepest


eye = np.array(eye, np.float32); fwd = np.array(target, np.float32) - eye; fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, np.array([0, 1, 0], np.float32)); right /= np.linalg.norm(right); up = np.cross(right, fwd)
tan = math.tan(math.pi / 8.0)
rng = np.random.default_rng(args.seed)


def fix(v):
    v = v.astype(np.float64); v[np.abs(v) < 1e-9] = 1e-9
    return v


trees = {2: collapse(2), args.width: collapse(args.width)}
n2, nw = len(trees[2]), len(trees[args.width])
fill = np.mean([len(c) for c in trees[args.width].values()])
print(f"{args.scene}: {n_entries} stream entries, {n2} binary internal nodes -> {nw} {args.width}-wide nodes ({fill:.2f} children per node); {int((~is_internal).sum())} leaf entries")
print(f"  stream bytes: compact binary {n_entries * 48 / 1e6:.2f} MB; wide nodes 64 B + leaf records 48 B: {(nw * 64 + int((~is_internal).sum()) * 48) / 1e6:.2f} MB")

# rays per 8x8 tile: primary, bounce (closest hit), shadow from the bounce's hit (any hit)
sets = {"primary (closest)": [], "GI bounce (closest)": [], "shadow from the bounce hit (any)": []}
for ty in range(0, H, 8):
    for tx in range(0, W, 8):
        lanes = {k: [] for k in sets}
        for y in range(ty, ty + 8):
            for x in range(tx, tx + 8):
                px = ((x + 0.5) / W * 2 - 1) * tan * (W / H); py = (1 - (y + 0.5) / H * 2) * tan
                d = fix((fwd + px * right + py * up) / np.linalg.norm(fwd + px * right + py * up))
                o = eye.astype(np.float64)
                rec = {}
                for w, tr in trees.items():
                    t, kinds, deep = walk(tr, o, d)
                    rec[w] = (kinds, deep)
                lanes["primary (closest)"].append(rec)
                if t is None: continue
                p = o + d * t; n = -d
                r = rng.normal(size=3); r /= np.linalg.norm(r)
                if r @ n < 0: r = -r
                r = fix(r); o2 = p + n * 1e-3
                rec = {}
                for w, tr in trees.items():
                    t2, kinds, deep = walk(tr, o2, r)
                    rec[w] = (kinds, deep)
                lanes["GI bounce (closest)"].append(rec)
                if t2 is None: continue
                p2 = o2 + r * t2 - r * 1e-3
                l = light if light is not None else p + np.array((0.0, 0.4, 0.0))   # dungeon: a torch a little above the primary hit
                dl = l - p2; dist = np.linalg.norm(dl)
                if dist < 1e-4: continue
                dl = fix(dl / dist)
                rec = {}
                for w, tr in trees.items():
                    _, kinds, deep = walk(tr, p2, dl, limit=dist, any_hit=True)
                    rec[w] = (kinds, deep)
                lanes["shadow from the bounce hit (any)"].append(rec)
        for k in sets: sets[k].append(lanes[k])

# cost model (VALU instructions per body): binary internal 2 x 14 + 12, wide 4 x 14 + 25, leaf 41; texels 2 / 4 / 3
VALU = {2: (40, 41), args.width: (14 * args.width + 25, 41)}
TEX = {2: (2, 3), args.width: (args.width, 3)}
LINES = {2: (1.25, 1.5), args.width: (1.0, 1.5)}   # a 48-B entry's 32-B head straddles a line a quarter of the time, its 48 B half of the time
for name, waves in sets.items():
    print(f"  {name}:")
    for w in (2, args.width):
        node = leaf = rays = iters = paid = need = 0; deep = 0; tex = lines = 0.0
        for lanes in waves:
            if not lanes: continue
            ks = [r[w][0] for r in lanes]
            deep = max(deep, max(r[w][1] for r in lanes))
            rays += len(ks)
            L = max(len(k) for k in ks)
            iters += L
            for i in range(L):
                a = any(len(k) > i and k[i] == 0 for k in ks); b = any(len(k) > i and k[i] == 1 for k in ks)
                paid += 64 * (VALU[w][0] * a + VALU[w][1] * b)
            for k in ks:
                nn = k.count(0); ll = len(k) - nn
                node += nn; leaf += ll
                need += VALU[w][0] * nn + VALU[w][1] * ll
                tex += TEX[w][0] * nn + TEX[w][1] * ll; lines += LINES[w][0] * nn + LINES[w][1] * ll
        nw_ = sum(1 for lanes in waves if lanes)
        print(f"    {w}-wide: {node / rays:6.1f} node + {leaf / rays:5.1f} leaf steps per ray | {tex / rays:6.1f} texels, {lines / rays:5.1f} lines per ray | "
              f"{iters / nw_:6.1f} loop iterations per wave | VALU paid per wave {paid / nw_ / 64:8.0f} (lane utilisation {need / paid:.2f}) | deepest stack {deep}")
