#!/usr/bin/env python3
"""How much of the traversal loop's work is lost to lane divergence, and how much of it regrouping rays could win back —
a host-side model of st_device.h's traverse() over the engine's own device BVH stream (no GPU needed).

Every ray is traversed with the product's rules (near child first, far child pushed, 24-entry stack, leaf runs) in numpy and its
sequence of steps is recorded as internal (box pair) / leaf (triangle) steps. A wave of 64 rays walks in lockstep, so step i of
the wave costs B if any lane's i-th step is an internal one plus T if any lane's is a leaf one (the if-if loop skips a body no
lane wants); a lane that has finished idles. utilisation = work the lanes need / work the waves pay.
Groupings compared: the kernels' 8x8 pixel tiles; rays sorted by direction octant + Morton code of the origin cell (what a
wavefront-style regrouping pass could do); rays sorted by their own step count (an upper bound no real sort reaches).

  python tools/regroup_sim.py [--scene dungeon|cornell] [--size W H] [--subdivide K]
"""
import argparse, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strolle_amd import Engine, scenes

B_COST, T_COST = 1.0, 1.2   # relative issue cost of a box-pair step and a triangle step (57 + 35 vs ~70 + ~40 instructions)

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dungeon")
ap.add_argument("--size", type=int, nargs=2, default=(480, 272))
ap.add_argument("--subdivide", type=int, default=0)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
W, H = args.size
assert W % 8 == 0 and H % 8 == 0

e = Engine(device=-1)
if args.scene == "cornell":
    scenes.build_cornell(e); eye, target = (0.0, 1.0, 3.2), (0.0, 1.0, 0.0)
else:
    scenes.build_dungeon(e, subdivide=args.subdivide); eye, target = (-5.75, 0.5, -16.8), (-5.75, 0.5, -17.0)
e.tick()
S = e.read_scene(4).reshape(-1, 4, 4).astype(np.float32)   # device stream: 4 texels per entry
SU = S.view(np.uint32)
n_entries = len(S)
is_internal = SU[:, 0, 3] == 0


def traverse(origin, direction, max_steps=4096):
    """closest-hit traversal of all rays at once; returns (t, per-ray list of step kinds as a uint8 matrix, step counts)"""
    n = len(origin)
    inv = (1.0 / direction).astype(np.float32)
    ptr = np.zeros(n, np.int64); sp = np.zeros(n, np.int64); stack = np.zeros((n, 24), np.int64)
    best = np.full(n, np.float32(3.4028235e38)); alive = np.ones(n, bool)
    kinds = np.full((n, 512), 2, np.uint8); steps = np.zeros(n, np.int64)
    for _ in range(max_steps):
        idx = np.flatnonzero(alive)
        if not len(idx): break
        ent = S[ptr[idx]]; entu = SU[ptr[idx]]
        internal = entu[:, 0, 3] == 0
        st = steps[idx]; ok = st < kinds.shape[1]
        kinds[idx[ok], st[ok]] = np.where(internal[ok], 0, 1); steps[idx] += 1
        pop = np.zeros(len(idx), bool)
        # internal nodes
        ii = np.flatnonzero(internal); g = idx[ii]
        if len(ii):
            o, iv = origin[g], inv[g]
            def box(lo, hi):
                t1 = (lo - o) * iv; t2 = (hi - o) * iv
                tmin = np.maximum(0.0, np.minimum(t1, t2).max(1)); tmax = np.minimum(np.float32(3.4028235e38), np.maximum(t1, t2).min(1))
                return np.where(tmin <= tmax, tmin, np.float32(3.4028235e38))
            near_d = box(ent[ii, 0, :3], ent[ii, 1, :3]); far_d = box(ent[ii, 2, :3], ent[ii, 3, :3])
            near_p = ptr[g] + 1; far_p = (entu[ii, 1, 3] // 64).astype(np.int64)
            swap = far_d < near_d
            near_p, far_p = np.where(swap, far_p, near_p), np.where(swap, near_p, far_p)
            near_d, far_d = np.where(swap, far_d, near_d), np.where(swap, near_d, far_d)
            push = (far_d < best[g]) & (sp[g] < 24)
            stack[g[push], sp[g[push]]] = far_p[push]; sp[g[push]] += 1
            go = near_d < best[g]
            ptr[g[go]] = near_p[go]
            pop[ii[~go]] = True
        # leaf entries
        li = np.flatnonzero(~internal); g = idx[li]
        if len(li):
            p0, e1, e2 = ent[li, 1, :3], ent[li, 2, :3], ent[li, 3, :3]
            d, o = direction[g], origin[g]
            pvec = np.cross(d, e2); det = (e1 * pvec).sum(1)
            okd = ~(np.abs(det) < np.float32(1.1920929e-07))
            with np.errstate(divide="ignore", invalid="ignore"):
                inv_det = 1.0 / det
                tvec = o - p0; u = (tvec * pvec).sum(1) * inv_det
                qvec = np.cross(tvec, e1); v = (d * qvec).sum(1) * inv_det; t = (e2 * qvec).sum(1) * inv_det
            hit = okd & ~((u < 0) | (u > 1) | (v < 0) | (u + v > 1) | (t <= 0) | (t >= best[g]))
            best[g[hit]] = t[hit]
            more = (entu[li, 0, 0] & 1) != 0
            ptr[g[more]] += 1
            pop[li[~more]] = True
        pi = idx[pop]
        can = sp[pi] > 0
        sp[pi[can]] -= 1; ptr[pi[can]] = stack[pi[can], sp[pi[can]]]
        alive[pi[~can]] = False
    return best, kinds, steps


def wave_stats(kinds, steps, order):
    """utilisation of 64-ray waves taken in `order`"""
    need = paid = 0.0
    for w in range(0, len(order) - len(order) % 64, 64):
        r = order[w:w + 64]
        k = kinds[r][:, :int(steps[r].max())]
        any_int = (k == 0).any(0); any_leaf = (k == 1).any(0)
        paid += 64 * (B_COST * any_int.sum() + T_COST * any_leaf.sum())
        need += B_COST * (k == 0).sum() + T_COST * (k == 1).sum()
    return need / paid


def morton(ix, iy, iz):
    def part(x):
        x = x.astype(np.uint64) & 0x3ff
        x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249
        return x
    return part(ix) | (part(iy) << 1) | (part(iz) << 2)


# primary rays (pixel centres through a pi/4 perspective, as scenes.camera_for)
eye = np.array(eye, np.float32); fwd = np.array(target, np.float32) - eye; fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, np.array([0, 1, 0], np.float32)); right /= np.linalg.norm(right); up = np.cross(right, fwd)
ys, xs = np.mgrid[0:H, 0:W]
tan = math.tan(math.pi / 8.0)
px = ((xs + 0.5) / W * 2 - 1) * tan * (W / H); py = (1 - (ys + 0.5) / H * 2) * tan
dirs = (fwd[None, None] + px[..., None] * right + py[..., None] * up).reshape(-1, 3).astype(np.float32)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
dirs[np.abs(dirs) < 1e-9] = 1e-9
org = np.broadcast_to(eye, dirs.shape).astype(np.float32).copy()
tile_order = (((ys // 8) * (W // 8) + xs // 8) * 64 + (ys % 8) * 8 + xs % 8).reshape(-1).argsort()   # the kernels' wave = 8x8 tile

t, kinds, steps = traverse(org, dirs)
hit = t < 3e38
print(f"{args.scene}: {n_entries} BVH entries, {W}x{H}; primary rays: {steps.mean():.1f} steps per ray (max {steps.max()}), {hit.mean():.2f} hit")
print(f"  primary, 8x8 tiles:                 lane-step utilisation {wave_stats(kinds, steps, tile_order):.3f}")

# secondary rays: uniform hemisphere about a pseudo-normal (towards the camera side), from the primary hit points
rng = np.random.default_rng(args.seed)
sel = np.flatnonzero(hit)
p = org[sel] + dirs[sel] * t[sel, None]
n = -dirs[sel]
r = rng.normal(size=(len(sel), 3)).astype(np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
r[(r * n).sum(1) < 0] *= -1
r[np.abs(r) < 1e-9] = 1e-9
o2 = (p + n * 1e-3).astype(np.float32)
t2, k2, s2 = traverse(o2, r)
print(f"  secondary (one uniform-hemisphere ray per hit pixel): {s2.mean():.1f} steps per ray (max {s2.max()})")
in_tile = tile_order[np.isin(tile_order, sel)]
pos = {v: i for i, v in enumerate(sel)}
tile2 = np.array([pos[v] for v in in_tile])
print(f"  secondary, pixel order of the tiles: utilisation {wave_stats(k2, s2, tile2):.3f}")
octant = ((r[:, 0] > 0).astype(np.uint64) << 2) | ((r[:, 1] > 0).astype(np.uint64) << 1) | (r[:, 2] > 0).astype(np.uint64)
lo, hi = o2.min(0), o2.max(0)
cell = np.clip(((o2 - lo) / np.maximum(hi - lo, 1e-6) * 63).astype(np.int64), 0, 63)
key = (octant << np.uint64(32)) | morton(cell[:, 0], cell[:, 1], cell[:, 2])
print(f"  secondary, sorted by direction octant + origin cell: {wave_stats(k2, s2, np.argsort(key, kind='stable')):.3f}")
# a finer key: the direction on a 16x16 octahedral grid interleaved with a coarser origin cell
ad = np.abs(r).sum(1, keepdims=True); oc = r[:, :2] / ad
neg = r[:, 2] < 0
oc[neg] = (1 - np.abs(oc[neg][:, ::-1])) * np.sign(oc[neg])
dcell = np.clip(((oc * 0.5 + 0.5) * 15.999).astype(np.int64), 0, 15)
key2 = (morton(dcell[:, 0], dcell[:, 1], np.zeros(len(r), np.int64)) << np.uint64(32)) | morton(cell[:, 0] >> 2, cell[:, 1] >> 2, cell[:, 2] >> 2)
print(f"  secondary, sorted by 16x16 direction cell + coarse origin cell: {wave_stats(k2, s2, np.argsort(key2, kind='stable')):.3f}")
key3 = (morton(cell[:, 0] >> 1, cell[:, 1] >> 1, cell[:, 2] >> 1) << np.uint64(32)) | morton(dcell[:, 0], dcell[:, 1], np.zeros(len(r), np.int64))
print(f"  secondary, sorted by origin cell, then direction cell:          {wave_stats(k2, s2, np.argsort(key3, kind='stable')):.3f}")
print(f"  secondary, sorted by step count (upper bound):      {wave_stats(k2, s2, np.argsort(s2, kind='stable')):.3f}")

# What a workgroup could do on its own (no global pass): the rays of ONE block — 256 consecutive rays of the tile order, 4 waves — exchanged
# through LDS so that each wave takes rays of similar direction (or of similar length, the bound). Blocks of 1024 for comparison.
def block_sorted(base_order, key_of, block):
    out = []
    for b in range(0, len(base_order), block):
        r = base_order[b:b + block]
        out.append(r[np.argsort(key_of[r], kind="stable")])
    return np.concatenate(out)
dir_key = morton(dcell[:, 0], dcell[:, 1], np.zeros(len(r), np.int64)).astype(np.int64)
for block in (256, 1024):
    print(f"  secondary, blocks of {block} rays regrouped by direction octant:   {wave_stats(k2, s2, block_sorted(tile2, octant.astype(np.int64), block)):.3f}")
    print(f"  secondary, blocks of {block} rays regrouped by 16x16 direction cell: {wave_stats(k2, s2, block_sorted(tile2, dir_key, block)):.3f}")
    print(f"  secondary, blocks of {block} rays regrouped by step count (bound):  {wave_stats(k2, s2, block_sorted(tile2, s2, block)):.3f}")

# Two dependent rays per lane (GI sampling: the bounce ray, then the shadow ray from its hit point): a wave that runs them as two loops pays
# max(steps of ray 1) + max(steps of ray 2); a wave whose lanes each start their second ray as soon as their first is through pays
# max(steps of ray 1 + steps of ray 2). The second ray here: another uniform-hemisphere ray, from the first one's hit point.
h2 = t2 < 3e38
p2 = o2 + r * t2[:, None]
r3 = rng.normal(size=(len(sel), 3)).astype(np.float32); r3 /= np.linalg.norm(r3, axis=1, keepdims=True)
r3[(r3 * -r).sum(1) < 0] *= -1
r3[np.abs(r3) < 1e-9] = 1e-9
t3, k3, s3 = traverse((p2 - r * 1e-3).astype(np.float32), r3)
s3 = np.where(h2, s3, 0)
two = fused = need = 0
for w in range(0, len(tile2) - len(tile2) % 64, 64):
    q = tile2[w:w + 64]
    two += s2[q].max() + s3[q].max(); fused += (s2[q] + s3[q]).max(); need += (s2[q] + s3[q]).mean()
print(f"  two dependent rays per lane: wave steps as two loops {two}, as one loop over both {fused} ({fused / two:.3f}); lane-step utilisation {need / two:.3f} -> {need / fused:.3f}")
