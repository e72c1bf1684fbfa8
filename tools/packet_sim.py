#!/usr/bin/env python3
"""What a wave-wide PACKET traversal of any-hit (shadow) rays would cost against the per-lane loop — a host-side model over the
engine's own device BVH stream (no GPU needed).

Any-hit rays (`Ray::intersect`, strolle-gpu/src/ray.rs:84-112) return a boolean only, so the order in which a ray's nodes are
visited is free. A packet walks ONE node per step for the whole wave: every lane still alive in the subtree tests both child
boxes (its own arithmetic), a child is entered when any such lane reaches it (lanes that do not are masked off for that
subtree — each lane therefore visits exactly the leaves its own box tests admit, i.e. the same triangles as the per-lane loop
would, minus what an earlier hit makes unnecessary), control flow is wave-uniform, the node is fetched once (scalar load).

Cost model (steps, not instructions): per-lane loop = sum over lockstep steps of [any lane at an internal node] + [any lane at
a leaf entry] (the if-if loop executes a body when any lane wants it); packet = number of entries the packet visits.

  python tools/packet_sim.py [--scene dungeon|cornell] [--size W H] [--kind di|gi]
"""
import argparse, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strolle_amd import Engine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dungeon")
ap.add_argument("--size", type=int, nargs=2, default=(480, 272))
ap.add_argument("--subdivide", type=int, default=0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--kind", default="di")
ap.add_argument("--waves", type=int, default=400)
args = ap.parse_args()
W, H = args.size
F32MAX = np.float32(3.4028235e38)

e = Engine(device=-1)
if args.scene == "cornell":
    scenes.build_cornell(e); eye, target = (0.0, 1.0, 3.2), (0.0, 1.0, 0.0)
    lights = np.array([[0.0, 1.5, 0.5]], np.float32)
else:
    scenes.build_dungeon(e, subdivide=args.subdivide); eye, target = (-5.75, 0.5, -16.8), (-5.75, 0.5, -17.0)
    lights = np.array([(-3.0, 0.75, -23.0), (-23.5, 0.75, -31.0), (1.25, 0.75, -10.5), (-3.15, 0.75, 1.25), (-3.25, 0.75, 20.25), (13.25, 0.75, -28.25)], np.float32)
e.tick()
S = e.read_scene(4).reshape(-1, 4, 4).astype(np.float32)
SU = S.view(np.uint32)
LO0, HI0, LO1, HI1 = S[:, 0, :3], S[:, 1, :3], S[:, 2, :3], S[:, 3, :3]
FAR = (SU[:, 1, 3] // 64).astype(np.int64)
INTERNAL = SU[:, 0, 3] == 0
MORE = (SU[:, 0, 0] & 1) != 0


def box(lo, hi, o, iv):
    t1 = (lo - o) * iv; t2 = (hi - o) * iv
    tmin = np.maximum(np.float32(0), np.minimum(t1, t2).max(-1)); tmax = np.minimum(F32MAX, np.maximum(t1, t2).min(-1))
    return np.where(tmin <= tmax, tmin, F32MAX)


def tri(k, o, d, lim):
    p0, e1, e2 = S[k, 1, :3], S[k, 2, :3], S[k, 3, :3]
    pvec = np.cross(d, e2); det = (e1 * pvec).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.float32(1) / det
        tvec = o - p0; u = (tvec * pvec).sum(-1) * inv
        qvec = np.cross(tvec, e1); v = (d * qvec).sum(-1) * inv; t = (e2 * qvec).sum(-1) * inv
    return ~(np.abs(det) < np.float32(1.1920929e-07)) & ~((u < 0) | (u > 1) | (v < 0) | (u + v > 1) | (t <= 0) | (t >= lim))


def closest(origin, direction):
    """closest-hit distance of every ray (vectorised over all rays; product rules)"""
    n = len(origin); inv = (np.float32(1) / direction).astype(np.float32)
    ptr = np.zeros(n, np.int64); sp = np.zeros(n, np.int64); stack = np.zeros((n, 64), np.int64)
    best = np.full(n, F32MAX); alive = np.ones(n, bool)
    while alive.any():
        idx = np.flatnonzero(alive); k = ptr[idx]; internal = INTERNAL[k]; pop = np.zeros(len(idx), bool)
        ii = np.flatnonzero(internal); g = idx[ii]
        if len(ii):
            kk = k[ii]
            nd = box(LO0[kk], HI0[kk], origin[g], inv[g]); fd = box(LO1[kk], HI1[kk], origin[g], inv[g])
            npn, fpn = kk + 1, FAR[kk]; sw = fd < nd
            npn, fpn = np.where(sw, fpn, npn), np.where(sw, npn, fpn); nd, fd = np.where(sw, fd, nd), np.where(sw, nd, fd)
            push = fd < best[g]; stack[g[push], sp[g[push]]] = fpn[push]; sp[g[push]] += 1
            go = nd < best[g]; ptr[g[go]] = npn[go]; pop[ii[~go]] = True
        li = np.flatnonzero(~internal); g = idx[li]
        if len(li):
            kk = k[li]
            p0, e1, e2 = S[kk, 1, :3], S[kk, 2, :3], S[kk, 3, :3]; d, o = direction[g], origin[g]
            pvec = np.cross(d, e2); det = (e1 * pvec).sum(1)
            with np.errstate(divide="ignore", invalid="ignore"):
                iv = np.float32(1) / det; tvec = o - p0; u = (tvec * pvec).sum(1) * iv
                qvec = np.cross(tvec, e1); v = (d * qvec).sum(1) * iv; t = (e2 * qvec).sum(1) * iv
            hit = ~(np.abs(det) < np.float32(1.1920929e-07)) & ~((u < 0) | (u > 1) | (v < 0) | (u + v > 1) | (t <= 0) | (t >= best[g]))
            best[g[hit]] = t[hit]
            more = MORE[kk]; ptr[g[more]] += 1; pop[li[~more]] = True
        pi = idx[pop]; can = sp[pi] > 0
        sp[pi[can]] -= 1; ptr[pi[can]] = stack[pi[can], sp[pi[can]]]; alive[pi[~can]] = False
    return best


def per_lane_wave(o, d, lim, order="near"):
    """the shipped any-hit loop for one wave in lockstep: returns (paid internal bodies, paid leaf bodies, needed internal, needed leaf, occluded[])"""
    n = len(o); iv = (np.float32(1) / d).astype(np.float32)
    ptr = np.zeros(n, np.int64); sp = np.zeros(n, np.int64); stack = np.zeros((n, 64), np.int64)
    alive = np.ones(n, bool); found = np.zeros(n, bool); occluder = np.full(n, -1, np.int64)
    paid_i = paid_l = need_i = need_l = 0
    while alive.any():
        idx = np.flatnonzero(alive); k = ptr[idx]; internal = INTERNAL[k]; pop = np.zeros(len(idx), bool)
        ii = np.flatnonzero(internal); g = idx[ii]
        if len(ii):
            paid_i += 1; need_i += len(ii); kk = k[ii]
            nd = box(LO0[kk], HI0[kk], o[g], iv[g]); fd = box(LO1[kk], HI1[kk], o[g], iv[g])
            npn, fpn = kk + 1, FAR[kk]; sw = (fd < nd) if order == "near" else ((nd >= lim[g]) & (fd < lim[g]))
            npn, fpn = np.where(sw, fpn, npn), np.where(sw, npn, fpn); nd, fd = np.where(sw, fd, nd), np.where(sw, nd, fd)
            push = fd < lim[g]; stack[g[push], sp[g[push]]] = fpn[push]; sp[g[push]] += 1
            go = nd < lim[g]; ptr[g[go]] = npn[go]; pop[ii[~go]] = True
        li = np.flatnonzero(~internal); g = idx[li]
        if len(li):
            paid_l += 1; need_l += len(li); kk = k[li]
            hit = tri(kk, o[g], d[g], lim[g])
            found[g[hit]] = True; alive[g[hit]] = False; occluder[g[hit]] = kk[hit]
            more = MORE[kk] & ~hit; ptr[g[more]] += 1; pop[li[~more & ~hit]] = True
        pi = idx[pop]; can = sp[pi] > 0
        sp[pi[can]] -= 1; ptr[pi[can]] = stack[pi[can], sp[pi[can]]]; alive[pi[~can]] = False
    return paid_i, paid_l, need_i, need_l, found, occluder


def packet_wave(o, d, lim, order="count"):
    """masked packet traversal of one wave: returns (internal nodes visited, leaf entries visited, occluded[])"""
    n = len(o); iv = (np.float32(1) / d).astype(np.float32)
    found = np.zeros(n, bool)
    stack = []; cur = 0; mask = np.ones(n, bool)
    vi = vl = 0
    while True:
        mask = mask & ~found
        if not mask.any():
            if not stack: break
            cur, mask = stack.pop(); continue
        if INTERNAL[cur]:
            vi += 1
            hl = (box(LO0[cur], HI0[cur], o, iv) < lim) & mask
            hr = (box(LO1[cur], HI1[cur], o, iv) < lim) & mask
            l, r = cur + 1, FAR[cur]
            if hl.any() and hr.any():
                if order == "count" and hr.sum() > hl.sum(): stack.append((l, hl)); cur, mask = r, hr
                else: stack.append((r, hr)); cur, mask = l, hl
            elif hl.any(): cur, mask = l, hl
            elif hr.any(): cur, mask = r, hr
            else: mask = np.zeros(n, bool)
        else:
            vl += 1
            hit = tri(cur, o, d, lim) & mask
            found |= hit
            if MORE[cur]: cur += 1
            else: mask = np.zeros(n, bool)
    return vi, vl, found


# primary hits
eye = np.array(eye, np.float32); fwd = np.array(target, np.float32) - eye; fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, np.array([0, 1, 0], np.float32)); right /= np.linalg.norm(right); up = np.cross(right, fwd)
ys, xs = np.mgrid[0:H, 0:W]
tan = math.tan(math.pi / 8.0)
px = ((xs + 0.5) / W * 2 - 1) * tan * (W / H); py = (1 - (ys + 0.5) / H * 2) * tan
dirs = (fwd[None, None] + px[..., None] * right + py[..., None] * up).reshape(-1, 3).astype(np.float32)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True); dirs[np.abs(dirs) < 1e-9] = 1e-9
org = np.broadcast_to(eye, dirs.shape).astype(np.float32).copy()
t = closest(org, dirs)
hitp = (org + dirs * (t - np.float32(0.01))[:, None]).astype(np.float32)
is_hit = t < 3e38
rng = np.random.default_rng(args.seed)
print(f"{args.scene}: {len(S)} entries, {W}x{H}, {is_hit.mean():.2f} of the primary rays hit")

if args.kind == "gi":   # GI sampling b: shadow rays from SECONDARY hits (uniform hemisphere bounce) to a light
    n_ = -dirs
    r = rng.normal(size=dirs.shape).astype(np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
    r[(r * n_).sum(1) < 0] *= -1; r[np.abs(r) < 1e-9] = 1e-9
    t2 = closest(hitp, r)
    is_hit &= t2 < 3e38
    hitp = (hitp + r * (np.minimum(t2, 1e30) - np.float32(0.01))[:, None]).astype(np.float32)

# light choice: proportional to the unshadowed 1/d^2 (what RIS over the light table converges to), origin jittered in the light's radius
dl = hitp[:, None, :] - lights[None]; d2 = (dl * dl).sum(-1)
wgt = np.where(np.sqrt(d2) < 35.0, 1.0 / np.maximum(d2, 1e-3), 0.0) + 1e-12
cdf = np.cumsum(wgt / wgt.sum(1, keepdims=True), 1)
pick = (rng.random(len(hitp))[:, None] > cdf).sum(1).clip(0, len(lights) - 1)
j = rng.normal(size=hitp.shape).astype(np.float32); j /= np.linalg.norm(j, axis=1, keepdims=True)
lo = (lights[pick] + j * np.float32(0.15) * rng.random((len(hitp), 1)).astype(np.float32)).astype(np.float32)
sd = hitp - lo; sl = np.linalg.norm(sd, axis=1).astype(np.float32); sd = (sd / sl[:, None]).astype(np.float32); sd[np.abs(sd) < 1e-9] = 1e-9

tiles = [(ty, tx) for ty in range(H // 8) for tx in range(W // 8)]
rng.shuffle(tiles)
tot = dict(lane_i=0, lane_l=0, need_i=0, need_l=0, pk_i=0, pk_l=0, pf_i=0, pf_l=0, occl=0, rays=0, mism=0, lights=0)
for ty, tx in tiles[:args.waves]:
    yy, xx = np.mgrid[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
    ids = (yy * W + xx).reshape(-1); ids = ids[is_hit[ids]]
    if not len(ids): continue
    o, d, lim = lo[ids], sd[ids], sl[ids]
    pi, pl, ni, nl, f0, occ = per_lane_wave(o, d, lim)
    li_, ll_, _, _, fl_, _ = per_lane_wave(o, d, lim, "left")
    tot["left_i"] = tot.get("left_i", 0) + li_; tot["left_l"] = tot.get("left_l", 0) + ll_; tot["mism"] += int((fl_ != f0).sum())
    vi, vl, f1 = packet_wave(o, d, lim, "count")
    fi, fl, f2 = packet_wave(o, d, lim, "first")
    tot["lane_i"] += pi; tot["lane_l"] += pl; tot["need_i"] += ni; tot["need_l"] += nl
    tot["pk_i"] += vi; tot["pk_l"] += vl; tot["pf_i"] += fi; tot["pf_l"] += fl
    tot["occl"] += int(f0.sum()); tot["rays"] += len(ids); tot["mism"] += int((f0 != f1).sum() + (f0 != f2).sum())
    tot["lights"] += len(np.unique(pick[ids]))
    # last-occluder cache, one entry per (tile, light): what a random occluded lane with that light found in the previous frame
    done = np.zeros(len(ids), bool)
    for L in np.unique(pick[ids]):
        m = (pick[ids] == L)
        cand = occ[m & f0]
        if not len(cand): continue
        k = int(rng.choice(cand))
        done |= m & tri(k, o, d, lim)
    tot["c_hit"] = tot.get("c_hit", 0) + int(done.sum())
    tot["c_all"] = tot.get("c_all", 0) + int(done.all())
    if (~done).any():
        ci, cl, _, _, _, _ = per_lane_wave(o[~done], d[~done], lim[~done])
        tot["c_i"] = tot.get("c_i", 0) + ci; tot["c_l"] = tot.get("c_l", 0) + cl
    tot["c_l"] = tot.get("c_l", 0) + 1
nw = min(args.waves, len(tiles))
print(f"kind {args.kind}: {tot['rays']} shadow rays in {nw} waves, {tot['occl'] / max(tot['rays'], 1):.2f} occluded, {tot['lights'] / nw:.2f} distinct lights per wave; boolean mismatches packet vs per-lane: {tot['mism']}")
print(f"  per-lane loop : {tot['lane_i'] / nw:7.1f} internal + {tot['lane_l'] / nw:6.1f} leaf bodies per wave (lane utilisation {(tot['need_i'] + tot['need_l']) / (64.0 * (tot['lane_i'] + tot['lane_l'])):.2f})")
print(f"  per-lane, left child first (no near/far sort): {tot['left_i'] / nw:7.1f} internal + {tot['left_l'] / nw:6.1f} leaf bodies per wave")
print(f"  per-lane + (tile, light) last-occluder cache: {tot['c_i'] / nw:7.1f} internal + {tot['c_l'] / nw:6.1f} leaf bodies per wave; {tot['c_hit'] / max(tot['rays'], 1):.2f} of the rays end at the cached triangle, {tot['c_all'] / nw:.2f} of the waves entirely  -> {(tot['lane_i'] + tot['lane_l']) / max(tot['c_i'] + tot['c_l'], 1):.2f}x fewer bodies")
print(f"  packet (larger child first): {tot['pk_i'] / nw:7.1f} internal + {tot['pk_l'] / nw:6.1f} leaf entries per wave  -> {(tot['lane_i'] + tot['lane_l']) / max(tot['pk_i'] + tot['pk_l'], 1):.2f}x fewer bodies")
print(f"  packet (left child first)  : {tot['pf_i'] / nw:7.1f} internal + {tot['pf_l'] / nw:6.1f} leaf entries per wave  -> {(tot['lane_i'] + tot['lane_l']) / max(tot['pf_i'] + tot['pf_l'], 1):.2f}x fewer bodies")

# ---- a WORLD-SPACE last-occluder cache: key = (light, endpoint cell); filled by an independent set of rays (the frame before),
# then every ray of this frame tests the triangle its key holds before traversing
print("world-space last-occluder cache (key: light x endpoint cell), filled by the previous frame's rays:")
ids_all = np.flatnonzero(is_hit)
sub = ids_all[rng.permutation(len(ids_all))[:min(len(ids_all), 64 * args.waves)]]
def occluders_of(o, d, lim):
    out = np.full(len(o), -1, np.int64); f = np.zeros(len(o), bool)
    for s in range(0, len(o), 64):
        _, _, _, _, ff, oc = per_lane_wave(o[s:s + 64], d[s:s + 64], lim[s:s + 64]); out[s:s + 64] = oc; f[s:s + 64] = ff
    return f, out
# previous frame: same pixels, other light jitter (and possibly another light)
pick_prev = (rng.random(len(hitp))[:, None] > cdf).sum(1).clip(0, len(lights) - 1)
j2 = rng.normal(size=hitp.shape).astype(np.float32); j2 /= np.linalg.norm(j2, axis=1, keepdims=True)
lo2 = (lights[pick_prev] + j2 * np.float32(0.15) * rng.random((len(hitp), 1)).astype(np.float32)).astype(np.float32)
sd2 = hitp - lo2; sl2 = np.linalg.norm(sd2, axis=1).astype(np.float32); sd2 = (sd2 / sl2[:, None]).astype(np.float32); sd2[np.abs(sd2) < 1e-9] = 1e-9
fprev, oprev = occluders_of(lo2[ids_all], sd2[ids_all], sl2[ids_all])
for cell in (0.125, 0.25, 0.5, 1.0):
    def key(lightsel, pts): 
        c = np.floor(pts / cell).astype(np.int64)
        return (lightsel.astype(np.int64) * 73856093) ^ (c[:, 0] * 19349663) ^ (c[:, 1] * 83492791) ^ (c[:, 2] * 2654435761)
    table = {}
    kp = key(pick_prev[ids_all], hitp[ids_all])
    for k_, f_, o_ in zip(kp, fprev, oprev):
        if f_: table[int(k_) & 0xfffff] = int(o_)
    kc = key(pick[sub], hitp[sub])
    ent = np.array([table.get(int(k_) & 0xfffff, -1) for k_ in kc])
    have = ent >= 0
    hitc = np.zeros(len(sub), bool)
    hitc[have] = tri(ent[have], lo[sub][have], sd[sub][have], sl[sub][have])
    # cost per wave of 8x8 tiles is not defined for a random subset; report ray-level rates
    print(f"  cell {cell:5.3f}: {have.mean():.2f} of the rays find an entry, {hitc.mean():.2f} end at it ({len(table)} entries in use)")

# ---- regrouping WITHIN a block: the 256 shadow rays of four horizontally adjacent tiles (one workgroup) sorted by the light they go to,
# then cut into four waves again (an LDS shuffle of ray records + two barriers on the device) — does the per-lane loop get cheaper?
print("block-level regrouping by light (4 adjacent tiles = 256 rays -> 4 waves):")
tot_a = tot_b = tot_c = 0; nblk = 0
blocks = [(ty, gx) for ty in range(H // 8) for gx in range(W // 32)]
rng.shuffle(blocks)
for ty, gx in blocks[:max(1, args.waves // 4)]:
    yy, xx = np.mgrid[ty * 8:ty * 8 + 8, gx * 32:gx * 32 + 32]
    # tile order: wave w = tile gx*4 + w
    ids_t = [((yy[:, w * 8:(w + 1) * 8]) * W + xx[:, w * 8:(w + 1) * 8]).reshape(-1) for w in range(4)]
    ids_t = [i[is_hit[i]] for i in ids_t]
    allids = np.concatenate(ids_t)
    if len(allids) < 64: continue
    a = sum(sum(per_lane_wave(lo[i], sd[i], sl[i])[:2]) for i in ids_t if len(i))
    order = allids[np.lexsort((allids, pick[allids]))]                      # by light, then pixel
    b = sum(sum(per_lane_wave(lo[order[s:s + 64]], sd[order[s:s + 64]], sl[order[s:s + 64]])[:2]) for s in range(0, len(order), 64))
    # by light, then by ray length (a proxy for traversal length known before tracing)
    order2 = allids[np.lexsort((sl[allids], pick[allids]))]
    c = sum(sum(per_lane_wave(lo[order2[s:s + 64]], sd[order2[s:s + 64]], sl[order2[s:s + 64]])[:2]) for s in range(0, len(order2), 64))
    tot_a += a; tot_b += b; tot_c += c; nblk += 1
print(f"  bodies per block: pixel tiles {tot_a / nblk:.1f}; sorted by light {tot_b / nblk:.1f} ({tot_a / tot_b:.2f}x fewer); by light then ray length {tot_c / nblk:.1f} ({tot_a / tot_c:.2f}x fewer)")
