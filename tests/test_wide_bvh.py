"""The WIDE stream (StTuning::wide_bvh; k_bvh.hip k_bvh_wide, st_device.h closest_hit_wide): the host collapses the binary tree of the device
stream into nodes of up to four children once per build. CPU tests of that topology (host-only engines): it must be a re-bracketing of the
SAME tree — every leaf run reachable exactly once, every child box one of the contract stream's own boxes — and a walk over it must find
what a brute-force search over all triangles finds (strolle-gpu/src/ray.rs:114-266 is the walk being replaced; triangle.rs:64-113 the hit test)."""
import numpy as np
import pytest

from strolle_amd import Engine, scenes

F32MAX = float(np.float32(3.4028235e38))


def _scene(name):
    e = Engine(device=-1)
    if name == "cornell":
        scenes.build_cornell(e)
    elif name == "dungeon":
        scenes.build_dungeon(e)
    else:
        scenes.build_random_soup(e, 700, seed=11, n_lights=1)
    e.tick()
    stream = e.read_scene(4).reshape(-1, 4, 4)
    topo = e.read_scene(14).view(np.uint32)
    leaf_entry = e.read_scene(15).view(np.uint32)
    e.close()
    return stream, int(topo[0]) & 0xff, topo[1:].reshape(-1, 8), leaf_entry   # (bits 8.. of the first word: the walk's worst-case pending entries)


@pytest.mark.parametrize("name", ["cornell", "soup", "dungeon"])
def test_wide_topology_is_a_rebracketing_of_the_binary_tree(name):
    S, root, topo, leaf_entry = _scene(name)
    SU = S.view(np.uint32)
    internal = SU[:, 0, 3] == 0
    n_entries = len(S)
    # leaf records: every leaf entry once, in stream order (a run's records stay consecutive)
    assert np.array_equal(leaf_entry, np.flatnonzero(~internal).astype(np.uint32))
    leaf_index = {int(e): i for i, e in enumerate(leaf_entry)}
    assert root == 0 and internal[0]
    seen_runs, seen_nodes, used_src = set(), set(), set()
    # binary reference: the runs (first entry of each) reachable from the root
    runs = set()
    stack = [0]
    while stack:
        k = stack.pop()
        if internal[k]:
            stack += [k + 1, int(SU[k, 1, 3]) // 64]
        else:
            runs.add(k)
    todo = [0]
    fill = []
    while todo:
        n = todo.pop()
        assert n not in seen_nodes, "a wide node is linked twice"
        seen_nodes.add(n)
        src, link = topo[n, :4], topo[n, 4:]
        live = src != 0xffffffff
        assert live[:2].all() and (np.diff(live.astype(int)) <= 0).all(), "children fill the slots from the left"
        fill.append(int(live.sum()))
        for s, l in zip(src[live], link[live]):
            s, l = int(s), int(l)
            assert s not in used_src, "a box of the contract stream is used twice"
            used_src.add(s)
            entry, slot = s >> 1, s & 1
            assert internal[entry]
            child = entry + 1 if slot == 0 else int(SU[entry, 1, 3]) // 64       # whose box that is
            if l & 1:
                assert not internal[child] and leaf_index[child] == l >> 1, "a leaf link points at the first record of the child's run"
                assert child not in seen_runs
                seen_runs.add(child)
            else:
                assert internal[child]
                todo.append(l >> 1)
    assert seen_nodes == set(range(len(topo))), "every wide node is reachable"
    assert seen_runs == runs, "every leaf run of the binary tree is reachable exactly once"
    assert np.mean(fill) > 2.5 or n_entries < 16, f"the collapse leaves {np.mean(fill):.2f} children per node"
    assert max(fill) <= 4


def _f16_out(lo, hi):
    """the conservative f16 rounding k_bvh_wide applies (lower bounds down, upper bounds up)"""
    lo16 = lo.astype(np.float16); lo16 = np.where(lo16.astype(np.float32) > lo, np.nextafter(lo16, np.float16(-np.inf)), lo16)
    hi16 = hi.astype(np.float16); hi16 = np.where(hi16.astype(np.float32) < hi, np.nextafter(hi16, np.float16(np.inf)), hi16)
    return lo16.astype(np.float64), hi16.astype(np.float64)


def _walk(S, SU, topo, leaf_entry, o, d, deepest=None):
    """closest hit over the wide topology with conservative f16 boxes, nearest child first; returns (t, triangle) or (None, None);
    deepest: a one-element list that receives the most entries the stack held"""
    inv = 1.0 / d
    best, tri = F32MAX, None
    stack, cur = [], 0
    steps = 0
    while True:
        steps += 1
        if not cur & 1:
            hits = []
            for s, l in zip(topo[cur >> 1, :4], topo[cur >> 1, 4:]):
                if s == 0xffffffff:
                    continue
                at = S[int(s) >> 1, 2 * (int(s) & 1):2 * (int(s) & 1) + 2, :3]
                lo, hi = _f16_out(at[0], at[1])
                t1, t2 = (lo - o) * inv, (hi - o) * inv
                tmin, tmax = max(0.0, float(np.minimum(t1, t2).max())), float(np.maximum(t1, t2).min())
                if tmin <= tmax and tmin < best:
                    hits.append((tmin, int(l)))
            hits.sort()
            if hits:
                stack += [h[1] for h in reversed(hits[1:])]
                if deepest is not None:
                    deepest[0] = max(deepest[0], len(stack))
                cur = hits[0][1]
                continue
        else:
            k = int(leaf_entry[cur >> 1])
            p0, e1, e2 = (S[k, i, :3].astype(np.float64) for i in (1, 2, 3))
            pvec = np.cross(d, e2); det = float(e1 @ pvec)
            if abs(det) >= 1.1920929e-07:
                tvec = o - p0; u = float(tvec @ pvec) / det
                qvec = np.cross(tvec, e1); v = float(d @ qvec) / det; t = float(e2 @ qvec) / det
                if not (u < 0 or u > 1 or v < 0 or u + v > 1 or t <= 0 or t >= best):
                    best, tri = t, int(SU[k, 0, 1])
            if SU[k, 0, 0] & 1:
                cur += 2
                continue
        if not stack:
            break
        cur = stack.pop()
    return (best, tri, steps) if tri is not None else (None, None, steps)


@pytest.mark.parametrize("name", ["cornell", "soup"])
def test_a_walk_over_the_wide_topology_finds_the_brute_force_hit(name):
    S, root, topo, leaf_entry = _scene(name)
    SU = S.view(np.uint32)
    leaves = np.flatnonzero(SU[:, 0, 3] != 0)
    P0, E1, E2 = (S[leaves, i, :3].astype(np.float64) for i in (1, 2, 3))
    rng = np.random.default_rng(3)
    lo, hi = P0.min(0) - 0.5, P0.max(0) + 0.5
    hits = 0
    for _ in range(160):
        o = rng.uniform(lo, hi)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        d[np.abs(d) < 1e-6] = 1e-6
        # brute force Moeller-Trumbore over every triangle
        pvec = np.cross(d, E2); det = (E1 * pvec).sum(1)
        ok = np.abs(det) >= 1.1920929e-07
        with np.errstate(divide="ignore", invalid="ignore"):
            tvec = o - P0; u = (tvec * pvec).sum(1) / det
            qvec = np.cross(tvec, E1); v = (d * qvec).sum(1) / det; t = (E2 * qvec).sum(1) / det
        ok &= ~((u < 0) | (u > 1) | (v < 0) | (u + v > 1) | (t <= 0))
        want = float(t[ok].min()) if ok.any() else None
        got, tri, steps = _walk(S, SU, topo, leaf_entry, o, d)
        assert steps < 4 * len(S)
        if want is None:
            assert got is None
        else:
            hits += 1
            assert got is not None and abs(got - want) <= 1e-9 * max(1.0, abs(want)), (got, want)
    assert hits > 20, "the rays do not meet the scene"


def test_no_ray_of_config_3s_scene_comes_near_the_24_entry_stack():
    """BASELINE config 3's stand-in (the dungeon subdivided twice, 208 k triangles) is 26 internal nodes deep, the reference's
    traversal stack holds 24 pending entries (strolle-gpu/src/lib.rs:76, ray.rs:176-180). The oracle counts what its rays need: no push is
    dropped at 24, the deepest stack any ray reaches is about half of that, and rendering with an unbounded stack (64) gives the same bits —
    heatmap integers, G-buffer, GI samples. (The product's contract walks take a 26-entry stack for this tree: test_c_abi.py.)"""
    from oracle_binding import OracleEngine, set_stack_limit, stack_stats
    from parity import assert_bits_equal
    from strolle_amd import Buffer, CameraMode
    size = (240, 136)
    results = []
    try:
        for limit in (24, 64):
            set_stack_limit(limit)
            stack_stats(reset=True)
            o = OracleEngine(); scenes.build_dungeon(o, subdivide=2); o.set_seed(0)
            planes = {}
            for mode, frames, bufs in ((CameraMode.BVH_HEATMAP, 1, (Buffer.DBG_USED_MEMORY,)),
                                       (CameraMode.GI_DIFFUSE, 2, (Buffer.PRIM_GBUFFER_D0_A, Buffer.GI_D0, Buffer.GI_D1, Buffer.GI_RESERVOIRS_1))):
                desc = scenes.dungeon_camera(size, mode, depth=1)
                cam = o.create_camera(desc)
                for _ in range(frames):
                    o.update_camera(cam, desc); o.tick(); frame = o.render_camera(cam)
                planes[mode.name] = [frame] + [o.read_buffer(cam, b) for b in bufs]
            results.append((planes, stack_stats()))
            o.close()
    finally:
        set_stack_limit(24)
    (p24, (dropped24, deepest24)), (p64, (dropped64, deepest64)) = results
    assert dropped24 == 0 and dropped64 == 0, f"{dropped24} pushes were dropped at the reference's stack size"
    assert 8 <= deepest24 <= 16 and deepest24 == deepest64, (deepest24, deepest64)
    for mode in p24:
        for a, b in zip(p24[mode], p64[mode]):
            assert_bits_equal(a, b, f"{mode}: 24-entry stack vs unbounded")


def test_the_sliver_bundle_needs_more_than_24_pending_entries():
    """scenes.build_sliver_bundle is what the GPU test of the wide walk's overflow report renders (test_gpu_fast_tolerance.py): here the host model
    shows that it IS adversarial — a ray down the bundle's axis keeps more than the reference's 24 entries (strolle-gpu/src/lib.rs:76) pending in the
    4-wide walk — while the binary contract tree stays within what its own walks hold (at most 32: no ST_ERR_BVH_TOO_DEEP from the tree itself)."""
    e = Engine(device=-1)
    scenes.build_sliver_bundle(e)
    e.tick()
    depth, stack = e.bvh_depth()
    assert 24 <= depth <= 32 and stack == depth, (depth, stack)
    S = e.read_scene(4).reshape(-1, 4, 4)
    topo = e.read_scene(14).view(np.uint32)
    leaf_entry = e.read_scene(15).view(np.uint32)
    e.close()
    assert (int(topo[0]) >> 8) > 24, "the topology's own worst case fits the stack"
    deepest = [0]
    d = np.array([1.0, 1e-6, 1e-6]); d /= np.linalg.norm(d)
    _walk(S, S.view(np.uint32), topo[1:].reshape(-1, 8), leaf_entry, np.array([-3.0, 0.012, -0.012]), d, deepest)
    assert deepest[0] > 24, f"the axis ray keeps only {deepest[0]} entries pending"
