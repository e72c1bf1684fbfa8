"""CPU tests of the product's boundary: the C-ABI library loads, exports every symbol include/strolle_hip.h
declares, and its host logic (scene stores, baking, BVH build/flatten, light table) equals the oracle's bit for bit.
No compute call is made (there is no GPU here); GPU-requiring calls must fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle_binding import OracleEngine
from parity import assert_bits_equal
from strolle_amd import Engine, Instance, Light, Material, Mesh, StrolleError, scenes
from strolle_amd.api import LIB_PATH, load_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "strolle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(st_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = load_library()
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_rust_facade_binds_exported_symbols():
    """rust/strolle-hip (the Rust facade a maintainer builds outside this box; no rustc here): every `st_*` function its
    `extern "C"` block declares is exported by the library and declared in the header, its `#[repr(C)]` structs list the
    header's fields in the header's order, and `Engine` offers every public method of the reference's engine
    (strolle/src/lib.rs:132-395)."""
    rust = os.path.join(ROOT, "rust", "strolle-hip", "src")
    ffi = open(os.path.join(rust, "ffi.rs")).read()
    bound = sorted(set(re.findall(r"pub fn (st_[a-z0-9_]+)\(", ffi)))
    assert len(bound) >= 25
    lib, declared = load_library(), set(declared_symbols())
    assert not [s for s in bound if not hasattr(lib, s)] and not [s for s in bound if s not in declared]
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "strolle_hip.h")).read(), flags=re.S)
    for name in ("StMeshTriangle", "StMaterial", "StLight", "StCamera", "StTuning", "StDistRect", "StDistUniqueId", "StDistGrid"):
        c_body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
        c_fields = [f.strip().split("[")[0] for decl in c_body.split(";") if decl.strip() for f in decl.strip().split(" ", 1)[1].split(",")]
        r_body = re.search(r"pub struct %s \{(.*?)\n\}" % name, ffi, re.S).group(1)
        r_fields = re.findall(r"pub (\w+):", r_body)
        assert r_fields == c_fields, (name, r_fields, c_fields)
    engine = open(os.path.join(rust, "lib.rs")).read()
    methods = ["new", "insert_mesh", "remove_mesh", "insert_material", "has_material", "remove_material", "insert_image", "remove_image", "insert_instance",
               "remove_instance", "insert_light", "remove_light", "update_sun", "create_camera", "update_camera", "render_camera", "delete_camera", "tick"]
    assert not [m for m in methods if not re.search(r"pub fn %s\b" % m, engine)]


def test_struct_sizes_match_header():
    from strolle_amd import api
    assert (C.sizeof(api.StMeshTriangle), C.sizeof(api.StMaterial), C.sizeof(api.StLight), C.sizeof(api.StCamera)) == (144, 88, 52, 160)


@pytest.mark.parametrize("scene", ["cornell", "soup", "dungeon", "dungeon33k"])
def test_host_engine_scene_buffers_equal_oracle(scene):
    # dungeon33k (every triangle split in four) is past the sizes at which baking and the BVH build go to the worker pool
    build = {"cornell": scenes.build_cornell, "soup": lambda e: scenes.build_random_soup(e, 5000, seed=3), "dungeon": scenes.build_dungeon,
             "dungeon33k": lambda e: scenes.build_dungeon(e, subdivide=1)}[scene]
    prod, orac = Engine(device=-1), OracleEngine()
    for e in (prod, orac):
        build(e)
        e.tick()
    names = ["bvh stream", "triangles (144 B)", "lights", "materials"]
    for what in range(4):
        assert_bits_equal(prod.read_scene(what), orac.read_scene(what), f"{scene}: {names[what]}")
    assert prod.world() == orac.world()


def test_bvh_stream_contract():
    """serializer.rs:20-110: root at 0, internal = 4 float4 with w==0 marker and right pointer, leaf flag bits."""
    prod = Engine(device=-1)
    scenes.build_cornell(prod); prod.tick()
    bvh = prod.read_scene(0).reshape(-1, 4)
    bits = bvh.view(np.uint32)
    assert bits[0, 3] == 0, "root is an internal node"
    seen_tris = set()
    ptr_stack = [0]
    while ptr_stack:
        p = ptr_stack.pop()
        if bits[p, 3] == 0:
            right = int(bits[p + 1, 3])
            assert p + 4 < right < len(bvh)
            ptr_stack += [p + 4, right]
            assert np.all(bvh[p, :3] <= bvh[p + 1, :3]) and np.all(bvh[p + 2, :3] <= bvh[p + 3, :3])
        else:
            while True:
                assert bits[p, 3] == 1
                seen_tris.add(int(bits[p, 1]))
                if bits[p, 0] & 1 == 0:
                    break
                p += 1
    assert seen_tris == set(range(32)), "every Cornell triangle is referenced exactly by the leaves"


@pytest.mark.parametrize("scene", ["cornell", "soup", "dungeon"])
def test_device_bvh_stream_is_the_serializers_stream_relaid(scene):
    """The device form of the BVH stream (st_types.h "device BVH stream", read_scene(4)): every entry 64 B, an internal node
    as the serializer wrote it with the far pointer turned into a byte offset, a leaf entry followed by its triangle's
    hit-test record (v0, v1 - v0, v2 - v0 of the reference's 144-B triangle). Walking both forms in lockstep must visit the
    same nodes, boxes, flags, triangles and materials."""
    prod = Engine(device=-1)
    {"cornell": scenes.build_cornell, "soup": lambda e: scenes.build_random_soup(e, 700, seed=3), "dungeon": scenes.build_dungeon}[scene](prod)
    prod.tick()
    ref = prod.read_scene(0).reshape(-1, 4); rb = ref.view(np.uint32)
    dev = prod.read_scene(4).reshape(-1, 4); db = dev.view(np.uint32)
    tris = prod.read_scene(1).reshape(-1, 36)   # 144-B triangles: d0..d8
    assert len(dev) % 4 == 0
    # entry k of the device stream <-> k-th entry (internal node or leaf entry) of the serializer's stream, in stream order
    starts, p = [], 0
    while p < len(ref):
        starts.append(p)
        p += 4 if rb[p, 3] == 0 else 1
    assert len(dev) == 4 * len(starts)
    where = {p: k for k, p in enumerate(starts)}
    for k, p in enumerate(starts):
        d = dev[4 * k:4 * k + 4]; u = db[4 * k:4 * k + 4]
        if rb[p, 3] == 0:
            assert u[0, 3] == 0
            assert np.array_equal(d[:, :3].view(np.uint32), ref[p:p + 4, :3].view(np.uint32)), "child boxes"
            assert u[1, 3] == 64 * where[int(rb[p + 1, 3])], "far pointer = byte offset of the far child's entry"
            assert where[p + 4] == k + 1, "the near child is the next entry"
        else:
            assert np.array_equal(u[0], rb[p]), "leaf entry texel"
            t = tris[int(rb[p, 1])]
            v0, v1, v2 = t[0:3], t[12:15], t[24:27]   # triangle.rs:9-21: (position, uv.x) (normal, uv.y) (tangent) per vertex
            assert np.array_equal(d[1, :3], v0) and np.array_equal(d[2, :3], v1 - v0) and np.array_equal(d[3, :3], v2 - v0)


def test_device_bvh_stream_edge_cases():
    """Empty scene, a scene whose tree is a single leaf, and a scene that changes between ticks (refit and rebuild): the
    device form always has four float4 per entry of the serializer's stream and follows it."""
    from strolle_amd import Instance, Material, Mesh

    def entries(e):
        ref = e.read_scene(0).reshape(-1, 4).view(np.uint32)
        n, p = 0, 0
        while p < len(ref):
            p += 4 if ref[p, 3] == 0 else 1
            n += 1
        return n

    e = Engine(device=-1)
    e.tick()
    assert e.read_scene(0).size == 0 and e.read_scene(4).size == 0
    e.insert_material(1, Material(base_color=(0.5, 0.5, 0.5, 1.0)))
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)
    nrm = np.tile(np.array([[[0, 0, 1]]], np.float32), (1, 3, 1))
    e.insert_mesh(1, Mesh(tri, nrm))
    eye = np.concatenate([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)], axis=1)
    e.insert_instance(1, Instance(1, 1, eye))
    e.tick()
    dev = e.read_scene(4).reshape(-1, 4)
    assert entries(e) == 1 and dev.shape == (4, 4), "one triangle: one leaf entry, no internal node"
    assert dev.view(np.uint32)[0, 3] != 0 and np.array_equal(dev[1:, :3], np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32))
    for refit in (False, True):
        e.set_bvh_refresh(refit)
        for k in range(3):
            more = tri + np.float32(2 + k)
            e.insert_mesh(10 + k, Mesh(more, nrm)); e.insert_instance(10 + k, Instance(10 + k, 1, eye))
            e.tick()
            assert e.read_scene(4).size == 16 * entries(e)
        for k in range(3):
            e.remove_instance(10 + k); e.remove_mesh(10 + k)
        e.tick()
        assert e.read_scene(4).size == 16 * entries(e) == 16
    e.close()


def test_light_table_remap_and_kill_equal_oracle():
    """lights.rs:97-154: removing a light shifts later slots, marks the killed slot 0xcafebabe and the remapped ones."""
    prod, orac = Engine(device=-1), OracleEngine()
    for e in (prod, orac):
        scenes.build_cornell(e)
        for i in range(2, 6):
            e.insert_light(i, Light.point((i, 1.0, 0.0), 0.1, (1.0, 1.0, 1.0), 10.0))
        e.tick()
        e.remove_light(3)
        e.insert_light(4, Light.spot((4, 2.0, 0.0), 0.2, (2.0, 1.0, 1.0), 10.0, (0.0, -1.0, 0.0), 0.5))
        e.tick()
    a, b = prod.read_scene(2), orac.read_scene(2)
    assert_bits_equal(a, b, "lights after remove/update")
    lights = a.reshape(-1, 7, 4).view(np.uint32)
    assert (lights[:, 3, 0] == 0xCAFEBABE).sum() == 1
    for e in (prod, orac):
        e.tick()
    assert_bits_equal(prod.read_scene(2), orac.read_scene(2), "slot flags cleared one frame later")
    assert prod.world() == orac.world()


def test_instance_update_and_removal_equal_oracle():
    prod, orac = Engine(device=-1), OracleEngine()
    from strolle_amd import Instance
    for e in (prod, orac):
        scenes.build_random_soup(e, 400, seed=8)
        e.tick()
        e.remove_instance(2)
        x = np.concatenate([np.eye(3, dtype=np.float32) * 0.5, np.array([[0.1], [0.2], [0.3]], np.float32)], axis=1)
        e.insert_instance(1, Instance(1, 3, x))  # same mesh, new material + transform
        e.tick()
    for what in range(4):
        assert_bits_equal(prod.read_scene(what), orac.read_scene(what), f"scene buffer {what} after edits")


def test_bvh_refresh_reuses_unchanged_subtrees_and_equals_a_fresh_build():
    """strolle/src/bvh/builder.rs:35-124: a refresh keeps the subtrees whose primitives did not change. Here the reused tree
    must be the tree a from-scratch build gives (the oracle engine rebuilds from scratch every tick), and moving one small
    instance of the dungeon must leave most primitives inside reused subtrees."""
    from strolle_amd import Instance
    npz = np.load(os.path.join(scenes.ASSETS, "dungeon.npz"))
    moved = 8                                  # dungeon mesh/instance handle 1+7: six triangles
    material = 1 + int(npz["material_7"])
    x = np.ascontiguousarray(npz["xform_7"].reshape(4, 3).T, np.float32)
    prod, orac = Engine(device=-1), OracleEngine()
    for e in (prod, orac):
        scenes.build_dungeon(e)
        e.tick()
    n0, r0 = prod.bvh_refresh()
    assert n0 == 8393 + 3 * 1536 and r0 == 0, "the first build has nothing to reuse"  # level.glb + the three tori
    for k in range(3):
        x = x.copy(); x[1, 3] += np.float32(0.125)
        for e in (prod, orac):
            e.insert_instance(moved, Instance(moved, material, x))
            e.tick()
        assert_bits_equal(prod.read_scene(0), orac.read_scene(0), f"BVH stream after move {k}")
        n, r = prod.bvh_refresh()
        assert n == n0 and r > n // 2, f"move {k}: {r} of {n} primitives reused"
    for e in (prod, orac):                     # removing it: fewer primitives, the rest still largely reusable
        e.remove_instance(moved)
        e.tick()
    assert_bits_equal(prod.read_scene(0), orac.read_scene(0), "BVH stream after the removal")
    n, r = prod.bvh_refresh()
    assert n == n0 - 6 and r > n // 2


def test_bvh_refit_mode_keeps_the_tree_while_instances_only_move():
    """ST_BVH_REFIT (SURVEY 8(f).2's alternative to rebuilding): moves refit the boxes of the existing tree, anything else
    rebuilds. The product refits the flat stream, the oracle its node tree; both must give the same stream, the boxes must
    contain their triangles, and the tree must be the old topology (a fresh build of the moved scene differs)."""
    from strolle_amd import Instance
    npz = np.load(os.path.join(scenes.ASSETS, "dungeon.npz"))
    prod, orac, fresh = Engine(device=-1), OracleEngine(), Engine(device=-1)
    for e in (prod, orac, fresh):
        scenes.build_dungeon(e)
    for e in (prod, orac):
        e.set_bvh_refresh(True)
    for e in (prod, orac, fresh):
        e.tick()
    assert prod.bvh_refits() == orac.bvh_refits() == (1, 0)
    topology = prod.read_scene(0).reshape(-1, 4).view(np.uint32)[:, 3].copy()    # .w: node kind, right pointers, leaf markers
    moved = [(8, 7), (1, 0), (17, 16)]
    for k in range(4):
        for handle, i in moved:
            x = np.ascontiguousarray(npz[f"xform_{i}"].reshape(4, 3).T, np.float32)
            x[:, 3] += np.float32(0.05 * (k + 1)) * np.array([1.0, 0.5, -0.25], np.float32)
            for e in (prod, orac, fresh):
                e.insert_instance(handle, Instance(handle, 1 + int(npz[f"material_{i}"]), x))
        for e in (prod, orac, fresh):
            e.tick()
        stream = prod.read_scene(0)
        assert_bits_equal(stream, orac.read_scene(0), f"refit {k}: BVH stream")
        assert_bits_equal(prod.read_scene(1), fresh.read_scene(1), f"refit {k}: triangles")
        assert np.array_equal(stream.reshape(-1, 4).view(np.uint32)[:, 3], topology), "a refit must not touch topology or leaf entries"
        assert not np.array_equal(stream, fresh.read_scene(0)), "the rebuilt tree of the moved scene is a different one"
        assert prod.bvh_refits() == orac.bvh_refits() == (1, k + 1)
    # every box contains the triangles below it: walk the stream
    s4 = prod.read_scene(0).reshape(-1, 4)
    tri = prod.read_scene(1).reshape(-1, 9, 4)
    def box_of(p):
        if s4[p].view(np.uint32)[3] == 0:
            lo_l, hi_l = box_of(p + 4); lo_r, hi_r = box_of(int(s4[p + 1].view(np.uint32)[3]))
            for (lo, hi), (blo, bhi) in (((lo_l, hi_l), (s4[p][:3], s4[p + 1][:3])), ((lo_r, hi_r), (s4[p + 2][:3], s4[p + 3][:3]))):
                assert np.all(blo <= lo) and np.all(bhi >= hi) and np.array_equal(blo, lo) and np.array_equal(bhi, hi), f"node {p}: box is not the tight union"
            return np.minimum(lo_l, lo_r), np.maximum(hi_l, hi_r)
        lo, hi = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
        while True:
            e = s4[p].view(np.uint32)
            v = tri[e[1]][[0, 3, 6], :3]
            lo, hi = np.minimum(lo, v.min(0)), np.maximum(hi, v.max(0))
            if not e[0] & 1:
                return lo, hi
            p += 1
    import sys
    sys.setrecursionlimit(10000)
    box_of(0)
    # anything but a move rebuilds: a material turning Blend changes leaf flags, a removed instance changes the leaves
    from strolle_amd import Material
    for e in (prod, orac):
        e.insert_material(1, Material(base_color=[1, 1, 1, 0.5], alpha_mode=1))
        e.insert_instance(8, Instance(8, 1 + int(npz["material_7"]), np.ascontiguousarray(npz["xform_7"].reshape(4, 3).T, np.float32)))
        e.tick()
    assert_bits_equal(prod.read_scene(0), orac.read_scene(0), "after a material change")
    assert prod.bvh_refits() == orac.bvh_refits() == (2, 4)
    for e in (prod, orac):
        e.remove_instance(8)
        e.tick()
    assert_bits_equal(prod.read_scene(0), orac.read_scene(0), "after a removal")
    assert prod.bvh_refits() == orac.bvh_refits() == (3, 4)
    # back in rebuild mode every change builds the tree a from-scratch build gives
    x = np.ascontiguousarray(npz["xform_0"].reshape(4, 3).T, np.float32)
    for e in (prod, orac, fresh):
        e.set_bvh_refresh(False) if e is not fresh else None
        e.remove_instance(8)
        e.insert_material(1, Material(base_color=[1, 1, 1, 0.5], alpha_mode=1))
        e.insert_instance(1, Instance(1, 1 + int(npz["material_0"]), x))
        e.tick()
    assert_bits_equal(prod.read_scene(0), fresh.read_scene(0), "rebuild mode again: product vs an engine that never refitted")
    assert_bits_equal(prod.read_scene(0), orac.read_scene(0), "rebuild mode again: product vs oracle")


@pytest.mark.parametrize("refit", [False, True])
def test_every_instance_moving_at_pool_size_equals_oracle(refit):
    """stress-bvh.rs in miniature, at a size where a refresh is baked on the worker pool (>= 16 k triangles): all instances
    of the 33 k-triangle dungeon move every tick; triangles, BVH stream and materials must stay equal to the oracle's, which
    bakes and builds on one thread."""
    from strolle_amd import Instance
    npz = np.load(os.path.join(scenes.ASSETS, "dungeon.npz"))
    prod, orac = Engine(device=-1), OracleEngine()
    for e in (prod, orac):
        scenes.build_dungeon(e, subdivide=1)
        e.set_bvh_refresh(refit)
        e.tick()
    for k in range(3):
        for i in range(int(npz["n_meshes"])):
            x = np.ascontiguousarray(npz[f"xform_{i}"].reshape(4, 3).T, np.float32)
            x[:, 3] += np.float32(0.01 * (k + 1) * (1 + i % 3))
            for e in (prod, orac):
                e.insert_instance(1 + i, Instance(1 + i, 1 + int(npz[f"material_{i}"]), x))
        for e in (prod, orac):
            e.tick()
        for what, name in enumerate(("BVH stream", "triangles", "lights", "materials")):
            assert_bits_equal(prod.read_scene(what), orac.read_scene(what), f"tick {k}: {name}")
    assert prod.bvh_refits() == orac.bvh_refits() == ((1, 3) if refit else (4, 0))


def _emulate_device_refit(stream, target, parent, local, items, batch_off, levels, entry_of_tri, tri_bounds, batch_limit=512):
    """k_bvh.hip on the host, in numpy: patch the leaf entries, then run the work list launch by launch, batch by batch. Inside a
    batch the items are taken in REVERSE order (the kernel's lanes arrive in no particular order; the result must not depend on
    it), child boxes live in a per-batch dictionary (the LDS slots) and only boxes stored by EARLIER launches are read from the
    stream — exactly what the kernel may rely on."""
    s = stream.reshape(-1, 4, 4).copy(); t = target.reshape(-1, 4, 4)
    su = s.view(np.uint32)
    leaf = su[:, 0, 3] != 0
    s[leaf, 1:] = t[leaf, 1:]                                            # k_bvh_patch_leaves (the moved records are in the target)
    assert np.array_equal(np.flatnonzero(leaf), np.sort(entry_of_tri[entry_of_tri != 0xffffffff])), "entry_of_tri names every leaf entry once"
    def grow(lo, hi, p): return np.minimum(lo, p), np.maximum(hi, p)
    stored = np.zeros((len(s), 2), bool)                                 # child slots written so far (by whom does not matter)
    for first, count in levels.reshape(-1, 2):
        written_now = []
        for b in range(first, first + count):
            work = items[batch_off[b]:batch_off[b + 1]]
            assert 0 < len(work) <= batch_limit
            slots, arrived = {}, {}
            for item in work[::-1]:
                e = int(item & 0x7fffffff); p = int(parent[e])
                lo, hi = np.full(3, np.float32(3.4028235e38)), np.full(3, np.float32(-3.4028235e38))
                if item >> 31:
                    assert stored[p >> 1, p & 1], "a finished task's box must have been stored by an earlier launch"
                    assert (p >> 1, p & 1) not in written_now, "... not by this one (no hand-off between workgroups inside a launch)"
                    lo, hi = s[p >> 1, 2 * (p & 1), :3].copy(), s[p >> 1, 2 * (p & 1) + 1, :3].copy(); have = True
                else:
                    k = e
                    while True:
                        tri = int(su[k, 0, 1])
                        for pt in (tri_bounds[2 * tri, :3], tri_bounds[2 * tri + 1, :3]): lo, hi = grow(lo, hi, pt)
                        if not su[k, 0, 0] & 1: break
                        k += 1
                    have = False
                while True:
                    pe, slot = p >> 1, p & 1
                    if not have:
                        s[pe, 2 * slot, :3] = lo; s[pe, 2 * slot + 1, :3] = hi; stored[pe, slot] = True; written_now.append((pe, slot))
                    l = int(local[pe] & 0x7fffffff)
                    assert l < batch_limit
                    slots[(l, slot)] = (lo, hi)
                    arrived[l] = arrived.get(l, 0) + 1
                    if arrived[l] == 1: break
                    assert arrived[l] == 2
                    lo, hi = np.full(3, np.float32(3.4028235e38)), np.full(3, np.float32(-3.4028235e38))
                    for c in (0, 1):
                        for pt in slots[(l, c)]: lo, hi = grow(lo, hi, pt)
                    p = int(parent[pe]); have = False
                    if p == 0xffffffff: break
                    if local[pe] >> 31:
                        s[p >> 1, 2 * (p & 1), :3] = lo; s[p >> 1, 2 * (p & 1) + 1, :3] = hi; stored[p >> 1, p & 1] = True; written_now.append((p >> 1, p & 1))
                        break
            assert all(v == 2 for v in arrived.values()), "every node of a batch must see both children arrive"
    internal = ~leaf
    assert stored[internal].all(), "every child box must have been rewritten"
    return s.reshape(-1)


@pytest.mark.parametrize("scene", ["cornell", "dungeon", "dungeon x4"])
def test_device_refit_work_list_reproduces_the_host_refit(scene):
    """ST_BVH_REFIT_DEVICE's host half (st_engine.cpp index_device_tree): the tree cut into 512-leaf tasks, batches and launches.
    A numpy emulation of k_bvh.hip run over that work list, starting from the stream of BEFORE the move, must arrive bit for bit at
    the device form of the host's own refit — with no box read inside a launch that the same launch wrote from another batch."""
    from strolle_amd import Instance
    e = Engine(device=-1)
    if scene == "cornell":
        scenes.build_cornell(e)
    else:
        scenes.build_dungeon(e, subdivide=1 if scene.endswith("x4") else 0)
    e.set_bvh_refresh(1)
    e.tick()
    before = e.read_scene(4).copy()
    u32 = lambda what: e.read_scene(what).view(np.uint32).copy()
    parent, local, items, batch_off, levels, entry_of_tri = (u32(w) for w in (7, 8, 9, 10, 11, 12))
    n_levels = len(levels) // 2
    assert n_levels == {"cornell": 1, "dungeon": 2, "dungeon x4": 2}[scene]
    assert batch_off[0] == 0 and batch_off[-1] == len(items) and np.all(np.diff(batch_off.astype(np.int64)) > 0)
    npz = np.load(os.path.join(scenes.ASSETS, "cornell.npz" if scene == "cornell" else "dungeon.npz"))
    for i in range(int(npz["n_meshes"])):
        if scene != "cornell" and i % 2: continue
        x = np.ascontiguousarray(npz[f"xform_{i}"].reshape(4, 3).T, np.float32)
        x[:, 3] += np.float32(0.02 * (1 + i % 4)) * np.array([1.0, -0.5, 0.25], np.float32)
        e.insert_instance(1 + i, Instance(1 + i, 1 + int(npz[f"material_{i}"]), x))
    e.tick()
    assert e.bvh_refits() == (1, 1)
    after = e.read_scene(4)
    assert not np.array_equal(before, after)
    for w, old in ((7, parent), (8, local), (9, items), (10, batch_off), (12, entry_of_tri)):
        assert np.array_equal(u32(w), old), "a refit keeps the tree, so its work list too"
    got = _emulate_device_refit(before, after, parent, local, items, batch_off, levels, entry_of_tri, e.read_scene(13).reshape(-1, 4))
    assert_bits_equal(got, after, "emulated device refit vs the host's refit")


def test_the_leaf_run_weight_of_the_hosts_tree():
    """st_debug_auto_tree: the surface-area-weighted mean leaf-run length of the host's binned-SAH tree, what ST_BVH_AUTO's choice of a scene's first tree rests on
    (above 3.4 the device builder's tree renders faster: profiles/r06_tree_choice*.txt). The Cornell box and the demo dungeon hang one or two triangles from
    nearly every leaf; the dungeon with every triangle split into 16 hangs runs of up to 140 coplanar ones from its large faces; 16 instanced copies of the level — as
    many triangles as that, none of them split — do not."""
    for build, lo, hi in ((scenes.build_cornell, 1.0, 2.0), (scenes.build_dungeon, 1.3, 2.2), (lambda e: scenes.build_dungeon(e, subdivide=2), 4.0, 5.0),
                          (lambda e: scenes.build_dungeon(e, copies=16), 1.3, 2.4)):
        e = Engine(device=-1)
        build(e); e.tick()
        weight, on_device = e.auto_tree()
        assert lo <= weight <= hi, weight
        assert on_device == (weight > 3.4)   # (the choice; a host-only engine has no device to build on and keeps the host's tree whatever it says)
        assert e.device_builds() == 0
        e.close()


def test_bvh_depth_is_reported():
    """st_debug_bvh_depth: the longest chain of internal nodes and the entries the launches' traversal stack holds for this tree — 24 as the
    reference's (strolle-gpu/src/lib.rs:76) while that is enough (Cornell box, demo dungeon), the chain's own length up to 32 for deeper trees
    (the synthetic 208 k-triangle dungeon: 26), so that no push is dropped."""
    for build, want in ((scenes.build_cornell, 24), (scenes.build_dungeon, 24), (lambda e: scenes.build_dungeon(e, subdivide=2), 26)):
        e = Engine(device=-1)
        build(e); e.tick()
        depth, stack = e.bvh_depth()
        assert stack == want and 5 <= depth <= stack, (depth, stack)
        e.close()


def _chain_scene(e, n):
    """n triangles whose centroids grow by 8x from one to the next: nearly every binned-SAH split (12 bins) peels the largest ones off — a tree
    that is almost a chain (coordinates stay between 2^-120 and 2^54 so that box areas stay finite floats)."""
    e.insert_material(1, Material(base_color=(0.8, 0.8, 0.8, 1.0)))
    pos = np.zeros((n, 3, 3), np.float32)
    for i in range(n):
        x = np.float32(2.0) ** (3 * i - 120)
        pos[i] = [[x, 0, 0], [x * 1.01, 0, 0], [x, x * 0.01, 0]]
    nrm = np.zeros_like(pos); nrm[..., 2] = 1
    e.insert_mesh(1, Mesh(pos, nrm)); e.insert_instance(1, Instance(1, 1, np.eye(4, dtype=np.float32)[:3]))


def test_a_tree_deeper_than_the_stack_fails_the_tick_loudly():
    """the traversal stack holds what the tree's deepest chain of internal nodes can need — 24 entries as the
    reference's (lib.rs:76), up to 32 for deeper trees (the subdivided dungeons: 25 / 26; no push is dropped and scenes.py no longer sets
    allow_deep_bvh). Only a tree deeper than 32 drops pushes: st_tick says ST_ERR_BVH_TOO_DEEP (once per build; the scene is uploaded all the
    same) unless StTuning::allow_deep_bvh accepts it."""
    from strolle_amd.api import ST_ERR_BVH_TOO_DEEP, StrolleError
    e = Engine(device=-1)
    scenes.build_dungeon(e, subdivide=1)
    assert e.tuning().allow_deep_bvh == 0
    e.tick()
    assert e.bvh_depth() == (25, 25), "the launches take a stack as deep as this tree needs"
    e.close()
    e = Engine(device=-1)
    _chain_scene(e, 46); e.tick()
    depth, stack = e.bvh_depth()
    assert 26 < depth <= 32 and stack == depth, (depth, stack)
    e.close()
    e = Engine(device=-1)
    _chain_scene(e, 58)
    with pytest.raises(StrolleError) as err:
        e.tick()
    depth, stack = e.bvh_depth()
    assert depth > 32 and stack == 32
    assert f"status {ST_ERR_BVH_TOO_DEEP}" in str(err.value) and f"{depth} internal nodes deep" in str(err.value) and "holds 32" in str(err.value)
    assert len(e.read_scene(0)) > 0, "the tick did its work before reporting"
    e.tick()                                     # reported once per build: a tick that rebuilds nothing is clean
    e.set_tuning(allow_deep_bvh=1)
    e.insert_light(99, Light.point((0.0, 1.0, 0.0), 0.1, (1.0, 1.0, 1.0), 5.0)); e.tick()
    e.close()


def test_tuning_round_trips_and_rejects_a_foreign_struct():
    """StTuning (include/strolle_hip.h): get / set round trip, defaults as documented, a wrong struct_size is refused."""
    from strolle_amd.api import StrolleError
    e = Engine(device=-1)
    t = e.tuning()
    assert t.struct_size == C.sizeof(type(t)) and t.overlap == 1 and t.fuse == 1 and t.tile_map == 1 and t.tile_map_denoise == 2
    assert t.anyhit_fast == 1 and t.allow_deep_bvh == 0
    e.set_tuning(fuse=0, tile_map=2, side_priority=-1)
    t2 = e.tuning()
    assert (t2.fuse, t2.tile_map, t2.side_priority, t2.overlap) == (0, 2, -1, 1)
    t2.struct_size = 12
    with pytest.raises(StrolleError):
        e.set_tuning(t2)
    with pytest.raises(StrolleError):
        e.set_tuning(tile_map=7)
    e.close()


def test_atlas_rectangles_are_released_and_reused():
    """images.rs:54-113: an image that is removed, or comes back with another size, gives its rectangle back. Rectangles of
    live images never overlap, stay inside the 8192 x 8192 atlas, and churn far beyond the atlas area never runs out of
    space while little is live. Placement is this project's policy (the reference's comes from the guillotiere crate), so
    the product is checked against the oracle's independent restatement of that policy."""
    rng = np.random.default_rng(11)
    prod, orac = Engine(device=-1), OracleEngine()
    live = {}
    inserted_area = 0
    for step in range(1200):
        if live and (len(live) >= 40 or rng.random() < 0.35):
            h = int(rng.choice(sorted(live)))
            del live[h]
            for e in (prod, orac): e.remove_image(h)
        else:
            h = int(rng.integers(1, 60))     # an id that is live: same size rewrites in place, another size re-allocates
            w, hh = (int(v) for v in rng.choice([64, 200, 512, 1024], 2))
            if h in live and rng.random() < 0.5:
                w, hh = live[h]
            img = np.full((hh, w, 4), step % 251, np.uint8)
            for e in (prod, orac): e.insert_image(h, img)
            live[h] = (w, hh)
            inserted_area += w * hh
        if step % 50 == 49 or step == 1199:
            rects = {h: prod.image_rect(h) for h in live}
            assert rects == {h: orac.image_rect(h) for h in live}, f"step {step}: product and oracle place images differently"
            for h, (x, y, w, hh) in rects.items():
                assert (w, hh) == live[h] and x + w <= 8192 and y + hh <= 8192
            items = sorted(rects.values())
            for i, a in enumerate(items):
                for b in items[i + 1:]:
                    assert a[0] + a[2] <= b[0] or b[0] + b[2] <= a[0] or a[1] + a[3] <= b[1] or b[1] + b[3] <= a[1], f"step {step}: {a} overlaps {b}"
    assert inserted_area > 8192 * 8192, "the churn has to exceed what a never-freeing allocator could hold"
    # a full atlas is reported (the reference warns and drops the image), and space comes back when images go away
    big = np.zeros((4096, 8192, 4), np.uint8)
    for h in list(live):
        for e in (prod, orac): e.remove_image(h)
    prod.insert_image(1, big); prod.insert_image(2, big)
    with pytest.raises(StrolleError, match="no more space in the atlas"):
        prod.insert_image(3, np.zeros((1, 1, 4), np.uint8))
    prod.remove_image(1)
    prod.insert_image(3, np.zeros((16, 16, 4), np.uint8))
    assert prod.image_rect(3) == (0, 0, 16, 16)


def test_gpu_calls_fail_loudly_without_a_device():
    prod = Engine(device=-1)
    scenes.build_cornell(prod)
    cam = prod.create_camera(scenes.cornell_camera((64, 64)))
    prod.tick()
    with pytest.raises(StrolleError, match="host-only"):
        prod.render_camera(cam, 0, 0)
    with pytest.raises(StrolleError):
        prod.read_buffer(cam, 0)
    with pytest.raises(StrolleError, match="camera does not exist"):
        prod.update_camera(999, scenes.cornell_camera((64, 64)))
    with pytest.raises(StrolleError, match="no triangles"):
        prod._check(prod._b.mesh_insert(prod._h, 77, None, 0))
    assert prod.has_material(1) and not prod.has_material(4242)
    prod.remove_light(4242); prod.remove_instance(4242); prod.remove_mesh(4242)  # unknown removes are silent no-ops


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "strolle_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f in ("api.py",) and "oracle" not in text.replace("no CPU", "").lower(), os.path.join(dirpath, f)
    import subprocess
    deps = subprocess.run(["ldd", LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_edit_sequences_keep_host_state_equal_to_oracle(seed):
    """Engine::tick under arbitrary histories (lib.rs:301-395): meshes, materials, images, instances and lights inserted,
    replaced and removed in random order — slot reuse in the triangle and material allocators, light-table remaps and kills,
    instances waiting for a mesh or material that arrives later — with every device-bound buffer compared after every tick."""
    from strolle_amd import Instance, Material, Mesh
    rng = np.random.default_rng(seed)
    prod, orac = Engine(device=-1), OracleEngine()
    engines = (prod, orac)
    meshes, materials, instances, lights, images = set(), set(), {}, set(), set()

    def rand_mesh():
        n = int(rng.integers(1, 40))
        c = rng.uniform(-2, 2, (n, 1, 3)).astype(np.float32)
        pos = c + rng.uniform(-0.3, 0.3, (n, 3, 3)).astype(np.float32)
        nrm = rng.standard_normal((n, 3, 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
        return Mesh(pos, nrm, rng.uniform(0, 1, (n, 3, 2)).astype(np.float32))

    def rand_xform():
        a = float(rng.uniform(0, 6.28))
        r = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32) * np.float32(rng.uniform(0.5, 1.5))
        return np.concatenate([r, rng.uniform(-1, 1, (3, 1)).astype(np.float32)], axis=1)

    for step in range(70):
        op = int(rng.integers(0, 11))
        if op == 0:
            h = int(rng.integers(1, 6)); m = rand_mesh(); meshes.add(h)
            for e in engines: e.insert_mesh(h, m)
        elif op == 1 and meshes:
            h = int(rng.choice(sorted(meshes))); meshes.discard(h)
            for e in engines: e.remove_mesh(h)
        elif op == 2:
            h = int(rng.integers(1, 6)); materials.add(h)
            tex = int(rng.choice(sorted(images))) if images and rng.random() < 0.5 else None
            mat = Material(base_color=rng.uniform(0, 1, 4).tolist(), perceptual_roughness=float(rng.uniform(0.05, 1)), metallic=float(rng.uniform(0, 1)),
                           emissive=rng.uniform(0, 2, 4).tolist(), alpha_mode=int(rng.integers(0, 2)), base_color_texture=tex)
            for e in engines: e.insert_material(h, mat)
        elif op == 3 and materials:
            h = int(rng.choice(sorted(materials))); materials.discard(h)
            for e in engines: e.remove_material(h)
        elif op in (4, 5):
            h = int(rng.integers(1, 8))
            inst = Instance(int(rng.integers(1, 6)), int(rng.integers(1, 6)), rand_xform())  # may name a mesh / material that does not exist (yet)
            instances[h] = inst
            for e in engines: e.insert_instance(h, inst)
        elif op == 6 and instances:
            h = int(rng.choice(sorted(instances))); instances.pop(h)
            for e in engines: e.remove_instance(h)
        elif op == 7:
            h = int(rng.integers(1, 7)); lights.add(h)
            l = Light.point(rng.uniform(-2, 2, 3).tolist(), float(rng.uniform(0.05, 0.3)), rng.uniform(0, 3, 3).tolist(), float(rng.uniform(5, 30))) if rng.random() < 0.6 \
                else Light.spot(rng.uniform(-2, 2, 3).tolist(), 0.1, rng.uniform(0, 3, 3).tolist(), 20.0, rng.uniform(-1, 1, 3).tolist(), float(rng.uniform(0.1, 1.0)))
            for e in engines: e.insert_light(h, l)
        elif op == 8 and lights:
            h = int(rng.choice(sorted(lights))); lights.discard(h)
            for e in engines: e.remove_light(h)
        elif op == 9:
            h = int(rng.integers(1, 4)); images.add(h)
            img = rng.integers(0, 256, (int(rng.integers(1, 20)), int(rng.integers(1, 20)), 4), dtype=np.uint8)
            for e in engines: e.insert_image(h, img)
        elif op == 10:
            az, alt = float(rng.uniform(0, 6.28)), float(rng.uniform(-1, 1.5))
            from strolle_amd import Sun
            for e in engines: e.update_sun(Sun(azimuth=az, altitude=alt))
        if step % 3 == 2 or step == 69:
            for e in engines: e.tick()
            for what, name in enumerate(["bvh stream", "triangles", "lights", "materials"]):
                assert_bits_equal(prod.read_scene(what), orac.read_scene(what), f"seed {seed} step {step}: {name}")
            assert prod.world() == orac.world(), f"seed {seed} step {step}: world"


def _compile_example(out_path, source="render_gltf.c"):
    import subprocess
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "examples", source), "-L", os.path.dirname(LIB_PATH), "-lstrolle_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + os.path.dirname(LIB_PATH), "-o", out_path]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_header_is_c99_and_the_c_example_links(tmp_path):
    """include/strolle_hip.h is a C header (the cgo / bindgen / ctypes side reads it as C): a strict C99 translation unit
    that includes it, and the C example that drives the whole ABI, must compile without warnings and link."""
    import subprocess
    probe = tmp_path / "probe.c"
    probe.write_text('#include "strolle_hip.h"\nint main(void) { StCamera c; StGltfOptions o; (void)c; (void)o; return 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(probe), "-o", str(tmp_path / "probe.o")],
                   check=True, capture_output=True, text=True)
    _compile_example(str(tmp_path / "render_gltf"))
    _compile_example(str(tmp_path / "dist_tiles"), "dist_tiles.c")   # the multi-GPU host loop of INTEGRATION.md section 6, as a program


def test_ctypes_structs_have_the_sizes_the_c_compiler_gives(tmp_path):
    """The Python mirror declares every struct of the header by hand; a C program that includes the header says what the
    sizes and a few telling offsets really are."""
    import subprocess
    from strolle_amd import api
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "strolle_hip.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu %zu %zu %zu %zu ", sizeof(StMeshTriangle), sizeof(StMaterial), sizeof(StLight), sizeof(StCamera), sizeof(StGltfOptions), sizeof(StGltfSummary), sizeof(StKernelProfile));\n'
                   '  printf("%zu %zu %zu %zu ", offsetof(StMaterial, base_color_texture), offsetof(StCamera, transform), offsetof(StGltfOptions, light_radius), offsetof(StKernelProfile, algorithmic_bytes));\n'
                   '  printf("%zu %zu %zu %zu %zu\\n", sizeof(StTuning), offsetof(StTuning, side_priority), offsetof(StTuning, device_bake), sizeof(StDistRect), sizeof(StDistUniqueId));\n'
                   '  return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True, text=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(api.StMeshTriangle), C.sizeof(api.StMaterial), C.sizeof(api.StLight), C.sizeof(api.StCamera), C.sizeof(api.StGltfOptions), C.sizeof(api.StGltfSummary),
            C.sizeof(api.StKernelProfile), api.StMaterial.base_color_texture.offset, api.StCamera.transform.offset, api.StGltfOptions.light_radius.offset,
            api.StKernelProfile.algorithmic_bytes.offset,
            C.sizeof(api.StTuning), api.StTuning.side_priority.offset, api.StTuning.device_bake.offset, C.sizeof(api.StDistRect), C.sizeof(api.StDistUniqueId)]
    assert got == want, (got, want)




def test_device_build_mode_without_a_device_builds_on_the_host():
    """ST_BVH_BUILD_DEVICE needs a device copy of the scene and nothing that observes the contract stream; a host-only engine has neither, so the
    mode falls back to the reference's rebuild: same stream, same refresh counts, no device builds."""
    a, b = Engine(device=-1), Engine(device=-1)
    b.set_bvh_refresh(3)
    for e in (a, b):
        scenes.build_cornell(e); e.tick()
        e.insert_light(77, Light.point((0.0, 1.0, 0.0), 0.1, (1.0, 1.0, 1.0), 5.0)); e.tick()
    assert_bits_equal(a.read_scene(0), b.read_scene(0), "BVH stream, ST_BVH_BUILD_DEVICE on a host-only engine")
    assert a.bvh_refits() == b.bvh_refits() and b.device_builds() == 0
    b.set_bvh_refresh(4)   # ST_BVH_AUTO, the default: on a host-only engine what ST_BVH_REBUILD does
    b.insert_light(78, Light.point((0.0, 1.2, 0.0), 0.1, (1.0, 1.0, 1.0), 5.0)); b.tick()
    assert b.device_builds() == 0
    with pytest.raises(StrolleError):
        b.set_bvh_refresh(5)
    for e in (a, b):
        e.close()
