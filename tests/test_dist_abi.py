"""Multi-GPU behind the C ABI (include/strolle_hip.h st_dist_*, strolle_amd/csrc/st_dist.cpp): the tile partition rule, the
window (tile + apron) a rank renders, and the gather of the tiles to rank 0 — driven here through the in-process transport on
host-only engines, i.e. without a GPU and without RCCL (same partition, same pack / unpack, same call sequence as the RCCL path)."""
import ctypes as C

import numpy as np
import pytest

from strolle_amd import Engine, StrolleError, scenes
from strolle_amd.api import OutputFormat, dist_partition, dist_window
from strolle_amd.distributed import tile_for_rank, tile_overhead, tile_window


@pytest.mark.parametrize("size", [(3840, 2160), (1920, 1080), (640, 360), (200, 120)])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8])
def test_partition_tiles_cover_the_frame_exactly_once(size, world):
    w, h = size
    cover = np.zeros((h, w), np.int32)
    tiles = [dist_partition(w, h, world, r) for r in range(world)]
    for x0, y0, x1, y1 in tiles:
        assert x0 % 16 == 0 and y0 % 8 == 0 and (x1 % 16 == 0 or x1 == w) and (y1 % 8 == 0 or y1 == h)
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()
    if world in (4, 8) and size == (3840, 2160):    # BASELINE.json configs 4 / 5: "4-tile" 2x2, "8-tile" 4x2
        assert tiles[0] == ((0, 0, 1920, 1080) if world == 4 else (0, 0, 960, 1080))
    if world == 2:
        assert tiles[0][2] == w, "two ranks: row bands (contiguous sends, no column seam)"
    # explicit column counts: row bands (cols = 1) and column strips (cols = world)
    bands = [dist_partition(w, h, world, r, cols=1) for r in range(world)]
    assert all(b[0] == 0 and b[2] == w for b in bands)


def test_window_adds_the_apron_only_towards_neighbours():
    w, h = 3840, 2160
    t = dist_partition(w, h, 8, 1)                       # (960, 0, 1920, 1080): neighbours left, right and below
    assert dist_window(w, h, t, 16) == (944, 0, 1936, 1096)
    assert dist_window(w, h, t, 0) == t
    corner = dist_partition(w, h, 8, 7)                  # bottom-right tile
    assert dist_window(w, h, corner, 16) == (2864, 1064, 3840, 2160)
    assert dist_window(w, h, (0, 0, w, h), 128) == (0, 0, w, h)
    # config 5 at 8 ranks must not pay more than 8 % in redundant pixels (row bands paid 15.6 %)
    mx, mean = tile_overhead(w, h, 8, 16)
    assert mx <= 0.08 and mean <= 0.05, (mx, mean)
    assert tile_overhead(w, h, 8, 16, cols=1)[0] > 0.10, "the row-band partition it replaces"
    assert tile_for_rank(w, h, 8, 3) == dist_partition(w, h, 8, 3) and tile_window(w, h, t, 16) == dist_window(w, h, t, 16)


def _engines(world, size, group, fmt=None):
    out = []
    for r in range(world):
        e = Engine(device=-1)
        scenes.build_cornell(e)
        cam = e.create_camera(scenes.cornell_camera(size))
        if fmt is not None:
            e.set_output_format(cam, fmt)
        e.dist_init_local(r, world, group)
        out.append((e, cam))
    return out


@pytest.mark.parametrize("world,cols,fmt,dtype", [(4, 0, None, np.float32), (8, 0, None, np.float32), (2, 0, None, np.float32), (3, 0, None, np.float32),
                                                  (4, 4, None, np.float32), (4, 0, OutputFormat.RGBA8_UNORM_SRGB, np.uint8), (6, 0, OutputFormat.RGBA16F, np.uint16)])
def test_local_transport_gathers_tiles_to_rank_0(world, cols, fmt, dtype):
    """Every rank hands over its tile of a frame it alone can have produced (the rank number + a position ramp); rank 0's assembled
    frame must be tile r from rank r, bit for bit — for bands (contiguous sends), 2-D tiles (packed) and the narrower formats."""
    w, h = 352, 200
    ranks = _engines(world, (w, h), group=1000 + world * 16 + cols, fmt=fmt)
    chan = 4
    frames = []
    yy, xx = np.mgrid[0:h, 0:w]
    for r in range(world):
        f = np.empty((h, w, chan), dtype)
        base = (r * 37 + xx * 3 + yy * 5)
        for c in range(chan):
            f[..., c] = (base + c).astype(dtype)
        frames.append(np.ascontiguousarray(f))
    full = np.zeros((h, w, chan), dtype)
    expect = np.zeros_like(full)
    for r, (e, cam) in enumerate(ranks):
        owned, window = e.dist_set_partition(cam, cols=cols, apron=16)
        assert owned == dist_partition(w, h, world, r, cols) and window == dist_window(w, h, owned, 16)
        x0, y0, x1, y1 = owned
        expect[y0:y1, x0:x1] = frames[r][y0:y1, x0:x1]
    for r in range(world - 1, -1, -1):     # non-root ranks first (in-process transport)
        e, cam = ranks[r]
        e.dist_gather(cam, frames[r].ctypes.data, full.ctypes.data if r == 0 else 0)
    assert np.array_equal(full, expect)
    # a second frame through the same mailbox, root assembling IN PLACE (full == its own frame)
    for f in frames:
        f += 1
    expect2 = frames[0].copy()
    for r in range(1, world):
        x0, y0, x1, y1 = dist_partition(w, h, world, r, cols)
        expect2[y0:y1, x0:x1] = frames[r][y0:y1, x0:x1]
    for r in range(world - 1, -1, -1):
        e, cam = ranks[r]
        e.dist_gather(cam, frames[r].ctypes.data, frames[0].ctypes.data if r == 0 else 0)
    assert np.array_equal(frames[0], expect2)
    for e, _ in ranks:
        e.dist_shutdown(); e.close()


def test_gather_says_when_a_rank_is_missing_and_when_nothing_was_set_up():
    (e0, c0), (e1, c1) = _engines(2, (64, 48), group=77)
    buf = np.zeros((48, 64, 4), np.float32)
    with pytest.raises(StrolleError, match="st_dist_set_partition"):
        e0.dist_gather(c0, buf.ctypes.data, buf.ctypes.data)
    e0.dist_set_partition(c0); e1.dist_set_partition(c1)
    with pytest.raises(StrolleError, match="has not handed over"):
        e0.dist_gather(c0, buf.ctypes.data, buf.ctypes.data)       # rank 1 has not sent
    e1.dist_gather(c1, buf.ctypes.data)
    e0.dist_gather(c0, buf.ctypes.data, buf.ctypes.data)
    with pytest.raises(StrolleError, match="has not handed over"):
        e0.dist_gather(c0, buf.ctypes.data, buf.ctypes.data)       # the same tile is not taken twice
    plain = Engine(device=-1)
    scenes.build_cornell(plain)
    cam = plain.create_camera(scenes.cornell_camera((64, 48)))
    with pytest.raises(StrolleError, match="st_dist_init"):
        plain.dist_set_partition(cam)
    with pytest.raises(StrolleError):
        plain.set_camera_window(cam, 8, 0, 64, 48)                 # columns must sit on multiples of 16
    plain.set_camera_window(cam, 16, 8, 48, 40)
    for e in (e0, e1, plain):
        e.close()


def test_a_partition_belongs_to_the_frame_size_it_was_computed_for():
    """st_camera_update with a new size rebuilds the camera and resets its window, while the cached partition kept the old tile —
    peers would send the old tile's bytes, the root expect the new one's. The gather now refuses until st_dist_set_partition is called
    again; a failed st_dist_set_partition leaves nothing behind; st_camera_delete takes the camera's partition with it."""
    (e0, c0), (e1, c1) = _engines(2, (64, 48), group=78)
    buf = np.zeros((96, 128, 4), np.float32)
    e0.dist_set_partition(c0); e1.dist_set_partition(c1)
    e1.dist_gather(c1, buf.ctypes.data); e0.dist_gather(c0, buf.ctypes.data, buf.ctypes.data)
    for e, c in ((e0, c0), (e1, c1)):
        e.update_camera(c, scenes.cornell_camera((128, 96)))      # a resize: buffers rebuilt, window back to the whole frame
    with pytest.raises(StrolleError, match="size changed"):
        e1.dist_gather(c1, buf.ctypes.data)
    with pytest.raises(StrolleError, match="size changed"):
        e0.dist_gather(c0, buf.ctypes.data, buf.ctypes.data)
    o0, _ = e0.dist_set_partition(c0); o1, _ = e1.dist_set_partition(c1)
    assert o0 == dist_partition(128, 96, 2, 0) and o1 == dist_partition(128, 96, 2, 1)
    e1.dist_gather(c1, buf.ctypes.data); e0.dist_gather(c0, buf.ctypes.data, buf.ctypes.data)
    # a window that no longer covers the rank's own tile is refused too
    e1.set_camera_window(c1, 0, 0, 128, 8)
    with pytest.raises(StrolleError, match="window no longer covers"):
        e1.dist_gather(c1, buf.ctypes.data)
    # a request that fails (3 columns do not divide 2 ranks) leaves no zero rectangle a later gather would accept
    e2 = Engine(device=-1); scenes.build_cornell(e2)
    c2 = e2.create_camera(scenes.cornell_camera((64, 48)))
    e2.dist_init_local(0, 2, 79)
    with pytest.raises(StrolleError, match="multiple of the column count"):
        e2.dist_set_partition(c2, cols=3)
    with pytest.raises(StrolleError, match="st_dist_set_partition has not been called"):
        e2.dist_gather(c2, buf.ctypes.data, buf.ctypes.data)
    # deleting the camera forgets its partition: a camera that reuses nothing of it has to set its own
    e2.dist_set_partition(c2)
    e2.delete_camera(c2)
    c3 = e2.create_camera(scenes.cornell_camera((64, 48)))
    with pytest.raises(StrolleError):
        e2.dist_gather(c3, buf.ctypes.data, buf.ctypes.data)
    for e in (e0, e1, e2):
        e.close()


def test_dist_init_refuses_a_truncated_unique_id():
    e = Engine(device=-1)
    with pytest.raises(StrolleError, match="128 bytes"):
        e.dist_init(0, 2, b"\x01" * 64)
    e.close()


# ---- cost-weighted grids (st_dist_grid / st_dist_grid_rebalance / st_dist_set_grid)
from strolle_amd.api import dist_grid, dist_grid_rebalance, dist_grid_tile  # noqa: E402


def _covers_exactly_once(tiles, w, h):
    cover = np.zeros((h, w), np.int32)
    for x0, y0, x1, y1 in tiles:
        assert x0 % 16 == 0 and y0 % 8 == 0 and x0 < x1 <= w and y0 < y1 <= h
        cover[y0:y1, x0:x1] += 1
    return bool((cover == 1).all())


@pytest.mark.parametrize("size,world,cols", [((3840, 2160), 8, 0), ((3840, 2160), 4, 0), ((1920, 1080), 8, 0), ((3840, 2160), 2, 0), ((1280, 720), 6, 0), ((3840, 2160), 8, 2)])
def test_a_rebalanced_grid_still_covers_the_frame_exactly_once(size, world, cols):
    w, h = size
    g = dist_grid(w, h, world, cols)
    assert g.tiles() == [dist_partition(w, h, world, r, cols) for r in range(world)], "the equal grid is st_dist_partition's split"
    assert _covers_exactly_once(g.tiles(), w, h)
    rng = np.random.default_rng(world * 7 + cols)
    for it in range(6):
        cost = rng.uniform(0.5, 2.0, world)
        g2 = dist_grid_rebalance(w, h, g, cost, max_step=0 if it % 2 else 16)
        assert _covers_exactly_once(g2.tiles(), w, h), g2.describe()
        if it % 2 == 0:   # a limited step: no edge moved further than 16 pixels
            a, b = g.describe(), g2.describe()
            assert max(abs(x - y) for x, y in zip(a["row_edges"], b["row_edges"])) <= 16
            assert max(abs(x - y) for ra, rb in zip(a["col_edges"], b["col_edges"]) for x, y in zip(ra, rb)) <= 16
        g = g2


def test_rebalancing_moves_work_away_from_the_expensive_tiles():
    """A synthetic cost field (cost per pixel 3 in the upper-left quadrant, 1 elsewhere), tiles costed by integrating it: after a few free
    rebalances max / mean falls from 1.6 to within 5 % of 1; equal costs leave the equal grid where it is."""
    w, h, world = 3840, 2160, 8
    yy, xx = np.mgrid[0:h, 0:w]
    density = np.where((xx < w // 2) & (yy < h // 2), 3.0, 1.0)

    def costs(g):
        return [float(density[y0:y1, x0:x1].sum()) for x0, y0, x1, y1 in g.tiles()]

    g = dist_grid(w, h, world)
    c0 = costs(g)
    assert max(c0) / np.mean(c0) > 1.5
    for _ in range(4):
        g = dist_grid_rebalance(w, h, g, costs(g))
    c1 = costs(g)
    assert max(c1) / np.mean(c1) < 1.05, (g.describe(), c1)
    assert _covers_exactly_once(g.tiles(), w, h)
    same = dist_grid_rebalance(w, h, dist_grid(w, h, world), [1.0] * world)
    assert same.describe() == dist_grid(w, h, world).describe()
    with pytest.raises(StrolleError, match="positive"):
        dist_grid_rebalance(w, h, g, [1.0] * 7 + [0.0])


def test_rebalancing_a_frame_whose_edge_is_off_the_grid_stays_on_the_grid():
    """a frame 1000 pixels wide (not a multiple of 16) or 1084 rows high (not a multiple of 8) whose cost sits at the far edge pushes the
    rebalanced edges against the minimum tile size; the back-to-front clamp used to land off the 16 x 8 grid (extent - min_size) and the result was refused.
    Every edge stays on the grid, tiles keep their minimum size, limited steps stay limited."""
    for (w, h, world, cols) in ((1000, 540, 8, 4), (1920, 1084, 8, 1), (1000, 1084, 6, 3)):
        g = dist_grid(w, h, world, cols)
        rows = world // (cols or 1)
        for it in range(8):
            cost = [1.0] * world
            cost[-1] = 60.0                                     # nearly all work in the last tile: edges crowd towards the far corner
            for r in range(rows):
                cost[r * g.cols + g.cols - 1] = max(cost[r * g.cols + g.cols - 1], 30.0)
            before = g.describe()
            g = dist_grid_rebalance(w, h, g, cost, max_step=0 if it >= 4 else 16)
            d = g.describe()
            assert _covers_exactly_once(g.tiles(), w, h), d
            assert all(e % 8 == 0 for e in d["row_edges"][:-1]) and all(e % 16 == 0 for row in d["col_edges"] for e in row[:-1]), d
            assert all(y1 - y0 >= 32 and x1 - x0 >= 64 for x0, y0, x1, y1 in g.tiles()), d
            if it < 4:
                assert max(abs(x - y) for x, y in zip(before["row_edges"], d["row_edges"])) <= 16
                assert max(abs(x - y) for ra, rb in zip(before["col_edges"], d["col_edges"]) for x, y in zip(ra, rb)) <= 16


def test_a_weighted_grid_gathers_like_the_equal_split():
    """st_dist_set_grid through the in-process transport: 4 ranks, a lopsided grid; rank 0's frame is tile r from rank r, bit for bit. A grid
    with the wrong tile count or edges off the pixel grid is refused."""
    w, h, world = 352, 200, 4
    ranks = _engines(world, (w, h), group=4242)
    g = dist_grid_rebalance(w, h, dist_grid(w, h, world), [4.0, 1.0, 1.0, 2.0])
    assert g.describe() != dist_grid(w, h, world).describe()
    yy, xx = np.mgrid[0:h, 0:w]
    frames = [np.ascontiguousarray(np.stack([(r * 37 + xx * 3 + yy * 5 + c).astype(np.float32) for c in range(4)], -1)) for r in range(world)]
    full = np.zeros((h, w, 4), np.float32); expect = np.zeros_like(full)
    for r, (e, cam) in enumerate(ranks):
        owned, window = e.dist_set_grid(cam, g, apron=16)
        assert owned == dist_grid_tile(g, r) and window == dist_window(w, h, owned, 16)
        x0, y0, x1, y1 = owned
        expect[y0:y1, x0:x1] = frames[r][y0:y1, x0:x1]
    for r in range(world - 1, -1, -1):
        e, cam = ranks[r]
        e.dist_gather(cam, frames[r].ctypes.data, full.ctypes.data if r == 0 else 0)
    assert np.array_equal(full, expect)
    e0, c0 = ranks[0]
    with pytest.raises(StrolleError, match="one tile per rank"):
        e0.dist_set_grid(c0, dist_grid(w, h, 2))
    bad = dist_grid(w, h, world); bad.row_edge[1] = 100
    with pytest.raises(StrolleError, match="multiples of 8"):
        e0.dist_set_grid(c0, bad)
    for e, _ in ranks:
        e.dist_shutdown(); e.close()
