"""Tile-parallel rendering across the GPUs of one node: one process per GPU, row-band partition of the
frame, one collective per frame (all-gather of the final HDR buffer over RCCL/xGMI).

The renderer's per-pixel passes are independent across ranks; the only exchange step is the gather of the
composed RGBA32F bands. The scene is replicated (every rank builds the same engine state deterministically).

Reference / BvhHeatmap modes read nothing outside their own pixel, so a band is bit-identical to the same rows
of a single-GPU frame (RNG is keyed on absolute pixel coordinates, strolle-gpu/src/noise/white.rs:15-19).
Image mode has neighbour taps (spatial resampling radius 128 px, wavelets, reprojection); each rank renders
its band plus `apron` extra rows on both sides so those taps find valid data, and only the band is gathered.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def weak_scaling_frame(base_size: Tuple[int, int], world_size: int) -> Tuple[int, int]:
    """Frame size whose pixel count is world_size x the single-GPU workload: width doubles from 4 ranks on,
    height takes the rest (1: 1920x1080, 2: 1920x2160, 4: 3840x2160 — BASELINE.json config 4 —, 8: 3840x4320)."""
    w, h = base_size
    wx = 2 if world_size >= 4 and world_size % 2 == 0 else 1
    return w * wx, h * (world_size // wx)


def band_for_rank(height: int, world_size: int, rank: int, align: int = 8) -> Tuple[int, int]:
    """Rows [y0, y1) owned by `rank`. Equal bands when the height divides evenly (kernels mask rows outside the window,
    so bands need not sit on 8-row tile boundaries); otherwise tile-aligned bands that differ by at most one tile row."""
    if height % world_size == 0:
        rows = height // world_size
        return rank * rows, (rank + 1) * rows
    tiles = (height + align - 1) // align
    per = tiles // world_size
    extra = tiles % world_size
    t0 = rank * per + min(rank, extra)
    t1 = t0 + per + (1 if rank < extra else 0)
    return min(t0 * align, height), min(t1 * align, height)


def render_window(height: int, band: Tuple[int, int], apron: int) -> Tuple[int, int]:
    """Rows a rank actually renders: its band widened by `apron` rows (clamped, kept on 8-row tile boundaries)."""
    y0 = max(0, (band[0] - apron) // 8 * 8)
    y1 = min(height, (band[1] + apron + 7) // 8 * 8)
    return y0, y1


def tile_for_rank(width: int, height: int, world_size: int, rank: int, cols: int = 0) -> Tuple[int, int, int, int]:
    """(x0, y0, x1, y1) rank `rank` owns under the library's partition rule (st_dist_partition; one definition, in C): the default
    grid is 1x1, two row bands, 2x2, 3x2, 4x2 ... (columns >= rows), tile edges on multiples of 16 pixels in x and 8 in y."""
    from .api import dist_partition
    return dist_partition(width, height, world_size, rank, cols)


def tile_window(width: int, height: int, tile: Tuple[int, int, int, int], apron: int) -> Tuple[int, int, int, int]:
    """What a rank renders: its tile widened by `apron` pixels towards its neighbours (st_dist_window)."""
    from .api import dist_window
    return dist_window(width, height, tile, apron)


def tile_overhead(width: int, height: int, world_size: int, apron: int, cols: int = 0):
    """Redundant pixels a rank renders around its tile, as a fraction of the tile: (max over ranks, mean)."""
    fr = []
    for r in range(world_size):
        t = tile_for_rank(width, height, world_size, r, cols); w = tile_window(width, height, t, apron)
        fr.append(((w[2] - w[0]) * (w[3] - w[1])) / ((t[2] - t[0]) * (t[3] - t[1])) - 1.0)
    return max(fr), sum(fr) / len(fr)


def gather_tiles_to_root(local_frame, full_frame, world_size: int, rank: int, cols: int = 0, dst: int = 0, group=None, tiles=None):
    """torch.distributed FALLBACK of st_dist_gather (the C ABI's RCCL gather is the product path; this one serves boxes without
    RCCL and the gloo tests): every rank sends its tile — packed, a tile is a strided view of the frame —, rank `dst` unpacks
    each into its place in `full_frame`. One gather, the only collective."""
    import torch
    import torch.distributed as dist

    height, width = local_frame.shape[0], local_frame.shape[1]
    if tiles is None:   # (`tiles`: every rank's tile of a cost-weighted grid, api.StDistGrid.tiles())
        tiles = [tile_for_rank(width, height, world_size, r, cols) for r in range(world_size)]
    x0, y0, x1, y1 = tiles[rank]
    send = local_frame[y0:y1, x0:x1].contiguous()
    if rank == dst:
        recv = [torch.empty((t[3] - t[1], t[2] - t[0]) + tuple(local_frame.shape[2:]), dtype=local_frame.dtype, device=local_frame.device) for t in tiles]
        same = all(r.shape == recv[0].shape for r in recv)
        if same:
            dist.gather(send, recv, dst=dst, group=group)
        else:   # ragged tiles: point-to-point
            ops = [dist.P2POp(dist.irecv, recv[r], r, group) for r in range(world_size) if r != dst]
            for w in dist.batch_isend_irecv(ops): w.wait()
            recv[dst] = send
        for r, t in enumerate(tiles):
            full_frame[t[1]:t[3], t[0]:t[2]] = recv[r]
        return full_frame
    same = all((t[3] - t[1], t[2] - t[0]) == (tiles[0][3] - tiles[0][1], tiles[0][2] - tiles[0][0]) for t in tiles)
    if same:
        dist.gather(send, None, dst=dst, group=group)
    else:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, dst, group)]): w.wait()
    return None


def gather_frame(local_frame, height: int, width: int, world_size: int, rank: int, group=None):
    """All-gather the owned band of every rank into a full frame (torch tensors, any backend).

    `local_frame` is the rank's full-size [H, W, 4] float32 tensor of which only its band is meaningful.
    Bands may differ by one tile row, so they are padded to the largest band for the collective."""
    import torch
    import torch.distributed as dist

    bands = [band_for_rank(height, world_size, r) for r in range(world_size)]
    max_rows = max(b[1] - b[0] for b in bands)
    y0, y1 = bands[rank]
    send = torch.zeros((max_rows, width, 4), dtype=local_frame.dtype, device=local_frame.device)
    send[: y1 - y0] = local_frame[y0:y1]
    recv = torch.empty((world_size, max_rows, width, 4), dtype=local_frame.dtype, device=local_frame.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group) if hasattr(dist, "all_gather_into_tensor") and local_frame.is_cuda \
        else dist.all_gather(list(recv.unbind(0)), send, group=group)
    out = torch.empty((height, width, 4), dtype=local_frame.dtype, device=local_frame.device)
    for r, (b0, b1) in enumerate(bands):
        out[b0:b1] = recv[r, : b1 - b0]
    return out


def gather_bands_to_root(local_frame, full_frame, height: int, world_size: int, rank: int, dst: int = 0, group=None):
    """The per-frame collective of the tile-parallel path: every rank sends its band, rank `dst` receives each band
    straight into its place in `full_frame` (no staging copy). With equal bands this is one point-to-point message
    per xGMI link into the root (RCCL implements gather as grouped send/recv), not a ring.
    `full_frame` is only used on `dst` and may alias `local_frame` there."""
    import torch.distributed as dist

    assert height % world_size == 0, "equal bands required; use gather_frame() for ragged partitions"
    y0, y1 = band_for_rank(height, world_size, rank)
    send = local_frame[y0:y1]
    if rank == dst:
        gather_list = []
        for r in range(world_size):
            b0, b1 = band_for_rank(height, world_size, r)
            gather_list.append(full_frame[b0:b1])
        if full_frame.data_ptr() == local_frame.data_ptr():
            send = send.clone()  # in-place: the root's own band must not alias its receive slot
        dist.gather(send, gather_list, dst=dst, group=group)
    else:
        dist.gather(send, None, dst=dst, group=group)
    return full_frame if rank == dst else None


def assemble_bands_numpy(bands_data, height: int, width: int) -> np.ndarray:
    """Host-side reference of gather_frame for tests: bands_data[r] is rank r's full-size frame."""
    world_size = len(bands_data)
    out = np.zeros((height, width, 4), np.float32)
    for r in range(world_size):
        y0, y1 = band_for_rank(height, world_size, r)
        out[y0:y1] = bands_data[r][y0:y1]
    return out
