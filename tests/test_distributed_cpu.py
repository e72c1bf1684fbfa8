"""world_size-2 gloo tests of the multi-GPU path's host logic (band partition + per-frame all-gather)."""
import os
import sys

import numpy as np
import pytest

from strolle_amd.distributed import assemble_bands_numpy, band_for_rank, render_window, weak_scaling_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("height,world", [(1080, 1), (1080, 2), (2160, 4), (4320, 8), (1083, 4), (77, 3)])
def test_bands_partition_the_frame(height, world):
    rows = []
    for r in range(world):
        y0, y1 = band_for_rank(height, world, r)
        assert 0 <= y0 <= y1 <= height
        if height % world:  # ragged partitions sit on 8-row tile boundaries; even ones are exactly equal
            assert y0 % 8 == 0 and (y1 % 8 == 0 or y1 == height)
        else:
            assert y1 - y0 == height // world
        rows += list(range(y0, y1))
    assert rows == list(range(height))


def test_weak_scaling_frames():
    assert weak_scaling_frame((1920, 1080), 1) == (1920, 1080)
    assert weak_scaling_frame((1920, 1080), 2) == (1920, 2160)
    assert weak_scaling_frame((1920, 1080), 4) == (3840, 2160)   # BASELINE.json config 4
    assert weak_scaling_frame((1920, 1080), 8) == (3840, 4320)
    for n in (1, 2, 4, 8):
        w, h = weak_scaling_frame((1920, 1080), n)
        assert w * h == n * 1920 * 1080
        for r in range(n):
            y0, y1 = band_for_rank(h, n, r)
            assert (y1 - y0) * w == 1920 * 1080


def test_render_window_has_apron_and_stays_on_tiles():
    assert render_window(2160, (540, 1080), 128) == (408, 1208)
    assert render_window(2160, (0, 540), 128) == (0, 672)
    assert render_window(2160, (1620, 2160), 128) == (1488, 2160)


WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from strolle_amd.distributed import band_for_rank, gather_frame, assemble_bands_numpy
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
H, W = 77, 24   # odd height: uneven bands
def frame_of(r):
    rng = np.random.default_rng(100 + r)
    return rng.standard_normal((H, W, 4)).astype(np.float32)
local = torch.from_numpy(frame_of(rank))
full = gather_frame(local, H, W, world, rank).numpy()
want = assemble_bands_numpy([frame_of(r) for r in range(world)], H, W)
assert np.array_equal(full, want), "gathered frame differs"
y0, y1 = band_for_rank(H, world, rank)
assert np.array_equal(full[y0:y1], frame_of(rank)[y0:y1])
# the bench's collective: equal bands gathered straight into rank 0's frame
from strolle_amd.distributed import gather_bands_to_root
H2 = 64
def frame2(r):
    return np.random.default_rng(200 + r).standard_normal((H2, W, 4)).astype(np.float32)
local2 = torch.from_numpy(frame2(rank))
full2 = torch.zeros((H2, W, 4)) if rank == 0 else None
got = gather_bands_to_root(local2, full2, H2, world, rank)
if rank == 0:
    assert np.array_equal(got.numpy(), assemble_bands_numpy([frame2(r) for r in range(world)], H2, W))
    # in-place variant: the root's render target doubles as the gathered frame
    inplace = torch.from_numpy(frame2(0))
    gather_bands_to_root(inplace, inplace, H2, world, rank)
    assert np.array_equal(inplace.numpy(), assemble_bands_numpy([frame2(r) for r in range(world)], H2, W))
else:
    assert got is None
    gather_bands_to_root(local2, None, H2, world, rank)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_gather_frame_gloo_world_size_2(tmp_path):
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), ROOT]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("ok") == 2
