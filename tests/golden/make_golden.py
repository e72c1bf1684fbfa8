#!/usr/bin/env python3
"""Regenerates the golden fixtures from the CPU oracle (run from the repo root).

The reference (Rust + rust-gpu + wgpu) cannot be built or run in this environment, so these vectors are
outputs of OUR oracle: they pin it against regressions and give the GPU path a committed target, but they are
not reference outputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_binding import OracleEngine  # noqa: E402
from strolle_amd import Buffer, CameraMode, scenes  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(mode, frames, depth=0):
    e = OracleEngine(); scenes.build_cornell(e); e.set_seed(1)
    desc = scenes.cornell_camera((64, 48), mode, depth=depth); cam = e.create_camera(desc)
    for _ in range(frames):
        e.update_camera(cam, desc); e.tick(); img = e.render_camera(cam)
    return e, cam, img


e, cam, img = run(CameraMode.BVH_HEATMAP, 1)
np.savez_compressed(os.path.join(OUT, "cornell_heatmap_64x48.npz"), image=img, used_memory=e.read_buffer(cam, Buffer.DBG_USED_MEMORY))
e, cam, img = run(CameraMode.REFERENCE, 3, depth=1)
np.savez_compressed(os.path.join(OUT, "cornell_reference_64x48.npz"), image=img)
e, cam, img = run(CameraMode.IMAGE, 7)
np.savez_compressed(os.path.join(OUT, "cornell_image_64x48.npz"), image=img)
print("golden fixtures written")
