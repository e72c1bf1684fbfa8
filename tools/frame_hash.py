#!/usr/bin/env python3
"""SHA-256 of what a build renders: Cornell `Image`, dungeon `Image`, dungeon `GiDiffuse` and `Reference`, a few frames each, the composed frame and the
temporal planes. Two builds of the library that print the same lines compute the same bits (tools/gpu_vop3_ab.sh runs it with STROLLE_HIP_LIB set to
either)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from strolle_amd import Buffer, CameraMode, Engine, scenes

size = (640, 360)
for name, build, camera, mode in (("cornell image", scenes.build_cornell, scenes.cornell_camera, CameraMode.IMAGE),
                                  ("dungeon image", scenes.build_dungeon, scenes.dungeon_camera, CameraMode.IMAGE),
                                  ("dungeon gi_diffuse", scenes.build_dungeon, scenes.dungeon_camera, CameraMode.GI_DIFFUSE),
                                  ("dungeon reference", scenes.build_dungeon, scenes.dungeon_camera, CameraMode.REFERENCE)):
    e = Engine(device=0, exact=False)
    e.keep_all_planes(True)
    build(e); e.set_seed(3)
    desc = camera(size, mode, depth=1) if mode == CameraMode.REFERENCE else camera(size, mode)
    cam = e.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    h = hashlib.sha256()
    for _ in range(7):
        e.update_camera(cam, desc); e.tick(); e.render_camera(cam, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        h.update(out.cpu().numpy().tobytes())
    for b in (Buffer.PRIM_GBUFFER_D0_A, Buffer.PRIM_GBUFFER_D1_A, Buffer.DI_RESERVOIRS_0, Buffer.GI_RESERVOIRS_0, Buffer.GI_RESERVOIRS_1, Buffer.DI_DIFF_PREV_COLORS, Buffer.GI_DIFF_PREV_COLORS, Buffer.REF_COLORS):
        try:
            h.update(np.ascontiguousarray(e.read_buffer(cam, b)).tobytes())
        except Exception as exc:   # a plane the mode does not keep
            h.update(str(type(exc)).encode())
    print(name, h.hexdigest(), flush=True)
    e.close()
