#!/usr/bin/env python3
"""SHA-256 of what the fast build renders over DEVICE-BUILT trees (refresh mode 3): the dungeon at 208 k triangles and at 13 k, a spawn and a despawn
between the frames. Two libraries that print the same lines build the same trees (the builder's sort is stable either way; STROLLE_HIP_LIB picks the library)."""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from strolle_amd import CameraMode, Engine, Instance, Mesh, scenes

rng = np.random.default_rng(2)
pos = (rng.uniform(-0.3, 0.3, (200, 1, 3)) + rng.uniform(-0.05, 0.05, (200, 3, 3))).astype(np.float32)
nrm = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
blob = Mesh(pos, np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32))
W, H = 640, 360
for subdivide in (2, 0):
    e = Engine(device=0); e.set_bvh_refresh(3)
    scenes.build_dungeon(e, subdivide=subdivide); e.set_seed(3); e.insert_mesh(7777, blob)
    desc = scenes.dungeon_camera((W, H), CameraMode.IMAGE)
    cam = e.create_camera(desc)
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    h = hashlib.sha256()
    for frame in range(8):
        if frame == 3:
            place = np.eye(4, dtype=np.float32)[:3].copy(); place[:, 3] = (-5.75, 0.6, -18.2)
            e.insert_instance(7000, Instance(7777, 2, place))
        if frame == 6: e.remove_instance(7000)
        e.update_camera(cam, desc); e.tick(stream); e.render_camera(cam, out.data_ptr(), stream); torch.cuda.synchronize()
        h.update(out.cpu().numpy().tobytes())
    print(f"dungeon subdivide {subdivide} device-built: {e.device_builds()} builds, frames {h.hexdigest()}")
    e.close()
