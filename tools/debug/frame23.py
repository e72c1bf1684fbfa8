"""Debug: which switch makes GI_RESERVOIRS_2 of a validation frame differ from the oracle after one whole fast frame."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle_binding import OracleEngine
from strolle_amd import Buffer, CameraMode, Engine, scenes
from test_gpu_fast_tolerance import lanes_outside_tolerance

size = (480, 270); F = int(os.environ.get("F", "23"))
FLOAT_BUFFERS = [b for b in Buffer if b != Buffer.DBG_USED_MEMORY]
orac = OracleEngine(); scenes.build_cornell(orac); orac.set_seed(0)
desc = scenes.cornell_camera(size, CameraMode.IMAGE)
co = orac.create_camera(desc)
for f in range(F):
    orac.update_camera(co, desc); orac.tick(); orac.render_camera(co, compose=False)
orac.update_camera(co, desc); orac.tick()
before = {b: orac.read_buffer(co, b) for b in FLOAT_BUFFERS}
orac.render_camera(co, compose=True)
want = {b: orac.read_buffer(co, b) for b in FLOAT_BUFFERS}
depth = want[Buffer.PRIM_GBUFFER_D0_B if F % 2 else Buffer.PRIM_GBUFFER_D0_A].reshape(-1, 4)[:, 0]
print("sky fraction", float((depth == 0).mean()))
for name, env, keep in [("lean", {}, False), ("keep", {}, True), ("lean_nofusecompose", {"ST_NO_FUSE_COMPOSE": "1"}, False), ("lean_nooverlap", {"ST_NO_OVERLAP": "1"}, False),
                        ("lean_noalias", {"ST_NO_GI_ALIAS": "1"}, False)]:
    os.environ.update(env)
    prod = Engine(device=0, exact=False)
    for k in env: os.environ.pop(k)
    scenes.build_cornell(prod); prod.set_seed(0)
    cp = prod.create_camera(desc)
    out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device="cuda:0")
    for f in range(F + 1):
        prod.update_camera(cp, desc); prod.tick()
    for b, d in before.items(): prod.write_buffer(cp, b, d)
    prod.keep_all_planes(keep)
    prod.render_camera(cp, out.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    for b in (Buffer.GI_RESERVOIRS_0, Buffer.GI_RESERVOIRS_1, Buffer.GI_RESERVOIRS_2, Buffer.GI_RESERVOIRS_3):
        got = prod.read_buffer(cp, b)
        bad = lanes_outside_tolerance(got, want[b]).reshape(-1, 16)
        px = bad.any(axis=1)
        stale = (got.reshape(-1, 16)[px].view(np.uint32) == before[b].reshape(-1, 16)[px].view(np.uint32)).all(axis=1)
        print(f"{name:20s} {b.name}: bad lanes {bad.mean():.2e}, bad px {px.sum()}, of which sky {int((depth[px] == 0).sum())}, of which got == state before the frame {int(stale.sum())}, per lane {bad.sum(axis=0).tolist()}")
    prod.close()
