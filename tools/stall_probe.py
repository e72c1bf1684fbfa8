#!/usr/bin/env python3
"""Where do the first ticks of bench.py's moving-geometry region spend their time?  (the driver's
`--steps 20 --warmup 5` run reported ms_per_step_geometry_moving = 2.548 ms where 60/12 gives 0.81.)

Replays what bench.py does before that region (static headline, then the moving-light region), then the geometry region itself with
a stream join after EVERY step, and prints per step: host time inside st_tick, host time inside st_render_camera, wall time until
the device was idle again. ST_TICK_TIMING=1 adds the tick's own breakdown on stderr.

    python tools/stall_probe.py [--warmup 5] [--steps 20] [--preroll 96]
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=96)
    ap.add_argument("--free-running", action="store_true", help="no per-step join: time the region the way bench.py does, after --warmup steps")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from strolle_amd import Instance, Light

    class Args:
        exact = False; seed = 0; cols = 0; apron = 16; py_gather = False
    job = bench.Job(torch, None, Args, "cornell", "image", (1920, 1080), 1, 0, 0, False)
    engine, cam = job.engine, job.cam
    job.run(a.preroll + a.warmup); torch.cuda.synchronize()
    el, _ = job.timed_region(a.steps)
    print(f"static: {el / a.steps * 1e3:.4f} ms/step")
    job.moving = True
    job.run(a.warmup); el, _ = job.timed_region(a.steps)
    job.moving = False
    print(f"moving light+camera: {el / a.steps * 1e3:.4f} ms/step")
    job.desc = job.scenes.cornell_camera((1920, 1080), job.mode, depth=1)
    engine.insert_light(1, Light.point((0.0, 1.5, 0.5), 0.15, (50.0 / (4.0 * math.pi),) * 3, 20.0))
    npz = np.load(os.path.join(job.scenes.ASSETS, "cornell.npz"))
    mesh = int(npz["n_meshes"]) - 1
    rest = np.ascontiguousarray(npz[f"xform_{mesh}"].reshape(4, 3).T, np.float32)

    def placed(i):
        x = rest.copy(); x[0, 3] += np.float32(0.15 * math.sin(i / 20.0))
        return Instance(1 + mesh, 1 + int(npz[f"material_{mesh}"]), x)
    engine.set_bvh_refresh(2)
    out = job.outs[0]
    if a.free_running:
        job.geometry = (1 + mesh, placed)
        job.run(a.warmup)
        el, _ = job.timed_region(a.steps)
        print(f"geometry moving, free running, warmup {a.warmup}: {el / a.steps * 1e3:.4f} ms/step")
        el, _ = job.timed_region(a.steps)
        print(f"geometry moving, free running, next {a.steps}: {el / a.steps * 1e3:.4f} ms/step")
        return
    print("step   insert   tick    render   until idle   (ms)")
    for i in range(a.warmup + a.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.insert_instance(1 + mesh, placed(job.frame_no)); job.frame_no += 1
        engine.update_camera(cam, job.desc)
        t1 = time.perf_counter()
        engine.tick(job.stream)
        t2 = time.perf_counter()
        engine.render_camera(cam, out.data_ptr(), job.stream)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f"{i:4d} {(t1 - t0) * 1e3:8.3f} {(t2 - t1) * 1e3:8.3f} {(t3 - t2) * 1e3:8.3f} {(t4 - t0) * 1e3:10.3f}" + ("   <- timed region starts" if i == a.warmup else ""))
    rebuilds, refits = engine.bvh_refits()
    print(f"rebuilds {rebuilds}, refits {refits}, device refits {engine.bvh_device_refits()}")


if __name__ == "__main__":
    main()
