#!/usr/bin/env python3
"""bench.py — headline benchmark: Mray/s + ms/frame, Cornell box 1920x1080, Image{denoise:true}
(1 spp ReSTIR DI + GI + SVGF), on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one frame: Engine::tick + CameraController::render (every pass of the pipeline) + composition.
Scene, buffers and temporal state are resident in HBM before the timed region; the output stays on the device.
The kernels run in the library's default (fast-arithmetic) build; --exact selects the bit-exact build.

N > 1, --scaling weak (default): the frame grows to N x (width x height) pixels (N=4 is BASELINE.json's 3840x2160 4-tile
config), each rank renders one tile of st_dist_partition's grid (2 ranks: row bands, 4: 2 x 2, 8: 4 x 2) + an apron, and the tiles
are gathered to rank 0 every frame — the only collective, st_dist_gather: RCCL through the C ABI (--py-gather: torch.distributed).
--scaling strong: the frame stays width x height and is split into N tiles — BASELINE.json configs 4 and 5 as written:
  ... bench.py --gpus 4 --scaling strong --width 3840 --height 2160 --scene cornell --mode reference     (config 4)
  ... bench.py --gpus 8 --scaling strong --width 3840 --height 2160 --scene dungeon --mode image         (config 5; N = 1, 2, 4, 8)
Extra regions, all in the same ONE JSON line (none of them changes `value` / `ms_per_step`, which stay the static headline):
  N = 1, headline workload:  `moving`  — the same K steps with the light orbiting as bevy-strolle/examples/cornell.rs:82-93 animates
                             it and the camera on a slow orbit (`ms_per_step_moving`); with one object moving every frame and the BVH refitted
                             on the device (`ms_per_step_geometry_moving`);
                             `present` — the same K steps through the facade's present path (RGBA8 target, two buffers,
                             st_camera_present_copy to page-locked host memory, previous frame polled: `ms_per_step_with_present`).
  N > 1:                     `multi_gpu.strong_config5` — BASELINE.json config 5 as written (dungeon 3840x2160 Image, ONE frame split
                             into N tiles + apron, gathered to rank 0), whatever --scaling the main region used; N = 4 also runs
                             config 4 (Cornell 3840x2160 Reference, 2 x 2 tiles) as `multi_gpu.strong_config4`.
  --no-extras skips them.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4-copy ceiling)
# profiler slots (st_kernels.h kernel_info names) that implement the reference's five `frame_denoising::wavelet` passes:
# name -> reference passes executed per launch
WAVELET_SLOTS = {"denoise_wavelet": 1, "denoise_wavelet x2 (strides 1+2)": 2, "denoise_wavelet+composition": 1}
WAVELET_SYMBOLS = ["denoise_wavelet_12", "denoise_wavelet_lds<1>", "denoise_wavelet_lds<2>", "denoise_wavelet_lds<4>", "denoise_wavelet_far<false>", "denoise_wavelet_far<true>"]
# In the lean frame the last a-trous pass also runs frame composition for its pixel: that launch executes two reference passes
# and is credited both (84 + 112 B per pixel, st_kernels.h) like every fused launch ("report the unfused figure as
# the algorithmic reference so fusion shows up as a gain"). `roofline.wavelet_only` restates the family without that launch.


def measure_copy_ceiling(torch, dev, engine=None):
    """This box's streaming ceiling, (bytes read + bytes written) / time of a 1 GiB device-to-device copy, two ways (SURVEY.md 8d asks
    for the measured ceiling beside the 8 TB/s spec figure): the library's own grid-stride float4 copy kernel (k_util.hip — the kind of
    kernel /opt/skills/guides/MI355X_MICROARCH.md measures 6.29 TB/s with) and torch's copy_ (what rounds 1-3 reported).
    Returns (own kernel GB/s or None, torch GB/s). `frac` figures against a measured ceiling use the LARGER of the two."""
    own = None
    if engine is not None:
        try:
            own = engine.copy_bandwidth(1 << 30, 6)
        except Exception:
            own = None
    n = (1 << 30) // 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    best = None
    for _ in range(6):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); dst.copy_(src); t1.record(); t1.synchronize()
        ms = t0.elapsed_time(t1)
        best = ms if best is None else min(best, ms)
    del src, dst
    torch.cuda.empty_cache()
    return own, 2.0 * (1 << 30) / (best * 1e-3) / 1e9


def cpu_baseline(args, size):
    """The CPU oracle ("port": our C++ restatement of the reference; the reference's lavapipe path cannot run here) timed on this
    host's cores on a BOUNDED sample of the same workload — same scene, mode and size as the GPU line: one warm-up frame, which
    also prices a frame, then as many timed frames as fit ~20 s (1 to 5); best frame reported."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleEngine, oracle_lib
    from strolle_amd import CameraMode, scenes
    lib = oracle_lib()
    cores = int(lib.or_num_threads())
    e = OracleEngine()
    mode = {"image": CameraMode.IMAGE, "gi_diffuse": CameraMode.GI_DIFFUSE, "reference": CameraMode.REFERENCE, "heatmap": CameraMode.BVH_HEATMAP}[args.mode]
    if args.scene == "cornell":
        scenes.build_cornell(e); desc = scenes.cornell_camera(size, mode, depth=1)
    else:
        scenes.build_dungeon(e, subdivide=2 if args.scene == "dungeon134k" else 0); desc = scenes.dungeon_camera(size, mode, depth=1)
    e.set_seed(args.seed)
    cam = e.create_camera(desc)
    t0 = time.perf_counter()
    e.update_camera(cam, desc); e.tick(); e.render_camera(cam)
    first = time.perf_counter() - t0           # includes the scene build on the first tick: an over-estimate, i.e. a cautious frame count
    timed = max(1, min(5, int(20.0 / max(first, 1e-3))))
    per_frame = []
    for _ in range(timed):
        e.ray_count(cam, reset=True)
        t0 = time.perf_counter()
        e.update_camera(cam, desc); e.tick(); e.render_camera(cam)
        dt = time.perf_counter() - t0
        per_frame.append((e.ray_count(cam) / dt, dt))
    best_rate, best_dt = max(per_frame)
    mean_dt = sum(d for _, d in per_frame) / timed
    return {"value": round(best_rate / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "ms_per_frame": round(best_dt * 1e3, 1), "mean_ms_per_frame": round(mean_dt * 1e3, 1),
            "sample": f"best of {timed} frame(s) (after 1 warm-up frame) of the same workload — {args.scene} {size[0]}x{size[1]} mode {args.mode} —, OpenMP over {cores} threads"}


def workload_key(scene, width, height, mode):
    return f"{scene}_{width}x{height}_{mode}"


_PMC_CACHE = {}


def load_pmc(key):
    """The committed rocprofv3 counter summary of THIS workload (profiles/pmc/<scene>_<W>x<H>_<mode>.json; separate `--pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` passes of this bench command, tools/gpu_profile_workload.sh + tools/summarize_profiles.py), or None. Another
    workload's bytes are never substituted."""
    if key not in _PMC_CACHE:
        path = os.path.join(ROOT, "profiles", "pmc", key + ".json")
        try:
            _PMC_CACHE[key] = json.load(open(path))
        except Exception:
            _PMC_CACHE[key] = None
    return _PMC_CACHE[key]


def static_traffic(symbols, key, run_slots=None):
    """HBM bytes per launch of the given profiler slots / kernel symbols from the workload's committed counter summary. Not measured
    by this process: hardware counters need the profiler. Returns (bytes or None, source note). The summary carries a stamp — the
    commit it was taken at and the launch structure (the profiler slots of one frame); when `run_slots` (this run's slots) differs
    from the stamped list the summary describes other launches and is refused."""
    pmc = load_pmc(key)
    if pmc is None:
        return None, f"no counter summary for this workload (profiles/pmc/{key}.json absent): null rather than another workload's bytes"
    stamp = pmc.get("_stamp", {})
    if run_slots is not None and stamp.get("launch_slots") is not None and sorted(stamp["launch_slots"]) != sorted(run_slots):
        return None, f"profiles/pmc/{key}.json was taken with another launch structure (commit {stamp.get('commit')}): refused"
    total, n = 0.0, 0
    for s in symbols:
        e = pmc.get(s)
        if e and e.get("hbm_bytes_per_launch") is not None:
            total += e["hbm_bytes_per_launch"] * e["launches_sampled"]; n += e["launches_sampled"]
    return (round(total / n) if n else None), f"profiles/pmc/{key}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at commit {stamp.get('commit')}; (2 x FETCH_SIZE + WRITE_SIZE) x 1024, calibration in profiles/README.md)"


def build_stamp():
    """The commit compiled INTO the library that ran (st_build_commit(), csrc/Makefile) — a property of the binary. Round 5 printed a side file
    the builder's own gpurun wrapper wrote, which said 9c646a9 in a line the driver measured on 9e6c1c1."""
    try:
        from strolle_amd.api import library_build_commit
        return library_build_commit()
    except Exception:
        return None


def n1_reference(key):
    """Single-GPU frame time of a BASELINE.json config measured by the builder on one MI355X (profiles/n1_reference.json) —
    the N = 1 point a strong-scaling figure of this run can be held against. Static, labelled as such."""
    path = os.path.join(ROOT, "profiles", "n1_reference.json")
    try:
        return json.load(open(path)).get(key)
    except Exception:
        return None


def apron_overhead(width, height, world, apron, cols=0):
    """Redundant pixels a rank renders around its tile (windows sit on the partition's 16 x 8 pixel grid), as a fraction of the tile:
    the largest over the ranks and the mean."""
    from strolle_amd.distributed import tile_overhead
    mx, mean = tile_overhead(width, height, world, apron, cols)
    return {"max": round(mx, 4), "mean": round(mean, 4)}


class Job:
    """One camera of one scene on this rank: engine, tile (+ apron), double-buffered render targets and the per-frame
    gather of the tiles to rank 0 on a communication stream."""

    def __init__(self, torch, dist, args, scene, mode_name, size, world, rank, local_rank, debug_shared):
        from strolle_amd import CameraMode, Engine, scenes
        from strolle_amd.distributed import tile_for_rank, tile_window
        self.torch, self.dist, self.world, self.rank, self.debug_shared = torch, dist, world, rank, debug_shared
        self.scenes = scenes
        self.scene, self.mode_name = scene, mode_name
        self.width, self.height = size
        self.engine = Engine(device=local_rank, exact=True if args.exact else None)
        mode = {"image": CameraMode.IMAGE, "gi_diffuse": CameraMode.GI_DIFFUSE, "reference": CameraMode.REFERENCE, "heatmap": CameraMode.BVH_HEATMAP}[mode_name]
        if scene == "cornell":
            scenes.build_cornell(self.engine)
            self.desc = scenes.cornell_camera(size, mode, depth=1)
        else:
            scenes.build_dungeon(self.engine, subdivide=2 if scene == "dungeon134k" else 0)
            self.desc = scenes.dungeon_camera(size, mode, depth=1)
        self.mode = mode
        self.engine.set_seed(args.seed)
        self.cam = self.engine.create_camera(self.desc)
        # the partition: st_dist_partition's tiles (2 ranks: row bands, 4: 2 x 2, 8: 4 x 2 — BASELINE.json's "4-tile" / "8-tile split"),
        # each rank rendering its tile + an apron of redundant pixels in the modes whose passes read neighbours
        self.cols = args.cols
        self.tile = tile_for_rank(self.width, self.height, world, rank, self.cols)
        self.band = (self.tile[1], self.tile[3])
        self.window = (0, 0, self.width, self.height)
        self.needs_apron = mode_name in ("image", "gi_diffuse")   # Reference / heatmap pixels read nothing but their own
        self.apron = (args.apron if self.needs_apron else 0) if world > 1 else 0
        self.c_abi_gather = world > 1 and not debug_shared and not args.py_gather
        if world > 1:
            if self.c_abi_gather:
                # the product path: RCCL through the C ABI (st_dist_*). The id travels over torch.distributed's store — plumbing, like the barrier
                # A rank whose transport does not come up must not leave the others waiting inside a collective: every rank reports, the
                # ranks agree (MIN over torch.distributed), and if any failed ALL fall back to the torch.distributed gather — and the line says so.
                from strolle_amd.api import dist_unique_id
                self.c_abi_error = None
                uid = [None]
                if rank == 0:
                    try: uid = [dist_unique_id()]
                    except Exception as exc: self.c_abi_error = f"st_dist_unique_id: {exc}"
                dist.broadcast_object_list(uid, src=0)
                if uid[0] is None:
                    self.c_abi_error = self.c_abi_error or "rank 0 could not create the RCCL id"
                else:
                    try: self.engine.dist_init(rank, world, uid[0])
                    except Exception as exc: self.c_abi_error = f"st_dist_init: {exc}"
                ok = torch.tensor([0 if self.c_abi_error else 1], dtype=torch.int32, device=f"cuda:{local_rank}")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok[0]) == 0:
                    if not self.c_abi_error:
                        self.engine.dist_shutdown()
                    print(f"[bench rank {rank}] st_dist transport unavailable ({self.c_abi_error or 'another rank failed'}): torch.distributed gather instead", file=sys.stderr)
                    self.c_abi_gather = False
            if self.c_abi_gather:
                owned, self.window = self.engine.dist_set_partition(self.cam, cols=self.cols, apron=self.apron)
                assert owned == self.tile
            else:
                self.window = tile_window(self.width, self.height, self.tile, self.apron)
                self.engine.set_camera_window(self.cam, *self.window)
        self.grid = None              # a cost-weighted grid (Job.set_grid) instead of the equal split
        self.dev = f"cuda:{local_rank}"
        # double-buffered render targets: frame i is gathered on `comm` while frame i+1 renders on `main`
        self.outs = [torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=self.dev) for _ in range(2 if world > 1 else 1)]
        self.full = torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=self.dev) if (world > 1 and rank == 0) else None
        self.main = torch.cuda.current_stream()
        self.comm = torch.cuda.Stream(device=self.dev) if world > 1 else None
        self.gathered = [None, None]  # events: gather that read outs[k] has finished
        self.gather_events = []       # (start, stop) of every gather on the comm stream
        self.stream = self.main.cuda_stream
        self.frame_no = 0
        self.moving = False
        self.moving_t0 = None
        self.geometry = None          # (handle, Instance factory): re-inserted with a new transform before every tick
        self.step_times = [] if (os.environ.get("ST_BENCH_STEP_TIMES") == "1" and world == 1) else None

    def set_grid(self, grid):
        """every rank's tile from `grid` (api.StDistGrid: st_dist_grid / st_dist_grid_rebalance) from the next frame on"""
        from strolle_amd.api import dist_grid_tile
        from strolle_amd.distributed import tile_window
        self.grid = grid
        self.tile = dist_grid_tile(grid, self.rank)
        self.band = (self.tile[1], self.tile[3])
        if self.c_abi_gather:
            self.engine.dist_wait(self.cam)   # gathers in flight were sized by the old tiles
            owned, self.window = self.engine.dist_set_grid(self.cam, grid, apron=self.apron)
            assert owned == self.tile
        else:
            self.window = tile_window(self.width, self.height, self.tile, self.apron)
            self.engine.set_camera_window(self.cam, *self.window)

    def animate(self):
        """bevy-strolle/examples/cornell.rs:82-93: the point light at (sin t / 2, 1.5, cos t / 2), t = elapsed seconds (one frame =
        1/60 s here); the camera — an orbit controller in the example — circles the box's centre at 0.1 rad/s."""
        import math
        from strolle_amd import Light
        if self.moving_t0 is None:
            self.moving_t0 = self.frame_no
        t = (self.frame_no - self.moving_t0) / 60.0
        self.engine.insert_light(1, Light.point((math.sin(t) / 2.0, 1.5, math.cos(t) / 2.0), 0.15, (50.0 / (4.0 * math.pi),) * 3, 20.0))
        a = 0.1 * t
        self.desc = self.scenes.camera_for((self.width, self.height), (3.2 * math.sin(a), 1.0, 3.2 * math.cos(a)), (0.0, 1.0, 0.0), self.mode, True, 1)

    def step(self):
        torch, world, rank = self.torch, self.world, self.rank
        k = self.frame_no % len(self.outs)
        if self.moving:
            self.animate()
        if self.geometry is not None:
            self.engine.insert_instance(self.geometry[0], self.geometry[1](self.frame_no))
        self.frame_no += 1
        out = self.outs[k]
        if world > 1 and self.gathered[k] is not None:
            self.main.wait_event(self.gathered[k])   # outs[k] may be overwritten only after its previous gather has read it
        self.engine.update_camera(self.cam, self.desc)   # Bevy calls update_camera every frame (bevy-strolle/src/stages/prepare.rs:300-340)
        if self.step_times is not None:   # ST_BENCH_STEP_TIMES=1: host time inside st_tick / st_render_camera per step (tools/stall_probe.py reads them)
            t0 = time.perf_counter(); self.engine.tick(self.stream); t1 = time.perf_counter()
            self.engine.render_camera(self.cam, out.data_ptr(), self.stream); t2 = time.perf_counter()
            self.step_times.append((self.frame_no, t0, t1 - t0, t2 - t1))
            return out
        self.engine.tick(self.stream)
        self.engine.render_camera(self.cam, out.data_ptr(), self.stream)
        if world == 1:
            return out
        if self.c_abi_gather:
            # st_dist_gather: grouped ncclSend / ncclRecv of the tiles on the engine's communication stream, behind this frame, under the next
            # (rendering into outs[k] again is ordered behind its gather by the engine)
            self.engine.dist_gather(self.cam, out.data_ptr(), self.full.data_ptr() if rank == 0 else 0, self.stream)
            return self.full if rank == 0 else out
        from strolle_amd.distributed import gather_tiles_to_root
        rendered = torch.cuda.Event(); rendered.record(self.main)
        self.comm.wait_event(rendered)
        with torch.cuda.stream(self.comm):
            g0 = torch.cuda.Event(enable_timing=True); g0.record(self.comm)
            if self.debug_shared:
                self.comm.synchronize()
                host_full = torch.zeros((self.height, self.width, 4)) if rank == 0 else None
                gather_tiles_to_root(out.cpu(), host_full, world, rank, self.cols, tiles=None if self.grid is None else self.grid.tiles())
                if rank == 0:
                    self.full.copy_(host_full)
            else:
                gather_tiles_to_root(out, self.full, world, rank, self.cols, tiles=None if self.grid is None else self.grid.tiles())   # --py-gather: the torch.distributed fallback
            done = torch.cuda.Event(enable_timing=True); done.record(self.comm)
        self.gather_events.append((g0, done))
        self.gathered[k] = done
        return self.full if rank == 0 else out

    def run(self, n):
        f = None
        for _ in range(n):
            f = self.step()
        return f

    def timed_region(self, steps):
        """EXACTLY `steps` steps between barrier + synchronize on both sides."""
        torch = self.torch
        # The interpreter's cyclic garbage collector stays out of the timed region (as timeit does it): a full collection of this process (torch +
        # numpy loaded) takes 35-47 ms, and one of them landing in a 16-ms region is what the driver's round-4 run reported as 2.548 ms per step for
        # the moving-geometry region — found with rocprofv3 --hip-trace (tools/gpu_stall_trace.sh): a 42-ms idle gap of the device with NO HIP call
        # in flight, between one frame's last kernel and the next tick's first copy.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            if self.world > 1:
                self.dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f = self.run(steps)
            torch.cuda.synchronize()
            if self.world > 1:
                self.dist.barrier()
            torch.cuda.synchronize()
            return time.perf_counter() - t0, f
        finally:
            if gc_was_on:
                gc.enable()

    def counted_rays(self):
        """rays of this rank's OWN tile: apron pixels are redundant work and not counted"""
        own = (self.tile[2] - self.tile[0]) * (self.tile[3] - self.tile[1])
        return self.engine.ray_count(self.cam) * own / ((self.window[2] - self.window[0]) * (self.window[3] - self.window[1]))

    def reduce(self, elapsed, rays):
        """max over ranks of the time, sum of the rays, every rank's ms per step"""
        torch, dist, world = self.torch, self.dist, self.world
        if world == 1:
            return elapsed, float(rays), None
        on = "cpu" if self.debug_shared else self.dev
        t = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device=on)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        each = [torch.zeros(1, dtype=torch.float64, device=on) for _ in range(world)]
        dist.all_gather(each, torch.tensor([elapsed], dtype=torch.float64, device=on))
        return float(tmax[0]), float(tsum[1]), [float(x[0]) for x in each]

    def gather_ms(self):
        if self.c_abi_gather:
            return self.engine.dist_gather_ms(self.cam)   # the last gather on the engine's communication stream (HIP events there)
        ev = self.gather_events
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None

    def gathered_bytes(self):
        """what arrives at rank 0 per frame: every tile but its own"""
        from strolle_amd.distributed import tile_for_rank
        t0 = tile_for_rank(self.width, self.height, self.world, 0, self.cols) if self.grid is None else self.grid.tiles()[0]
        return (self.width * self.height - (t0[2] - t0[0]) * (t0[3] - t0[1])) * 16

    def balance(self, rounds, frames):
        """Cost-weighted tiles: `rounds` times, every rank times `frames` frames of its own (no barrier inside), the ranks exchange the
        figures (one float each over torch.distributed: plumbing), and all compute the same new grid with st_dist_grid_rebalance — edges move
        by at most the apron per round, so every pixel a rank newly owns was rendered by it before (as apron) and has warm history.
        Returns what happened per round."""
        from strolle_amd.api import dist_grid, dist_grid_rebalance
        torch, dist = self.torch, self.dist
        log = []
        grid = self.grid if self.grid is not None else dist_grid(self.width, self.height, self.world, self.cols)
        on = "cpu" if self.debug_shared else self.dev
        for _ in range(rounds):
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter(); self.run(frames); torch.cuda.synchronize()
            mine = (time.perf_counter() - t0) / frames * 1e3
            each = [torch.zeros(1, dtype=torch.float64, device=on) for _ in range(self.world)]
            dist.all_gather(each, torch.tensor([mine], dtype=torch.float64, device=on))
            ms = [float(x[0]) for x in each]
            log.append({"per_rank_ms": [round(v, 4) for v in ms], "max_over_mean": round(max(ms) / (sum(ms) / len(ms)), 4), "grid": grid.describe()})
            try:
                grid = dist_grid_rebalance(self.width, self.height, grid, ms, max_step=max(self.apron, 16))
            except Exception as exc:   # a frame too small to move its edges (every rank gets the same answer: the inputs are the same)
                log[-1]["stopped"] = str(exc)
                break
            self.set_grid(grid)
        return log

    def close(self):
        self.torch.cuda.synchronize()
        self.engine.close()
        self.outs = []; self.full = None


def strong_config(torch, dist, args, world, rank, local_rank, debug_shared, scene, mode_name, size, key, steps=None, profile=False):
    """A BASELINE.json multi-GPU config as written — ONE frame of `size` split into `world` tiles (st_dist_partition; + apron in Image mode),
    tiles gathered to rank 0 every frame — timed like the main region (barrier + synchronize on both sides, max over ranks)."""
    job = Job(torch, dist, args, scene, mode_name, size, world, rank, local_rank, debug_shared)
    steps = max(2, min(args.steps, 30)) if steps is None else steps   # (as the line's main region: EXACTLY the K steps asked for)
    job.run(min(args.preroll, 48))
    balance_log = job.balance(args.balance_rounds, 6) if (args.balance_rounds and job.needs_apron and world > 1) else []
    job.run(args.warmup)
    torch.cuda.synchronize()
    job.engine.ray_count(job.cam, reset=True); job.gather_events.clear()
    elapsed, frame = job.timed_region(steps)
    elapsed, rays_total, per_rank = job.reduce(elapsed, job.counted_rays())
    finite = bool(torch.isfinite(frame).all())
    out = {"workload": f"{scene} {size[0]}x{size[1]} mode {mode_name}, strong: one frame in {world} tiles (st_dist_partition), gathered to rank 0",
           "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 4), "Mray_per_s": round(rays_total / elapsed / 1e6, 2),
           "per_rank_ms": None if per_rank is None else [round(x / steps * 1e3, 4) for x in per_rank],
           "gather_ms": None if job.gather_ms() is None else round(job.gather_ms(), 4),
           "gathered_bytes_per_frame": job.gathered_bytes(),
           "tile": [job.tile[2] - job.tile[0], job.tile[3] - job.tile[1]], "band_rows": job.band[1] - job.band[0], "apron_rows": job.apron,
           "gather": "st_dist_gather (RCCL through the C ABI)" if job.c_abi_gather else "torch.distributed fallback",
           "apron_overhead_frac": apron_overhead(job.width, job.height, world, job.apron, job.cols),
           "balance": {"rounds": balance_log, "final_grid": None if job.grid is None else job.grid.describe(),
                       "what": "cost-weighted tiles (st_dist_grid_rebalance): each round every rank times 6 frames, the ranks exchange one float each, edges move by at most the apron"} if balance_log else None,
           "n1_ms_reference": n1_reference(key), "n1_ms_reference_source": "profiles/n1_reference.json (builder-run single-GPU figure, not measured by this process)",
           "rays_per_frame": round(rays_total / steps), "width": size[0], "height": size[1],
           "frame_finite": finite}
    if profile and not args.no_profile:
        # this rank's launches of the same K steps again with per-kernel events (as the N = 1 line's regions 2 / 2b / 2c): the `roofline` object of a
        # line whose main region is this config is then this config's own dominant kernel (on rank 0's tile), not the headline scene's
        eng = job.engine
        eng.profile_enable(1); eng.profile_read(reset=True)
        el_p, _ = job.timed_region(steps)
        prof = eng.profile_read(reset=True)
        eng.profile_enable(1 | 4); eng.profile_read(reset=True); job.timed_region(steps)
        grouped = [q for q in eng.profile_read(reset=True) if q["name"].startswith("a-trous chain")]
        eng.profile_enable(1 | 8); eng.profile_read(reset=True); job.timed_region(steps)
        kernel_events = {q["name"]: q for q in eng.profile_read(reset=True)}
        eng.profile_enable(0)
        for q in prof:
            q["traversal_bytes"] = 0.0   # (the traversal-byte region is the headline line's; B bytes only here)
        lit = None
        try:
            import numpy as np
            from strolle_amd import Buffer
            w = job.window
            d0 = eng.read_buffer(job.cam, Buffer.PRIM_GBUFFER_D0_A).reshape(size[1], size[0], 4)[w[1]:w[3], w[0]:w[2], 0]
            lit = float(np.count_nonzero(d0) / d0.size)
        except Exception:
            lit = None
        out["_profile"] = {"prof": prof, "grouped": grouped, "kernel_events": kernel_events, "profiled_ms": el_p / steps * 1e3, "lit_fraction": lit}
    n1 = out["n1_ms_reference"]
    out["speedup_vs_n1"] = round(n1 / out["ms_per_step"], 3) if (isinstance(n1, (int, float)) and n1 > 0 and tuple(size) == (3840, 2160)) else None
    out["speedup_vs_n1_note"] = "n1_ms_reference / ms_per_step — the N = 1 figure is profiles/n1_reference.json's (one MI355X, builder-run), not this run's; null when the frame is not the config's 3840x2160" 
    job.close()
    return out


def emulate_tiles(torch, args):
    """`--emulate-tiles N` (one GPU): the N tile windows (+ apron) of a frame of --width x --height, rendered one after another with
    st_camera_set_window, each from its own engine with its own temporal state, next to the full frame — the only evidence ONE GPU can give
    for the tile split's balance: per-tile ms, max / mean, and predicted_speedup = T(full frame) / max_i T(tile_i) (the gather, 8.3 MB per
    tile per frame, rides behind the next frame and is not in the figure). PREDICTED FROM ONE GPU: no scaling curve has been measured."""
    from strolle_amd.distributed import tile_for_rank, tile_window
    n = args.emulate_tiles
    size = (args.width, args.height)
    steps = max(3, min(args.steps, 30))
    apron = args.apron if args.mode in ("image", "gi_diffuse") else 0
    from strolle_amd.api import dist_grid, dist_grid_rebalance

    def one(window):
        job = Job(torch, None, args, args.scene, args.mode, size, 1, 0, 0, False)
        if window is not None:
            job.engine.set_camera_window(job.cam, *window)
        job.run(min(args.preroll, 48)); job.run(args.warmup)
        torch.cuda.synchronize(); job.engine.ray_count(job.cam, reset=True)
        el, _ = job.timed_region(steps)
        rays = job.engine.ray_count(job.cam)
        job.close()
        return el / steps * 1e3, rays / steps

    full_ms, full_rays = one(None)
    grid = dist_grid(size[0], size[1], n, args.cols)
    rounds = []
    for it in range(1 + (args.balance_rounds if apron else 0)):   # round 0: the equal split; then st_dist_grid_rebalance from the measured tile times
        tiles, windows, per_tile = grid.tiles(), [], []
        for t in tiles:
            w = tile_window(size[0], size[1], t, apron)
            ms, _ = one(w)
            windows.append(list(w)); per_tile.append(round(ms, 4))
        mean = sum(per_tile) / n
        rounds.append({"grid": grid.describe(), "per_tile_ms": per_tile, "tile_rects": [list(t) for t in tiles], "tile_windows": windows,
                       "max_over_mean": round(max(per_tile) / mean, 4), "sum_of_tiles_over_full_frame": round(sum(per_tile) / full_ms, 4),
                       "predicted_speedup": round(full_ms / max(per_tile), 3)})
        grid = dist_grid_rebalance(size[0], size[1], grid, per_tile)
    best = max(rounds, key=lambda r: r["predicted_speedup"])
    return {"what": f"{args.scene} {size[0]}x{size[1]} mode {args.mode}: the {n} tile windows (apron {apron}) rendered one after another on ONE GPU (st_camera_set_window), each with its own engine and history; round 0 = st_dist_partition's equal split, later rounds = st_dist_grid_rebalance from the previous round's tile times",
            "label": "predicted from one GPU - no curve measured", "tiles": n, "steps": steps, "full_frame_ms": round(full_ms, 4), "ideal_speedup": n,
            "equal_split": {k: rounds[0][k] for k in ("per_tile_ms", "max_over_mean", "predicted_speedup", "sum_of_tiles_over_full_frame")},
            "balanced": {k: best[k] for k in ("per_tile_ms", "max_over_mean", "predicted_speedup", "sum_of_tiles_over_full_frame", "grid")},
            "rounds": rounds,
            "note": "sum_of_tiles_over_full_frame > 1 is what the apron's redundant pixels and a smaller launch's tail cost; predicted_speedup = T(full frame) / max_i T(tile_i) assumes the gather stays hidden behind the next frame (st_dist_gather on its own stream)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)  # two full 6-frame GI cycles (strolle-gpu/src/frame.rs:19-21)
    ap.add_argument("--preroll", type=int, default=96,
                    help="frames rendered as part of the setup, before the warm-up steps: the renderer's temporal state (reservoir "
                         "sample counts, which stop the preview passes from drawing neighbours once they reach 8; the denoiser's "
                         "16-frame history) needs several dozen frames to settle, and the metric is the steady-state rate of a static "
                         "camera. Measured: frames 12..72 of a new camera run 2.3 %% slower than every later block of 60. Reported in "
                         "the JSON line; 0 = none")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--apron", type=int, default=16, help="extra pixels rendered around a rank's tile in Image mode (N > 1)")
    ap.add_argument("--cols", type=int, default=0, help="N > 1: columns of the tile grid (0 = st_dist_partition's default: 2 ranks row bands, 4: 2 x 2, 8: 4 x 2; 1 = row bands)")
    ap.add_argument("--py-gather", action="store_true", help="N > 1: gather through torch.distributed (the fallback) instead of st_dist_gather (RCCL through the C ABI)")
    ap.add_argument("--scene", choices=["cornell", "dungeon", "dungeon134k"], default="cornell",
                    help="cornell = the headline workload; dungeon = BASELINE.json config 3's scene (level.glb + the demo's three tori); dungeon134k = the same surface subdivided twice (synthetic ~100k-triangle stand-in)")
    ap.add_argument("--mode", choices=["image", "gi_diffuse", "reference", "heatmap"], default="image")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: weak = the frame grows with N (per-GPU work fixed), strong = the frame stays --width x --height. Left out with the default workload "
                         "(the driver's command), the N > 1 line IS BASELINE.json's config 5 — dungeon 3840x2160 Image, ONE frame in N cost-balanced tiles, strong — "
                         "and the weak region rides along under multi_gpu.weak_scaling_extra")
    ap.add_argument("--exact", action="store_true", help="run the bit-exact build of the kernels (ST_ARITH_EXACT) instead of the default fast build")
    ap.add_argument("--dump-frame", default=None, help="rank 0 saves the last (gathered) frame as .npy — tests compare it with a single-GPU render")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing inside the timed region")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra regions (moving light + camera, moving geometry, present path; N > 1: BASELINE configs 5 / 4 as written)")
    ap.add_argument("--extras-size", type=int, nargs=2, default=(3840, 2160), metavar=("W", "H"), help="frame of the N > 1 strong-scaling extras (tests shrink it)")
    ap.add_argument("--emulate-tiles", type=int, default=0, metavar="N",
                    help="one GPU: render the N tile windows (+ apron) of a --width x --height frame one after another and print per-tile ms, max / mean and the speed-up N GPUs could reach at best (prints its own JSON line; no headline)")
    ap.add_argument("--balance-rounds", type=int, default=4, help="N > 1 strong-scaling extras (Image modes) and --emulate-tiles: rounds of cost-weighted tile rebalancing (st_dist_grid_rebalance) before the timed region; 0 = the equal split")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from strolle_amd import CameraMode, Engine, scenes
    from strolle_amd.distributed import weak_scaling_frame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # ST_BENCH_DEBUG_SHARED_GPU=1 (functional check only, never a result): all ranks render on cuda:0 and the gather goes
    # through gloo + host copies, so that the N > 1 control flow can be exercised on a single-GPU box.
    debug_shared = os.environ.get("ST_BENCH_DEBUG_SHARED_GPU") == "1"
    if debug_shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if debug_shared:
            dist.init_process_group("gloo")
        else:
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL; binds the communicator to this rank's GPU
            except TypeError:   # a torch without the device_id keyword
                dist.init_process_group("nccl")

    if args.emulate_tiles:
        if world != 1:
            raise SystemExit("--emulate-tiles runs on one GPU")
        out = emulate_tiles(torch, args)
        out["build_stamp"] = build_stamp()
        print(json.dumps(out))
        return
    # The driver's N > 1 command (no --scaling, the default workload): the line's value / ms_per_step / config.workload are BASELINE.json config 5 as
    # written — the north_star's scaling curve (strong: the frame is fixed, N tiles) — and the weak region below is demoted to an extra. N = 1 stays the headline.
    config5_main = world > 1 and args.scaling is None and (args.scene, args.mode, args.width, args.height) == ("cornell", "image", 1920, 1080)
    if args.scaling is None:
        args.scaling = "weak"
    base = (args.width, args.height)
    width, height = weak_scaling_frame(base, world) if args.scaling == "weak" else base
    rccl_ranks, backend = None, None
    if world > 1:   # an actual collective on the backend the gather will use: every rank must have joined it
        backend = dist.get_backend()
        one = torch.ones(1, dtype=torch.float32, device="cpu" if debug_shared else f"cuda:{local_rank}")
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        assert rccl_ranks == dist.get_world_size() == world
    job = Job(torch, dist, args, args.scene, args.mode, (width, height), world, rank, local_rank, debug_shared)
    engine, cam, band, window, dev = job.engine, job.cam, job.band, job.window, job.dev
    needs_apron = job.needs_apron
    step, timed_region = job.step, lambda: job.timed_region(args.steps)

    job.run(args.preroll)   # setup: bring the camera's temporal accumulators to their steady state (see --preroll)
    job.run(args.warmup)
    torch.cuda.synchronize()
    engine.ray_count(cam, reset=True)
    job.gather_events.clear()

    # region 1: EXACTLY K steps, no instrumentation -> value / ms_per_step
    elapsed, frame = timed_region()
    rays = job.counted_rays()
    gather_ms = job.gather_ms()
    # region 2: the same K steps again with HIP events around every run of same-slot launches (on the launch stream, pass
    # graph serial) -> per-kernel average durations for the roofline object. Kept out of region 1: it costs ~10 %.
    prof, profiled_ms, grouped, kernel_events = [], None, [], {}
    if not args.no_profile:
        engine.profile_enable(1)   # ST_PROFILE_TIMING
        engine.profile_read(reset=True)
        elapsed_p, _ = timed_region()
        prof = engine.profile_read(reset=True)
        profiled_ms = elapsed_p / args.steps * 1e3
        # region 2b: the same once more with the a-trous chain's four back-to-back launches under ONE event pair
        # (ST_PROFILE_GROUP_ATROUS): an event between two kernels makes the second wait for a barrier packet, 3-15 us each
        engine.profile_enable(1 | 4)
        engine.profile_read(reset=True)
        timed_region()
        grouped = [q for q in engine.profile_read(reset=True) if q["name"].startswith("a-trous chain")]
        # region 2c: once more with every launch carrying its own start / stop events (ST_PROFILE_KERNEL_EVENTS: the dispatch's
        # timestamps through hipExtLaunchKernelGGL, no event packet between kernels) -> per-kernel durations as rocprofv3 sees them
        engine.profile_enable(1 | 8)
        engine.profile_read(reset=True)
        timed_region()
        kernel_events = {q["name"]: q for q in engine.profile_read(reset=True)}
        # region 3, untimed: a few more frames with the traversal-byte counters on (ST_PROFILE_TRAVERSAL_BYTES; the kernels
        # sum the reference's `used_memory` over their rays, which costs ~10 us per tracing launch and is therefore off in
        # regions 1 and 2) -> the LDS / L2-served A part of the algorithmic bytes, scaled to K steps
        engine.profile_enable(2)
        engine.profile_read(reset=True)
        extra = max(3, min(args.steps, 6)) // 3 * 3   # whole GI schedules (three frames)
        for _ in range(extra):
            step()
        torch.cuda.synchronize()
        trav = {q["name"]: q["traversal_bytes"] * args.steps / extra for q in engine.profile_read(reset=True)}
        engine.profile_enable(0)
        for q in prof:
            q["traversal_bytes"] = trav.get(q["name"], 0.0)
            q["algorithmic_bytes"] += q["traversal_bytes"]
    elapsed, rays_total, per_rank = job.reduce(elapsed, rays)
    per_rank_ms = None if per_rank is None else [round(x / args.steps * 1e3, 4) for x in per_rank]
    finite = bool(torch.isfinite(frame).all())
    if rank == 0 and args.dump_frame:
        import numpy as np
        np.save(args.dump_frame, frame.cpu().numpy())
    headline = (args.scene, args.mode) == ("cornell", "image")
    # share of the pixels whose primary ray hit something (G-buffer depth != 0): a sky pixel leaves every denoiser kernel after one texel but
    # is credited a lit pixel's algorithmic bytes, so `roofline.frac` has to be read next to this figure (`roofline.frac_lit`)
    lit_fraction = None
    if args.mode in ("image", "gi_diffuse"):
        try:
            from strolle_amd import Buffer
            import numpy as np
            d0 = engine.read_buffer(cam, Buffer.PRIM_GBUFFER_D0_A).reshape(height, width, 4)[window[1]:window[3], window[0]:window[2], 0]
            lit_fraction = float(np.count_nonzero(d0) / d0.size)
        except Exception:
            lit_fraction = None
    engine_exact = engine.exact
    bvh_tree = ("device builder (k_lbvh.hip: LBVH, 4-wide; ST_BVH_AUTO picked it: the host's tree hangs long leaf runs on large faces, st_debug_auto_tree)" if engine.device_builds() > 0
                else "host, binned SAH (the reference's tree: strolle/src/bvh/builder.rs)")   # which tree the timed frames walked — read BEFORE bvh_depth(), a debug read that rebuilds a stale host tree
    bvh_depth = engine.bvh_depth()   # read here: the N > 1 extras close this job's engine
    extras = {}
    if not args.no_extras and world == 1 and headline:
        # -- the headline under motion (bevy-strolle/examples/cornell.rs animates its light every frame; the static figure
        #    above is the renderer's cheapest state: no reservoir is invalidated, preview resampling draws no neighbour)
        job.moving = True
        job.run(args.warmup)
        torch.cuda.synchronize(); engine.ray_count(cam, reset=True)
        el_m, frame_m = job.timed_region(args.steps)
        rays_m = engine.ray_count(cam)
        job.moving = False
        extras["ms_per_step_moving"] = round(el_m / args.steps * 1e3, 4)
        extras["moving"] = {"what": "same K steps; point light at (sin t / 2, 1.5, cos t / 2), t advancing 1/60 s per frame (cornell.rs:82-93), camera orbiting the box at 0.1 rad/s; light + camera updated through insert_light / update_camera before every tick",
                            "Mray_per_s": round(rays_m / el_m / 1e6, 2), "rays_per_frame": round(rays_m / args.steps), "frame_finite": bool(torch.isfinite(frame_m).all())}
        # -- geometry moving (examples/stress-bvh.rs in miniature): one object of the scene is re-inserted with a new transform before
        #    every tick; the tree is refitted on the device (ST_BVH_REFIT_DEVICE), st_tick sends the moved triangles only
        import math
        import numpy as np
        from strolle_amd import Instance, Light
        job.desc = job.scenes.cornell_camera((width, height), job.mode, depth=1)                          # camera and light back at t = 0
        engine.insert_light(1, Light.point((0.0, 1.5, 0.5), 0.15, (50.0 / (4.0 * math.pi),) * 3, 20.0))
        npz = np.load(os.path.join(job.scenes.ASSETS, "cornell.npz"))
        mesh = int(npz["n_meshes"]) - 1
        rest = np.ascontiguousarray(npz[f"xform_{mesh}"].reshape(4, 3).T, np.float32)
        def placed(i):
            x = rest.copy(); x[0, 3] += np.float32(0.15 * math.sin(i / 20.0))
            return Instance(1 + mesh, 1 + int(npz[f"material_{mesh}"]), x)
        engine.set_bvh_refresh(2)
        job.geometry = (1 + mesh, placed)
        # The first ticks of this mode do one-off work (a rebuild that indexes the tree for the device refit, each device copy's first full
        # upload, the mesh store, the first launch of k_bvh_bake's code object: tools/stall_probe.py shows them in ticks 0-2), so the warm-up
        # is never shorter than 8 ticks; and the region is timed TWICE back to back, both figures in the line. (The driver's round-4 run
        # reported 2.548 ms here: a 35-ms garbage collection of the interpreter inside the region — Job.timed_region keeps the collector out now.)
        job.run(max(args.warmup, 8))
        torch.cuda.synchronize(); engine.ray_count(cam, reset=True)
        el_g1, frame_g = job.timed_region(args.steps)
        rays_g = engine.ray_count(cam)
        el_g2, frame_g = job.timed_region(args.steps)
        el_g = min(el_g1, el_g2); rays_g = rays_g if el_g1 <= el_g2 else engine.ray_count(cam) - rays_g
        job.geometry = None
        if job.step_times is not None:   # where did the host spend its time in the two regions? (steps, longest tick / render call and the step it fell in)
            last = job.step_times[-2 * args.steps:]
            gaps = [b[1] - a[1] for a, b in zip(last, last[1:])]
            worst_tick = max(last, key=lambda r: r[2]); worst_render = max(last, key=lambda r: r[3]); worst_gap = max(range(len(gaps)), key=lambda i: gaps[i])
            print(f"[bench] geometry regions, host side: longest st_tick {worst_tick[2] * 1e3:.3f} ms (step {worst_tick[0] - last[0][0]}), longest st_render_camera {worst_render[3] * 1e3:.3f} ms (step {worst_render[0] - last[0][0]}), "
                  f"longest step-to-step gap {gaps[worst_gap] * 1e3:.3f} ms (after step {worst_gap}); per step tick/render ms: " + " ".join(f"{r[2] * 1e3:.2f}/{r[3] * 1e3:.2f}" for r in last), file=sys.stderr)
        rebuilds, refits = engine.bvh_refits()
        engine.insert_instance(1 + mesh, Instance(1 + mesh, 1 + int(npz[f"material_{mesh}"]), rest))
        engine.set_bvh_refresh(0)
        extras["ms_per_step_geometry_moving"] = round(el_g / args.steps * 1e3, 4)
        extras["geometry_moving"] = {"what": f"same K steps; instance {1 + mesh} of the scene re-inserted with a new transform before every tick, BVH refitted on the device (k_bvh.hip; ST_BVH_REFIT_DEVICE)",
                                     "regions_ms_per_step": [round(el_g1 / args.steps * 1e3, 4), round(el_g2 / args.steps * 1e3, 4)],
                                     "regions_note": "two back-to-back regions of K steps; ms_per_step_geometry_moving is the faster one, both are given",
                                     "Mray_per_s": round(rays_g / el_g / 1e6, 2), "bvh_rebuilds": rebuilds, "bvh_refits": refits, "bvh_device_refits": engine.bvh_device_refits(),
                                     "frame_finite": bool(torch.isfinite(frame_g).all())}
        # -- the facade's present path (rust/strolle-hip/src/present.rs, examples/render_gltf.c): RGBA8 target, two device
        #    frames + two page-locked host frames alternate, frame N-1 is polled (never a stream join) while frame N renders
        from strolle_amd import OutputFormat
        job.desc = job.scenes.cornell_camera((width, height), job.mode, depth=1)
        import math
        from strolle_amd import Light
        engine.insert_light(1, Light.point((0.0, 1.5, 0.5), 0.15, (50.0 / (4.0 * math.pi),) * 3, 20.0))   # light back at t = 0
        engine.set_output_format(cam, OutputFormat.RGBA8_UNORM_SRGB)
        dev8 = [torch.zeros((height, width, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
        host8 = [torch.zeros((height, width, 4), dtype=torch.uint8).pin_memory() for _ in range(2)]
        waited = [0]

        def present_step(i):
            k = i & 1
            engine.update_camera(cam, job.desc); engine.tick(job.stream)
            engine.render_camera(cam, dev8[k].data_ptr(), job.stream)
            engine.present_copy(cam, dev8[k].data_ptr(), host8[k].data_ptr(), dev8[k].numel(), job.stream)
            if i > 0 and not engine.present_ready(cam, host8[k ^ 1].data_ptr()):
                waited[0] += 1
                engine.present_ready(cam, host8[k ^ 1].data_ptr(), wait=True)
        for i in range(args.warmup + 12):
            present_step(i)
        torch.cuda.synchronize(); waited[0] = 0
        t0 = time.perf_counter()
        for i in range(args.steps):
            present_step(i)
        engine.present_ready(cam, host8[(args.steps - 1) & 1].data_ptr(), wait=True)   # the last frame has to arrive too
        torch.cuda.synchronize()
        el_p = time.perf_counter() - t0
        arrived = bool((host8[(args.steps - 1) & 1] == dev8[(args.steps - 1) & 1].cpu()).all())   # what landed on the host is the frame that was composed
        engine.set_output_format(cam, OutputFormat.RGBA32F)
        extras["ms_per_step_with_present"] = round(el_p / args.steps * 1e3, 4)
        extras["present"] = {"what": "same K steps composed as RGBA8 sRGB into two alternating device frames, st_camera_present_copy to two page-locked host frames, the previous frame's copy polled before the next tick (one frame of latency, no stream join)",
                             "bytes_per_frame": width * height * 4, "polls_that_found_the_copy_pending": waited[0], "polls_note": "the host enqueues frames faster than the GPU renders them; a pending poll blocks on that ONE copy (never on the render stream), which paces the host one frame ahead",
                             "host_frame_nonzero": bool(host8[(args.steps - 1) & 1].any()), "host_frame_equals_device_frame": arrived}
        del dev8, host8
    strong = {}
    if world > 1 and (config5_main or not args.no_extras):
        job.close()
        ex = tuple(args.extras_size)
        strong["strong_config5"] = strong_config(torch, dist, args, world, rank, local_rank, debug_shared, "dungeon", "image", ex, "config5_dungeon_3840x2160_image_ms",
                                                 steps=args.steps if config5_main else None, profile=config5_main)
        if world == 4 and not args.no_extras:
            strong["strong_config4"] = strong_config(torch, dist, args, world, rank, local_rank, debug_shared, "cornell", "reference", ex, "config4_cornell_3840x2160_reference_ms")
    copy_own, copy_torch = measure_copy_ceiling(torch, dev, None if strong else engine) if (rank == 0 and not debug_shared) else (None, None)
    copy_ceiling = max([c for c in (copy_own, copy_torch) if c], default=None)
    key = workload_key(args.scene, width, height, args.mode)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        result = {
            "metric": "Mray/s (primary + shadow + GI rays traced per second, whole job)",
            "value": round(rays_total / elapsed / 1e6, 2), "unit": "Mray/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_frames": args.preroll, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"Cornell box {width}x{height}, CameraMode::Image{{denoise:true}} (1 spp ReSTIR DI+GI + SVGF), static camera, point light at t=0"
                                    if headline else f"{args.scene} {width}x{height}, mode {args.mode} (NOT the headline workload)"),
                       "scene": {"cornell": "cornell (32 triangles, 2 light slots)", "dungeon": scenes.DUNGEON_DESCRIPTION,
                                 "dungeon134k": "SYNTHETIC: the dungeon with every triangle split into 16, same materials and lights"}[args.scene],
                       "arithmetic": "exact (bit-identical to the CPU oracle)" if engine_exact else "fast (hardware rcp/sqrt/exp/log, FMA contraction; traversal exact; tolerances in tests/test_gpu_fast_tolerance.py)",
                       "width": width, "height": height,
                       "bvh_tree": bvh_tree,
                       "bvh_deepest_internal_chain": bvh_depth[0], "bvh_stack_entries": bvh_depth[1],   # (of the HOST's tree) deeper than the stack = dropped pushes (st_debug_bvh_depth): the contract walks' stack is as deep as the tree needs, up to 32
                       "dropped_pushes": "none: contract walks hold the tree's deepest chain (tests/test_c_abi.py); the wide walk keeps 24 entries and renders the same bits with 48 (tests/test_gpu_fast_tolerance.py test_the_wide_walk_drops_no_push); the oracle drops none at 24 and its deepest stack on config 3's scene is 13 (tests/test_wide_bvh.py)" if bvh_depth[0] <= bvh_depth[1] else "POSSIBLE: the tree is deeper than the 32-entry stack",
                       "per_gpu_rows": band[1] - band[0], "apron_rows": (args.apron if needs_apron else 0) if world > 1 else 0,
                       "rays_per_frame": round(rays_total / args.steps), "frame_finite": finite,
                       **({"DEBUG_NOT_A_RESULT": "ranks share cuda:0, gather through gloo + host copies"} if debug_shared else {}),
                       "partition": "single GPU" if world == 1 else f"{world} tiles of {job.tile[2] - job.tile[0]}x{job.tile[3] - job.tile[1]} (st_dist_partition), per-frame RCCL gather of the RGBA32F tiles to rank 0 overlapped with the next frame"},
        }
        if world > 1:
            result["multi_gpu"] = {"rccl_ranks": rccl_ranks, "backend": backend, "rccl_ranks_note": "sum of ones over an all-reduce on that backend before the first frame (nccl = RCCL on ROCm)",
                                   "per_rank_ms_per_step": per_rank_ms, "gather_ms_on_comm_stream_rank0": None if gather_ms is None else round(gather_ms, 4),
                                   "gathered_bytes_per_frame": job.gathered_bytes(), "tile": [job.tile[2] - job.tile[0], job.tile[3] - job.tile[1]],
                                   "gather": "st_dist_gather (RCCL through the C ABI: grouped ncclSend / ncclRecv on the engine's communication stream)" if job.c_abi_gather else "torch.distributed fallback (gloo / --py-gather)",
                                   "apron_overhead_frac": apron_overhead(width, height, world, job.apron, job.cols),
                                   "main_region": "weak scaling: NOT a BASELINE config for N > 1 (the frame grows with N); the BASELINE configs as written are strong_config5 / strong_config4 below" if args.scaling == "weak" else "strong scaling of --width x --height",
                                   "hardware_scaling_curve": "none measured by the builder (gpurun boxes have one GPU); whatever the driver's N = 1, 2, 4, 8 runs print is the first",
                                   **strong}
        result.update(extras)
        if config5_main:
            # the N > 1 line of the driver's command: BASELINE.json config 5 as written is the line; what the main region above measured (weak scaling of
            # the headline workload: N x 1080p Cornell) moves under multi_gpu.weak_scaling_extra
            c5 = strong["strong_config5"]
            mg = result["multi_gpu"]
            mg["weak_scaling_extra"] = {"what": "the headline workload grown to N x 1920x1080 pixels, one tile per rank + apron, gathered to rank 0 (NOT a BASELINE config for N > 1)",
                                        "value_Mray_per_s": result["value"], "ms_per_step": result["ms_per_step"], "workload": result["config"]["workload"],
                                        "width": width, "height": height, "rays_per_frame": result["config"]["rays_per_frame"], "per_rank_ms_per_step": mg.pop("per_rank_ms_per_step"),
                                        "gather_ms_on_comm_stream_rank0": mg.pop("gather_ms_on_comm_stream_rank0"), "gathered_bytes_per_frame": mg.pop("gathered_bytes_per_frame"),
                                        "tile": mg.pop("tile"), "apron_overhead_frac": mg.pop("apron_overhead_frac"), "partition": result["config"]["partition"]}
            mg["main_region"] = "strong_config5: BASELINE.json config 5 as written (strong scaling) — value / ms_per_step / config of this line are its figures"
            result["value"], result["ms_per_step"], result["scaling"] = c5["Mray_per_s"], c5["ms_per_step"], "strong"
            result["speedup_vs_n1"] = c5["speedup_vs_n1"]
            result["speedup_vs_n1_note"] = c5["speedup_vs_n1_note"] + "; a prediction from one GPU exists (profiles/r05_tile_balance.json: 5.98x equal split, 6.63x balanced at N = 8) — predicted from one GPU, no curve measured by the builder"
            result["config"].update({"workload": f"BASELINE.json config 5: dungeon {c5['width']}x{c5['height']}, CameraMode::Image{{denoise:true}} (ReSTIR DI+GI + SVGF), ONE frame in {world} cost-balanced tiles "
                                                 f"(st_dist_grid_rebalance) + apron {c5['apron_rows']}, tiles gathered to rank 0 by st_dist_gather (RCCL ncclSend / ncclRecv through the C ABI)",
                                     "scene": scenes.DUNGEON_DESCRIPTION, "width": c5["width"], "height": c5["height"], "rays_per_frame": c5["rays_per_frame"], "frame_finite": c5["frame_finite"],
                                     "per_gpu_rows": c5["band_rows"], "apron_rows": c5["apron_rows"],
                                     "partition": f"{world} tiles, final grid {None if not c5.get('balance') else c5['balance']['final_grid']}, gather: {c5['gather']}"})
            for k in ("bvh_deepest_internal_chain", "bvh_stack_entries", "dropped_pushes"):   # (those describe the headline scene's tree)
                result["config"].pop(k, None)
            # the per-kernel table and the roofline object below: this config's own launches on rank 0's tile (strong_config's profile regions), not the weak region's
            pr = c5.pop("_profile", None)
            prof, grouped, kernel_events, profiled_ms, lit_fraction = (pr["prof"], pr["grouped"], pr["kernel_events"], pr["profiled_ms"], pr["lit_fraction"]) if pr else ([], [], {}, None, None)
            key = f"config5_dungeon_{c5['width']}x{c5['height']}_image_tile_1_of_{world}"   # (no counter summary exists for a tile: roofline.traffic is null)
            ms = c5["ms_per_step"]
        result["build_stamp"] = build_stamp()
        if copy_ceiling is not None:
            result["hbm_copy_ceiling_GBps_measured"] = round(copy_ceiling, 1)
            result["hbm_copy_ceiling"] = {"own_float4_copy_kernel_GBps": None if copy_own is None else round(copy_own, 1), "torch_copy_GBps": round(copy_torch, 1),
                                          "guide_GBps": 6290.0, "note": "frac_of_measured_copy_ceiling uses the larger of the two measured figures; `frac` is always against the 8 TB/s spec peak"}
        if prof:
            by = {p["name"]: p for p in prof}
            # the dominant kernel: the slot — or, for the a-trous passes, the family of slots — with the largest total time
            fam = [by[n] for n in WAVELET_SLOTS if n in by]
            fam_ms = sum(p["total_ms"] for p in fam)
            top = max(prof, key=lambda p: p["total_ms"])
            if fam and fam_ms >= max(p["total_ms"] for p in prof if p["name"] not in WAVELET_SLOTS):
                launches = sum(p["launches"] for p in fam)
                passes = sum(p["launches"] * WAVELET_SLOTS[p["name"]] for p in fam)
                alg = sum(p["algorithmic_bytes"] for p in fam)
                trav = sum(p["traversal_bytes"] for p in fam)
                composed = "denoise_wavelet+composition" in by
                name = f"denoise_wavelet ({passes // args.steps} a-trous passes per frame" + (" + frame composition (run inside the last pass's launch)" if composed else "") + f" in {launches // args.steps} launches)"
                tot_ms, symbols = fam_ms, WAVELET_SYMBOLS
            else:
                launches, alg, trav, tot_ms, name, symbols = top["launches"], top["algorithmic_bytes"], top["traversal_bytes"], top["total_ms"], top["name"], [top["name"]]
            per_slot = {"avg_launch_ms": round(tot_ms / launches, 5), "achieved": round((alg - trav) / (tot_ms * 1e-3) / 1e9, 2), "frac": round((alg - trav) / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            timing = "one HIP-event pair per profiler slot (three pairs over the chain's four launches)"
            if grouped and grouped[0]["launches"] == launches and fam and fam_ms >= max(p["total_ms"] for p in prof if p["name"] not in WAVELET_SLOTS):
                tot_ms = grouped[0]["total_ms"]   # same launches, same bytes; one interval around the four launches of a frame
                timing = "ONE HIP-event pair around the chain's four back-to-back launches of a frame (ST_PROFILE_GROUP_ATROUS)"
            one_pair = {"avg_launch_ms": round(tot_ms / launches, 5), "achieved": round((alg - trav) / (tot_ms * 1e-3) / 1e9, 2), "frac": round((alg - trav) / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            ke = [kernel_events[p["name"]] for p in (fam if fam and name.startswith("denoise_wavelet (") else [top]) if p["name"] in kernel_events]
            if ke and sum(q["launches"] for q in ke) == launches and sum(q["total_ms"] for q in ke) > 0:
                tot_ms = sum(q["total_ms"] for q in ke)   # same launches, same bytes; each kernel's own dispatch timestamps
                timing = "per-launch start / stop events attached to the dispatch (hipExtLaunchKernelGGL, ST_PROFILE_KERNEL_EVENTS): the kernels' own durations, as rocprofv3's kernel trace reports them"
            avg_ms = tot_ms / launches
            b_bytes = (alg - trav) / launches
            achieved = b_bytes / (avg_ms * 1e-3) / 1e9   # screen-space (HBM) bytes only: traversal bytes are cache- / LDS-served
            run_slots = sorted(p["name"] for p in prof)
            traffic, source = static_traffic(symbols, key, run_slots)
            wavelet_only = None
            if fam and "denoise_wavelet+composition" in by:   # the family without the launch that also composes the frame
                rest = [p for p in fam if p["name"] != "denoise_wavelet+composition"]
                r_alg = sum(p["algorithmic_bytes"] - p["traversal_bytes"] for p in rest); r_ms = sum(p["total_ms"] for p in rest)
                if r_ms > 0:
                    wavelet_only = {"passes_per_frame": sum(p["launches"] * WAVELET_SLOTS[p["name"]] for p in rest) // args.steps, "launches_per_frame": sum(p["launches"] for p in rest) // args.steps,
                                    "achieved": round(r_alg / (r_ms * 1e-3) / 1e9, 2), "frac": round(r_alg / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "ms_per_frame": round(r_ms / args.steps, 5)}
            result["roofline"] = {"bound": "hbm", "kernel": name, "timing": timing, "with_one_event_pair_around_the_chain": one_pair, "with_one_event_pair_per_slot": per_slot, "wavelet_only": wavelet_only, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": source,
                                  # the same kernel three ways, so that ONE line says how flattering the scene is: `frac` credits every pixel the
                                  # pass's algorithmic bytes; `frac_counter` = the counters' HBM bytes per launch (`traffic`) over the same duration;
                                  # `frac_lit` credits only the pixels that are not sky (`lit_pixel_fraction` of the frame)
                                  "frac_counter": None if not traffic else round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                  "lit_pixel_fraction": None if lit_fraction is None else round(lit_fraction, 4),
                                  "frac_lit": None if lit_fraction is None else round(achieved * lit_fraction / HBM_PEAK_GBS, 5),
                                  "frac_note": "frac = algorithmic bytes of ALL pixels / time / 8 TB/s; frac_counter = HBM bytes by the FETCH_SIZE / WRITE_SIZE counters / time / 8 TB/s; frac_lit = algorithmic bytes of the lit (non-sky) pixels only: a sky pixel leaves each denoiser kernel after one texel",
                                  "frac_of_measured_copy_ceiling": None if not copy_ceiling else round(achieved / copy_ceiling, 5),
                                  "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": round(b_bytes),
                                  "traversal_bytes_per_launch_not_hbm": round(trav / launches),
                                  "note": "HIP events on the launch stream around each run of back-to-back launches of the slot, over a second region of the same K steps; algorithmic bytes = compulsory screen-space plane bytes of the reference passes the launches execute (SURVEY.md 8d / DESIGN.md section 4)"}
            # the whole of frame_denoising.rs after reprojection: estimate_variance + the five a-trous passes (five launches). With
            # the variance pass's long-history branch riding in the reproject stages its share of the work moved into the
            # strides-1+2 launch's inputs, so the wavelet-only figure above and this one are both given.
            den = [by[n] for n in list(WAVELET_SLOTS) + ["denoise_variance"] if n in by]   # (+ composition when it rides in the last pass)
            if den and sum(p["total_ms"] for p in den) > 0:
                den_alg = sum(p["algorithmic_bytes"] - p["traversal_bytes"] for p in den)
                den_ms = sum(p["total_ms"] for p in den)
                den_rate = den_alg / (den_ms * 1e-3) / 1e9
                result["roofline_denoiser"] = {"kernel": "estimate_variance + 5 a-trous passes (frame_denoising.rs:80-361)", "bound": "hbm",
                                               "achieved": round(den_rate, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(den_rate / HBM_PEAK_GBS, 5),
                                               "ms_per_frame": round(den_ms / args.steps, 5), "algorithmic_bytes_per_frame": round(den_alg / args.steps)}
            tot = sum(p["total_ms"] for p in prof)
            def counter_rate(p):  # the slot's HBM bytes per launch by the committed counter passes over this run's launch time
                per_launch, _ = static_traffic([p["name"]], key, run_slots)
                return round(per_launch / (p["total_ms"] / p["launches"] * 1e-3) / 1e9, 1) if per_launch and p["total_ms"] > 0 else None
            result["kernels"] = {p["name"]: {"ms_per_frame": round(p["total_ms"] / args.steps, 5), "launches_per_frame": round(p["launches"] / args.steps, 2),
                                            "us_per_launch": round(p["total_ms"] / p["launches"] * 1e3, 2),
                                            "B_only_GBps": round((p["algorithmic_bytes"] - p["traversal_bytes"]) / (p["total_ms"] * 1e-3) / 1e9, 1) if p["total_ms"] > 0 else None,
                                            "counter_GBps": counter_rate(p),
                                            "traversal_GBps_cache_served": round(p["traversal_bytes"] / (p["total_ms"] * 1e-3) / 1e9, 1) if p["total_ms"] > 0 and p["traversal_bytes"] else 0.0}
                                 for p in sorted(prof, key=lambda p: -p["total_ms"])}
            result["kernels_note"] = ("B_only_GBps: compulsory screen-space bytes of the REFERENCE passes a launch executes (unfused accounting, SURVEY.md 8d) "
                                      "over its duration - an equivalent rate: a fused launch never moves part of those bytes, so it can exceed the HBM peak; "
                                      "counter_GBps: the same launch's FETCH_SIZE / WRITE_SIZE bytes from profiles/pmc/<workload>.json (static, an upper bound for gather kernels) "
                                      "over this run's duration - the figure to hold against the 8 TB/s peak")
            if kernel_events:
                for name_, q in kernel_events.items():
                    if name_ in result["kernels"] and q["launches"]:
                        result["kernels"][name_]["us_per_launch_kernel_events"] = round(q["total_ms"] / q["launches"] * 1e3, 2)
            result["gpu_kernel_ms_per_frame"] = round(tot / args.steps, 4)
            # both components of the algorithmic bytes for the whole frame, against the unprofiled frame time
            a_bytes = sum(p["traversal_bytes"] for p in prof) / args.steps
            b_bytes = sum(p["algorithmic_bytes"] - p["traversal_bytes"] for p in prof) / args.steps
            result["frame_bytes"] = {"screen_space_B": round(b_bytes), "traversal_A": round(a_bytes),
                                     "B_GBps": round(b_bytes / (ms * 1e-3) / 1e9, 1),
                                     "B_frac_of_peak": round(b_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "B_frac_of_measured_copy_ceiling": None if not copy_ceiling else round(b_bytes / (ms * 1e-3) / 1e9 / copy_ceiling, 4),
                                     "note": "B = compulsory screen-space plane bytes of the reference's passes (unfused accounting; HBM), A = the reference's used_memory traversal bytes (cache- or LDS-served on these scenes: not HBM traffic, not added to B)"}
            # the frame's REAL traffic: the committed counter passes' bytes per launch x this run's launches per frame. This — not
            # B_frac_of_peak, an unfused-accounting equivalent rate that fusion inflates — is the achieved HBM bandwidth.
            counted, missing = 0.0, []
            for q in prof:
                per_launch, _ = static_traffic([q["name"]], key, run_slots)
                if per_launch is None:
                    missing.append(q["name"])
                else:
                    counted += per_launch * q["launches"] / args.steps
            have_counters = len(missing) < len(prof)
            result["frame_bytes"].update({"counter_GB_per_frame": round(counted / 1e9, 4) if have_counters else None, "counter_GBps": round(counted / (ms * 1e-3) / 1e9, 1) if have_counters else None,
                                          "counter_frac_of_peak": round(counted / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if have_counters else None,
                                          "counter_frac_of_measured_copy_ceiling": None if (not copy_ceiling or not have_counters) else round(counted / (ms * 1e-3) / 1e9 / copy_ceiling, 4),
                                          "counter_source": static_traffic([], key, run_slots)[1],
                                          "counter_slots_without_data": missing,
                                          "counter_note": "sum over slots of (2 x FETCH_SIZE + WRITE_SIZE) per launch from the workload's committed counter summary (static: an earlier rocprofv3 --pmc run of this command; an upper bound for the gather kernels) x launches per frame of THIS run, over THIS run's ms_per_step"})
            result["ms_per_step_with_event_timing"] = round(profiled_ms, 4)
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args, (width, height))
            except Exception as ex:  # the baseline must never sink the GPU number
                result["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
