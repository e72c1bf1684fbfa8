#!/usr/bin/env python3
"""How well do two launches of the frame share the chip? Runs on the GPU box.

Two engines with the same scene render on two streams of one device; each enqueues only ONE launch of the whole frame's graph
(st_debug_set_launch_filter: the frame is built with every fusion, one launch of it is enqueued), over and over. For every
pair (X on stream 1, Y on stream 2) the tool reports
    G = work done while both streams were busy / time both streams were busy        (work in units of "seconds when run alone")
G = 1: the pair time-shares the chip, overlapping buys nothing; G = 2: each runs as if alone. The frame's two-stream schedule
(st_render.cpp) gains exactly what its pairs' G allows, which is why this exists.

Launch ordinals of the whole Image frame in serial order: 0 prim_visibility+reprojection, 1 di_sampling+temporal, 2 di_spatial,
3 di_resolving+reproject, 4 / 5 the GI head (sampling then temporal on even tracing frames and on validation frames, temporal then
spatial on odd tracing frames), 6 gi_preview_both, 7 gi_preview late + resolving, 8 variance, 9 wavelets 1+2, 10 wavelet 4,
11 wavelet 8, 12 wavelet 16 + composition.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

UNITS = [  # name, ordinal by frame % 6 (None: the unit does not run on that kind of frame)
    ("prim",        lambda f: 0),
    ("di_sampling", lambda f: 1),
    ("di_spatial",  lambda f: 2),
    ("di_resolve",  lambda f: 3),
    ("gi_sampling", lambda f: 4 if f % 6 in (0, 2, 4, 5) else None),
    ("gi_spatial",  lambda f: 5 if f % 6 in (1, 3) else None),
    ("gi_temporal", lambda f: 5 if f % 6 in (0, 2, 4, 5) else 4),
    ("gi_preview",  lambda f: 6),
    ("gi_late",     lambda f: 7),
    ("variance",    lambda f: 8),
    ("wavelet12",   lambda f: 9),
    ("wavelet4",    lambda f: 10),
    ("wavelet8",    lambda f: 11),
    ("wavelet16c",  lambda f: 12),
]


class Side:
    def __init__(self, torch, scene, size, device=0):
        from strolle_amd import CameraMode, Engine, scenes
        self.torch = torch
        self.engine = Engine(device=device)
        if scene == "cornell":
            scenes.build_cornell(self.engine); self.desc = scenes.cornell_camera(size, CameraMode.IMAGE, depth=1)
        else:
            scenes.build_dungeon(self.engine); self.desc = scenes.dungeon_camera(size, CameraMode.IMAGE, depth=1)
        self.engine.set_seed(1234)
        self.cam = self.engine.create_camera(self.desc)
        self.out = torch.zeros((size[1], size[0], 4), dtype=torch.float32, device=f"cuda:{device}")
        self.stream = torch.cuda.Stream(device=device)
        self.frame = 0
        self.engine.tick(self.stream.cuda_stream)

    def render(self, unit=None):
        """one render call; with a unit only that launch is enqueued. Returns whether a launch was enqueued."""
        if unit is None:
            self.engine.set_launch_filter(~0)
            launched = True
        else:
            ordinal = UNITS[unit][1](self.frame)
            self.engine.set_launch_filter(0 if ordinal is None else 1 << ordinal)
            launched = ordinal is not None
        self.engine.render_camera(self.cam, self.out.data_ptr(), self.stream.cuda_stream)
        self.frame += 1
        return launched

    def launches(self, unit, n):
        """render calls until n launches of the unit are enqueued"""
        done = 0
        while done < n:
            done += 1 if self.render(unit) else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="dungeon"); ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--budget-ms", type=float, default=8.0, help="work per stream and pair, in alone-milliseconds")
    ap.add_argument("--units", default="", help="comma-separated subset of unit names")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    import torch
    size = (a.width, a.height)
    A, B = Side(torch, a.scene, size), Side(torch, a.scene, size)
    for s in (A, B):
        for _ in range(96): s.render()
    torch.cuda.synchronize()
    units = [i for i, u in enumerate(UNITS) if not a.units or u[0] in a.units.split(",")]

    def ev(): return torch.cuda.Event(enable_timing=True)
    alone = {}
    for u in units:   # per-launch time alone (second of two rounds)
        for rnd in range(2):
            e0, e1 = ev(), ev(); n = 12
            A.launches(u, 2); torch.cuda.synchronize()
            e0.record(A.stream); A.launches(u, n); e1.record(A.stream); torch.cuda.synchronize()
            alone[u] = e0.elapsed_time(e1) / n
    print(f"{a.scene} {a.width}x{a.height}: alone, us per launch: " + ", ".join(f"{UNITS[u][0]} {alone[u] * 1e3:.0f}" for u in units), flush=True)

    G = {}
    for x in units:
        for y in units:
            if y < x: continue
            nx = max(2, round(a.budget_ms / alone[x])); ny = max(2, round(a.budget_ms / alone[y]))
            A.launches(x, 1); B.launches(y, 1); torch.cuda.synchronize()
            s0, ax, by = ev(), ev(), ev()
            s0.record(A.stream); B.stream.wait_event(s0)   # common start
            ix = iy = 0
            while ix < nx or iy < ny:   # interleaved so that neither queue runs dry
                if ix < nx and ix * ny <= iy * nx: A.launches(x, 1); ix += 1
                elif iy < ny: B.launches(y, 1); iy += 1
                else: A.launches(x, 1); ix += 1
            ax.record(A.stream); by.record(B.stream); torch.cuda.synchronize()
            tx, ty = s0.elapsed_time(ax), s0.elapsed_time(by)
            wx, wy = nx * alone[x], ny * alone[y]
            both = min(tx, ty); tail = abs(tx - ty)   # the stream that finishes last runs its tail alone, at full speed
            g = (wx + wy - tail) / both
            G[(x, y)] = g
    names = [UNITS[u][0] for u in units]
    w = max(len(n) for n in names) + 1
    print(" " * w + " ".join(f"{n[:7]:>7s}" for n in names))
    for x in units:
        print(f"{UNITS[x][0]:<{w}s}" + " ".join(f"{G[(min(x, y), max(x, y))]:7.2f}" for y in units))
    if a.json:
        json.dump({"scene": a.scene, "size": size, "alone_us": {UNITS[u][0]: alone[u] * 1e3 for u in units},
                   "G": {f"{UNITS[x][0]}|{UNITS[y][0]}": g for (x, y), g in G.items()}}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
