"""Pin of the CPU oracle against the REAL Strolle ("parity unpinned" until this runs).

The reference cannot be built in this repository's container (Rust + rust-gpu nightly + wgpu + a Vulkan device), so the
dumps have to come from outside: tools/reference_dump/ holds a patch for the reference (deterministic per-pass seeds =
this repository's pass_seed chain, COPY_SRC on every per-camera buffer, `Engine::dump_camera_buffers`), an example program
that renders the Cornell box with it and writes every buffer, and the scene exporter. A maintainer with the reference's
toolchain runs (tools/reference_dump/README.md):

    python tools/reference_dump/export_scene.py /tmp/scene.bin 64 48
    (in the patched reference)  STROLLE_SEED=0 cargo run --release -p strolle --example dump_cornell -- /tmp/scene.bin <repo>/tests/golden/reference 64 48

and this test then compares the oracle with what the reference produced:
  * BVH heatmap: the colour ramp of `used_memory` — equal colours <=> equal integer counts (bvh_heatmap.rs:3-77) — exactly
    (|d| <= 1e-6: the ramp is + - * / only);
  * Reference { depth: 1 }: hits, rays and accumulated colours after frames 1 and 4 within the per-channel tolerance
    1e-3 + 1e-3 |x| on >= 99 % of the channels and PSNR >= 40 dB (sin / cos / acos / pow are driver-precision in the reference);
  * Image (the port casts primary rays where the reference rasterises, DESIGN.md deviation 2, so pixels on silhouettes may
    see another surface; everything downstream of a different ReSTIR sample diverges pixel by pixel, so the resampling
    planes are compared in distribution):
      - G-buffer of every dumped frame: depth within 1e-3 relative and the octahedral normal within 2e-3 on >= 98 % of the
        INTERIOR pixels (pixels whose 3x3 neighbourhood in the oracle is one continuous surface: neither sky nor a depth
        step), metallic / roughness / reflectance bytes equal on >= 98 % of them;
      - reservoir statistics, frames 5..7: the histograms of the DI reservoirs' m and w (DI_RESERVOIRS_0 lanes 0, 1) and of the
        GI reservoirs' m and w (GI_RESERVOIRS_0 lanes 3, 7) over the lit pixels are within a total-variation distance of 0.10
        of the oracle's (a wrong MIS weight or M cap moves these histograms wholesale);
      - mean radiance of the DI and GI sample planes within 5 % over frames 3..7, and the GI : DI ratio of those means within
        5 % of the oracle's — the oracle's ReSTIR GI comes out 4-10 % brighter than one path-traced bounce (DESIGN.md section
        2); if the reference's GI does not, this is the assertion that says so. Both ratios are reported either way.
A dump of the dungeon (export_scene.py ... dungeon -> tests/golden/reference_dungeon) is compared the same way.
Without the dumps the comparison is SKIPPED, loudly; `test_dump_loader_round_trip` keeps the loader and the comparison code
honest by running them on a dump written by the oracle itself in the harness's file format.
"""
import os

import numpy as np
import pytest

from oracle_binding import OracleEngine
from parity import psnr
from strolle_amd import Buffer, CameraMode, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP_DIRS = {"cornell": os.path.join(ROOT, "tests", "golden", "reference"), "dungeon": os.path.join(ROOT, "tests", "golden", "reference_dungeon")}
SEED = 0
SCENES = {"cornell": (scenes.build_cornell, scenes.cornell_camera), "dungeon": (scenes.build_dungeon, scenes.dungeon_camera)}


def load_dump(directory):
    """{(run, frame, buffer name): float32 array}, (width, height), instances ticked — from a dump_cornell output directory."""
    out, size, ticked = {}, None, None
    for line in open(os.path.join(directory, "manifest.txt")):
        fields = line.split()
        run, frame, name, w, h, nbytes = fields[:6]
        data = np.fromfile(os.path.join(directory, f"{run}_f{frame}_{name}.bin"), dtype="<f4")
        assert data.nbytes == int(nbytes), (run, frame, name)
        out[(run, int(frame), name)] = data
        size = (int(w), int(h))
        for f in fields[6:]:
            if f.startswith("instances_ticked="):
                ticked = int(f.split("=")[1])
    return out, size, ticked


def count_instances(scene):
    class Counter:   # the scene builders' engine interface, counting insert_instance calls
        n = 0
        def __getattr__(self, name):
            def call(*a, **k):
                if name == "insert_instance":
                    self.n += 1
            return call
    c = Counter(); SCENES[scene][0](c)
    return c.n


def oracle_runs(size, scene="cornell", instances_ticked=None):
    """The same three runs on the oracle: {(run, frame, buffer name): array}. The harness ticks once per inserted instance
    before it creates the camera (insertion order, see its module comment), which advances the frame counter."""
    build, camera_fn = SCENES[scene]
    if instances_ticked is None:
        instances_ticked = count_instances(scene)
    got = {}
    runs = [("heatmap", CameraMode.BVH_HEATMAP, 0, 1, {1}), ("reference", CameraMode.REFERENCE, 1, 4, {1, 4}), ("image", CameraMode.IMAGE, 0, 7, set(range(1, 8)))]
    for run, mode, depth, frames, dump_at in runs:
        e = OracleEngine()
        build(e); e.set_seed(SEED)
        for _ in range(instances_ticked):
            e.tick()   # frame numbers as in the harness (one tick per inserted instance)
        desc = camera_fn(size, mode, depth=depth)
        cam = e.create_camera(desc)
        for frame in range(1, frames + 1):
            e.update_camera(cam, desc); e.tick(); e.render_camera(cam)
            if frame in dump_at:
                for b in Buffer:
                    if b == Buffer.DBG_USED_MEMORY:
                        continue
                    if (run == "heatmap" and b != Buffer.REF_COLORS) or (run == "reference" and not b.name.startswith("REF_")):
                        continue
                    got[(run, frame, b.name)] = e.read_buffer(cam, b)
        e.close()
    return got


def interior_mask(depth, size):
    """Pixels whose 3x3 neighbourhood is one continuous surface in `depth` ([H*W]): no sky, no depth step above 5 %."""
    w, h = size
    d = depth.reshape(h, w)
    pad = np.pad(d, 1, mode="edge")
    ok = d > 0
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            n = pad[dy:dy + h, dx:dx + w]
            ok &= (n > 0) & (np.abs(n - d) <= 0.05 * np.maximum(d, 1e-6))
    ok[0, :] = ok[-1, :] = False; ok[:, 0] = ok[:, -1] = False
    return ok.reshape(-1)


def tv_distance(a, b, bins):
    """total-variation distance of two samples' histograms over common bins"""
    ha, _ = np.histogram(a, bins=bins); hb, _ = np.histogram(b, bins=bins)
    ha = ha / max(ha.sum(), 1); hb = hb / max(hb.sum(), 1)
    return 0.5 * float(np.abs(ha - hb).sum())


def compare_image(reference, oracle, size, report):
    """The Image-mode criteria of the module docstring."""
    frames = sorted({f for (run, f, name) in oracle if run == "image"})
    for f in frames:
        name = "PRIM_GBUFFER_D0_B" if f % 2 else "PRIM_GBUFFER_D0_A"   # (frame numbers of the run; parity only matters jointly for both sides)
        for cand in (name, "PRIM_GBUFFER_D0_A", "PRIM_GBUFFER_D0_B"):
            r, o = reference[("image", f, cand)].reshape(-1, 4), oracle[("image", f, cand)].reshape(-1, 4)
            if (o[:, 0] != 0).any():
                break
        inside = interior_mask(o[:, 0], size)
        if inside.sum() < 16:
            continue
        depth_ok = np.abs(r[inside, 0] - o[inside, 0]) <= 1e-3 * np.abs(o[inside, 0])
        normal_ok = (np.abs(r[inside, 1] - o[inside, 1]) <= 2e-3) & (np.abs(r[inside, 2] - o[inside, 2]) <= 2e-3)
        bytes_ok = r[inside, 3].view(np.uint32) == o[inside, 3].view(np.uint32)
        report[f"image_f{f}_gbuffer_interior"] = {"pixels": int(inside.sum()), "depth": float(depth_ok.mean()), "normal": float(normal_ok.mean()), "material_bytes": float(bytes_ok.mean())}
        assert depth_ok.mean() >= 0.98 and normal_ok.mean() >= 0.98 and bytes_ok.mean() >= 0.98, (f, report[f"image_f{f}_gbuffer_interior"])
    m_bins = np.concatenate([np.arange(0.0, 33.0, 1.0), [64.0, 129.0, 1e9]])
    w_bins = np.concatenate([[-1e9], np.linspace(0.0, 6.0, 49), [1e9]])
    for f in [f for f in frames if f >= 5]:
        for plane, lanes, (m_lane, w_lane) in (("DI_RESERVOIRS_0", 8, (0, 1)), ("GI_RESERVOIRS_0", 16, (3, 7))):
            r, o = reference[("image", f, plane)].reshape(-1, lanes), oracle[("image", f, plane)].reshape(-1, lanes)
            lit_r, lit_o = r[:, m_lane] > 0, o[:, m_lane] > 0
            report[f"image_f{f}_{plane}_lit_fraction"] = (float(lit_r.mean()), float(lit_o.mean()))
            assert abs(lit_r.mean() - lit_o.mean()) <= 0.03, (f, plane, "share of non-empty reservoirs")
            for what, lane, bins in (("m", m_lane, m_bins), ("w", w_lane, w_bins)):
                tv = tv_distance(r[lit_r, lane], o[lit_o, lane], bins)
                report[f"image_f{f}_{plane}_{what}_tv"] = tv
                assert tv <= 0.10, (f, plane, what, tv)
    means = {}
    for name in ("DI_DIFF_SAMPLES", "GI_DIFF_SAMPLES"):
        r = np.mean([reference[("image", f, name)].reshape(-1, 4)[:, :3].mean() for f in frames if f >= 3])
        o = np.mean([oracle[("image", f, name)].reshape(-1, 4)[:, :3].mean() for f in frames if f >= 3])
        means[name] = (float(r), float(o))
        report[f"image_{name}_mean_ratio"] = float(r / o) if o else float("nan")
        assert abs(r / o - 1.0) <= 0.05, (name, r, o)
    gi_di_ref = means["GI_DIFF_SAMPLES"][0] / means["DI_DIFF_SAMPLES"][0]
    gi_di_orc = means["GI_DIFF_SAMPLES"][1] / means["DI_DIFF_SAMPLES"][1]
    report["image_gi_to_di_ratio"] = {"reference": gi_di_ref, "oracle": gi_di_orc}
    assert abs(gi_di_ref / gi_di_orc - 1.0) <= 0.05, ("GI : DI radiance ratio", gi_di_ref, gi_di_orc)


def compare(reference, oracle, size=None):
    """Raises AssertionError on the first criterion of the module docstring that fails; returns a small report."""
    report = {}
    a, b = reference[("heatmap", 1, "REF_COLORS")], oracle[("heatmap", 1, "REF_COLORS")]
    assert a.shape == b.shape
    report["heatmap_max_abs_diff"] = float(np.abs(a - b).max())
    assert report["heatmap_max_abs_diff"] <= 1e-6, "BVH heatmap colours (= used_memory integers) differ"
    for frame in (1, 4):
        for name in ("REF_HITS", "REF_RAYS", "REF_COLORS"):
            r, o = reference[("reference", frame, name)], oracle[("reference", frame, name)]
            ok = np.abs(r - o) <= 1e-3 + 1e-3 * np.abs(o)
            ok |= np.isnan(r) & np.isnan(o)
            report[f"reference_f{frame}_{name}_within"] = float(ok.mean())
            assert ok.mean() >= 0.99, (frame, name, float(ok.mean()))
        r, o = reference[("reference", frame, "REF_COLORS")].reshape(-1, 4), oracle[("reference", frame, "REF_COLORS")].reshape(-1, 4)
        peak = float(max(np.percentile(o[:, :3] / np.maximum(o[:, 3:], 1), 99.9), 1e-3))
        p = psnr(np.clip(r[:, :3] / np.maximum(r[:, 3:], 1), 0, peak), np.clip(o[:, :3] / np.maximum(o[:, 3:], 1), 0, peak), peak)
        report[f"reference_f{frame}_psnr"] = p
        assert p >= 40.0, (frame, p)
    compare_image(reference, oracle, size, report)
    return report


@pytest.mark.parametrize("scene", ["cornell", "dungeon"])
def test_oracle_matches_the_reference_dump(scene):
    directory = DUMP_DIRS[scene]
    if not os.path.exists(os.path.join(directory, "manifest.txt")):
        pytest.skip(f"NO REFERENCE DUMP: {os.path.relpath(directory, ROOT)}/manifest.txt is absent, so the oracle stays UNPINNED against the real "
                    "Strolle for traversal / shading / resampling. Produce it with tools/reference_dump (needs the reference's Rust toolchain).")
    reference, size, ticked = load_dump(directory)
    print(compare(reference, oracle_runs(size, scene, ticked), size))


def test_dump_loader_round_trip(tmp_path):
    """The harness's file format, the loader and the comparison, exercised end to end with the oracle standing in for the
    reference (24 x 16 pixels keeps it to a few seconds)."""
    size = (24, 16)
    n_instances = count_instances("cornell")
    assert n_instances == 8
    oracle = oracle_runs(size)
    lines = []
    for (run, frame, name), data in oracle.items():
        data.astype("<f4").tofile(tmp_path / f"{run}_f{frame}_{name}.bin")
        lines.append(f"{run} {frame} {name} {size[0]} {size[1]} {data.nbytes} instances_ticked={n_instances}")
    (tmp_path / "manifest.txt").write_text("\n".join(lines) + "\n")
    loaded, got_size, ticked = load_dump(str(tmp_path))
    assert got_size == size and ticked == n_instances and set(loaded) == set(oracle)
    report = compare(loaded, oracle, size)
    assert report["heatmap_max_abs_diff"] == 0.0 and report["reference_f4_psnr"] == float("inf")
    assert report["image_f7_GI_RESERVOIRS_0_m_tv"] == 0.0 and report["image_f7_gbuffer_interior"]["depth"] == 1.0
    # corrupted dumps must fail, each on its own criterion: one wrong heatmap texel ...
    bad = dict(loaded)
    bad[("heatmap", 1, "REF_COLORS")] = loaded[("heatmap", 1, "REF_COLORS")].copy()
    bad[("heatmap", 1, "REF_COLORS")][5] += 0.25
    with pytest.raises(AssertionError):
        compare(bad, oracle, size)
    # ... a G-buffer whose interior depths are 1 % off ...
    bad = dict(loaded)
    for f in range(1, 8):
        for name in ("PRIM_GBUFFER_D0_A", "PRIM_GBUFFER_D0_B"):
            g = loaded[("image", f, name)].copy().reshape(-1, 4); g[:, 0] *= 1.01; bad[("image", f, name)] = g.reshape(-1)
    with pytest.raises(AssertionError):
        compare(bad, oracle, size)
    # ... reservoirs whose sample counts are twice the reference's (a wrong M accounting) ...
    bad = dict(loaded)
    for f in (5, 6, 7):
        g = loaded[("image", f, "GI_RESERVOIRS_0")].copy().reshape(-1, 16); g[:, 3] *= 2.0; bad[("image", f, "GI_RESERVOIRS_0")] = g.reshape(-1)
    with pytest.raises(AssertionError):
        compare(bad, oracle, size)
    # ... and a GI estimate 8 % too bright against the direct one
    bad = dict(loaded)
    for f in range(1, 8):
        g = loaded[("image", f, "GI_DIFF_SAMPLES")].copy(); g *= 1.08; bad[("image", f, "GI_DIFF_SAMPLES")] = g
    with pytest.raises(AssertionError):
        compare(bad, oracle, size)


def test_scene_export_covers_both_benchmark_scenes(tmp_path):
    """tools/reference_dump/export_scene.py: the Cornell box and the dungeon of demo.rs (45 textures, Blend materials, the three
    tori, six lights) serialise into the file dump_cornell.rs reads — counts and sizes as the scene builders give them."""
    import struct, sys
    sys.path.insert(0, os.path.join(ROOT, "tools", "reference_dump"))
    import export_scene
    for scene, want in (("cornell", dict(images=0, materials=8, instances=8, triangles=32, lights=1)), ("dungeon", dict(images=45, materials=48, instances=48, triangles=13001, lights=6))):
        rec, data = export_scene.export(scene, (64, 48))
        got = dict(images=len(rec.images), materials=len(rec.materials), instances=len(rec.instances), triangles=sum(len(rec.meshes[i.mesh_handle].positions) for i in rec.instances), lights=len(rec.lights))
        assert got == want, (scene, got)
        assert data[:4] == b"STSC" and struct.unpack_from("<II", data, 4) == (2, want["images"])
        expected = 12 + sum(8 + rec.images[h].size for h in rec.images) + 4 + want["materials"] * 56 + 4 + want["instances"] * 56 + want["triangles"] * 96 + 4 + want["lights"] * 32 + 8 + 128
        assert len(data) == expected, (scene, len(data), expected)
        assert got["instances"] == count_instances(scene)


@pytest.mark.skipif(not os.path.isdir("/root/reference/strolle"), reason="the reference's sources exist in the build container only")
def test_the_reference_dump_patch_still_applies(tmp_path):
    """tools/reference_dump/strolle_deterministic_dump.patch against the reference as it lies under /root/reference (read-only: a
    scratch copy of the files the patch names is patched, with --dry-run first): the pin harness must stay runnable by a maintainer."""
    import re, shutil, subprocess
    patch = os.path.join(ROOT, "tools", "reference_dump", "strolle_deterministic_dump.patch")
    text = open(patch).read()
    files = sorted(set(re.findall(r"^\+\+\+ b/(\S+)", text, re.M)) | set(re.findall(r"^--- a/(\S+)", text, re.M)))
    assert files, "the patch names no files"
    for f in files:
        src = os.path.join("/root/reference", f)
        dst = tmp_path / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        if os.path.exists(src):
            shutil.copy(src, dst)
    dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert dry.returncode == 0, dry.stdout[-2000:] + dry.stderr[-2000:]
    real = subprocess.run(["patch", "-p1", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert real.returncode == 0 and "mod dump" in open(tmp_path / "strolle" / "src" / "lib.rs").read() and (tmp_path / "strolle" / "src" / "dump.rs").exists()
