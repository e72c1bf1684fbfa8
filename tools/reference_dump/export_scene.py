#!/usr/bin/env python3
"""Writes the scene file tools/reference_dump/dump_cornell.rs reads, from the same inputs this repository's tests use
(strolle_amd/scenes.py: assets/cornell.npz, the point light of cornell.rs:45-54 at t = 0, the camera of cornell.rs:76-78).

  python tools/reference_dump/export_scene.py out/scene.bin [width height]     (camera aspect; default 64 48)

Layout (little endian): "STSC", u32 1; u32 n_materials x 12 f32 (base_color, emissive, perceptual_roughness, metallic,
reflectance, ior); u32 n_instances x { u32 material index, u32 n_triangles, 12 f32 transform (x, y, z axes, translation),
n x 24 f32 (positions 9, normals 9, uvs 6) }; u32 n_lights x 8 f32 (position, radius, colour, range);
2 f32 sun (azimuth, altitude); 16 + 16 f32 camera transform and projection, column-major."""
import math
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from strolle_amd import scenes  # noqa: E402


def main():
    out = sys.argv[1]
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 48)
    npz = np.load(os.path.join(scenes.ASSETS, "cornell.npz"))
    buf = bytearray(b"STSC") + struct.pack("<I", 1)
    n_mat = len(npz["material_metallic"])
    buf += struct.pack("<I", n_mat)
    for i in range(n_mat):
        vals = list(npz["material_base_color"][i]) + list(npz["material_emissive"][i]) + [1.0] + [float(npz["material_perceptual_roughness"][i]), float(npz["material_metallic"][i]), 0.5, 1.0]
        buf += struct.pack("<12f", *[float(v) for v in vals])
    n = int(npz["n_meshes"])
    buf += struct.pack("<I", n)
    for i in range(n):
        pos = np.asarray(npz[f"positions_{i}"], np.float32).reshape(-1, 3, 3)
        nrm = np.asarray(npz[f"normals_{i}"], np.float32).reshape(-1, 3, 3)
        uv = np.asarray(npz[f"uvs_{i}"], np.float32).reshape(-1, 3, 2)
        buf += struct.pack("<II", int(npz[f"material_{i}"]), len(pos))
        buf += np.asarray(npz[f"xform_{i}"], np.float32).reshape(12).tobytes()   # rows of the 4x3 array = x, y, z axes, translation
        buf += np.concatenate([pos.reshape(len(pos), 9), nrm.reshape(len(pos), 9), uv.reshape(len(pos), 6)], axis=1).astype("<f4").tobytes()
    intensity = 50.0 / (4.0 * math.pi)   # extract.rs:285-297
    buf += struct.pack("<I", 1) + struct.pack("<8f", 0.0, 1.5, 0.5, 0.15, intensity, intensity, intensity, 20.0)
    buf += struct.pack("<2f", 0.0, -1.0)   # cornell.rs:87
    cam = scenes.cornell_camera((w, h)).to_c()
    buf += struct.pack("<16f", *cam.transform) + struct.pack("<16f", *cam.projection)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    open(out, "wb").write(buf)
    print(f"wrote {out}: {len(buf)} bytes, {n} instances, camera {w}x{h}")


if __name__ == "__main__":
    main()
