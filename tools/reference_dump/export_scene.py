#!/usr/bin/env python3
"""Writes the scene file tools/reference_dump/dump_cornell.rs reads, from the same scene builders this repository's tests
and bench.py use (strolle_amd/scenes.py): the calls `build_cornell` / `build_dungeon` make on an engine are recorded and
serialised, so the reference is handed exactly the meshes, materials, textures, instances (in insertion order), lights,
sun and camera the oracle and the product get.

  python tools/reference_dump/export_scene.py out/scene.bin [width height [cornell|dungeon]]     (default 64 48 cornell)

Layout, version 2 (little endian): "STSC", u32 2;
  u32 n_images    x { u32 width, u32 height, width*height*4 bytes RGBA8 (sRGB) };
  u32 n_materials x { 12 f32 (base_color 4, emissive 4, perceptual_roughness, metallic, reflectance, ior),
                      u32 base_color_texture (0 = none, else 1 + image index), u32 alpha_mode (0 opaque, 1 blend) };
  u32 n_instances x { u32 material index, u32 n_triangles, 12 f32 transform (x, y, z axes, translation),
                      n x 24 f32 (positions 9, normals 9, uvs 6) }   — one mesh per instance, in insertion order;
  u32 n_lights    x 8 f32 (position, radius, colour, range);
  2 f32 sun (azimuth, altitude); 16 + 16 f32 camera transform and projection, column-major."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from strolle_amd import scenes  # noqa: E402


class Recorder:
    """The part of the Engine interface the scene builders use; keeps what it is given."""

    def __init__(self):
        self.images, self.materials, self.meshes, self.instances, self.lights, self.sun = {}, {}, {}, [], {}, (0.0, 0.35)

    def set_blue_noise(self, rgba):
        pass   # the reference embeds its own copy of the same texture (strolle/src/noise.rs:40-50)

    def insert_image(self, handle, rgba, srgb=True):
        assert srgb, "the scene file stores sRGB base-colour textures only"
        self.images[handle] = np.ascontiguousarray(rgba, np.uint8)

    def insert_material(self, handle, m):
        self.materials[handle] = m

    def insert_mesh(self, handle, mesh):
        self.meshes[handle] = mesh

    def insert_instance(self, handle, inst):
        self.instances.append(inst)

    def insert_light(self, handle, light):
        assert light.kind == 0, "point lights only"
        self.lights[handle] = light

    def update_sun(self, sun):
        self.sun = (sun.azimuth, sun.altitude)


def serialise(rec: Recorder, camera) -> bytes:
    buf = bytearray(b"STSC") + struct.pack("<I", 2)
    image_ids = sorted(rec.images)
    buf += struct.pack("<I", len(image_ids))
    for h in image_ids:
        img = rec.images[h]
        buf += struct.pack("<II", img.shape[1], img.shape[0]) + img.tobytes()
    material_ids = sorted(rec.materials)
    buf += struct.pack("<I", len(material_ids))
    for h in material_ids:
        m = rec.materials[h]
        assert not (m.emissive_texture or m.metallic_roughness_texture or m.normal_map_texture), "only base-colour textures are exported"
        buf += struct.pack("<12f", *[float(v) for v in list(m.base_color) + list(m.emissive)], float(m.perceptual_roughness), float(m.metallic), float(m.reflectance), float(m.ior))
        buf += struct.pack("<II", 1 + image_ids.index(m.base_color_texture) if m.base_color_texture else 0, int(m.alpha_mode))
    buf += struct.pack("<I", len(rec.instances))
    for inst in rec.instances:
        mesh = rec.meshes[inst.mesh_handle]
        n = len(mesh.positions)
        buf += struct.pack("<II", material_ids.index(inst.material_handle), n)
        buf += inst.xform12().astype("<f4").tobytes()   # x, y, z axes, translation
        buf += np.concatenate([mesh.positions.reshape(n, 9), mesh.normals.reshape(n, 9), mesh.uvs.reshape(n, 6)], axis=1).astype("<f4").tobytes()
    light_ids = sorted(rec.lights)
    buf += struct.pack("<I", len(light_ids))
    for h in light_ids:
        l = rec.lights[h]
        buf += struct.pack("<8f", *[float(v) for v in l.position], float(l.radius), *[float(v) for v in l.color], float(l.range))
    buf += struct.pack("<2f", *rec.sun)
    cam = camera.to_c()
    buf += struct.pack("<16f", *cam.transform) + struct.pack("<16f", *cam.projection)
    return bytes(buf)


def export(scene: str, size):
    rec = Recorder()
    if scene == "cornell":
        scenes.build_cornell(rec); camera = scenes.cornell_camera(size)
    elif scene == "dungeon":
        scenes.build_dungeon(rec); camera = scenes.dungeon_camera(size)
    else:
        raise SystemExit(f"unknown scene {scene!r}")
    return rec, serialise(rec, camera)


def main():
    out = sys.argv[1]
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 48)
    scene = sys.argv[4] if len(sys.argv) > 4 else "cornell"
    rec, data = export(scene, (w, h))
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    open(out, "wb").write(data)
    print(f"wrote {out}: {len(data)} bytes, {scene}: {len(rec.instances)} instances, {sum(len(rec.meshes[i.mesh_handle].positions) for i in rec.instances)} triangles, "
          f"{len(rec.materials)} materials, {len(rec.images)} images, {len(rec.lights)} lights, camera {w}x{h}")


if __name__ == "__main__":
    main()
