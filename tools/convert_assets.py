#!/usr/bin/env python3
"""Convert the reference's benchmark assets into flat .npz scene files.

Runs in the build container only (it reads /root/reference, which does not
exist on the GPU box); the outputs under assets/ are committed.

  cornell.zip  -> assets/cornell.npz   (32 triangles, 8 materials)
  demo.zip     -> assets/dungeon.npz   (level.glb: triangles, materials, 64x64 textures)
  blue-noise.png -> assets/blue_noise.npy (256x256x4 u8; strolle/src/noise.rs:40-50)

Scene file layout (per mesh i): positions_i [n,3,3], normals_i [n,3,3],
uvs_i [n,3,2] (object space, one row per triangle), xform_i [12] (column-major
3x4 world transform = product of the glTF node matrices, f32), material_i.
Meshes are emitted in glTF node-traversal order; that order fixes triangle ids
(the reference iterates a HashMap there: strolle/src/instances.rs:80).

Cornell credit: "Cornell Box- Original" by t-ly (sketchfab.com/t-ly), CC-BY-4.0.
"""
import io
import json
import struct
import sys
import zipfile

import numpy as np

REF = "/root/reference"
COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def node_matrix(node):
    if "matrix" in node:
        return np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T  # glTF is column-major
    m = np.eye(4)
    t = node.get("translation", [0, 0, 0])
    r = node.get("rotation", [0, 0, 0, 1])
    s = node.get("scale", [1, 1, 1])
    x, y, z, w = r
    rot = np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])
    m[:3, :3] = rot @ np.diag(s)
    m[:3, 3] = t
    return m


def read_accessor(gltf, buffers, idx):
    acc = gltf["accessors"][idx]
    bv = gltf["bufferViews"][acc["bufferView"]]
    dt = np.dtype(COMPONENT[acc["componentType"]])
    nc = NCOMP[acc["type"]]
    start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * nc
    buf = buffers[bv["buffer"]]
    out = np.zeros((acc["count"], nc), dtype=dt)
    for i in range(acc["count"]):
        out[i] = np.frombuffer(buf, dtype=dt, count=nc, offset=start + i * stride)
    return out


def convert_gltf(gltf, buffers, images_rgba):
    out = {}
    mats = []
    for m in gltf.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        tex = pbr.get("baseColorTexture", {}).get("index", None)
        img = -1
        if tex is not None:
            img = gltf["textures"][tex].get("source", -1)
        mats.append(dict(
            base_color=pbr.get("baseColorFactor", [1, 1, 1, 1]),
            metallic=pbr.get("metallicFactor", 1.0),
            perceptual_roughness=pbr.get("roughnessFactor", 1.0),  # bevy_gltf: perceptual_roughness = roughness_factor
            emissive=m.get("emissiveFactor", [0, 0, 0]),
            alpha_mode={"OPAQUE": 0, "MASK": 1, "BLEND": 1}[m.get("alphaMode", "OPAQUE")],
            alpha_cutoff=m.get("alphaCutoff", 0.5),
            is_mask=m.get("alphaMode", "OPAQUE") == "MASK",
            image=img,
        ))
    out["material_base_color"] = np.array([m["base_color"] for m in mats], dtype=np.float32).reshape(-1, 4)
    out["material_metallic"] = np.array([m["metallic"] for m in mats], dtype=np.float32)
    out["material_perceptual_roughness"] = np.array([m["perceptual_roughness"] for m in mats], dtype=np.float32)
    out["material_emissive"] = np.array([m["emissive"] for m in mats], dtype=np.float32).reshape(-1, 3)
    out["material_alpha_mode"] = np.array([m["alpha_mode"] for m in mats], dtype=np.uint32)
    out["material_image"] = np.array([m["image"] for m in mats], dtype=np.int32)
    for i, img in enumerate(images_rgba):
        out[f"image_{i}"] = img

    n_mesh = 0

    def visit(node_idx, parent):
        nonlocal n_mesh
        node = gltf["nodes"][node_idx]
        world = parent @ node_matrix(node)
        if "mesh" in node:
            for prim in gltf["meshes"][node["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue
                pos = read_accessor(gltf, buffers, prim["attributes"]["POSITION"]).astype(np.float32)
                nrm = read_accessor(gltf, buffers, prim["attributes"]["NORMAL"]).astype(np.float32)
                if "TEXCOORD_0" in prim["attributes"]:
                    uv = read_accessor(gltf, buffers, prim["attributes"]["TEXCOORD_0"]).astype(np.float32)
                else:
                    uv = np.zeros((len(pos), 2), dtype=np.float32)
                if "indices" in prim:
                    idx = read_accessor(gltf, buffers, prim["indices"]).astype(np.int64).reshape(-1)
                else:
                    idx = np.arange(len(pos))
                idx = idx[: len(idx) // 3 * 3].reshape(-1, 3)
                i = n_mesh
                out[f"positions_{i}"] = pos[idx]
                out[f"normals_{i}"] = nrm[idx]
                out[f"uvs_{i}"] = uv[idx]
                w32 = world.astype(np.float32)
                out[f"xform_{i}"] = np.concatenate([w32[:3, 0], w32[:3, 1], w32[:3, 2], w32[:3, 3]]).astype(np.float32)
                out[f"material_{i}"] = np.int32(prim.get("material", 0))
                n_mesh += 1
        for c in node.get("children", []):
            visit(c, world)

    scene = gltf["scenes"][gltf.get("scene", 0)]
    for n in scene["nodes"]:
        visit(n, np.eye(4))
    out["n_meshes"] = np.int32(n_mesh)
    out["n_images"] = np.int32(len(images_rgba))
    return out


def main():
    from PIL import Image

    # ---- blue noise
    img = Image.open(f"{REF}/strolle/assets/blue-noise.png").convert("RGBA")
    bn = np.asarray(img, dtype=np.uint8)
    assert bn.shape == (256, 256, 4)
    np.save("assets/blue_noise.npy", bn)

    # ---- cornell
    zf = zipfile.ZipFile(f"{REF}/bevy-strolle/assets/cornell.zip")
    gltf = json.loads(zf.read("cornell/scene.gltf"))
    buffers = [zf.read("cornell/" + b["uri"]) for b in gltf["buffers"]]
    np.savez_compressed("assets/cornell.npz", **convert_gltf(gltf, buffers, []))

    # ---- dungeon (GLB container)
    zf = zipfile.ZipFile(f"{REF}/bevy-strolle/assets/demo.zip")
    glb = zf.read("demo/level.glb")
    magic, version, length = struct.unpack_from("<III", glb, 0)
    assert magic == 0x46546C67
    off = 12
    chunks = []
    while off < length:
        clen, ctype = struct.unpack_from("<II", glb, off)
        chunks.append((ctype, glb[off + 8: off + 8 + clen]))
        off += 8 + clen
    gltf = json.loads(chunks[0][1])
    buffers = [chunks[1][1]]
    images = []
    for im in gltf.get("images", []):
        bv = gltf["bufferViews"][im["bufferView"]]
        raw = buffers[bv["buffer"]][bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]]
        images.append(np.asarray(Image.open(io.BytesIO(raw)).convert("RGBA"), dtype=np.uint8))
    np.savez_compressed("assets/dungeon.npz", **convert_gltf(gltf, buffers, images))
    d = np.load("assets/dungeon.npz")
    ntri = sum(len(d[f"positions_{i}"]) for i in range(int(d["n_meshes"])))
    print("dungeon: meshes", int(d["n_meshes"]), "triangles", ntri, "images", int(d["n_images"]))
    c = np.load("assets/cornell.npz")
    print("cornell: meshes", int(c["n_meshes"]), "triangles", sum(len(c[f"positions_{i}"]) for i in range(int(c["n_meshes"]))))


if __name__ == "__main__":
    sys.exit(main())
