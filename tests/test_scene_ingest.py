"""Scene ingest above the C ABI (st_scene_load_gltf / st_decode_png). Host-only: no GPU needed.

The loader is checked three ways: (1) the PNG decoder against an independent encoder written here (every colour type,
bit depth, filter type and Adam7) and against PIL where it is installed; (2) a synthetic glTF exercising the features
the benchmark assets do not (TRS + matrix hierarchies, strided and normalised accessors, u8/u16/u32 indices, missing
normals, Mask/Blend materials, the default material, data URIs, external files, skipped primitive modes) against the
same scene inserted through the Python API; (3) in the build container only, the reference's own benchmark assets
against the committed .npz conversions of them."""
import base64
import json
import os
import struct
import zipfile
import zlib

import numpy as np
import pytest

from parity import assert_bits_equal
from strolle_amd import Engine, Instance, Material, Mesh, StrolleError, scenes
from strolle_amd.api import decode_png

REFERENCE = "/root/reference/bevy-strolle/assets"


# ------------------------------------------------------------------------------------------------ a PNG encoder for the tests
def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _filter_rows(rows, bpp, choose):
    """rows: list of bytes (unfiltered scanlines of one (sub)image); choose(y) -> filter type."""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        f = choose(y)
        out.append(f)
        for k, v in enumerate(row):
            a = row[k - bpp] if k >= bpp else 0
            b = prev[k]
            c = prev[k - bpp] if k >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[f]
            out.append((v - pred) & 255)
        prev = row
    return bytes(out)


def _pack_row(samples, depth):
    """samples: 1-D array of sample values for one scanline -> packed bytes (MSB first, 16-bit big endian)."""
    if depth == 8:
        return bytes(samples.astype(np.uint8))
    if depth == 16:
        return samples.astype(">u2").tobytes()
    per = 8 // depth
    pad = (-len(samples)) % per
    s = np.concatenate([samples, np.zeros(pad, samples.dtype)]).reshape(-1, per).astype(np.uint32)
    shifts = np.arange(per - 1, -1, -1, dtype=np.uint32) * depth
    return bytes((s << shifts).sum(axis=1).astype(np.uint8))


def encode_png(samples, ctype, depth, plte=None, trns=None, interlace=False, choose=lambda y: y % 5, idat_split=0):
    """samples: [h, w, channels] integer array of full-depth sample values (palette indices for ctype 3)."""
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)] if interlace else [(0, 0, 1, 1)]
    raw = b""
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = [_pack_row(sub[y].reshape(-1), depth) for y in range(sub.shape[0])]
        raw += _filter_rows(rows, bpp, choose)

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)

    z = zlib.compress(raw, 6)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        out += chunk(b"PLTE", bytes(np.asarray(plte, np.uint8).reshape(-1)))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    out += chunk(b"tEXt", b"Comment\x00made by tests/test_scene_ingest.py")
    if idat_split:
        for i in range(0, len(z), idat_split):
            out += chunk(b"IDAT", z[i:i + idat_split])
    else:
        out += chunk(b"IDAT", z)
    return out + chunk(b"IEND", b"")


def expected_rgba(samples, ctype, depth, plte=None, trns=None):
    h, w, ch = samples.shape
    s = samples.astype(np.uint32)
    if depth == 16:
        to8 = lambda v: (v >> 8).astype(np.uint8)
    elif depth == 8:
        to8 = lambda v: v.astype(np.uint8)
    else:
        to8 = lambda v: (v * 255 // ((1 << depth) - 1)).astype(np.uint8)
    out = np.zeros((h, w, 4), np.uint8)
    if ctype == 3:
        pal = np.asarray(plte, np.uint8).reshape(-1, 3)
        alpha = np.full(len(pal), 255, np.uint8)
        if trns is not None:
            alpha[:len(trns)] = np.frombuffer(bytes(trns), np.uint8)
        out[..., :3] = pal[s[..., 0]]
        out[..., 3] = alpha[s[..., 0]]
    elif ctype in (0, 4):
        out[..., 0] = out[..., 1] = out[..., 2] = to8(s[..., 0])
        if ctype == 4:
            out[..., 3] = to8(s[..., 1])
        else:
            out[..., 3] = 255
            if trns is not None:
                key = struct.unpack(">H", bytes(trns)[:2])[0]
                out[..., 3][s[..., 0] == key] = 0
    else:
        for c in range(3):
            out[..., c] = to8(s[..., c])
        if ctype == 6:
            out[..., 3] = to8(s[..., 3])
        else:
            out[..., 3] = 255
            if trns is not None:
                key = struct.unpack(">HHH", bytes(trns)[:6])
                out[..., 3][(s[..., 0] == key[0]) & (s[..., 1] == key[1]) & (s[..., 2] == key[2])] = 0
    return out


PNG_CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("ctype,depth", PNG_CASES)
def test_png_decoder_every_colour_type_depth_filter(ctype, depth, interlace):
    rng = np.random.default_rng(ctype * 100 + depth + (1000 if interlace else 0))
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    for (w, h) in ((1, 1), (5, 3), (13, 11), (64, 64), (33, 9)):
        n_pal = min(1 << depth, 200)
        samples = rng.integers(0, n_pal if ctype == 3 else (1 << depth), (h, w, channels))
        plte = rng.integers(0, 256, (n_pal, 3)) if ctype == 3 else None
        trns = None
        if ctype == 3:
            trns = bytes(rng.integers(0, 256, n_pal // 2 + 1).astype(np.uint8))
        elif ctype == 0:
            trns = struct.pack(">H", int(samples[0, 0, 0]))
        elif ctype == 2:
            trns = struct.pack(">HHH", *[int(v) for v in samples[h // 2, w // 2]])
        png = encode_png(samples, ctype, depth, plte, trns, interlace, idat_split=7 if w == 13 else 0)
        got = decode_png(png)
        want = expected_rgba(samples, ctype, depth, plte, trns)
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"ctype {ctype} depth {depth} {w}x{h} interlace {interlace}"


def test_png_decoder_matches_pil():
    Image = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(4)
    for mode, shape in (("RGBA", (37, 21, 4)), ("RGB", (64, 64, 3)), ("L", (17, 40)), ("LA", (9, 9, 2)), ("P", (50, 31))):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        # smooth gradient mixed in so that the encoder's adaptive filtering picks different filters per row
        a = ((a.astype(np.int32) // 8) + np.arange(shape[1])[None, :].reshape((1, shape[1]) + (1,) * (len(shape) - 2)) * 3).astype(np.uint8)
        img = Image.fromarray(a, mode="L" if mode == "P" else mode)
        if mode == "P":
            img = img.convert("P", palette=Image.ADAPTIVE, colors=64)
        buf = io.BytesIO()
        img.save(buf, format="PNG", optimize=True)
        want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"), np.uint8)
        assert np.array_equal(decode_png(buf.getvalue()), want), mode


def test_png_decoder_rejects_damage_without_crashing():
    rng = np.random.default_rng(9)
    samples = rng.integers(0, 256, (24, 31, 4))
    good = encode_png(samples, 6, 8)
    assert np.array_equal(decode_png(good), expected_rgba(samples, 6, 8))
    with pytest.raises(StrolleError, match="signature"):
        decode_png(b"not a png at all")
    with pytest.raises(StrolleError):
        decode_png(good[:len(good) // 2])
    flipped = bytearray(good); flipped[60] ^= 0x10
    with pytest.raises(StrolleError, match="CRC"):
        decode_png(bytes(flipped))
    # random damage: every outcome is fine except a crash or a hang (CRCs recomputed so that the damage reaches the inflater)
    def rechunk(data):
        out, off = bytearray(data[:8]), 8
        while off + 12 <= len(data):
            n = struct.unpack(">I", data[off:off + 4])[0]
            body = data[off + 4:off + 8 + n]
            out += data[off:off + 4] + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)
            off += 12 + n
        return bytes(out)
    for i in range(300):
        d = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            d[int(rng.integers(8, len(d)))] = int(rng.integers(0, 256))
        try:
            decode_png(rechunk(bytes(d)))
        except StrolleError:
            pass


# ------------------------------------------------------------------------------------------------ synthetic glTF
def _quat(axis, angle):
    axis = np.asarray(axis, np.float64); axis /= np.linalg.norm(axis)
    return [*(axis * np.sin(angle / 2)).tolist(), float(np.cos(angle / 2))]


def _node_matrix(node):
    """The composition the loader documents: f64, M = T * R * S, or the column-major `matrix`."""
    if "matrix" in node:
        return np.array(node["matrix"], np.float64).reshape(4, 4).T
    x, y, z, w = node.get("rotation", [0, 0, 0, 1])
    rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                    [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                    [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4)
    m[:3, :3] = rot * np.array(node.get("scale", [1, 1, 1]), np.float64)[None, :]
    m[:3, 3] = node.get("translation", [0, 0, 0])
    return m


def _matmul(a, b):
    """Plain triple loop: the loader sums k = 0..3 in order, without fused multiply-adds."""
    out = np.zeros((4, 4))
    for i in range(4):
        for j in range(4):
            s = a[i, 0] * b[0, j]
            for k in range(1, 4):
                s = s + a[i, k] * b[k, j]
            out[i, j] = s
    return out


class SyntheticScene:
    """Builds a glTF document + binary buffer, and the list of engine calls the loader is expected to make."""

    def __init__(self, seed):
        rng = np.random.default_rng(seed)
        self.bin = bytearray()
        self.views, self.accessors = [], []
        textures_rgba = [rng.integers(0, 256, (8, 8, 4)), rng.integers(0, 256, (5, 16, 4))]
        self.pngs = [encode_png(t, 6, 8) for t in textures_rgba]
        self.textures_rgba = [t.astype(np.uint8) for t in textures_rgba]

        def view(data, stride=None):
            while len(self.bin) % 4:
                self.bin.append(0)
            v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
            if stride:
                v["byteStride"] = stride
            self.bin += data
            self.views.append(v)
            return len(self.views) - 1

        def accessor(view_index, component, count, kind, offset=0, normalized=False):
            a = {"bufferView": view_index, "componentType": component, "count": count, "type": kind}
            if offset:
                a["byteOffset"] = offset
            if normalized:
                a["normalized"] = True
            self.accessors.append(a)
            return len(self.accessors) - 1

        # mesh 0: interleaved (stride 32: pos 12, normal 12, uv 8), u16 indices
        nv = 12
        pos0 = rng.uniform(-1, 1, (nv, 3)).astype(np.float32)
        nrm0 = rng.standard_normal((nv, 3)).astype(np.float32)
        uv0 = rng.uniform(0, 1, (nv, 2)).astype(np.float32)
        inter = np.concatenate([pos0, nrm0, uv0], axis=1).astype(np.float32)
        v_inter = view(inter.tobytes(), stride=32)
        idx0 = rng.integers(0, nv, 8 * 3).astype(np.uint16)
        a_idx0 = accessor(view(idx0.tobytes()), 5123, len(idx0), "SCALAR")
        prim0 = {"attributes": {"POSITION": accessor(v_inter, 5126, nv, "VEC3"), "NORMAL": accessor(v_inter, 5126, nv, "VEC3", 12),
                                "TEXCOORD_0": accessor(v_inter, 5126, nv, "VEC2", 24)}, "indices": a_idx0, "material": 1}
        # mesh 0, second primitive: no normals, no indices, normalised u16 UVs, tangents; 7 vertices -> 2 triangles
        pos1 = rng.uniform(-2, 2, (7, 3)).astype(np.float32)
        uv1 = rng.integers(0, 65536, (7, 2)).astype(np.uint16)
        tan1 = rng.standard_normal((7, 4)).astype(np.float32)
        prim1 = {"attributes": {"POSITION": accessor(view(pos1.tobytes()), 5126, 7, "VEC3"),
                                "TEXCOORD_0": accessor(view(uv1.tobytes()), 5123, 7, "VEC2", normalized=True),
                                "TANGENT": accessor(view(tan1.tobytes()), 5126, 7, "VEC4")}, "material": 0}
        # a line list: skipped
        prim_lines = {"attributes": {"POSITION": prim1["attributes"]["POSITION"]}, "mode": 1}
        # mesh 1: u8 indices, u32 indices, no material (default material)
        pos2 = rng.uniform(-1, 1, (6, 3)).astype(np.float32)
        nrm2 = rng.standard_normal((6, 3)).astype(np.float32)
        a_pos2, a_nrm2 = accessor(view(pos2.tobytes()), 5126, 6, "VEC3"), accessor(view(nrm2.tobytes()), 5126, 6, "VEC3")
        idx2 = rng.integers(0, 6, 4 * 3).astype(np.uint8)
        idx3 = rng.integers(0, 6, 3 * 3).astype(np.uint32)
        prim2 = {"attributes": {"POSITION": a_pos2, "NORMAL": a_nrm2}, "indices": accessor(view(idx2.tobytes()), 5121, len(idx2), "SCALAR"), "material": 2}
        prim3 = {"attributes": {"POSITION": a_pos2, "NORMAL": a_nrm2}, "indices": accessor(view(idx3.tobytes()), 5125, len(idx3), "SCALAR")}
        v_img0 = view(self.pngs[0])

        self.materials = [
            {"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.7, 0.6, 0.4], "metallicFactor": 0.25, "roughnessFactor": 0.75, "baseColorTexture": {"index": 0}},
             "emissiveFactor": [0.5, 0.25, 2.0], "alphaMode": "BLEND"},
            {"pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.2, 0.3, 0.45], "baseColorTexture": {"index": 1}}, "alphaMode": "MASK", "alphaCutoff": 0.5,
             "normalTexture": {"index": 0}},
            {"pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.9, 0.1, 0.3]}, "alphaMode": "MASK", "alphaCutoff": 0.25, "emissiveTexture": {"index": 1}},
        ]
        self.nodes = [
            {"children": [1, 3], "matrix": [1.0, 0.0, 0.0, 0.0, 0.0, 2.220446049250313e-16, -1.0, 0.0, 0.0, 1.0, 2.220446049250313e-16, 0.0, 0.5, -0.25, 3.0, 1.0]},
            {"children": [2], "rotation": _quat([1, 2, 3], 0.7), "scale": [0.04, 1.5, 2.0], "translation": [-2.47702, -14.1602, 0.02125]},
            {"mesh": 0, "rotation": _quat([0, 1, 0], -1.1), "translation": [0.1, 0.2, 0.3]},
            {"mesh": 1, "scale": [2.0, 2.0, 2.0], "children": [5, 6, 7, 8]},
            {"mesh": 1, "translation": [9.0, 9.0, 9.0], "extensions": {"KHR_lights_punctual": {"light": 0}}},   # not reachable from the scene: must not be loaded
            {"translation": [0.5, 2.0, -1.0], "extensions": {"KHR_lights_punctual": {"light": 0}}},
            {"translation": [-1.0, 3.0, 0.25], "rotation": _quat([1, 0.2, 0], 1.1), "extensions": {"KHR_lights_punctual": {"light": 1}}},
            {"extensions": {"KHR_lights_punctual": {"light": 2}}},                                           # directional: skipped
            {"translation": [1.0, 1.0, 1.0], "extensions": {"KHR_lights_punctual": {"light": 3}}},          # too faint: skipped
        ]
        self.lights = [{"type": "point", "color": [1.0, 0.5, 0.25], "intensity": 40.0, "range": 12.5},
                       {"type": "spot", "intensity": 300.0, "spot": {"innerConeAngle": 0.2, "outerConeAngle": 0.6}},
                       {"type": "directional", "intensity": 3.0},
                       {"type": "point", "intensity": 0.00005}]
        self.doc = {
            "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": self.nodes,
            "meshes": [{"primitives": [prim0, prim1, prim_lines]}, {"primitives": [prim2, prim3]}],
            "materials": self.materials, "accessors": self.accessors, "bufferViews": self.views,
            "textures": [{"source": 0}, {"source": 1}],
            "extensionsUsed": ["KHR_lights_punctual"], "extensions": {"KHR_lights_punctual": {"lights": self.lights}},
            "images": [{"bufferView": v_img0, "mimeType": "image/png"}, {"uri": "data:image/png;base64," + base64.b64encode(self.pngs[1]).decode()}],
        }
        # ---- what the loader is expected to hand to the engine
        uv1f = uv1.astype(np.float32) / np.float32(65535.0)
        e1, e2 = pos1[[1, 4]] - pos1[[0, 3]], pos1[[2, 5]] - pos1[[0, 3]]
        flat = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2], e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], 1).astype(np.float32)
        flat = flat / np.sqrt((flat[:, 0] * flat[:, 0] + flat[:, 1] * flat[:, 1]) + flat[:, 2] * flat[:, 2])[:, None]
        i0 = idx0.astype(np.int64).reshape(-1, 3)
        self.expected_meshes = [
            (Mesh(pos0[i0], nrm0[i0], uv0[i0]), 1, [0, 1, 2]),
            (Mesh(pos1[:6].reshape(2, 3, 3), np.repeat(flat[:, None, :], 3, axis=1), uv1f[:6].reshape(2, 3, 2), tan1[:6].reshape(2, 3, 4)), 0, [0, 1, 2]),
            (Mesh(pos2[idx2.astype(np.int64).reshape(-1, 3)], nrm2[idx2.astype(np.int64).reshape(-1, 3)]), 2, [0, 3]),
            (Mesh(pos2[idx3.astype(np.int64).reshape(-1, 3)], nrm2[idx3.astype(np.int64).reshape(-1, 3)]), 3, [0, 3]),
        ]

    def write(self, directory, glb):
        doc = dict(self.doc)
        if glb:
            doc["buffers"] = [{"byteLength": len(self.bin)}]
            js = json.dumps(doc).encode()
            js += b" " * ((-len(js)) % 4)
            binary = bytes(self.bin) + b"\0" * ((-len(self.bin)) % 4)
            body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binary), 0x004E4942) + binary
            data = struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body
            path = os.path.join(directory, "scene.glb")
        else:
            doc["buffers"] = [{"byteLength": len(self.bin), "uri": "scene%20data.bin"}]
            with open(os.path.join(directory, "scene data.bin"), "wb") as f:
                f.write(bytes(self.bin))
            data = json.dumps(doc, indent=1).encode()
            path = os.path.join(directory, "scene.gltf")
        with open(path, "wb") as f:
            f.write(data)
        return path, data

    def insert_expected(self, engine, first_handle=1, first_image=1000, subdivide=0):
        for i, rgba in enumerate(self.textures_rgba):
            engine.insert_image(first_image + i, rgba, srgb=True)
        f32 = lambda v: float(np.float32(v))
        mats = [
            Material(base_color=[f32(0.8), f32(0.7), f32(0.6), f32(0.4)], metallic=0.25, perceptual_roughness=0.75, emissive=[0.5, 0.25, 2.0, 1.0], alpha_mode=1,
                     base_color_texture=first_image, reflectance=0.5, ior=1.0),
            Material(base_color=[f32(0.1), f32(0.2), f32(0.3), 0.0], metallic=1.0, perceptual_roughness=1.0, emissive=[0, 0, 0, 1.0], alpha_mode=1,
                     base_color_texture=first_image + 1, normal_map_texture=first_image),
            Material(base_color=[f32(0.9), f32(0.9), f32(0.1), 1.0], metallic=1.0, perceptual_roughness=1.0, emissive=[0, 0, 0, 1.0], alpha_mode=1, emissive_texture=first_image + 1),
            Material(base_color=[1.0, 1.0, 1.0, 1.0], metallic=1.0, perceptual_roughness=1.0, emissive=[0, 0, 0, 1.0], alpha_mode=0),   # glTF default material
        ]
        for i, m in enumerate(mats):
            engine.insert_material(first_handle + i, m)
        from strolle_amd import Light
        chain_to = {5: [0, 3, 5], 6: [0, 3, 6]}
        for k, (node, light) in enumerate(((5, self.lights[0]), (6, self.lights[1]))):
            world = np.eye(4)
            for n in chain_to[node]:
                world = _matmul(world, _node_matrix(self.nodes[n]))
            position = [float(np.float32(world[i, 3])) for i in range(3)]
            color = [float(np.float32(np.float32(c) * np.float32(light["intensity"]))) for c in light.get("color", [1.0, 1.0, 1.0])]
            if light["type"] == "point":
                engine.insert_light(first_handle + k, Light.point(position, 0.0, color, light.get("range", 20.0)))
            else:
                d = -world[:3, 2]; d = d / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
                engine.insert_light(first_handle + k, Light.spot(position, 0.0, color, light.get("range", 20.0), [float(np.float32(v)) for v in d], light["spot"]["outerConeAngle"]))
        for i, (mesh, material, chain) in enumerate(self.expected_meshes):
            world = np.eye(4)
            for n in chain:
                world = _matmul(world, _node_matrix(self.nodes[n]))
            if subdivide:
                pos, nrm, uv = scenes._subdivide(mesh.positions, mesh.normals, mesh.uvs, subdivide)
                mesh = Mesh(pos, nrm, uv)
            engine.insert_mesh(first_handle + i, mesh)
            engine.insert_instance(first_handle + i, Instance(first_handle + i, first_handle + material, world[:3, :].astype(np.float32)))


@pytest.mark.parametrize("container", ["gltf", "glb", "memory"])
def test_loader_equals_the_same_scene_inserted_through_the_api(tmp_path, container):
    scene = SyntheticScene(seed=21)
    path, data = scene.write(str(tmp_path), glb=container != "gltf")
    a, b = Engine(device=-1), Engine(device=-1)
    summary = a.load_gltf(data if container == "memory" else path, first_handle=7, first_image_handle=300)
    assert summary == dict(meshes=4, triangles=8 + 2 + 4 + 3, materials=4, images=2, images_dropped=0, primitives_skipped=1, lights=2, lights_skipped=2)
    scene.insert_expected(b, first_handle=7, first_image=300)
    for e in (a, b):
        e.tick()
    for what, name in enumerate(("BVH stream", "triangles", "lights", "materials")):
        assert_bits_equal(a.read_scene(what), b.read_scene(what), f"{container}: {name}")


def test_loader_subdivision_matches_the_python_route(tmp_path):
    scene = SyntheticScene(seed=5)
    # tangents are not subdivided by scenes._subdivide: compare on the primitives without them
    del scene.doc["meshes"][0]["primitives"][1]["attributes"]["TANGENT"]
    m = scene.expected_meshes[1][0]
    scene.expected_meshes[1] = (Mesh(m.positions, m.normals, m.uvs), 0, [0, 1, 2])
    path, _ = scene.write(str(tmp_path), glb=True)
    a, b = Engine(device=-1), Engine(device=-1)
    assert a.load_gltf(path, subdivide=2)["triangles"] == 17 * 16
    scene.insert_expected(b, subdivide=2)
    for e in (a, b):
        e.tick()
    for what in range(4):
        assert_bits_equal(a.read_scene(what), b.read_scene(what), f"subdivided scene buffer {what}")


def test_loader_reports_errors_instead_of_guessing(tmp_path):
    e = Engine(device=-1)
    with pytest.raises(StrolleError, match="status 7"):
        e.load_gltf(str(tmp_path / "missing.glb"))
    with pytest.raises(StrolleError, match="status 8"):
        e.load_gltf(b'{"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [')
    scene = SyntheticScene(seed=2)
    _, good = scene.write(str(tmp_path), glb=True)
    with pytest.raises(StrolleError, match="status 8"):
        e.load_gltf(good[:len(good) - 40])
    # an index that points outside the vertex array
    doc = json.loads(json.dumps(scene.doc))
    doc["accessors"][doc["meshes"][1]["primitives"][0]["attributes"]["POSITION"]]["count"] = 3
    doc["accessors"][doc["meshes"][1]["primitives"][0]["attributes"]["NORMAL"]]["count"] = 3
    doc["buffers"] = [{"byteLength": len(scene.bin), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(scene.bin)).decode()}]
    with pytest.raises(StrolleError, match="out of range"):
        e.load_gltf(json.dumps(doc).encode())
    # texture formats the loader does not read are named, not skipped
    doc = json.loads(json.dumps(scene.doc))
    doc["buffers"] = [{"byteLength": len(scene.bin), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(scene.bin)).decode()}]
    doc["images"][1] = {"uri": "data:image/webp;base64," + base64.b64encode(b"RIFF\x24\x00\x00\x00WEBPVP8 " + bytes(24)).decode()}
    with pytest.raises(StrolleError, match="status 9.*glTF image 1.*neither a PNG nor a JPEG"):
        e.load_gltf(json.dumps(doc).encode())
    doc["images"][1] = {"uri": "data:image/jpeg;base64," + base64.b64encode(b"\xff\xd8\xff\xe0" + bytes(32)).decode()}   # a JPEG in name only
    with pytest.raises(StrolleError, match="status 8.*JPEG"):
        e.load_gltf(json.dumps(doc).encode())
    doc = json.loads(json.dumps(scene.doc))
    doc["extensionsRequired"] = ["KHR_draco_mesh_compression"]
    with pytest.raises(StrolleError, match="status 9"):
        e.load_gltf(json.dumps(doc).encode())
    # a cycle in the node graph ends in an error, not in a stack overflow
    doc = json.loads(json.dumps(scene.doc))
    doc["buffers"] = [{"byteLength": len(scene.bin), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(scene.bin)).decode()}]
    doc["nodes"][4]["children"] = [0]
    doc["nodes"][3]["children"] = [4]
    doc["meshes"] = [{"primitives": []}, {"primitives": []}]
    with pytest.raises(StrolleError, match="cycle"):
        e.load_gltf(json.dumps(doc).encode())


def test_loader_survives_random_damage(tmp_path):
    """Every outcome is acceptable except a crash: bounds are checked before any read."""
    scene = SyntheticScene(seed=3)
    _, good = scene.write(str(tmp_path), glb=True)
    json_len = struct.unpack("<I", good[12:16])[0]
    rng = np.random.default_rng(17)
    outcomes = {"ok": 0, "error": 0}
    for i in range(400):
        d = bytearray(good)
        lo, hi = (20, 20 + json_len) if i % 2 == 0 else (20 + json_len + 8, len(d))
        for _ in range(int(rng.integers(1, 4))):
            d[int(rng.integers(lo, hi))] = int(rng.integers(32, 127)) if i % 2 == 0 else int(rng.integers(0, 256))
        e = Engine(device=-1)
        try:
            e.load_gltf(bytes(d))
            e.tick()
            outcomes["ok"] += 1
        except StrolleError:
            outcomes["error"] += 1
        e.close()
    assert outcomes["ok"] and outcomes["error"], outcomes


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference's assets exist in the build container only")
@pytest.mark.parametrize("name", ["cornell", "dungeon", "dungeon-subdivided"])
def test_reference_assets_load_to_the_committed_conversions(tmp_path, name):
    """assets/*.npz were converted from these files by tools/convert_assets.py; the C++ loader has to arrive at the same
    engine state from the original files, and its PNG decoder at the same texels as PIL did for the conversion."""
    if name == "cornell":
        zipfile.ZipFile(os.path.join(REFERENCE, "cornell.zip")).extractall(tmp_path)
        path, kw, npz, overrides = str(tmp_path / "cornell" / "scene.gltf"), {}, "cornell.npz", None
    else:
        zipfile.ZipFile(os.path.join(REFERENCE, "demo.zip")).extractall(tmp_path)
        kw = dict(reflectance=0.0, perceptual_roughness=1.0, subdivide=1 if "sub" in name else 0)
        path, npz, overrides = str(tmp_path / "demo" / "level.glb"), "dungeon.npz", dict(reflectance=0.0, perceptual_roughness=1.0)
    converted = np.load(os.path.join(scenes.ASSETS, npz))
    a, b = Engine(device=-1), Engine(device=-1)
    summary = a.load_gltf(path, **kw)
    scenes._insert_gltf(b, converted, material_overrides=overrides, subdivide=kw.get("subdivide", 0))
    for e in (a, b):
        e.tick()   # (the subdivided level is 25 internal nodes deep: the launches take a 25-entry stack, no ST_ERR_BVH_TOO_DEEP)
    for what, label in enumerate(("BVH stream", "triangles", "lights", "materials")):
        assert_bits_equal(a.read_scene(what), b.read_scene(what), f"{name}: {label}")
    assert summary["meshes"] == int(converted["n_meshes"]) and summary["images"] == int(converted["n_images"])
    if name == "dungeon":
        glb = open(path, "rb").read()
        doc = json.loads(glb[20:20 + struct.unpack("<I", glb[12:16])[0]])
        binary = glb[20 + struct.unpack("<I", glb[12:16])[0] + 8:]
        for i, image in enumerate(doc["images"]):
            view = doc["bufferViews"][image["bufferView"]]
            png = binary[view.get("byteOffset", 0): view.get("byteOffset", 0) + view["byteLength"]]
            assert np.array_equal(decode_png(png), converted[f"image_{i}"]), f"texture {i}"


# ------------------------------------------------------------------------------------------------ JPEG
def _test_picture(rng, w, h):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 90 * np.cos(x / 3.0 + y / 11.0), (x * 3 + y * 5) % 256], -1) + rng.normal(0, 12, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("progressive", [False, True])
@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_jpeg_decoder_matches_libjpeg_bit_for_bit(subsampling, progressive):
    """st_jpeg.h follows the arithmetic every mainstream decoder shares (13-bit integer IDCT, triangle-filter chroma
    upsampling, 16-bit fixed-point YCbCr -> RGB), so its output can be compared with libjpeg(-turbo) through PIL exactly:
    baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, optimised and default Huffman tables, restart intervals, sizes that
    are not multiples of the MCU, single-pixel rows and columns, greyscale."""
    Image = pytest.importorskip("PIL.Image")
    import io
    from strolle_amd.api import decode_image
    rng = np.random.default_rng(10 * subsampling + progressive)
    for (w, h) in ((64, 64), (33, 17), (1, 1), (7, 50), (200, 123), (3, 3), (2, 9), (17, 1)):
        for quality, extra in ((30, {}), (92, dict(optimize=True)), (75, dict(restart_marker_blocks=2))):
            for grey in (False, True):
                a = _test_picture(rng, w, h)
                img = Image.fromarray(a[..., 0] if grey else a)
                kw = dict(quality=quality, progressive=progressive, **extra)
                if not grey:
                    kw["subsampling"] = subsampling
                buf = io.BytesIO()
                try:
                    img.save(buf, "JPEG", **kw)
                except TypeError:       # an older Pillow without restart_marker_blocks
                    kw.pop("restart_marker_blocks", None); buf = io.BytesIO(); img.save(buf, "JPEG", **kw)
                want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"), np.uint8)
                got = decode_image(buf.getvalue())
                assert np.array_equal(got, want), f"{w}x{h} q{quality} {extra} grey={grey}"


def test_jpeg_textures_load_through_the_gltf_loader(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(8)
    scene = SyntheticScene(seed=12)
    jpegs = []
    for i, (w, h) in enumerate(((24, 16), (9, 31))):
        buf = io.BytesIO()
        Image.fromarray(_test_picture(rng, w, h)).save(buf, "JPEG", quality=80, progressive=bool(i), subsampling=2 - i)
        jpegs.append(buf.getvalue())
        scene.textures_rgba[i] = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"), np.uint8)
    doc = scene.doc
    view = doc["bufferViews"][doc["images"][0]["bufferView"]]
    while len(scene.bin) % 4:
        scene.bin.append(0)
    view["byteOffset"], view["byteLength"] = len(scene.bin), len(jpegs[0])
    scene.bin += jpegs[0]
    doc["images"][0]["mimeType"] = "image/jpeg"
    doc["images"][1] = {"uri": "data:image/jpeg;base64," + base64.b64encode(jpegs[1]).decode()}
    path, _ = scene.write(str(tmp_path), glb=True)
    a, b = Engine(device=-1), Engine(device=-1)
    assert a.load_gltf(path)["images"] == 2
    scene.insert_expected(b)
    for e in (a, b):
        e.tick()
    for what in range(4):
        assert_bits_equal(a.read_scene(what), b.read_scene(what), f"scene buffer {what}")
    assert a.image_rect(1000) == (0, 0, 24, 16) and a.image_rect(1001)[2:] == (9, 31)


def test_jpeg_decoder_rejects_what_it_does_not_read_and_survives_damage():
    Image = pytest.importorskip("PIL.Image")
    import io
    from strolle_amd.api import decode_image
    rng = np.random.default_rng(6)
    a = _test_picture(rng, 40, 30)
    buf = io.BytesIO(); Image.fromarray(a).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(StrolleError, match="status 9.*four-component"):
        decode_image(buf.getvalue())
    with pytest.raises(StrolleError, match="status 9.*neither"):
        decode_image(b"RIFF\x00\x00\x00\x00WEBPVP8 ")
    good = io.BytesIO(); Image.fromarray(a).save(good, "JPEG", quality=70, progressive=True)
    good = good.getvalue()
    with pytest.raises(StrolleError):
        decode_image(good[:200])
    arithmetic = bytearray(good); arithmetic[good.index(b"\xff\xc2") + 1] = 0xCA
    with pytest.raises(StrolleError, match="status 9.*arithmetic"):
        decode_image(bytes(arithmetic))
    outcomes = [0, 0]
    for i in range(400):
        d = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            d[int(rng.integers(2, len(d)))] = int(rng.integers(0, 256))
        try:
            decode_image(bytes(d)); outcomes[0] += 1
        except StrolleError:
            outcomes[1] += 1
    assert outcomes[0] and outcomes[1], outcomes
