"""Test-side writers for the model-import tests: PNG (all five row filters, RGB / RGBA / palette / grey), glTF 2.0 as .gltf + .bin,
.gltf with data: URIs and .glb, plus independent numpy restatements of what the loader must produce (Lanczos3 resize of
image 0.25.9, builder.rs:191-292 texture-layer assignment). Nothing here is used by the product."""
import base64
import json
import struct
import zlib
import numpy as np


# ---------------------------------------------------------------------------------------------------------------- PNG
def _chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)


def _filter_rows(rows, bpp, filters):
    """rows [h, row_bytes] u8 -> filtered scanlines with a leading filter-type byte; filters: int or sequence cycled per row."""
    h, n = rows.shape
    out = np.zeros((h, n + 1), np.uint8)
    prev = np.zeros(n, np.int32)
    for y in range(h):
        ft = filters if isinstance(filters, int) else filters[y % len(filters)]
        cur = rows[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if n > bpp else np.zeros(n, np.int32)
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if n > bpp else np.zeros(n, np.int32)
        b = prev
        if ft == 0: f = cur
        elif ft == 1: f = cur - a
        elif ft == 2: f = cur - b
        elif ft == 3: f = cur - ((a + b) >> 1)
        else:
            p = a + b - c
            pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
            f = cur - pred
        out[y, 0] = ft
        out[y, 1:] = (f & 255).astype(np.uint8)
        prev = cur
    return out


def png_bytes(img, filters=(0, 1, 2, 3, 4), palette=None, depth=8, trns=None, grey=False, sixteen=False, idat_split=3):
    """img: [h, w, 3|4] u8 (colour), or [h, w] u8 indices with `palette` [n, 3] (depth 1/2/4/8), or [h, w] grey."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    extra = b""
    if palette is not None:
        ctype, bits = 3, depth
        idx = img.astype(np.uint8)
        if depth == 8:
            rows = idx
        else:
            per = 8 // depth
            pad = (-w) % per
            ip = np.pad(idx, ((0, 0), (0, pad)))
            ip = ip.reshape(h, -1, per).astype(np.uint32)
            shifts = np.array([8 - depth * (k + 1) for k in range(per)], np.uint32)
            rows = (ip << shifts).sum(axis=2).astype(np.uint8)
        extra += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
        if trns is not None:
            extra += _chunk(b"tRNS", bytes(trns))
        bpp = 1
    elif grey:
        ctype, bits, rows, bpp = 0, 8, img.astype(np.uint8), 1
    elif sixteen:
        ctype, bits = 2, 16
        rows = np.repeat(img.reshape(h, -1), 2, axis=1).astype(np.uint8)
        bpp = 6
    else:
        ch = img.shape[2]
        ctype, bits, rows, bpp = (2 if ch == 3 else 6), 8, img.reshape(h, w * ch).astype(np.uint8), ch
        if trns is not None:      # colour key for RGB: 3 x u16
            extra += _chunk(b"tRNS", struct.pack(">HHH", *trns))
    raw = _filter_rows(np.ascontiguousarray(rows), bpp, filters).tobytes()
    z = zlib.compress(raw, 6)
    parts = [z[i * len(z) // idat_split:(i + 1) * len(z) // idat_split] for i in range(idat_split)]
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bits, ctype, 0, 0, 0)) + extra
    for p in parts:
        out += _chunk(b"IDAT", p)
    return out + _chunk(b"IEND", b"")


# ---------------------------------------------------------------------------------------------------------------- Lanczos3 (image 0.25.9)
def _axis_taps(n_in, n_out):
    f32 = np.float32
    ratio = f32(n_in) / f32(n_out)
    sratio = f32(1.0) if ratio < 1 else ratio
    support = f32(3.0) * sratio
    taps = []
    for o in range(n_out):
        x = (f32(o) + f32(0.5)) * ratio
        left = int(min(max(np.floor(x - support), 0), n_in - 1))
        right = int(min(max(np.ceil(x + support), left + 1), n_in))
        x = x - f32(0.5)
        i = np.arange(left, right).astype(f32)
        t = (i - x) / sratio

        def sinc(v):
            a = (v * f32(np.pi)).astype(f32)
            with np.errstate(invalid="ignore", divide="ignore"):
                return np.where(v == 0, f32(1), np.sin(a, dtype=f32) / a).astype(f32)
        wgt = np.where(np.abs(t) < 3, sinc(t) * sinc((t / f32(3)).astype(f32)), f32(0)).astype(f32)
        s = f32(0)
        for v in wgt:
            s = f32(s + v)
        taps.append((left, (wgt / s).astype(f32)))
    return taps


def lanczos3_resize(rgba, dw, dh):
    """[h, w, 4] u8 -> [dh, dw, 4] u8, float32 arithmetic in the loop order of image's vertical_sample / horizontal_sample."""
    f32 = np.float32
    sh, sw = rgba.shape[:2]
    if (sw, sh) == (dw, dh):
        return rgba.copy()
    src = rgba.astype(f32)
    tmp = np.zeros((dh, sw, 4), f32)
    for oy, (left, w) in enumerate(_axis_taps(sh, dh)):
        acc = np.zeros((sw, 4), f32)
        for i, wi in enumerate(w):
            acc = (acc + (src[left + i] * wi).astype(f32)).astype(f32)
        tmp[oy] = acc
    out = np.zeros((dh, dw, 4), np.uint8)
    for ox, (left, w) in enumerate(_axis_taps(sw, dw)):
        acc = np.zeros((dh, 4), f32)
        for i, wi in enumerate(w):
            acc = (acc + (tmp[:, left + i] * wi).astype(f32)).astype(f32)
        v = np.clip(acc, 0, 255)
        out[:, ox] = np.floor(v + f32(0.5)).astype(np.uint8)      # round half away from zero for v >= 0
    return out


# ---------------------------------------------------------------------------------------------------------------- glTF writer
_COMP = {np.dtype(np.float32): 5126, np.dtype(np.uint8): 5121, np.dtype(np.uint16): 5123, np.dtype(np.uint32): 5125}
_TYPE = {1: "SCALAR", 2: "VEC2", 3: "VEC3", 4: "VEC4"}


class GltfWriter:
    def __init__(self):
        self.bin = bytearray()
        self.j = {"asset": {"version": "2.0"}, "buffers": [], "bufferViews": [], "accessors": [], "meshes": [], "materials": [],
                  "images": [], "textures": [], "samplers": [], "nodes": [], "scenes": [{"nodes": []}], "scene": 0}

    def _align(self, n=4):
        while len(self.bin) % n:
            self.bin.append(0)

    def view(self, data, stride=None):
        self._align()
        off = len(self.bin)
        self.bin += bytes(data)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.j["bufferViews"].append(v)
        return len(self.j["bufferViews"]) - 1

    def accessor(self, arr, normalized=False, view=None, byte_offset=0):
        arr = np.ascontiguousarray(arr)
        ncomp = 1 if arr.ndim == 1 else arr.shape[1]
        if view is None:
            view = self.view(arr.tobytes())
        a = {"bufferView": view, "componentType": _COMP[arr.dtype], "count": int(arr.shape[0]), "type": _TYPE[ncomp]}
        if byte_offset:
            a["byteOffset"] = byte_offset
        if normalized:
            a["normalized"] = True
        if arr.dtype == np.float32 and ncomp == 3:
            a["min"] = [float(x) for x in arr.min(axis=0)]; a["max"] = [float(x) for x in arr.max(axis=0)]
        self.j["accessors"].append(a)
        return len(self.j["accessors"]) - 1

    def interleaved(self, arrays):
        """float32 arrays of equal length packed vertex by vertex into one strided bufferView -> accessor ids."""
        n = arrays[0].shape[0]
        rec = np.concatenate([np.ascontiguousarray(a, np.float32).reshape(n, -1) for a in arrays], axis=1)
        stride = rec.shape[1] * 4
        v = self.view(rec.tobytes(), stride=stride)
        ids, off = [], 0
        for a in arrays:
            ids.append(self.accessor(np.ascontiguousarray(a, np.float32), view=v, byte_offset=off))
            off += a.shape[1] * 4
        return ids

    def image(self, png, mode="view", name="tex"):
        """mode: 'view' (bufferView), 'datauri', or ('file', path_written_by_caller_relative)."""
        if mode == "view":
            self.j["images"].append({"bufferView": self.view(png), "mimeType": "image/png"})
        elif mode == "datauri":
            self.j["images"].append({"uri": "data:image/png;base64," + base64.b64encode(png).decode()})
        else:
            self.j["images"].append({"uri": mode[1]})
        self.j["textures"].append({"source": len(self.j["images"]) - 1})
        return len(self.j["textures"]) - 1

    def material(self, **kw):
        self.j["materials"].append(kw)
        return len(self.j["materials"]) - 1

    def primitive(self, mesh, attributes, indices=None, material=None, mode=None):
        while len(self.j["meshes"]) <= mesh:
            self.j["meshes"].append({"primitives": []})
        p = {"attributes": attributes}
        if indices is not None: p["indices"] = indices
        if material is not None: p["material"] = material
        if mode is not None: p["mode"] = mode
        self.j["meshes"][mesh]["primitives"].append(p)

    def _clean(self):
        j = {k: v for k, v in self.j.items() if not (isinstance(v, list) and len(v) == 0)}
        return j

    def save_glb(self, path):
        self._align()
        j = self._clean()
        j["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(j).encode()
        js += b" " * ((-len(js)) % 4)
        body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(self.bin), 0x004E4942) + bytes(self.bin)
        with open(path, "wb") as f:
            f.write(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)

    def save_gltf(self, path, embed=False):
        j = self._clean()
        if embed:
            j["buffers"] = [{"byteLength": len(self.bin), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()}]
        else:
            bname = str(path).rsplit("/", 1)[-1].rsplit(".", 1)[0] + " data.bin"        # a space: exercises percent-decoding
            with open(str(path).rsplit("/", 1)[0] + "/" + bname, "wb") as f:
                f.write(bytes(self.bin))
            j["buffers"] = [{"byteLength": len(self.bin), "uri": bname.replace(" ", "%20")}]
        with open(path, "w") as f:
            json.dump(j, f, indent=1)


def uv_sphere(n_lat=12, n_lon=24, radius=0.5):
    """positions [n,3], normals [n,3], uvs [n,2], tangents [n,4], indices [m] u32."""
    th = np.linspace(0.0, np.pi, n_lat + 1)[:, None]; ph = np.linspace(0.0, 2 * np.pi, n_lon + 1)[None, :]
    n = np.stack([np.sin(th) * np.cos(ph), np.cos(th) * np.ones_like(ph), np.sin(th) * np.sin(ph)], axis=-1).reshape(-1, 3)
    uv = np.stack([np.broadcast_to(ph / (2 * np.pi), (n_lat + 1, n_lon + 1)), np.broadcast_to(th / np.pi, (n_lat + 1, n_lon + 1))], axis=-1).reshape(-1, 2)
    tg = np.stack([-np.sin(ph) * np.ones_like(th), np.zeros_like(th * ph), np.cos(ph) * np.ones_like(th), np.ones_like(th * ph)], axis=-1).reshape(-1, 4)
    idx = []
    for i in range(n_lat):
        for k in range(n_lon):
            a = i * (n_lon + 1) + k; b = a + n_lon + 1
            if i > 0: idx += [a, a + 1, b]          # outward-facing (counter-clockwise seen from outside)
            if i < n_lat - 1: idx += [a + 1, b + 1, b]
    return (n * radius).astype(np.float32), n.astype(np.float32), uv.astype(np.float32), tg.astype(np.float32), np.asarray(idx, np.uint32)


def make_textures(seed=7):
    """Five deterministic 1024 x 1024 RGBA8 images: base colour, normal map, occlusion, emissive, metallic-roughness."""
    y, x = np.mgrid[0:1024, 0:1024].astype(np.float32) / 1024.0
    rng = np.random.default_rng(seed)

    def pack(r, g, b, a=None):
        a = np.full_like(r, 1.0) if a is None else a
        return (np.clip(np.stack([r, g, b, a], axis=-1), 0, 1) * 255 + 0.5).astype(np.uint8)
    base = pack(0.5 + 0.5 * np.sin(20 * x), 0.5 + 0.5 * np.sin(14 * y + 1), 0.6 + 0.3 * np.sin(9 * (x + y)))
    nx, ny = 0.25 * np.sin(40 * x), 0.25 * np.cos(34 * y)
    normal = pack(0.5 + 0.5 * nx, 0.5 + 0.5 * ny, 0.5 + 0.5 * np.sqrt(np.maximum(0, 1 - nx * nx - ny * ny)))
    occl = pack(0.6 + 0.4 * np.sin(6 * x) * np.sin(6 * y), 0 * x, 0 * x)
    emis = pack((np.sin(30 * x) > 0.95) * 0.8, (np.sin(30 * y) > 0.95) * 0.6, 0 * x)
    mr = pack(0 * x, 0.15 + 0.8 * (0.5 + 0.5 * np.sin(11 * x + 3 * y)), 0.5 + 0.5 * np.cos(7 * y))
    noise = rng.integers(0, 3, size=base.shape, dtype=np.uint8)
    base[..., :3] = np.clip(base[..., :3].astype(np.int32) + noise[..., :3] - 1, 0, 255).astype(np.uint8)
    return base, normal, occl, emis, mr


# ---------------------------------------------------------------------------------------------------------------- builder.rs:191-292 restated
def assign_layers(materials_tex, n_color0, n_data0):
    """materials_tex: per material (base, normal, occlusion, emissive, mr) IMAGE indices or None. Returns the per-material layer
    ids and the image order appended to the colour and the data array (first use wins, base/emissive -> colour, others -> data)."""
    cmap, dmap, corder, dorder, out = {}, {}, [], [], []

    def take(m, order, base, img):
        if img is None:
            return 0xFFFF
        if img not in m:
            m[img] = base + len(order); order.append(img)
        return m[img]
    for (b, n, o, e, mr) in materials_tex:
        ib = take(cmap, corder, n_color0, b)
        inn = take(dmap, dorder, n_data0, n)
        io = take(dmap, dorder, n_data0, o)
        ie = take(cmap, corder, n_color0, e)
        imr = take(dmap, dorder, n_data0, mr)
        out.append((ib, inn, io, ie, imr))
    return out, corder, dorder
