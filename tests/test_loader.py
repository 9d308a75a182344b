"""Model import: src/scene/loader.rs:9-181, src/scene/builder.rs:191-314, src/scene/scenes.rs:246-322 through the
C ABI (frt_model_*, frt_scene_add_gltf_*, frt_scene_create_gltf_scene). The files are written by tests/_gltf.py; expected values are
the arrays that went into them or independent numpy restatements. Host-only: no GPU needed."""
import json
import numpy as np
import pytest
import _gltf


def _sphere_model(tmp_path, frt, kind="glb", tex_mode="view", interleaved=False, index_dtype=np.uint16, uv_dtype=np.float32):
    pos, nrm, uv, tg, idx = _gltf.uv_sphere()
    w = _gltf.GltfWriter()
    texs = _gltf.make_textures()
    tids = [w.image(_gltf.png_bytes(t if k != 1 else t[..., :3], filters=(0, 1, 2, 3, 4)), mode=tex_mode) for k, t in enumerate(texs)]
    m0 = w.material(pbrMetallicRoughness={"baseColorFactor": [0.9, 0.8, 0.7, 1.0], "metallicFactor": 0.25, "roughnessFactor": 0.6,
                                          "baseColorTexture": {"index": tids[0]}, "metallicRoughnessTexture": {"index": tids[4]}},
                    normalTexture={"index": tids[1]}, occlusionTexture={"index": tids[2]}, emissiveTexture={"index": tids[3]},
                    emissiveFactor=[0.5, 0.4, 0.3])
    m1 = w.material(pbrMetallicRoughness={"baseColorTexture": {"index": tids[0]}}, normalTexture={"index": tids[0]})
    if uv_dtype == np.float32:
        uvq, uvn = uv, False
    else:
        mx = np.iinfo(uv_dtype).max
        uvq, uvn = np.round(uv * mx).astype(uv_dtype), True
    if interleaved:
        a_pos, a_nrm, a_tg = w.interleaved([pos, nrm, tg])
    else:
        a_pos, a_nrm, a_tg = w.accessor(pos), w.accessor(nrm), w.accessor(tg)
    a_uv = w.accessor(uvq, normalized=uvn)
    w.primitive(0, {"POSITION": a_pos, "NORMAL": a_nrm, "TEXCOORD_0": a_uv, "TANGENT": a_tg}, indices=w.accessor(idx.astype(index_dtype)), material=m0)
    quad = np.array([[-1, 0, -1], [1, 0, -1], [-1, 0, 1], [1, 0, 1]], np.float32)
    w.primitive(1, {"POSITION": w.accessor(quad)}, indices=w.accessor(np.array([0, 2, 1, 1, 2, 3], np.uint8)), material=m1)
    w.primitive(1, {"POSITION": w.accessor(quad[[0, 2, 1]] + np.float32(2.0))})       # non-indexed, no material
    path = tmp_path / ("model." + ("glb" if kind == "glb" else "gltf"))
    if kind == "glb": w.save_glb(str(path))
    else: w.save_gltf(str(path), embed=(kind == "embedded"))
    uv_expect = uv if uv_dtype == np.float32 else (uvq.astype(np.float32) / np.float32(np.iinfo(uv_dtype).max))
    return path, dict(pos=pos, nrm=nrm, uv=uv_expect, tg=tg, idx=idx, quad=quad, texs=texs)


def _check_model(frt, model, ref):
    c = model.counts()
    assert (c["geometries"], c["materials"], c["images"], c["warnings"]) == (3, 2, 5, 0)
    g, mi = model.geometry(0)
    assert mi == 0
    np.testing.assert_array_equal(g.positions[:, :3], ref["pos"]); assert (g.positions[:, 3] == 1.0).all()
    enc = np.stack([frt.geometry.encode_octahedral_normal(n) for n in ref["nrm"]])
    np.testing.assert_array_equal(g.attributes[:, 0:2], enc)
    np.testing.assert_array_equal(g.attributes[:, 2:4], ref["uv"])
    np.testing.assert_array_equal(g.attributes[:, 4:8], ref["tg"])
    np.testing.assert_array_equal(g.indices, ref["idx"])
    g1, mi1 = model.geometry(1)
    assert mi1 == 1 and g1.indices.tolist() == [0, 2, 1, 1, 2, 3]
    np.testing.assert_array_equal(g1.positions[:, :3], ref["quad"])
    up = frt.geometry.encode_octahedral_normal([0, 1, 0])
    assert (g1.attributes[:, 0:2] == up).all() and (g1.attributes[:, 2:4] == 0).all() and (g1.attributes[:, 4:8] == [1, 0, 0, 1]).all()   # loader.rs:127-146 defaults
    g2, mi2 = model.geometry(2)
    assert mi2 == 0 and g2.indices.tolist() == [0, 1, 2]                                                                               # loader.rs:160-163, :176
    m0, m1 = model.material(0), model.material(1)
    assert list(m0.base_color) == [np.float32(0.9), np.float32(0.8), np.float32(0.7), 1.0]
    assert m0.metallic == 1.0 and m0.roughness == np.float32(0.6)            # Material::metallic(x) sets metallic = 1 (material.rs:54-58), then .roughness()
    assert list(m0.emissive_factor) == [0.5, np.float32(0.4), np.float32(0.3)] and m0.light_index == -1 and m0.transmission == 0.0 and m0.ior == 1.0
    assert (m0.tex_info_0, m0.tex_info_1, m0.tex_info_2) == (0 | (1 << 16), 2 | (3 << 16), 4 | (0xFFFF << 16))       # image indices
    assert m1.metallic == 1.0 and m1.roughness == 1.0 and list(m1.base_color) == [1, 1, 1, 1]                        # glTF defaults
    assert (m1.tex_info_0, m1.tex_info_1, m1.tex_info_2) == (0, 0xFFFFFFFF, 0xFFFFFFFF)
    for k, t in enumerate(ref["texs"]):
        want = t.copy()
        if k == 1: want[..., 3] = 255                                       # written as RGB -> to_rgba8
        np.testing.assert_array_equal(model.image(k), want)                # 1024 x 1024 in = plain copy out


@pytest.mark.parametrize("kind,tex_mode,interleaved,index_dtype,uv_dtype", [
    ("glb", "view", False, np.uint16, np.float32),
    ("gltf", "datauri", True, np.uint32, np.uint16),
    ("embedded", "datauri", False, np.uint16, np.uint8),
])
def test_load_gltf_roundtrip(frt, tmp_path, kind, tex_mode, interleaved, index_dtype, uv_dtype):
    path, ref = _sphere_model(tmp_path, frt, kind, tex_mode, interleaved, index_dtype, uv_dtype)
    _check_model(frt, frt.loader.load_gltf(path), ref)


def test_external_image_file_and_default_material(frt, tmp_path):
    w = _gltf.GltfWriter()
    img = _gltf.make_textures()[0]
    (tmp_path / "my tex.png").write_bytes(_gltf.png_bytes(img, filters=4))
    w.image(b"", mode=("file", "my%20tex.png"))
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    w.primitive(0, {"POSITION": w.accessor(tri)})
    p = tmp_path / "a.gltf"; w.save_gltf(str(p))
    m = frt.loader.load_gltf(p)
    assert m.counts() == {"geometries": 1, "materials": 1, "images": 1, "warnings": 0}
    d = m.material(0)                                                       # loader.rs:101-104: Material::new([1,1,1,1])
    assert list(d.base_color) == [1, 1, 1, 1] and d.metallic == 0.0 and d.roughness == 0.5 and d.tex_info_0 == 0xFFFFFFFF
    np.testing.assert_array_equal(m.image(0), img)


def _one_image_model(tmp_path, png, name="i.glb"):
    w = _gltf.GltfWriter()
    w.image(png)
    w.primitive(0, {"POSITION": w.accessor(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32))})
    p = tmp_path / name; w.save_glb(str(p))
    return p


@pytest.mark.parametrize("depth", [1, 2, 4, 8])
def test_png_palette_depths(frt, tmp_path, depth):
    rng = np.random.default_rng(depth)
    ncol = 1 << depth
    pal = rng.integers(0, 256, size=(ncol, 3), dtype=np.uint8)
    idx = rng.integers(0, ncol, size=(1024, 1024), dtype=np.uint8)
    trns = None if depth == 2 else rng.integers(0, 256, size=max(1, ncol // 2), dtype=np.uint8)
    m = frt.loader.load_gltf(_one_image_model(tmp_path, _gltf.png_bytes(idx, palette=pal, depth=depth, trns=trns, filters=(0, 2))))
    want = np.concatenate([pal[idx], np.full((1024, 1024, 1), 255, np.uint8)], axis=2)
    if trns is not None:
        a = np.full(ncol, 255, np.uint8); a[:len(trns)] = trns
        want[..., 3] = a[idx]
    np.testing.assert_array_equal(m.image(0), want)


def test_png_colour_key_and_each_filter(frt, tmp_path):
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, size=(1024, 1024, 3), dtype=np.uint8)
    rgb[100:200, 50:80] = (10, 20, 30)
    for ft in range(5):
        m = frt.loader.load_gltf(_one_image_model(tmp_path, _gltf.png_bytes(rgb, filters=ft, trns=(10, 20, 30), idat_split=1 + ft), f"f{ft}.glb"))
        got = m.image(0)
        np.testing.assert_array_equal(got[..., :3], rgb)
        key = (rgb == (10, 20, 30)).all(axis=2)
        np.testing.assert_array_equal(got[..., 3], np.where(key, 0, 255))


def test_unsupported_images_become_white_with_a_warning(frt, tmp_path):
    """loader.rs:35-44: formats other than R8G8B8 / R8G8B8A8 -> white TEXTURE_WIDTH x TEXTURE_HEIGHT; so do undecodable files."""
    g = np.zeros((8, 8), np.uint8)
    cases = {"grey": _gltf.png_bytes(g, grey=True), "sixteen": _gltf.png_bytes(np.zeros((4, 4, 3), np.uint8), sixteen=True),
             "jpeg": b"\xff\xd8\xff\xe0\x00\x10" + b"\0" * 32, "junk": b"not an image at all"}
    for name, data in cases.items():
        m = frt.loader.load_gltf(_one_image_model(tmp_path, data, name + ".glb"))
        assert (m.image(0) == 255).all(), name
        ws = m.warnings()
        assert len(ws) == 1 and "white" in ws[0], (name, ws)


@pytest.mark.parametrize("shape", [(48, 64), (1500, 700), (1024, 512), (3, 5)])
def test_lanczos3_resize_matches_the_f32_restatement(frt, tmp_path, shape):
    """image 0.25.9 resize_exact(1024, 1024, Lanczos3): vertical pass to f32, horizontal pass, clamp, round. Up- and down-scaling."""
    h, w = shape
    rng = np.random.default_rng(h * 31 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), rng.integers(0, 256, (h, w)), rng.integers(0, 256, (h, w))], axis=-1).astype(np.uint8)
    m = frt.loader.load_gltf(_one_image_model(tmp_path, _gltf.png_bytes(img, filters=(1, 4))))
    got = m.image(0).astype(np.int32)
    want = _gltf.lanczos3_resize(img, 1024, 1024).astype(np.int32)
    d = np.abs(got - want)
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (d.max(), (d != 0).mean())     # libm sinf vs numpy sin: a last-bit weight difference may flip a rounding


def test_loader_errors(frt, tmp_path):
    def expect(path, text):
        with pytest.raises(frt.FrtError) as e:
            frt.loader.load_gltf(path)
        assert text in str(e.value), str(e.value)
    expect(tmp_path / "missing.glb", "cannot read")
    (tmp_path / "bad.gltf").write_text("{ \"asset\": ")
    expect(tmp_path / "bad.gltf", "json")
    (tmp_path / "v1.glb").write_bytes(b"glTF" + (1).to_bytes(4, "little") + (12).to_bytes(4, "little"))
    expect(tmp_path / "v1.glb", "version")
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    for name, edit, text in [
        ("mode", lambda j: j["meshes"][0]["primitives"][0].__setitem__("mode", 1), "TRIANGLES"),
        ("sparse", lambda j: j["accessors"][0].__setitem__("sparse", {"count": 1}), "sparse"),
        ("range", lambda j: j["accessors"][1].__setitem__("count", 9), "exceeds"),
        ("nopos", lambda j: j["meshes"][0]["primitives"][0]["attributes"].pop("POSITION"), "POSITION"),
        ("view", lambda j: j["bufferViews"][0].__setitem__("byteLength", 10 ** 6), "exceeds"),
    ]:
        w = _gltf.GltfWriter()
        w.primitive(0, {"POSITION": w.accessor(tri)}, indices=w.accessor(np.array([0, 1, 2], np.uint16)))
        p = tmp_path / (name + ".gltf"); w.save_gltf(str(p), embed=True)
        j = json.loads(p.read_text()); edit(j); p.write_text(json.dumps(j))
        expect(p, text)
    w = _gltf.GltfWriter()
    w.primitive(0, {"POSITION": w.accessor(tri)}, indices=w.accessor(np.array([0, 1, 7], np.uint16)))
    p = tmp_path / "oob.glb"; w.save_glb(str(p))
    expect(p, "index out of range")
    with pytest.raises(frt.FrtError):
        frt.scenes.create_gltf_scene(tmp_path / "missing.glb", np.eye(4), np.eye(4))


def _mat_tex(m):
    f = lambda v: None if v == 0xFFFF else v
    return (f(m.tex_info_0 & 0xFFFF), f(m.tex_info_0 >> 16), f(m.tex_info_1 & 0xFFFF), f(m.tex_info_1 >> 16), f(m.tex_info_2 & 0xFFFF))


def test_add_gltf_to_scene_assigns_texture_layers_like_builder_rs(frt, tmp_path):
    path, ref = _sphere_model(tmp_path, frt)
    model = frt.loader.load_gltf(path)
    b = frt.SceneBuilder()
    floor = b.add_material(frt.material_new([0.5, 0.5, 0.5, 1]))
    mat_ids = b.add_gltf_materials(model)
    mesh_ids = b.add_gltf_meshes(model)
    assert mat_ids.tolist() == [floor + 1, floor + 2] and mesh_ids.tolist() == [0, 1, 2]
    T = np.eye(4, dtype=np.float32); T[3, :3] = (0.1, 0.2, 0.3)
    b.add_gltf_instances(model, mesh_ids, mat_ids, T.reshape(16))
    b.build()
    want, corder, dorder = _gltf.assign_layers([_mat_tex(model.material(i)) for i in range(2)], 3, 3)    # 3 default layers per array (builder.rs:41-91)
    assert corder == [0, 3] and dorder == [1, 2, 4, 0]       # image 0 is used as base colour AND (by material 1) as a normal map
    mats = b.get("materials")
    for k, w_ in enumerate(want):
        got = frt.Material.from_buffer_copy(mats[mat_ids[k]].tobytes())
        assert _mat_tex(got) == tuple(None if v == 0xFFFF else v for v in w_), (k, _mat_tex(got), w_)
    inst = b.get("instances")
    assert inst[:, 0].tolist() == [0, 1, 2] and inst[:, 1].tolist() == [mat_ids[0], mat_ids[1], mat_ids[0]]
    np.testing.assert_array_equal(inst[:, 5:21].view(np.float32), np.tile(T.reshape(16), (3, 1)))
    assert b.counts()["tris"] == ref["idx"].size // 3 + 2 + 1


def test_create_gltf_scene(frt, tmp_path):
    """scenes.rs:246-322: plane + light plane meshes, floor material, quad light (material + light + instance), then the model."""
    path, ref = _sphere_model(tmp_path, frt)
    L = np.eye(4, dtype=np.float32); L[1, 1] = -1.0; L[2, 2] = -1.0; L[3, 1] = 5.0       # translate(0,5,0) * rotate_x(pi)
    M = np.eye(4, dtype=np.float32) * np.float32(2.0); M[3, 3] = 1.0
    s = frt.scenes.create_gltf_scene(path, M.reshape(16), L.reshape(16))
    c = s.counts()
    assert (c["meshes"], c["materials"], c["lights"], c["instances"]) == (2 + 3, 1 + 1 + 2, 1, 2 + 3)
    assert c["tris"] == 2 + 2 + ref["idx"].size // 3 + 2 + 1
    mats = [frt.Material.from_buffer_copy(r.tobytes()) for r in s.get("materials")]
    assert mats[0].roughness == np.float32(0.99) and list(mats[0].base_color)[:3] == [np.float32(0.73)] * 3
    assert mats[1].light_index == 0 and list(mats[1].emissive_factor) == [15.0, 15.0, 15.0]
    light = frt.Light.from_buffer_copy(s.get("lights")[0].tobytes())
    assert light.type_ == 0 and list(light.emission) == [1.0, 1.0, 1.0, 15.0] and list(light.position) == [0.0, 5.0, 0.0]
    inst = s.get("instances")
    assert inst[:, 0].tolist() == [0, 1, 2, 3, 4] and inst[:, 1].tolist() == [0, 1, 2, 3, 2]
    np.testing.assert_array_equal(inst[2:, 5:21].view(np.float32), np.tile(M.reshape(16), (3, 1)))
    floor = inst[0, 5:21].view(np.float32).reshape(4, 4)
    assert floor[0, 0] == 10.0 and floor[3, 1] == -1.0


def test_load_obj_subset(frt, tmp_path):
    """Extension (no OBJ path in the reference): quads are fan-triangulated, negative indices, missing normals -> smooth normals,
    v flipped (OBJ's v runs upward), one default material."""
    p = tmp_path / "cube.obj"
    p.write_text("""# unit cube, quads, no normals
mtllib none.mtl
v -0.5 -0.5 -0.5
v  0.5 -0.5 -0.5
v  0.5  0.5 -0.5
v -0.5  0.5 -0.5
v -0.5 -0.5  0.5
v  0.5 -0.5  0.5
v  0.5  0.5  0.5
v -0.5  0.5  0.5
vt 0 0
vt 1 0
vt 1 1
vt 0 1
f 1/1 4/4 3/3 2/2
f 5/1 6/2 7/3 8/4
f 1/1 2/2 6/3 5/4
f 2/1 3/2 7/3 6/4
f 3/1 4/2 8/3 7/4
f -4/1 -1/2 -5/3 -8/4
""")
    m = frt.loader.load_gltf(p)
    assert m.counts()["geometries"] == 1 and m.counts()["materials"] == 1 and len(m.warnings()) == 1
    g, mi = m.geometry(0)
    assert g.indices.size == 36 and mi == 0
    tri = g.positions[g.indices.reshape(-1, 3), :3]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    c = tri.mean(axis=1)
    assert ((n * c).sum(axis=1) > 0).all()                     # every face winds outward
    assert set(np.unique(g.attributes[:, 3]).tolist()) == {0.0, 1.0}
    touching = (tri == g.positions[0, :3]).all(axis=2).any(axis=1)             # area-weighted sum of the face normals around position 0
    sn = n[touching].sum(axis=0); sn /= np.linalg.norm(sn)
    assert (sn < 0).all()                                                       # the (-,-,-) corner
    same_pos = (g.positions[:, :3] == g.positions[0, :3]).all(axis=1)
    np.testing.assert_allclose(g.attributes[same_pos, 0:2], np.tile(frt.geometry.encode_octahedral_normal(sn), (same_pos.sum(), 1)), atol=1e-6)
    q = tmp_path / "n.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 0 0 1\nvn 0 1 0\nf 1//1 3//1 2//1\n")
    g2, _ = frt.loader.load_gltf(q).geometry(0)
    assert (g2.attributes[:, 0:2] == frt.geometry.encode_octahedral_normal([0, 1, 0])).all() and (g2.attributes[:, 2:4] == 0).all()
    with pytest.raises(frt.FrtError):
        (tmp_path / "e.obj").write_text("v 0 0 0\n"); frt.loader.load_gltf(tmp_path / "e.obj")


@pytest.mark.parametrize("subsampling,quality,extra", [(0, 92, {}), (1, 85, {}), (2, 75, {}), (2, 90, {"restart_marker_rows": 1}), (0, 95, {"optimize": True}),
                                                       (0, 90, {"progressive": True}), (2, 80, {"progressive": True}), (1, 60, {"progressive": True, "restart_marker_rows": 2})])
def test_baseline_jpeg_textures(frt, tmp_path, subsampling, quality, extra):
    """Baseline and progressive JPEG (4:4:4, 4:2:2, 4:2:0, restart markers, optimised Huffman tables) against Pillow's libjpeg decode of the same
    bytes. The standard does not fix IDCT rounding or chroma upsampling, so decoders differ by a level here and there."""
    PIL = pytest.importorskip("PIL.Image")
    import io
    img = _gltf.make_textures()[0][..., :3]
    buf = io.BytesIO()
    try:
        PIL.fromarray(img).save(buf, "JPEG", quality=quality, subsampling=subsampling, **extra)
    except TypeError:
        pytest.skip("this Pillow cannot write the requested JPEG variant")
    data = buf.getvalue()
    want = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB")).astype(np.int32)
    m = frt.loader.load_gltf(_one_image_model(tmp_path, data))
    assert m.warnings() == []
    got = m.image(0)
    assert (got[..., 3] == 255).all()
    d = np.abs(got[..., :3].astype(np.int32) - want)
    assert d.mean() < 0.6 and d.max() <= 6, (d.mean(), d.max())


def test_jpeg_variants_the_loader_does_not_decode(frt, tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    import io
    img = _gltf.make_textures()[0][:64, :64, :3]
    for name, kw, conv in (("grey", {}, "L"), ("grey-progressive", {"progressive": True}, "L")):
        buf = io.BytesIO(); PIL.fromarray(img).convert(conv).save(buf, "JPEG", quality=90, **kw)
        m = frt.loader.load_gltf(_one_image_model(tmp_path, buf.getvalue(), name + ".glb"))
        assert (m.image(0) == 255).all() and len(m.warnings()) == 1 and "white" in m.warnings()[0], name
    # odd size + resize path
    odd = _gltf.make_textures()[0][:333, :517, :3]
    buf = io.BytesIO(); PIL.fromarray(odd).save(buf, "JPEG", quality=90, subsampling=2)
    m = frt.loader.load_gltf(_one_image_model(tmp_path, buf.getvalue(), "odd.glb"))
    want = np.asarray(PIL.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    want = _gltf.lanczos3_resize(np.concatenate([want, np.full(want.shape[:2] + (1,), 255, np.uint8)], axis=2), 1024, 1024).astype(np.int32)
    d = np.abs(m.image(0).astype(np.int32) - want)
    assert m.warnings() == [] and d.mean() < 0.6 and d.max() <= 8, (d.mean(), d.max())


def test_loader_survives_mutated_files(frt, tmp_path):
    """Robustness: truncated and bit-flipped GLB / PNG / JPEG inputs must end in a clean error or a fallback, never a crash."""
    path, _ = _sphere_model(tmp_path, frt)
    blob = bytearray(path.read_bytes()[:400000])           # keep the test quick: JSON chunk + geometry + part of the first image
    rng = np.random.default_rng(11)
    survived = 0
    for trial in range(60):
        b = bytearray(blob)
        if trial % 3 == 0:
            b = b[:int(rng.integers(12, len(b)))]
        else:
            for _ in range(int(rng.integers(1, 30))):
                b[int(rng.integers(0, min(len(b), 3000)))] = int(rng.integers(0, 256))     # the JSON / header region
        p = tmp_path / f"m{trial}.glb"; p.write_bytes(bytes(b))
        try:
            frt.loader.load_gltf(p).counts(); survived += 1
        except frt.FrtError:
            pass
    PIL = pytest.importorskip("PIL.Image")
    import io
    buf = io.BytesIO(); PIL.fromarray(_gltf.make_textures()[0][:96, :96, :3]).save(buf, "JPEG", quality=80)
    jp = buf.getvalue(); pn = _gltf.png_bytes(_gltf.make_textures()[0][:64, :64])
    for trial in range(80):
        src = bytearray(jp if trial % 2 else pn)
        for _ in range(int(rng.integers(1, 12))):
            src[int(rng.integers(2, len(src)))] = int(rng.integers(0, 256))
        if trial % 5 == 0:
            src = src[:int(rng.integers(8, len(src)))]
        m = frt.loader.load_gltf(_one_image_model(tmp_path, bytes(src), f"i{trial}.glb"))
        assert m.image(0).shape == (1024, 1024, 4)


def test_gltf_showcase_scenes(frt, tmp_path):
    """scenes.rs:324-520 wrappers (assets are not shipped: the synthetic model stands in). Transforms as glam composes them."""
    path, ref = _sphere_model(tmp_path, frt)
    s = frt.scenes.create_avocado_scene(path)
    inst = s.get("instances")
    M = inst[2, 5:21].view(np.float32).reshape(4, 4)
    assert np.allclose(M, np.diag([20, 20, 20, 1]).astype(np.float32))
    L = inst[1, 5:21].view(np.float32).reshape(4, 4)                  # translate(0,5,0) * rotation_x(pi): the light faces down
    assert L[3, 1] == 5.0 and abs(L[1, 1] + 1.0) < 1e-6 and abs(L[2, 2] + 1.0) < 1e-6
    light = frt.Light.from_buffer_copy(s.get("lights")[0].tobytes())
    assert abs(light.position[1] - 5.0) < 1e-6
    h = frt.scenes.create_damaged_helmet_scene(path).get("instances")[2, 5:21].view(np.float32).reshape(4, 4)
    assert abs(h[1, 2] - 1.0) < 1e-6 and abs(h[2, 1] + 1.0) < 1e-6    # rotation_x(pi/2): +Y -> +Z
    v = frt.scenes.create_multi_material_model_scene(path).get("instances")[2, 5:21].view(np.float32).reshape(4, 4)
    assert abs(v[0, 0] + 0.5) < 1e-6 and abs(v[2, 2] + 0.5) < 1e-6 and v[1, 1] == 0.5
    c = frt.scenes.create_chocolate_truffle_scene(path)
    n = c.counts()
    assert n["lights"] == 3 and n["meshes"] == 2 + 3 + 1 and n["instances"] == 1 + 3 + 3 and n["materials"] == 1 + 2 + 3
    mats = [frt.Material.from_buffer_copy(r.tobytes()) for r in c.get("materials")]
    assert mats[0].metallic == 1.0 and mats[0].roughness == np.float32(0.8)             # Material::metallic(0.8) quirk
    assert mats[1].roughness == np.float32(0.25) and mats[2].roughness == np.float32(0.25)   # both bright: satin, metallic left at 1
    lights = [frt.Light.from_buffer_copy(r.tobytes()) for r in c.get("lights")]
    assert [l.type_ for l in lights] == [1, 1, 1] and list(lights[0].position) == [8.0, 4.0, 2.0] and lights[0].emission[3] == 80.0
    fb = frt.scenes.create_chocolate_truffle_scene(tmp_path / "missing.glb", fallback=path)   # falls back to the avocado scene
    assert fb.counts()["lights"] == 1


def test_out_of_range_texture_and_light_indices_never_reach_the_gpu(frt, tmp_path):
    """A texture whose `source` names an image that does not exist (the gltf crate rejects such a document) leaves the slot empty and
    is reported; a material whose texture layer or light index does not exist makes SceneBuilder.build() fail instead of becoming an
    out-of-bounds read in sample_layer."""
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    w = _gltf.GltfWriter()
    w.primitive(0, {"POSITION": w.accessor(tri)}, indices=w.accessor(np.array([0, 1, 2], np.uint16)), material=0)
    p = tmp_path / "badtex.gltf"; w.save_gltf(str(p), embed=True)
    j = json.loads(p.read_text())
    j["textures"] = [{"source": 9999}]
    j["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}, "normalTexture": {"index": 0}}]
    p.write_text(json.dumps(j))
    m = frt.loader.load_gltf(p)
    assert _mat_tex(m.material(0)) == (None, None, None, None, None)
    assert any("9999" in x and "does not exist" in x for x in m.warnings())
    # builder level: a hand-made material with a layer / light that does not exist
    for field, value, text in (("tex_info_0", 57, "base colour texture layer 57"), ("tex_info_2", 0xFFFF0000 | 9, "metallic-roughness texture layer 9"),
                               ("light_index", 3, "light_index 3")):
        b = frt.SceneBuilder()
        mesh = b.add_mesh(frt.geometry.create_plane())
        mat = frt.material_new((1, 1, 1, 1))
        if field == "tex_info_0":
            mat.tex_info_0 = 0xFFFF0000 | value
        else:
            setattr(mat, field, value)
        b.add_instance(mesh, b.add_material(mat), np.eye(4, dtype=np.float32))
        with pytest.raises(frt.FrtError) as e:
            b.build()
        assert text in str(e.value), str(e.value)
