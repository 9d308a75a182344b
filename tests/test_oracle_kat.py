"""T0 known-answer tests of the oracle (SURVEY.md §8c pins, derived from the reference text) and of its numeric contract."""
import ctypes as C
import numpy as np
import pytest


def test_pcg_hash_pins(orc):
    # restir.wgsl:132-136
    pins = {0: 129708002, 1: 2831084092, 2: 2055130248, 12345: 4099845390, 0xFFFFFFFF: 3861530882}
    for k, v in pins.items():
        assert orc.L.orc_pcg_hash(k) == v
    # temporal seed (restir.wgsl:797-798) and spatial seed (restir_spatial.wgsl:866-867), pixel 1000 frame 3
    assert orc.L.orc_pcg_hash((1000 + 3 * 927163) & 0xFFFFFFFF) == 1637827857
    assert orc.L.orc_pcg_hash((1000 + 3 * 0x12345678) & 0xFFFFFFFF) == 1189794563


def test_struct_sizes(orc):
    s = (C.c_uint32 * 8)()
    orc.L.orc_struct_sizes(s)
    assert list(s) == [288, 32, 64, 64, 16, 32, 16, 32]


def test_mesh_counts(orc):
    # geometry.rs: plane 4v/2t, cube 24/12, icosphere(3) 642/1280, crystal 48/16; icosphere(2) 162/320
    for which, sub, want in [(0, 0, (4, 2)), (1, 0, (24, 12)), (2, 3, (642, 1280)), (3, 0, (48, 16)), (2, 2, (162, 320)), (2, 4, (2562, 5120))]:
        c = (C.c_uint32 * 2)()
        orc.L.orc_mesh_counts(which, sub, c)
        assert tuple(c) == want


def test_octahedral_examples(orc):
    # geometry.rs:56-76
    for n, want in [((0, 1, 0), (0, 1)), ((0, 0, -1), (1, 1)), ((-1, 0, 0), (-1, 0)), ((0, 0, 1), (0, 0)), ((0, -1, 0), (0, -1))]:
        a = np.array(n, np.float32); out = np.zeros(2, np.float32)
        orc.L.orc_encode_octahedral(a.ctypes.data, out.ctypes.data)
        assert tuple(out) == want


def test_unorm8_quantisation(orc):
    # G-buffer albedo rgba8unorm (renderer.rs:129-131)
    for v, q in [(0.73, 186), (0.65, 166), (0.05, 13), (0.12, 31), (0.45, 115), (0.15, 38), (0.0, 0), (1.0, 255), (2.0, 255), (-1.0, 0)]:
        assert orc.L.orc_f32_to_unorm8(v) == q
    assert orc.L.orc_f32_to_unorm8(float("nan")) == 0


def test_f16_conversion_matches_ieee(orc):
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2**32, 200000, dtype=np.uint64).astype(np.uint32)
    vals = bits.view(np.float32)
    vals = np.concatenate([vals, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.9802326e-8,
                                           6.1035156e-5, 6.0975552e-5, 1.0, 10.0, 200.0, np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    for v, w in zip(vals, want):
        if np.isnan(v):
            continue
        assert orc.L.orc_f32_to_f16(float(v)) == int(w), v
    h = rng.integers(0, 2**16, 20000, dtype=np.uint32).astype(np.uint16)
    for b in h:
        f = np.array([b], np.uint16).view(np.float16).astype(np.float32)[0]
        g = orc.L.orc_f16_to_f32(int(b))
        assert (np.isnan(f) and np.isnan(g)) or f == g


def _ulp_err(got, want):
    want = np.asarray(want, np.float64)
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    return np.max(np.abs(np.asarray(got, np.float64) - want) / ulp)


def test_contract_elementary_functions_accuracy(orc):
    # fixed polynomial algorithms: must be close to the true functions (a few ulp); the exact bits are the contract
    x = np.linspace(0, 2 * np.pi, 5001, dtype=np.float32)
    s = [orc.L.orc_sin(float(v)) for v in x]; c = [orc.L.orc_cos(float(v)) for v in x]
    assert np.max(np.abs(np.array(s) - np.sin(x.astype(np.float64)))) < 2e-7
    assert np.max(np.abs(np.array(c) - np.cos(x.astype(np.float64)))) < 2e-7
    e = np.linspace(-20, 20, 2001, dtype=np.float32)
    assert _ulp_err([orc.L.orc_exp2(float(v)) for v in e], np.exp2(e.astype(np.float64))) < 4
    l = np.exp(np.linspace(-20, 20, 2001)).astype(np.float32)
    got = np.array([orc.L.orc_log2(float(v)) for v in l])
    assert np.max(np.abs(got - np.log2(l.astype(np.float64)))) < 4e-6
    b = np.linspace(0.001, 1.0, 500, dtype=np.float32)
    for y in (5.0, 20.0, 1 / 2.2):
        got = np.array([orc.L.orc_pow(float(v), y) for v in b])
        want = np.power(b.astype(np.float64), y)
        ok = want > 1e-30                      # below that f32 underflows towards 0
        assert np.max(np.abs(got[ok] - want[ok]) / want[ok]) < 2e-5 and np.all(got[~ok] < 1e-29)
    assert orc.L.orc_pow(0.0, 5.0) == 0.0 and orc.L.orc_pow(1.0, 5.0) == 1.0 and orc.L.orc_pow(-1.0, 5.0) == 0.0
    assert orc.L.orc_exp(0.0) == 1.0


def test_light_table(orc):
    # LightUniform rows of SURVEY.md §8(a): builder.rs:316-429 applied to scenes.rs:93-117
    L = orc.cornell().get("lights")
    f = L.view(np.float32)
    assert L[0, 3] == 0 and L[1, 3] == 1                                 # type_: quad, sphere
    np.testing.assert_allclose(f[0, 0:3], [0, 0.99, 0], atol=0)
    np.testing.assert_allclose(f[0, 4:7], [0.25, 0, 0], atol=1e-7)
    np.testing.assert_allclose(f[0, 8:11], [0, 0, 0.25], atol=1e-7)
    assert abs(f[0, 7] - 0.25) < 1e-7
    np.testing.assert_allclose(f[0, 12:16], [1, 1, 1, 10])
    np.testing.assert_allclose(f[1, 0:3], np.array([0.4, -0.5, 0.3], np.float32))
    assert abs(f[1, 8] - 0.05) < 1e-8 and abs(f[1, 7] - 0.0314159) < 1e-6
    np.testing.assert_allclose(f[1, 12:16], np.array([0.02, 0.02, 0.9, 10], np.float32))


def test_cornell_counts_and_materials(orc):
    s = orc.cornell()
    c = s.counts()
    assert (c["tris"], c["instances"], c["materials"], c["lights"], c["meshes"]) == (1320, 9, 8, 2, 4)
    m = s.get("materials").view(np.float32)
    mi = s.get("materials")
    # scenes.rs:31-48: metal = Material::metallic(0.01) -> metallic 1, roughness 0.01 (material.rs:54-58 quirk); glass(1.5)
    assert m[4, 8] == 1.0 and m[4, 7] == np.float32(0.01)
    assert m[5, 9] == 1.0 and m[5, 10] == np.float32(1.5) and m[5, 7] == 0.0
    assert mi[3, 12] == 0xFFFF0001 and mi[6, 12] == 0xFFFF0000           # checker uses colour layer 1, lights layer 0
    assert mi.view(np.int32)[6, 11] == 0 and mi.view(np.int32)[7, 11] == 1 and mi.view(np.int32)[0, 11] == -1


def test_frame0_invariants(orc):
    """background pixels, light-quad pixels after the temporal stage, zero motion, accum definition."""
    s = orc.cornell()
    W = H = 96
    r = s.renderer(W, H, 8, False, 8)
    cam = orc.camera(1.0, 0, 2)
    r.render(cam)
    pos = r.read(0, 0).view(np.float32)
    res_t = r.read(4, 0)
    raw = r.read(5, 0).view(np.float16)
    bg = pos[..., 3] < 0
    assert bg.any()
    assert not res_t[bg].any() and not raw[bg].astype(np.float32).any()          # restir.wgsl:805-811, restir_spatial.wgsl:874-884
    light = pos[..., 3] == 6.0                                                     # quad-light material id
    assert light.any()
    rt = res_t.view(np.float32)[light]
    ru = res_t.view(np.uint32)[light]
    assert np.all(rt[:, 7] == 10.0) and np.all(ru[:, 2] == 1) and np.all(rt[:, 3] == 1.0) and not rt[:, 4:7].any()   # restir.wgsl:543-552
    assert not r.read(3, 0).any()                                                  # static camera: motion == 0 exactly
    acc = r.read(7, 0).view(np.float32)
    assert np.all(acc[..., 3] == 1.0) and np.all(acc[..., :3] >= 0) and not np.isnan(acc).any()
