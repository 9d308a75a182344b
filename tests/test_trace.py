"""T2: acceleration structures against the brute-force loop over all triangles (BASELINE.json configs[0])."""
import numpy as np
import pytest


def _rays(n, seed, scene_box=1.0):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.98, 0.98, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return o, d.astype(np.float32)


def _edge_rays():
    # axis-parallel rays, rays lying exactly in wall planes (d component == 0, origin on the plane), rays through shared
    # triangle edges / vertices of the quads, and rays from the camera to the box corners
    o, d = [], []
    for y in (1.0, -1.0, 0.99):
        for dz in (-1.0, 1.0):
            o.append((1.0, y, 0.3474734)); d.append((-0.76039785, 0.0, dz * 0.64945745))
            o.append((0.0, y, 0.0)); d.append((1.0, 0.0, 0.0)); o.append((0.0, y, 0.0)); d.append((0.0, 0.0, dz))
    for a in range(3):
        for sgn in (-1.0, 1.0):
            v = [0.0, 0.0, 0.0]; v[a] = sgn
            o.append((0.1, -0.2, 0.3)); d.append(tuple(v)); o.append((0.0, 0.0, 0.0)); d.append(tuple(v))
    for cx in (-1.0, 0.0, 1.0):
        for cy in (-1.0, 0.0, 1.0):
            t = np.array([cx, cy, -1.0]) - np.array([0, 0, 3.0]); t /= np.linalg.norm(t)
            o.append((0.0, 0.0, 3.0)); d.append(tuple(t))
    o.append((0.0, -1.0, 0.0)); d.append((0.70710678, 0.0, -0.70710678))       # along the floor diagonal (shared edge)
    return np.array(o, np.float32), np.array(d, np.float32)


@pytest.mark.parametrize("which", ["cornell", "restir"])
def test_bvh_equals_bruteforce(frt, orc, hostcheck, which):
    fs = frt.scenes.create_cornell_box() if which == "cornell" else frt.scenes.create_restir_scene()
    os_ = orc.cornell() if which == "cornell" else orc.restir_scene()
    os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    n = 20000 if which == "cornell" else 3000
    o, d = _rays(n, 7)
    if which == "restir":
        o = (o * np.array([5, 1, 5], np.float32)).astype(np.float32)
    eo, ed = _edge_rays()
    o = np.concatenate([o, eo]); d = np.concatenate([d, ed])
    for tmin, tmax in ((0.001, 100.0), (0.0001, 0.7)):
        tb, ib, uvb, fb, _ = os_.trace_closest(o, d, tmin, tmax, False)          # ground truth: loop over all triangles
        tv, iv, uvv, fv, st = os_.trace_closest(o, d, tmin, tmax, True)          # oracle walk of the imported BVH2
        th, ih, uvh, fh = hostcheck.trace(fs, o, d, tmin, tmax, any_hit=False)   # product traversal (pair nodes), host build
        tq, iq, uvq, fq = hostcheck.trace(fs, o, d, tmin, tmax, any_hit=False, quantized=True)   # ... over the 16-bit pair nodes of the resident kernels
        t4, i4, uv4, f4 = hostcheck.trace(fs, o, d, tmin, tmax, any_hit=False, quantized=2)      # ... over the quad nodes (four children per node)
        t8, i8, uv8, f8 = hostcheck.trace(fs, o, d, tmin, tmax, any_hit=False, quantized=3)      # ... over the 8-wide nodes with grid boxes (trace8)
        for (t, i, uv, f) in ((tv, iv, uvv, fv), (th, ih, uvh, fh), (tq, iq, uvq, fq), (t4, i4, uv4, f4), (t8, i8, uv8, f8)):
            assert np.array_equal(i, ib) and np.array_equal(t, tb)
            hit = ib != 0xFFFFFFFF
            assert np.array_equal(uv[hit], uvb[hit]) and np.array_equal(f[hit], fb[hit])
        assert (ib != 0xFFFFFFFF).mean() > (0.2 if tmax > 1 else 0.02)
        ob = os_.trace_any(o, d, tmin, tmax, False)
        assert np.array_equal(os_.trace_any(o, d, tmin, tmax, True), ob)
        for quantized in (False, True, 2, 3):
            _, ia, _, _ = hostcheck.trace(fs, o, d, tmin, tmax, any_hit=True, quantized=quantized)
            assert np.array_equal((ia != 0xFFFFFFFF).astype(np.uint8), ob)
        assert np.array_equal(ob.astype(bool), tb >= 0)                           # any-hit <=> a closest hit exists


def test_in_plane_reconnection_rays_are_not_lost(frt, orc, hostcheck):
    """Regression: d.y == 0 with the origin on the ceiling plane (spatial-reuse visibility rays in corner pixels) used to
    make the fma-form slab test evaluate inf - inf and drop the ceiling triangles."""
    fs = frt.scenes.create_cornell_box(); os_ = orc.cornell()
    o = np.array([[1.0, 1.0, 0.34747338]] * 4, np.float32)
    d = np.array([[-0.76039785, 0, -0.64945745], [-0.08538333, 0, -0.9963482], [-0.3049969, 0, -0.95235336], [-0.8957124, 0, -0.4446341]], np.float32)
    tmax = np.array([0.32927045, 0.03921813, 0.026023595, 0.24082603], np.float32)
    want = os_.trace_any(o, d, 0.0001, tmax, False)
    for quantized in (False, True, 2, 3):
        _, tri, _, _ = hostcheck.trace(fs, o, d, 0.0001, tmax, any_hit=True, quantized=quantized)
        assert np.array_equal((tri != 0xFFFFFFFF).astype(np.uint8), want)


def test_hit_semantics(orc):
    s = orc.cornell()
    o = np.array([[0.8, 0.5, 3], [0, 0, 3], [0, -0.999, 0]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 1, 0]], np.float32)
    t, tri, uv, fr, _ = s.trace_closest(o, d, 0.001, 1000.0, False)
    assert t[0] == 4.0 and fr[0] == 1          # back wall at z = -1, front face towards the camera
    assert t[1] == -1.0 and tri[1] == 0xFFFFFFFF   # open front: miss
    ti = s.get("tri_instance")
    assert ti[tri[2]] == 5 and abs(t[2] - 1.989) < 1e-5 and fr[2] == 1    # light quad at y = 0.99 faces down
    # tmin / tmax are exclusive
    t2, _, _, _, _ = s.trace_closest(o[:1], d[:1], 0.001, 4.0, False)
    assert t2[0] == -1.0


def test_quad_tree_covers_the_scene_and_fits_the_stack(frt, hostcheck):
    """The quad tree the default kernels walk (frt_bvh.cpp: build_quad_nodes): every triangle slot under exactly one leaf, the same leaves as
    the binary tree, and the deepest possible traversal stack — found by walking every root-to-leaf path with all children hit — within
    the 32 entries of the kernels' LDS stacks, also for a tree at the builder's depth limit."""
    fs = frt.scenes.create_cornell_box()
    q = hostcheck.quad_stats(fs); b = fs.bvh_stats()
    assert q["leaves"] == b["leaves"] and q["triangles"] == fs.get("bvh2_tri_index").size
    assert q["stack_walked"] == q["stack_need"] <= 31 and q["nodes"] < b["pair_nodes"] * 0.6 and q["children_x100"] > 300
    # a scene large enough for the builder's insertion-optimisation pass (frt_bvh_opt.hpp; 8192 triangles and more): the re-emitted tree must be a
    # valid canonical BVH2 — children adjacent and behind their parent, every box containing its children's, every triangle under one leaf —
    # within the depth the traversal stack is sized for, and its quad tree must fit the stack like any other
    big = frt.scenes.create_restir_scene()
    nodes = big.get("bvh2_nodes"); idx = big.get("bvh2_tri_index"); bb = big.bvh_stats()
    assert idx.size >= 8192 and bb["depth"] <= 30
    lo = nodes[:, 0:3].view(np.float32); hi = nodes[:, 4:7].view(np.float32); left = nodes[:, 3]; cnt = nodes[:, 7]
    inner = np.nonzero(cnt == 0)[0]
    assert np.all(left[inner] > inner) and np.all(left[inner] + 1 < len(nodes))
    for c in (left[inner], left[inner] + 1):
        assert np.all(lo[inner] <= lo[c]) and np.all(hi[inner] >= hi[c])
    kids = np.concatenate([left[inner], left[inner] + 1])
    assert np.array_equal(np.sort(kids), np.arange(1, len(nodes)))              # every node but the root is the child of exactly one node
    leaves = np.nonzero(cnt > 0)[0]
    covered = np.concatenate([np.arange(left[l], left[l] + cnt[l]) for l in leaves])
    assert np.array_equal(np.sort(covered), np.arange(idx.size)) and np.array_equal(np.sort(idx), np.arange(idx.size))
    qb = hostcheck.quad_stats(big)
    assert qb["stack_walked"] == qb["stack_need"] <= 31 and qb["leaves"] == bb["leaves"] and qb["triangles"] == idx.size
    # a deep, lopsided tree: triangle sizes and positions in geometric progression make the SAH peel one triangle per level
    n = 120
    pos = np.zeros((3 * n, 4), np.float32); pos[:, 3] = 1.0
    for i in range(n):
        x = 2.0 ** i
        pos[3 * i, :3] = (x, 0, 0); pos[3 * i + 1, :3] = (x * 1.05, 0.1 * x, 0); pos[3 * i + 2, :3] = (x, 0, 0.1 * x)
    att = np.zeros((3 * n, 8), np.float32)
    geo = frt.geometry.Geometry(pos, att, np.arange(3 * n, dtype=np.uint32))
    sb = frt.SceneBuilder()
    mesh = sb.add_mesh(geo)
    mat = sb.add_material(frt.material_new([0.8, 0.8, 0.8, 1.0]))
    sb.add_instance(mesh, mat, np.eye(4, dtype=np.float32).T)
    deep = sb.build()
    qd = hostcheck.quad_stats(deep); bd = deep.bvh_stats()
    assert bd["depth"] >= 24, bd
    assert qd["stack_walked"] == qd["stack_need"] <= 31 and qd["leaves"] == bd["leaves"] and qd["triangles"] == n, (qd, bd)
    rng = np.random.default_rng(3)
    pick = rng.integers(0, n, 2000)
    tgt = (pos[3 * pick, :3] + pos[3 * pick + 1, :3] + pos[3 * pick + 2, :3]) / 3.0
    o = (tgt + np.array([0.3, 1.0, 0.2], np.float32) * (2.0 ** pick)[:, None]).astype(np.float32)
    d = (tgt - o).astype(np.float64); d /= np.linalg.norm(d, axis=1, keepdims=True)
    t2, i2, _, _ = hostcheck.trace(deep, o, d.astype(np.float32), 0.0, 3e38, any_hit=False)
    t4, i4, _, _ = hostcheck.trace(deep, o, d.astype(np.float32), 0.0, 3e38, any_hit=False, quantized=2)
    assert np.array_equal(i2, i4) and np.array_equal(t2, t4) and (i2 != 0xFFFFFFFF).mean() > 0.2
    # the 8-wide tree of the same lopsided scene (coordinates over 36 binary orders of magnitude: every node has its own grid)
    wd = hostcheck.wide8_stats(deep)
    assert wd["defects"] == 0 and wd["triangles"] == n and wd["stack_walked"] <= wd["stack_need"] <= 8, wd
    t8, i8, _, _ = hostcheck.trace(deep, o, d.astype(np.float32), 0.0, 3e38, any_hit=False, quantized=3)
    assert np.array_equal(i2, i8) and np.array_equal(t2, t8)


def test_wide8_tree_covers_the_scene_and_needs_a_shallow_stack(frt, hostcheck):
    """The 8-wide tree with grid boxes (csrc/frt_bvh8.hpp; frt_trace.hpp: trace8): every triangle slot under exactly one leaf child, the same leaves as
    the binary tree, every grid box containing its child's float box (the planes trace8 reconstructs bracket it: the wide tree may only prune less),
    inner children numbered contiguously in slot order, and a traversal stack as deep as the TREE (one word per level), not as the sum of its fan-outs."""
    for make, max_need in ((frt.scenes.create_cornell_box, 5), (frt.scenes.create_restir_scene, 6)):
        fs = make()
        w = hostcheck.wide8_stats(fs); b = fs.bvh_stats(); t = fs.tree_stats(); q = hostcheck.quad_stats(fs)
        assert w["defects"] == 0, w
        assert w["leaves"] == b["leaves"] and w["triangles"] == fs.counts()["tris"] == t["wide8_tri_slots"]
        assert w["nodes"] == t["wide8_nodes"] and w["stack_need"] == t["wide8_stack_need"] and w["levels"] == t["wide8_depth"]
        assert w["stack_walked"] <= w["stack_need"] <= max_need < q["stack_need"]
        assert w["nodes"] < q["nodes"] * 0.65 and w["children_x100"] > 400
        # the triangle slots are a permutation of the binary tree's
        a = fs.get("tri_slots8"); c = fs.get("tri_slots")
        assert np.array_equal(np.sort(a[:, 3].view(np.uint32)), np.arange(a.shape[0])) and np.array_equal(np.sort(c[:, 3].view(np.uint32)), np.arange(c.shape[0]))
        order = np.argsort(a[:, 3].view(np.uint32)); order_c = np.argsort(c[:, 3].view(np.uint32))
        assert a[order].tobytes() == c[order_c].tobytes()


def test_heavily_overlapping_geometry_builds_in_bounded_time(frt, hostcheck):
    """20,000 nearly coincident triangles: every box overlaps every other, so the insertion-optimisation pass (frt_bvh_opt.hpp, scenes of 8,192
    triangles and more) gets no pruning from its bound; its search is cut after a fixed number of candidates, which keeps the pass linear. The tree
    must still be a valid one: pair-node and quad-node walks agree and the rays that aim at the pile hit it."""
    import time
    n = 20000
    rng = np.random.default_rng(1)
    pos = np.zeros((3 * n, 4), np.float32); pos[:, 3] = 1.0
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    pos[:, :3] = np.tile(base, (n, 1)) + rng.normal(0, 1e-3, (3 * n, 3)).astype(np.float32)
    geo = frt.geometry.Geometry(pos, np.zeros((3 * n, 8), np.float32), np.arange(3 * n, dtype=np.uint32))
    sb = frt.SceneBuilder()
    mesh = sb.add_mesh(geo)
    mat = sb.add_material(frt.material_new([0.8, 0.8, 0.8, 1.0]))
    sb.add_instance(mesh, mat, np.eye(4, dtype=np.float32).T)
    t0 = time.perf_counter()
    scene = sb.build()
    assert time.perf_counter() - t0 < 30.0
    st = scene.bvh_stats(); q = hostcheck.quad_stats(scene)
    assert st["depth"] <= 30 and q["stack_walked"] == q["stack_need"] <= 31 and q["triangles"] == n
    o = np.tile(np.array([[0.25, 0.25, 1.0]], np.float32), (64, 1)) + rng.normal(0, 0.05, (64, 3)).astype(np.float32)
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (64, 1))
    t2, i2, _, _ = hostcheck.trace(scene, o, d, 0.0, 100.0, any_hit=False)
    t4, i4, _, _ = hostcheck.trace(scene, o, d, 0.0, 100.0, any_hit=False, quantized=2)
    assert np.array_equal(i2, i4) and np.array_equal(t2, t4) and (i2 != 0xFFFFFFFF).mean() > 0.5
    w = hostcheck.wide8_stats(scene)
    if w["nodes"]:      # (a pile of coincident boxes may need a deeper stack than trace8 has: the renderer then keeps the quad walk)
        assert w["defects"] == 0 and w["triangles"] == n
        t8, i8, _, _ = hostcheck.trace(scene, o, d, 0.0, 100.0, any_hit=False, quantized=3)
        assert np.array_equal(i2, i8) and np.array_equal(t2, t8)
