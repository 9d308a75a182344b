"""T1: the product's host scene model (libfrt.so, no GPU needed) against the oracle's restatement, bit for bit."""
import numpy as np
import pytest


@pytest.mark.parametrize("which", ["cornell", "restir"])
def test_scene_arrays_bit_equal(frt, orc, which):
    fs = frt.scenes.create_cornell_box() if which == "cornell" else frt.scenes.create_restir_scene()
    os_ = orc.cornell() if which == "cornell" else orc.restir_scene()
    fc, oc = fs.counts(), os_.counts()
    for k in ("tris", "instances", "materials", "lights", "meshes", "attributes", "indices"):
        assert fc[k] == oc[k]
    for k in ("tris", "tri_instance", "materials", "lights", "attributes", "indices", "mesh_infos", "instances"):
        assert fs.get(k).tobytes() == os_.get(k).tobytes(), k


def test_restir_scene_counts(frt):
    c = frt.scenes.create_restir_scene().counts()
    # scenes.rs:133-223: 2 planes + 100 icosphere(2) lights + cube
    assert c["lights"] == 100 and c["instances"] == 103 and c["materials"] == 103 and c["tris"] == 2 * 2 + 100 * 320 + 12


@pytest.mark.parametrize("which,sub", [(0, 0), (1, 0), (2, 3), (3, 0), (2, 1)])
def test_geometry_generators_bit_equal(frt, orc, which, sub):
    g = {0: frt.geometry.create_plane, 1: frt.geometry.create_cube, 3: frt.geometry.create_crystal}.get(which)
    geo = g() if g else frt.geometry.create_sphere(sub)
    pos, att, idx = orc.mesh(which, sub)
    assert geo.positions.tobytes() == pos.tobytes() and geo.attributes.tobytes() == att.tobytes() and geo.indices.tobytes() == idx.tobytes()


def test_plane_front_face_is_plus_y(frt):
    # geometry.rs:87, :114: triangle 0 of the plane has geometric normal +Y = shading normal
    g = frt.geometry.create_plane()
    p = g.positions[:, :3]
    i = g.indices[:3]
    n = np.cross(p[i[1]] - p[i[0]], p[i[2]] - p[i[0]])
    assert n[1] > 0 and n[0] == 0 and n[2] == 0


@pytest.mark.parametrize("aspect", [1.0, 16.0 / 9.0])
def test_camera_uniform_bit_equal(frt, orc, aspect):
    for frame in (0, 5):
        a = bytes(frt.CameraController().build_uniform(aspect, frame, 2))
        assert a == orc.camera(aspect, frame, 2).tobytes()
    cu = frt.CameraController().build_uniform(aspect, 3, 2)
    assert list(cu.view_pos) == [0.0, 0.0, 3.0, 1.0] and cu.frame_count == 3 and cu.num_lights == 2
    assert list(cu.view_proj) == list(cu.prev_view_proj)            # static camera, jitter == 0 (camera.rs:202-203, :234-238)
    vi = np.array(cu.view_inverse, np.float32).reshape(4, 4)
    np.testing.assert_allclose(vi[3, :3], [0, 0, 3], atol=1e-6)      # origin = view_inv[3].xyz (gbuffer.wgsl:103)


def test_builder_api_matches_factory(frt, orc):
    """Build the Cornell Box through the public SceneBuilder calls (the drop-in surface) and compare with the factory."""
    import math
    b = frt.SceneBuilder()
    plane = b.add_mesh(frt.geometry.create_plane()); cube = b.add_mesh(frt.geometry.create_cube())
    sphere = b.add_mesh(frt.geometry.create_sphere(3)); crystal = b.add_mesh(frt.geometry.create_crystal())
    ref = frt.scenes.create_cornell_box()
    mats = ref.get("materials")
    inst = ref.get("instances")
    for k in range(6):
        m = frt.Material.from_buffer_copy(mats[k].tobytes()); b.add_material(m)
    # instances in reference order; lights via register_* (which add materials 6 and 7)
    for k in range(9):
        mesh, mat = int(inst[k, 0]), int(inst[k, 1])
        xf = inst[k, 5:21].view(np.float32)
        if k == 5:
            b.register_quad_light(mesh, xf, [1.0, 1.0, 1.0], 10.0)
        elif k == 7:
            b.register_sphere_light(mesh, xf, np.array([0.02, 0.02, 0.9], np.float32), 10.0)
        else:
            b.add_instance(mesh, mat, xf, 0x1)
    b.build()
    for k in ("tris", "tri_instance", "materials", "lights", "instances", "bvh2_nodes", "bvh2_tri_index"):
        assert b.get(k).tobytes() == ref.get(k).tobytes(), k


def test_bvh_is_valid(frt):
    for s in (frt.scenes.create_cornell_box(), frt.scenes.create_restir_scene()):
        nodes = s.get("bvh2_nodes"); idx = s.get("bvh2_tri_index"); tris = s.get("tris")
        st = s.bvh_stats()
        assert st["depth"] <= 30 and st["max_leaf"] <= 4
        assert sorted(idx.tolist()) == list(range(len(tris)))
        f = nodes.view(np.float32)
        v0 = tris[:, 0:3]; v1 = v0 + tris[:, 3:6]; v2 = v0 + tris[:, 6:9]
        lo = np.minimum(np.minimum(v0, v1), v2); hi = np.maximum(np.maximum(v0, v1), v2)
        seen = np.zeros(len(tris), int)
        stack = [(0, 1)]
        maxd = 0
        while stack:
            n, d = stack.pop(); maxd = max(maxd, d)
            bmin, bmax = f[n, 0:3], f[n, 4:7]
            left, count = int(nodes[n, 3]), int(nodes[n, 7])
            if count:
                ids = idx[left:left + count]; seen[ids] += 1
                assert np.all(lo[ids] > bmin) and np.all(hi[ids] < bmax)          # strictly inside the padded box
            else:
                for c in (left, left + 1):
                    assert np.all(f[c, 0:3] >= bmin) and np.all(f[c, 4:7] <= bmax)
                    stack.append((c, d + 1))
        assert np.all(seen == 1) and maxd == st["depth"]


def test_camera_build_uniform_matches_oracle(frt, orc):
    """camera.rs:207-256 for arbitrary controller states, jitter (projection shear, :224-228) and previous view-projection
    (:233-238), and get_halton_jitter (:182-205): product host code vs the oracle's separate restatement, bit for bit."""
    import numpy as np
    rng = np.random.default_rng(7)
    for k in range(40):
        pos = rng.uniform(-2, 2, 3).astype(np.float32); yaw = float(np.float32(rng.uniform(-3.2, 3.2))); pitch = float(np.float32(rng.uniform(-1.5, 1.5)))
        aspect = float(np.float32(rng.uniform(0.5, 2.5))); W, H = int(rng.integers(16, 4000)), int(rng.integers(16, 2200))
        jit = frt.CameraController.get_halton_jitter(k, W, H, 1.0 if k % 3 else 0.0)
        assert jit == orc.halton_jitter(k, W, H, 1.0 if k % 3 else 0.0)
        ctl = frt.CameraController(pos, yaw, pitch)
        prev = None
        for f in range(3):
            cu = ctl.build_uniform(aspect, f, 2, jit)
            want, unj = orc.camera_build(pos, yaw, pitch, prev, aspect, f, 2, jit)
            assert bytes(cu) == want.tobytes() and np.array_equal(np.asarray(ctl.unjittered_view_proj, np.float32), unj)
            ctl.commit_frame(); prev = unj           # state.rs:172
            ctl.position[0] += 0.05; pos = np.asarray(ctl.position, np.float32)
    # the default pose is the reference's initial controller state (camera.rs:40-42)
    assert bytes(frt.CameraController().build_uniform(16 / 9, 5, 2)) == orc.camera(16 / 9, 5, 2).tobytes()
    # jitter shears exactly two entries of the projection: view_proj changes, view_inverse does not
    a = frt.CameraController().build_uniform(1.5, 0, 2); b = frt.CameraController().build_uniform(1.5, 0, 2, (0.01, -0.02))
    assert list(a.view_inverse) == list(b.view_inverse) and list(a.view_proj) != list(b.view_proj) and list(a.prev_view_proj) == list(b.prev_view_proj)
