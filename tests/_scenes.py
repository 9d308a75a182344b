"""Synthetic mesh scenes for BASELINE.json configs[3] / [4], built through the PUBLIC builder calls of both the product
(frt.SceneBuilder) and the oracle (orc_scene_*), from the same numpy arrays.

The Stanford Bunny and Sponza are not available offline, so stand-ins of comparable size are used:
  config 3: icosphere(6) (81,920 triangles) displaced radially by a fixed sum of three sinusoids, inside the Cornell walls
  config 4: a colonnade of 12 x 4 icosphere(4) "capitals" (245,760 triangles) + cube columns in a 10 x 4 x 4 hall, one quad light
"""
import ctypes as C
import numpy as np


def _mat(tx, ty, tz, sx, sy, sz):
    m = np.zeros((4, 4), np.float32)          # column-major: m[c, r]
    m[0, 0], m[1, 1], m[2, 2], m[3, 3] = sx, sy, sz, 1.0
    m[3, 0], m[3, 1], m[3, 2] = tx, ty, tz
    return m.reshape(16)


class DualBuilder:
    """Issues every builder call to both libraries."""

    def __init__(self, frt, orc):
        self.frt, self.orc = frt, orc
        self.fb = frt.SceneBuilder()
        self.oh = orc.L.orc_scene_create()

    def add_mesh(self, pos4, attrs, idx):
        pos4 = np.ascontiguousarray(pos4, np.float32); attrs = np.ascontiguousarray(attrs, np.float32); idx = np.ascontiguousarray(idx, np.uint32)
        g = self.frt.geometry.Geometry(pos4, attrs, idx)
        a = self.fb.add_mesh(g)
        b = self.orc.L.orc_scene_add_mesh(self.oh, pos4.ctypes.data, pos4.shape[0], attrs.ctypes.data, idx.ctypes.data, idx.size)
        assert a == b
        return a

    def add_material(self, mat):
        a = self.fb.add_material(mat)
        b = self.orc.L.orc_scene_add_material(self.oh, C.byref(mat))
        assert a == b
        return a

    def add_instance(self, mesh, mat, m):
        m = np.ascontiguousarray(m, np.float32)
        self.fb.add_instance(mesh, mat, m)
        self.orc.L.orc_scene_add_instance(self.oh, mesh, mat, m.ctypes.data)

    def add_light(self, light):
        self.fb.add_light(light)
        self.orc.L.orc_scene_add_light(self.oh, C.byref(light))

    def build(self, share_bvh=True):
        """share_bvh: the oracle walks the PRODUCT's canonical BVH2 (north_star: "CPU tracer over the same BVH"). False: the oracle is handed nothing
        the product built — its renderer must then be created with use_bvh = False (brute force over all triangles)."""
        from _oracle import OrcScene
        self.fb.build()
        self.orc.L.orc_scene_build(self.oh)
        osc = OrcScene(self.orc, self.oh)
        if share_bvh:
            osc.set_bvh(self.fb.get("bvh2_nodes"), self.fb.get("bvh2_tri_index"))
        return self.fb, osc


def _geo(frt, name, *a):
    g = getattr(frt.geometry, name)(*a)
    return g.positions.copy(), g.attributes.copy(), g.indices.copy()


def _quad_light(frt, pos, half, emission):
    l = frt.Light()
    l.position[:] = pos; l.type_ = 0
    l.u[:] = (half, 0, 0); l.v[:] = (0, 0, half); l.area = 4.0 * half * half
    l.emission[:] = emission
    return l


def _emissive(frt, light_index, rgb, intensity):
    m = frt.material_new([1, 1, 1, 1])
    m.light_index = light_index
    m.emissive_factor[:] = [c * intensity for c in rgb]
    m.tex_info_0 = 0xFFFF0000
    return m


def bumpy_sphere_in_box(frt, orc, subdiv=6, share_bvh=True):
    """config 3 stand-in. subdiv 6 -> 81,920 triangles (+ 12 wall/light triangles)."""
    b = DualBuilder(frt, orc)
    plane = b.add_mesh(*_geo(frt, "create_plane"))
    pos, att, idx = _geo(frt, "create_sphere", subdiv)
    p = pos[:, :3].astype(np.float64) * 2.0           # unit directions (radius 0.5 -> 1)
    disp = 1.0 + 0.08 * np.sin(7.0 * p[:, 0]) * np.sin(5.0 * p[:, 1]) + 0.05 * np.sin(11.0 * p[:, 2] + 1.0) + 0.03 * np.sin(17.0 * p[:, 0] * p[:, 1])
    pos[:, :3] = (p * disp[:, None] * 0.5).astype(np.float32)
    blob = b.add_mesh(pos, att, idx)
    white = b.add_material(frt.material_new([0.73, 0.73, 0.73, 1.0]))
    red = b.add_material(frt.material_new([0.65, 0.05, 0.05, 1.0]))
    green = b.add_material(frt.material_new([0.12, 0.45, 0.15, 1.0]))
    lm = b.add_material(_emissive(frt, 0, (1, 1, 1), 10.0))
    ref = frt.scenes.create_cornell_box().get("instances")      # reuse the wall transforms of scenes.rs:50-90
    for k, mat in ((0, white), (1, white), (2, white), (3, red), (4, green)):
        b.add_instance(plane, mat, ref[k, 5:21].view(np.float32))
    b.add_instance(plane, lm, ref[5, 5:21].view(np.float32))
    b.add_light(_quad_light(frt, (0, 0.99, 0), 0.25, (1, 1, 1, 10)))
    b.add_instance(blob, white, _mat(0, -0.4, 0, 0.6, 0.6, 0.6))
    return b.build(share_bvh)


def colonnade(frt, orc, nx=12, nz=4, subdiv=4):
    """config 4 stand-in. 12 x 4 x icosphere(4) = 245,760 triangles + columns + hall."""
    b = DualBuilder(frt, orc)
    plane = b.add_mesh(*_geo(frt, "create_plane"))
    cube = b.add_mesh(*_geo(frt, "create_cube"))
    sph = b.add_mesh(*_geo(frt, "create_sphere", subdiv))
    stone = b.add_material(frt.material_new([0.7, 0.68, 0.6, 1.0]))
    floor = frt.material_new([0.73, 0.73, 0.73, 1.0]); floor.roughness = 0.99; floor.tex_info_0 = 0xFFFF0001
    floor = b.add_material(floor)
    lm = b.add_material(_emissive(frt, 0, (1, 0.95, 0.85), 30.0))
    b.add_instance(plane, floor, _mat(0, -1, 0, 10, 1, 4))
    ceil = _mat(0, 1.0, 0, 10, 1, 4); ceil[5] = -1.0; ceil[10] = -4.0        # flip about x: normal down
    b.add_instance(plane, stone, ceil)
    lq = _mat(0, 0.98, 0, 2.0, 1, 1.0); lq[5] = -1.0; lq[10] = -1.0
    b.add_instance(plane, lm, lq)
    l = _quad_light(frt, (0, 0.98, 0), 0.5, (1, 0.95, 0.85, 30)); l.u[:] = (1.0, 0, 0); l.v[:] = (0, 0, 0.5); l.area = 2.0
    b.add_light(l)
    for i in range(nx):
        for k in range(nz):
            x = (i - (nx - 1) / 2) * 0.8
            z = (k - (nz - 1) / 2) * 0.9
            b.add_instance(cube, stone, _mat(x, -0.35, z, 0.18, 1.3, 0.18))
            b.add_instance(sph, stone, _mat(x, 0.42, z, 0.45, 0.3, 0.45))
    return b.build()


def moving_camera_uniforms(frt, aspect, num_lights, frames, step=(0.02, 0.01, -0.03), yaw_step=0.01):
    """CameraUniform sequence for a camera that moves and turns every frame (the half of temporal / post logic the static
    benchmark never reaches: non-zero motion vectors, reprojection to other pixels, the TAA clamp branch).
    Built in numpy following camera.rs:207-256: view_proj, inverses, and prev_view_proj = the previous frame's view_proj."""
    out = []
    prev_vp = None
    for f in range(frames):
        eye = np.array([0.0 + step[0] * f, 0.0 + step[1] * f, 3.0 + step[2] * f])
        yaw = -np.pi / 2 + yaw_step * f
        fwd = np.array([np.cos(yaw), 0.0, np.sin(yaw)])
        s = np.cross(fwd, [0, 1, 0]); s /= np.linalg.norm(s)
        u = np.cross(s, fwd)
        view = np.eye(4)
        view[0, :3], view[1, :3], view[2, :3] = s, u, -fwd
        view[0, 3], view[1, 3], view[2, 3] = -eye @ s, -eye @ u, eye @ fwd
        h = 1.0 / np.tan(np.radians(45.0) / 2); r = 100.0 / (0.1 - 100.0)
        proj = np.zeros((4, 4)); proj[0, 0] = h / aspect; proj[1, 1] = h; proj[2, 2] = r; proj[2, 3] = r * 0.1; proj[3, 2] = -1.0
        vp = proj @ view
        cu = frt.CameraUniform()
        col = lambda m: np.asarray(m, np.float32).T.reshape(16)      # row-major math -> column-major storage
        cu.view_proj[:] = col(vp); cu.view_inverse[:] = col(np.linalg.inv(view)); cu.proj_inverse[:] = col(np.linalg.inv(proj))
        cu.prev_view_proj[:] = col(prev_vp if prev_vp is not None else vp)
        cu.view_pos[:] = [eye[0], eye[1], eye[2], 1.0]
        cu.frame_count, cu.num_lights = f, num_lights
        out.append(cu)
        prev_vp = vp
    return out


def gltf_scene(frt, orc, path, model_transform, light_transform):
    """scenes.rs:246-322 through the product (frt.scenes.create_gltf_scene) and, for the oracle, the same scene re-issued call by
    call from the loaded model: the meshes from frt.loader, materials / lights / instances as the product built them, and the
    texture layers in the order tests/_gltf.py::assign_layers (an independent restatement of builder.rs:191-292) derives."""
    import _gltf
    from _oracle import OrcScene
    fs = frt.scenes.create_gltf_scene(path, model_transform, light_transform)
    model = frt.loader.load_gltf(path)
    oh = orc.L.orc_scene_create()

    def add_mesh(g):
        pos = np.ascontiguousarray(g.positions, np.float32); att = np.ascontiguousarray(g.attributes, np.float32); idx = np.ascontiguousarray(g.indices, np.uint32)
        return orc.L.orc_scene_add_mesh(oh, pos.ctypes.data, pos.shape[0], att.ctypes.data, idx.ctypes.data, idx.size)
    add_mesh(frt.geometry.create_plane()); add_mesh(frt.geometry.create_plane())
    n = model.counts()
    for i in range(n["geometries"]):
        add_mesh(model.geometry(i)[0])

    def tex(m):
        f = lambda v: None if v == 0xFFFF else v
        return (f(m.tex_info_0 & 0xFFFF), f(m.tex_info_0 >> 16), f(m.tex_info_1 & 0xFFFF), f(m.tex_info_1 >> 16), f(m.tex_info_2 & 0xFFFF))
    _, corder, dorder = _gltf.assign_layers([tex(model.material(i)) for i in range(n["materials"])], 3, 3)
    for kind, order in ((0, corder), (1, dorder)):
        for img in order:
            t = np.ascontiguousarray(model.image(img))
            orc.L.orc_scene_add_texture(oh, kind, t.ctypes.data)
    for row in fs.get("materials"):
        r = np.ascontiguousarray(row); orc.L.orc_scene_add_material(oh, r.ctypes.data)
    for row in fs.get("lights"):
        r = np.ascontiguousarray(row); orc.L.orc_scene_add_light(oh, r.ctypes.data)
    for row in fs.get("instances"):
        m = np.ascontiguousarray(row[5:21]); orc.L.orc_scene_add_instance(oh, int(row[0]), int(row[1]), m.ctypes.data)
    orc.L.orc_scene_build(oh)
    osc = OrcScene(orc, oh)
    osc.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    return fs, osc


def random_scene(frt, orc, seed):
    """A closed room with randomly placed, rotated and (sometimes mirrored) cubes / icospheres / crystals in random diffuse, glossy,
    metal, glass and textured materials, a quad light and up to two sphere lights: material / lobe / light-type combinations the two
    named scenes do not contain. Deterministic in `seed`."""
    rng = np.random.default_rng(seed)
    b = DualBuilder(frt, orc)
    plane = b.add_mesh(*_geo(frt, "create_plane"))
    shapes = [b.add_mesh(*_geo(frt, "create_cube")), b.add_mesh(*_geo(frt, "create_sphere", 2)), b.add_mesh(*_geo(frt, "create_crystal"))]
    sphere = shapes[1]

    def material(kind):
        m = frt.material_new([float(x) for x in rng.uniform(0.15, 0.95, 3)] + [1.0])
        if kind == "diffuse": m.roughness = float(rng.uniform(0.3, 1.0))
        elif kind == "glossy": m.roughness = float(rng.uniform(0.06, 0.3))
        elif kind == "metal": m.metallic = 1.0; m.roughness = float(rng.uniform(0.02, 0.5))
        elif kind == "glass": m.metallic = 0.0; m.roughness = 0.0; m.ior = float(rng.uniform(1.3, 1.7)); m.transmission = 1.0
        elif kind == "checker": m.roughness = float(rng.uniform(0.4, 1.0)); m.tex_info_0 = 0xFFFF0001
        return m
    kinds = ["diffuse", "glossy", "metal", "glass", "checker", "diffuse"]
    mats = [b.add_material(material(k)) for k in kinds]
    walls = [b.add_material(material("diffuse")) for _ in range(3)] + [b.add_material(material("checker"))]
    ref = frt.scenes.create_cornell_box().get("instances")
    for k in range(5):
        b.add_instance(plane, walls[int(rng.integers(0, len(walls)))], ref[k, 5:21].view(np.float32))
    # lights: the Cornell quad light + 0..2 small sphere lights
    lights = 0
    b.add_instance(plane, b.add_material(_emissive(frt, lights, (1, 1, 1), 10.0)), ref[5, 5:21].view(np.float32))
    b.add_light(_quad_light(frt, (0, 0.99, 0), 0.25, (1, 1, 1, 10))); lights += 1
    for _ in range(int(rng.integers(0, 3))):
        pos = [float(x) for x in rng.uniform(-0.6, 0.6, 3)]
        radius = float(np.float32(rng.uniform(0.03, 0.08)))
        rgb = [float(x) for x in rng.uniform(0.05, 1.0, 3)]
        l = frt.Light()
        l.position[:] = pos; l.type_ = 1; l.u[:] = (0, 0, 0); l.v[:] = (radius, 0, 0)
        l.area = float(np.float32(4.0 * np.pi) * np.float32(radius) * np.float32(radius)); l.emission[:] = rgb + [12.0]
        b.add_instance(sphere, b.add_material(_emissive(frt, lights, rgb, 12.0)), _mat(*pos, 2 * radius, 2 * radius, 2 * radius))
        b.add_light(l); lights += 1
    for _ in range(7):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        s = rng.uniform(0.15, 0.45, 3)
        if rng.random() < 0.25: s[0] = -s[0]                       # mirrored instance: front faces flip (frt_scene.hpp InstanceRec.flip)
        M = np.eye(4); M[:3, :3] = R * s[None, :]; M[:3, 3] = rng.uniform(-0.6, 0.6, 3) * (1, 0.8, 1)
        b.add_instance(shapes[int(rng.integers(0, 3))], mats[int(rng.integers(0, len(mats)))], np.asarray(M.T, np.float32).reshape(16))
    fs, os_ = b.build()
    return fs, os_, lights
