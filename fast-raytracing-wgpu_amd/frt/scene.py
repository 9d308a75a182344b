"""SceneBuilder mirror (src/scene/builder.rs) over frt_scene_*."""
import ctypes as C
import numpy as np
from ._lib import lib, check, Material, Light, FrtError


def material_new(base_color):
    """Material::new (material.rs:31-47)."""
    m = Material()
    a = np.asarray(base_color, np.float32)
    lib().frt_material_default(a.ctypes.data, C.byref(m))
    return m


class SceneBuilder:
    def __init__(self, handle=None):
        self._destroy = lib().frt_scene_destroy
        self._h = handle if handle is not None else lib().frt_scene_create()
        if not self._h:
            raise FrtError("scene creation failed: " + lib().frt_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)     # bound at construction: module globals may already be gone at interpreter exit
            self._h = None

    # builder.rs:123
    def add_mesh(self, geo):
        pos = np.ascontiguousarray(geo.positions, np.float32)
        att = np.ascontiguousarray(geo.attributes, np.float32)
        idx = np.ascontiguousarray(geo.indices, np.uint32)
        return check(lib().frt_scene_add_mesh(self._h, pos.ctypes.data, pos.shape[0], att.ctypes.data, idx.ctypes.data, idx.size))

    # builder.rs:117
    def add_material(self, mat):
        return check(lib().frt_scene_add_material(self._h, C.byref(mat)))

    # builder.rs:181 (the mask argument is ignored by the reference as well)
    def add_instance(self, mesh_id, mat_id, transform_colmajor, _mask=0x1):
        m = np.ascontiguousarray(transform_colmajor, np.float32).reshape(16)
        return check(lib().frt_scene_add_instance(self._h, mesh_id, mat_id, m.ctypes.data))

    def add_light(self, light):
        return check(lib().frt_scene_add_light(self._h, C.byref(light)))

    # builder.rs:316 / :353
    def register_quad_light(self, mesh_id, transform_colmajor, color, intensity):
        m = np.ascontiguousarray(transform_colmajor, np.float32).reshape(16)
        c = np.asarray(color, np.float32)
        return check(lib().frt_scene_register_quad_light(self._h, mesh_id, m.ctypes.data, c.ctypes.data, float(intensity)))

    def register_sphere_light(self, mesh_id, transform_colmajor, color, intensity):
        m = np.ascontiguousarray(transform_colmajor, np.float32).reshape(16)
        c = np.asarray(color, np.float32)
        return check(lib().frt_scene_register_sphere_light(self._h, mesh_id, m.ctypes.data, c.ctypes.data, float(intensity)))

    # builder.rs:93 / :105
    def add_color_texture(self, rgba8):
        t = np.ascontiguousarray(rgba8, np.uint8).reshape(1024, 1024, 4)
        return check(lib().frt_scene_add_texture(self._h, 0, t.ctypes.data))

    def add_data_texture(self, rgba8):
        t = np.ascontiguousarray(rgba8, np.uint8).reshape(1024, 1024, 4)
        return check(lib().frt_scene_add_texture(self._h, 1, t.ctypes.data))

    # builder.rs:191-292 / :294-300 / :302-314 — `model` is a frt.loader.Model
    def add_gltf_materials(self, model):
        ids = np.zeros(max(model.counts()["materials"], 1), np.uint32)
        n = check(lib().frt_scene_add_gltf_materials(self._h, model._h, ids.ctypes.data))
        return ids[:n].copy()

    def add_gltf_meshes(self, model):
        ids = np.zeros(max(model.counts()["geometries"], 1), np.uint32)
        n = check(lib().frt_scene_add_gltf_meshes(self._h, model._h, ids.ctypes.data))
        return ids[:n].copy()

    def add_gltf_instances(self, model, mesh_ids, mat_ids, transform_colmajor):
        me = np.ascontiguousarray(mesh_ids, np.uint32); ma = np.ascontiguousarray(mat_ids, np.uint32)
        m = np.ascontiguousarray(transform_colmajor, np.float32).reshape(16)
        return check(lib().frt_scene_add_gltf_instances(self._h, model._h, me.ctypes.data, me.size, ma.ctypes.data, ma.size, m.ctypes.data))

    # builder.rs:431
    def build(self):
        check(lib().frt_scene_build(self._h))
        return self

    # ---- introspection
    def counts(self):
        c = (C.c_uint32 * 8)()
        check(lib().frt_scene_counts(self._h, c))
        return dict(zip(("tris", "instances", "materials", "lights", "meshes", "attributes", "indices", "bvh2_nodes"), list(c)))

    @property
    def num_lights(self):
        return self.counts()["lights"]

    def tree_stats(self):
        s = (C.c_uint32 * 8)()
        check(lib().frt_scene_tree_stats(self._h, s))
        return dict(zip(("quad_nodes", "quad_stack_need", "wide8_nodes", "wide8_stack_need", "wide8_depth", "wide8_children", "wide8_tri_slots", "quad_fold"), list(s)))

    def get(self, what):
        n = self.counts()
        if what in ("quad_nodes", "wide8_nodes", "tri_slots8", "tri_slots"):
            t = self.tree_stats()
            which, shape, dt = {"quad_nodes": (10, (t["quad_nodes"], 32), np.float32), "wide8_nodes": (11, (t["wide8_nodes"], 20), np.uint32),
                                "tri_slots8": (12, (t["wide8_tri_slots"], 12), np.float32), "tri_slots": (13, (n["tris"], 12), np.float32)}[what]
            out = np.zeros(shape, dt)
            check(lib().frt_scene_get(self._h, which, out.ctypes.data))
            return out
        spec = {"tris": (0, (n["tris"], 9), np.float32), "tri_instance": (1, (n["tris"],), np.uint32),
                "materials": (2, (n["materials"], 16), np.uint32), "lights": (3, (n["lights"], 16), np.uint32),
                "attributes": (4, (n["attributes"], 8), np.float32), "indices": (5, (n["indices"],), np.uint32),
                "mesh_infos": (6, (n["meshes"], 4), np.uint32), "instances": (7, (n["instances"], 30), np.uint32),
                "bvh2_nodes": (8, (n["bvh2_nodes"], 8), np.uint32), "bvh2_tri_index": (9, (n["tris"],), np.uint32)}[what]
        out = np.zeros(spec[1], spec[2])
        check(lib().frt_scene_get(self._h, spec[0], out.ctypes.data))
        return out

    def bvh_stats(self):
        s = (C.c_uint32 * 4)()
        check(lib().frt_scene_bvh_stats(self._h, s))
        return dict(zip(("depth", "leaves", "max_leaf", "pair_nodes"), list(s)))
