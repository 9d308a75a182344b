"""src/scene/loader.rs:9-181 load_gltf over frt_model_* (parsing, PNG decode and the Lanczos3 resize happen in libfrt.so)."""
import ctypes as C
import numpy as np
from ._lib import lib, check, Material, FrtError
from .geometry import Geometry


class Model:
    """The (geometries, materials, images, material_indices) tuple load_gltf returns, held by the library."""

    def __init__(self, path):
        self._destroy = lib().frt_model_destroy
        self._h = lib().frt_model_load(str(path).encode())
        if not self._h:
            raise FrtError(lib().frt_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    def counts(self):
        c = (C.c_uint32 * 4)()
        check(lib().frt_model_counts(self._h, c))
        return dict(zip(("geometries", "materials", "images", "warnings"), list(c)))

    def geometry(self, i):
        nv, ni, mi = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().frt_model_geometry_counts(self._h, i, C.byref(nv), C.byref(ni), C.byref(mi)))
        pos = np.zeros((nv.value, 4), np.float32); att = np.zeros((nv.value, 8), np.float32); idx = np.zeros(ni.value, np.uint32)
        check(lib().frt_model_geometry_get(self._h, i, pos.ctypes.data, att.ctypes.data, idx.ctypes.data))
        return Geometry(pos, att, idx), mi.value

    def material(self, i):
        m = Material()
        check(lib().frt_model_material_get(self._h, i, C.byref(m)))
        return m

    def set_material(self, i, m):
        check(lib().frt_model_material_set(self._h, i, C.byref(m)))

    def image(self, i):
        out = np.zeros((1024, 1024, 4), np.uint8)
        check(lib().frt_model_image_get(self._h, i, out.ctypes.data))
        return out

    def warnings(self):
        return [lib().frt_model_warning(self._h, i).decode() for i in range(self.counts()["warnings"])]


def load_gltf(path):
    """.gltf / .glb (and .obj, an extension). Raises FrtError with the loader's message on failure."""
    return Model(path)
