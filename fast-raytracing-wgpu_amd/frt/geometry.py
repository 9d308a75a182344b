"""src/geometry.rs generators through frt_geometry_create (computed in libfrt.so)."""
import ctypes as C
import numpy as np
from ._lib import lib, check

PLANE, CUBE, SPHERE, CRYSTAL = 0, 1, 2, 3


class Geometry:
    """positions [n,4] f32, attributes [n,8] f32 (oct normal 2, uv 2, tangent 4), indices [m] u32 (geometry.rs:12-18)."""

    def __init__(self, positions, attributes, indices):
        self.positions, self.attributes, self.indices = positions, attributes, indices


def _create(which, subdiv=0):
    nv, ni = C.c_uint32(), C.c_uint32()
    check(lib().frt_geometry_create(which, subdiv, C.byref(nv), C.byref(ni), None, None, None))
    pos = np.zeros((nv.value, 4), np.float32)
    att = np.zeros((nv.value, 8), np.float32)
    idx = np.zeros(ni.value, np.uint32)
    check(lib().frt_geometry_create(which, subdiv, None, None, pos.ctypes.data, att.ctypes.data, idx.ctypes.data))
    return Geometry(pos, att, idx)


def create_plane():      # geometry.rs:79 create_plane_blas
    return _create(PLANE)


def create_cube():       # geometry.rs:120 create_cube_blas
    return _create(CUBE)


def create_sphere(subdivisions):   # geometry.rs:222 create_sphere_blas
    return _create(SPHERE, subdivisions)


def create_crystal():    # geometry.rs:350 create_crystal_blas
    return _create(CRYSTAL)


def encode_octahedral_normal(n):   # geometry.rs:56
    a = np.asarray(n, np.float32)
    out = np.zeros(2, np.float32)
    lib().frt_encode_octahedral_normal(a.ctypes.data, out.ctypes.data)
    return out
