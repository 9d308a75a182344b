"""ctypes access to tests/hostcheck (host instantiation of the product's device functions). Test infrastructure only."""
import ctypes as C
import numpy as np

BPP = {0: 16, 1: 16, 2: 4, 3: 8, 4: 32, 5: 8, 6: 4, 7: 16}


class HostCheck:
    def __init__(self, path):
        L = self.L = C.CDLL(path)
        P, U32 = C.c_void_p, C.c_uint32
        L.hc_create.restype = P; L.hc_create.argtypes = [P, U32, U32, U32, C.c_int]
        L.hc_destroy.argtypes = [P]
        L.hc_render.argtypes = [P, P, C.c_int]
        L.hc_read.restype = C.c_int; L.hc_read.argtypes = [P, C.c_int, C.c_int, P]
        L.hc_rays.argtypes = [P, P]
        L.hc_set_jitter.argtypes = [P, C.c_float, C.c_float]
        L.hc_trace.argtypes = [P, C.c_int, U32, P, P, C.c_float, P, P, P, P, P, C.c_int]
        L.hc_quad_stats.argtypes = [P, P]
        L.hc_wide8_stats.argtypes = [P, P]

    def renderer(self, scene, w, h, max_depth=8, nthreads=8, state_machine=False):
        return HcRenderer(self, scene, w, h, max_depth, nthreads, state_machine)

    def quad_stats(self, scene):
        out = (C.c_uint32 * 6)()
        self.L.hc_quad_stats(scene._h, out)
        return dict(zip(("nodes", "stack_need", "stack_walked", "leaves", "triangles", "children_x100"), list(out)))

    def wide8_stats(self, scene):
        out = (C.c_uint32 * 8)()
        self.L.hc_wide8_stats(scene._h, out)
        return dict(zip(("nodes", "stack_need", "stack_walked", "leaves", "triangles", "children_x100", "defects", "levels"), list(out)))

    def trace(self, scene, o, d, tmin, tmax, any_hit=False, quantized=False):
        """quantized: False pair nodes (trace), True 16-bit pair nodes (trace_q), 2 quad nodes (trace4), 3 8-wide nodes with grid boxes (trace8)"""
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); n = o.shape[0]
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        t = np.zeros(n, np.float32); tri = np.zeros(n, np.uint32); uv = np.zeros((n, 2), np.float32); fr = np.zeros(n, np.uint8)
        self.L.hc_trace(scene._h, int(any_hit), n, o.ctypes.data, d.ctypes.data, tmin, tmax.ctypes.data, t.ctypes.data, tri.ctypes.data,
                        uv.ctypes.data, fr.ctypes.data, int(quantized))
        return t, tri, uv, fr


class HcRenderer:
    def __init__(self, hc, scene, w, h, max_depth, nthreads, state_machine=False):
        self.L, self.scene, self.w, self.hgt, self.sm = hc.L, scene, w, h, int(state_machine)
        self.h = self.L.hc_create(scene._h, w, h, max_depth, nthreads)

    def __del__(self):
        if self.h:
            self.L.hc_destroy(self.h); self.h = None

    def render(self, cam):
        cam = np.ascontiguousarray(np.frombuffer(bytes(cam), np.uint8))
        self.L.hc_render(self.h, cam.ctypes.data, self.sm)

    def set_jitter(self, jitter):
        self.L.hc_set_jitter(self.h, float(jitter[0]), float(jitter[1]))

    def read(self, buf, index=0):
        out = np.zeros((self.hgt, self.w, BPP[buf]), np.uint8)
        assert self.L.hc_read(self.h, buf, index, out.ctypes.data) == 0
        return out

    def rays(self):
        r = (C.c_ulonglong * 2)()
        self.L.hc_rays(self.h, r)
        return int(r[0]), int(r[1])
