"""ctypes access to the CPU oracle (oracle/_build/liborc.so). Test infrastructure only."""
import ctypes as C
import numpy as np

BPP = {0: 16, 1: 16, 2: 4, 3: 8, 4: 32, 5: 8, 6: 4, 7: 16}   # same selectors as FRT_BUF_*
PH_GBUFFER, PH_TEMPORAL, PH_SPATIAL, PH_POST, PH_ALL = 1, 2, 4, 8, 15


class Oracle:
    def __init__(self, path):
        L = self.L = C.CDLL(path)
        P, U32 = C.c_void_p, C.c_uint32
        sig = {
            "orc_pcg_hash": (U32, [U32]), "orc_encode_octahedral": (None, [P, P]),
            "orc_f32_to_f16": (C.c_uint16, [C.c_float]), "orc_f16_to_f32": (C.c_float, [C.c_uint16]), "orc_f32_to_unorm8": (C.c_uint8, [C.c_float]),
            "orc_sin": (C.c_float, [C.c_float]), "orc_cos": (C.c_float, [C.c_float]), "orc_exp2": (C.c_float, [C.c_float]),
            "orc_log2": (C.c_float, [C.c_float]), "orc_pow": (C.c_float, [C.c_float, C.c_float]), "orc_exp": (C.c_float, [C.c_float]),
            "orc_struct_sizes": (None, [P]), "orc_mesh_counts": (None, [C.c_int, U32, P]), "orc_mesh_get": (None, [C.c_int, U32, P, P, P]),
            "orc_scene_create": (P, []), "orc_scene_destroy": (None, [P]),
            "orc_scene_create_cornell_box": (P, []), "orc_scene_create_restir_scene": (P, []),
            "orc_scene_add_mesh": (C.c_int, [P, P, U32, P, P, U32]), "orc_scene_add_material": (C.c_int, [P, P]),
            "orc_scene_add_instance": (C.c_int, [P, U32, U32, P]), "orc_scene_add_light": (C.c_int, [P, P]),
            "orc_scene_add_texture": (C.c_int, [P, C.c_int, P]), "orc_scene_build": (C.c_int, [P]),
            "orc_scene_counts": (None, [P, P]), "orc_scene_get": (None, [P, C.c_int, P]),
            "orc_scene_set_bvh": (C.c_int, [P, P, U32, P, U32]), "orc_camera_default": (None, [C.c_float, U32, U32, P]),
            "orc_camera_build": (None, [P, C.c_float, C.c_float, P, C.c_float, U32, U32, C.c_float, C.c_float, P, P]),
            "orc_camera_halton_jitter": (None, [U32, U32, U32, C.c_float, P]), "orc_renderer_set_jitter": (None, [P, C.c_float, C.c_float]),
            "orc_set_text_mode": (None, [C.c_int]), "orc_get_text_mode": (C.c_int, []),
            "orc_trace_closest": (None, [P, C.c_int, U32, P, P, C.c_float, C.c_float, P, P, P, P, P]),
            "orc_trace_any": (None, [P, C.c_int, U32, P, P, C.c_float, P, P]),
            "orc_renderer_create": (P, [P, U32, U32, U32, C.c_int, C.c_int]), "orc_renderer_destroy": (None, [P]),
            "orc_renderer_reset": (None, [P]), "orc_renderer_restart_counter": (None, [P]), "orc_renderer_render": (None, [P, P]),
            "orc_renderer_render_phases": (None, [P, P, C.c_int, U32, U32]), "orc_renderer_end_frame": (None, [P]),
            "orc_renderer_frame_count": (U32, [P]), "orc_renderer_read": (C.c_int, [P, C.c_int, C.c_int, P]),
            "orc_renderer_write_rows": (C.c_int, [P, C.c_int, C.c_int, U32, U32, P]),
            "orc_renderer_read_rows": (C.c_int, [P, C.c_int, C.c_int, U32, U32, P]),
            "orc_renderer_stats": (None, [P, P]), "orc_renderer_time_frames": (C.c_double, [P, P, U32]),
        }
        for n, (r, a) in sig.items():
            f = getattr(L, n); f.restype = r; f.argtypes = a

    def set_text_mode(self, on):
        """Literal evaluation of the WGSL (true division, pow = exp2(y log2 x), libm) instead of the numeric contract. Process-wide."""
        self.L.orc_set_text_mode(int(bool(on)))

    # ---- scene
    def cornell(self):
        return OrcScene(self, self.L.orc_scene_create_cornell_box())

    def restir_scene(self):
        return OrcScene(self, self.L.orc_scene_create_restir_scene())

    def camera(self, aspect, frame, nlights):
        buf = np.zeros(288, np.uint8)
        self.L.orc_camera_default(aspect, frame, nlights, buf.ctypes.data)
        return buf

    def camera_build(self, position, yaw, pitch, prev_view_proj, aspect, frame, nlights, jitter=(0.0, 0.0)):
        """camera.rs:207-256 -> (288 uniform bytes, unjittered view_proj as 16 f32). prev_view_proj=None: first frame (IDENTITY)."""
        buf = np.zeros(288, np.uint8); unj = np.zeros(16, np.float32)
        pos = np.ascontiguousarray(position, np.float32)
        prev = None if prev_view_proj is None else np.ascontiguousarray(prev_view_proj, np.float32)
        self.L.orc_camera_build(pos.ctypes.data, yaw, pitch, None if prev is None else prev.ctypes.data, aspect, frame, nlights,
                                jitter[0], jitter[1], buf.ctypes.data, unj.ctypes.data)
        return buf, unj

    def halton_jitter(self, index, width, height, scale=0.0):
        out = np.zeros(2, np.float32)
        self.L.orc_camera_halton_jitter(index, width, height, scale, out.ctypes.data)
        return float(out[0]), float(out[1])

    def mesh(self, which, subdiv=0):
        c = (C.c_uint32 * 2)()
        self.L.orc_mesh_counts(which, subdiv, c)
        pos = np.zeros((c[0], 4), np.float32); att = np.zeros((c[0], 8), np.float32); idx = np.zeros(c[1] * 3, np.uint32)
        self.L.orc_mesh_get(which, subdiv, pos.ctypes.data, att.ctypes.data, idx.ctypes.data)
        return pos, att, idx


class OrcScene:
    def __init__(self, o, h):
        self.o, self.h = o, h

    def __del__(self):
        if self.h:
            self.o.L.orc_scene_destroy(self.h); self.h = None

    def counts(self):
        c = (C.c_uint32 * 8)()
        self.o.L.orc_scene_counts(self.h, c)
        return dict(zip(("tris", "instances", "materials", "lights", "meshes", "attributes", "indices", "bvh2_nodes"), list(c)))

    def get(self, what):
        n = self.counts()
        spec = {"tris": (0, (n["tris"], 9), np.float32), "tri_instance": (1, (n["tris"],), np.uint32),
                "materials": (2, (n["materials"], 16), np.uint32), "lights": (3, (n["lights"], 16), np.uint32),
                "attributes": (4, (n["attributes"], 8), np.float32), "indices": (5, (n["indices"],), np.uint32),
                "mesh_infos": (6, (n["meshes"], 4), np.uint32), "instances": (7, (n["instances"], 30), np.uint32)}[what]
        out = np.zeros(spec[1], spec[2])
        self.o.L.orc_scene_get(self.h, spec[0], out.ctypes.data)
        return out

    def set_bvh(self, nodes, tri_index):
        nodes = np.ascontiguousarray(nodes); tri_index = np.ascontiguousarray(tri_index, np.uint32)
        rc = self.o.L.orc_scene_set_bvh(self.h, nodes.ctypes.data, nodes.shape[0], tri_index.ctypes.data, tri_index.size)
        assert rc == 0

    def trace_closest(self, o, d, tmin, tmax, use_bvh):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); n = o.shape[0]
        t = np.zeros(n, np.float32); tri = np.zeros(n, np.uint32); uv = np.zeros((n, 2), np.float32); fr = np.zeros(n, np.uint8)
        st = (C.c_uint64 * 4)()
        self.o.L.orc_trace_closest(self.h, int(use_bvh), n, o.ctypes.data, d.ctypes.data, tmin, tmax, t.ctypes.data, tri.ctypes.data,
                                   uv.ctypes.data, fr.ctypes.data, st)
        return t, tri, uv, fr, list(st)

    def trace_any(self, o, d, tmin, tmax, use_bvh):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); n = o.shape[0]
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        occ = np.zeros(n, np.uint8)
        self.o.L.orc_trace_any(self.h, int(use_bvh), n, o.ctypes.data, d.ctypes.data, tmin, tmax.ctypes.data, occ.ctypes.data)
        return occ

    def renderer(self, w, h, max_depth=8, use_bvh=True, nthreads=8):
        return OrcRenderer(self, w, h, max_depth, use_bvh, nthreads)


class OrcRenderer:
    def __init__(self, scene, w, h, max_depth, use_bvh, nthreads):
        self.s, self.w, self.hgt = scene, w, h
        self.L = scene.o.L
        self.h = self.L.orc_renderer_create(scene.h, w, h, max_depth, int(use_bvh), nthreads)

    def __del__(self):
        if self.h:
            self.L.orc_renderer_destroy(self.h); self.h = None

    def render(self, cam):
        cam = np.ascontiguousarray(np.frombuffer(bytes(cam), np.uint8))
        self.L.orc_renderer_render(self.h, cam.ctypes.data)

    def render_phases(self, cam, phases, y0, y1):
        cam = np.ascontiguousarray(np.frombuffer(bytes(cam), np.uint8))
        self.L.orc_renderer_render_phases(self.h, cam.ctypes.data, phases, y0, y1)

    def end_frame(self):
        self.L.orc_renderer_end_frame(self.h)

    def set_jitter(self, jitter):      # PostParams.jitter, renderer.rs:376
        self.L.orc_renderer_set_jitter(self.h, float(jitter[0]), float(jitter[1]))

    @property
    def frame_count(self):
        return self.L.orc_renderer_frame_count(self.h)

    def reset(self):                # zero every buffer and the counter (a fresh renderer)
        self.L.orc_renderer_reset(self.h)

    def restart_counter(self):      # state.rs:152
        self.L.orc_renderer_restart_counter(self.h)

    def read(self, buf, index=0):
        out = np.zeros((self.hgt, self.w, BPP[buf]), np.uint8)
        assert self.L.orc_renderer_read(self.h, buf, index, out.ctypes.data) == 0
        return out

    def read_rows(self, buf, index, y0, y1):
        out = np.zeros((y1 - y0, self.w, BPP[buf]), np.uint8)
        assert self.L.orc_renderer_read_rows(self.h, buf, index, y0, y1, out.ctypes.data) == 0
        return out

    def write_rows(self, buf, index, y0, y1, data):
        data = np.ascontiguousarray(data, np.uint8)
        assert self.L.orc_renderer_write_rows(self.h, buf, index, y0, y1, data.ctypes.data) == 0

    def stats(self):
        st = (C.c_uint64 * 20)()
        self.L.orc_renderer_stats(self.h, st)
        v = list(st)
        keys = ("total", "gbuffer", "temporal", "spatial", "post")
        return {k: dict(zip(("closest", "any", "nodes", "tris"), v[4 * i:4 * i + 4])) for i, k in enumerate(keys)}

    def time_frames(self, cams):
        cams = np.ascontiguousarray(cams, np.uint8)
        return self.L.orc_renderer_time_frames(self.h, cams.ctypes.data, cams.size // 288)
