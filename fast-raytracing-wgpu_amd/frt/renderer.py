"""Renderer mirror (src/renderer.rs) over frt_renderer_*. Every pixel is produced by the HIP kernels in libfrt.so."""
import ctypes as C
import numpy as np
from ._lib import (lib, check, FrtError, RenderOpts, Stats, CameraUniform, BUF_BPP, BUF_ACCUM, BUF_DISPLAY, PHASE_ALL, FLAG_USE_STREAM)


class Renderer:
    def __init__(self, scene, width, height, max_depth=8, device=0, stream=None, rows=None, arena=None, arena_bytes=0, flags=0, motion_halo=0,
                 queue_capacity=0, cuts=None):
        """Renderer::new (renderer.rs:206). rows=(begin,end) restricts this renderer to an image strip; motion_halo = rows of
        previous-frame state kept valid beyond the strip for a moving camera (frt.dist.StripPlan(motion_halo=...))."""
        o = RenderOpts()
        o.max_depth, o.device, o.flags, o.queue_capacity = max_depth, device, flags, queue_capacity
        if cuts is not None:        # frt_render_opts.cut_depths: [] = never cut
            c = list(cuts)[:4] or [0xFFFFFFFF]
            for k, v in enumerate(c):
                o.cut_depths[k] = v
        if stream is not None:      # a caller-owned stream handle; 0 is the legacy default stream (torch's default current stream)
            o.stream = stream or None
            o.flags |= FLAG_USE_STREAM
        if rows is not None:
            o.row_begin, o.row_end = rows
            o.motion_halo_rows = motion_halo
        if arena is not None:
            o.device_arena, o.arena_bytes = arena, arena_bytes
        self.width, self.height = width, height
        self._scene = scene     # keep the scene alive
        self._destroy = lib().frt_renderer_destroy
        self._h = lib().frt_renderer_create(scene._h, width, height, C.byref(o))
        if not self._h:
            raise FrtError("renderer creation failed: " + lib().frt_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)     # bound at construction: module globals may already be gone at interpreter exit
            self._h = None

    @staticmethod
    def arena_bytes(width, height):
        return int(lib().frt_renderer_arena_bytes(width, height))

    def aspect_ratio(self):      # renderer.rs:202
        return self.width / self.height

    @property
    def frame_count(self):       # renderer.rs:198
        return int(lib().frt_renderer_frame_count(self._h))

    def render(self, camera_uniform, jitter=None):      # renderer.rs:349 (jitter -> PostParams.jitter, :361-379)
        if jitter is None:
            check(lib().frt_renderer_render(self._h, C.byref(camera_uniform)))
        else:
            check(lib().frt_renderer_render_jittered(self._h, C.byref(camera_uniform), float(jitter[0]), float(jitter[1])))

    def set_jitter(self, jitter):
        check(lib().frt_renderer_set_jitter(self._h, float(jitter[0]), float(jitter[1])))

    def fence(self):
        """Order the renderer's stream behind its internal second stream (no host wait); call before using buffer_info pointers."""
        check(lib().frt_renderer_fence(self._h))

    def order_edge_stream(self):
        """The edge stream behind the open frame's T-merge, now (frt_renderer_order_edge_stream): for transfers placed in that stream."""
        check(lib().frt_renderer_order_edge_stream(self._h))

    def stream_handle(self, which=0):
        return lib().frt_renderer_stream(self._h, which) or 0

    def render_phases(self, camera_uniform, phases=PHASE_ALL):
        check(lib().frt_renderer_render_phases(self._h, C.byref(camera_uniform), phases))

    def end_frame(self):
        check(lib().frt_renderer_end_frame(self._h))

    def sync(self):
        check(lib().frt_renderer_sync(self._h))

    def reset(self):             # state.rs:152 / renderer.rs:346
        check(lib().frt_renderer_reset(self._h))

    def clear(self):
        check(lib().frt_renderer_clear(self._h))

    def read_buffer(self, buf, index=0):
        bpp = BUF_BPP[buf]
        out = np.zeros((self.height, self.width, bpp), np.uint8)
        check(lib().frt_renderer_read_buffer(self._h, buf, index, out.ctypes.data))
        return out

    def read_rows(self, buf, index, y0, y1):
        out = np.zeros((y1 - y0, self.width, BUF_BPP[buf]), np.uint8)
        check(lib().frt_renderer_read_rows(self._h, buf, index, y0, y1, out.ctypes.data))
        return out

    def write_rows(self, buf, index, y0, y1, data):
        data = np.ascontiguousarray(data, np.uint8)
        assert data.size == (y1 - y0) * self.width * BUF_BPP[buf]
        check(lib().frt_renderer_write_rows(self._h, buf, index, y0, y1, data.ctypes.data))

    def read_display(self):
        return self.read_buffer(BUF_DISPLAY)

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        check(lib().frt_renderer_read_accum(self._h, out.ctypes.data))
        return out

    def buffer_info(self, buf, index=0):
        p, bpp = C.c_void_p(), C.c_uint32()
        check(lib().frt_renderer_buffer_info(self._h, buf, index, C.byref(p), C.byref(bpp)))
        return p.value, bpp.value

    def phase_rows(self):
        r = (C.c_uint32 * 8)()
        check(lib().frt_renderer_phase_rows(self._h, r))
        v = list(r)
        return {"gbuffer": (v[0], v[1]), "temporal": (v[2], v[3]), "spatial": (v[4], v[5]), "post": (v[6], v[7])}

    def set_timing(self, on):
        check(lib().frt_renderer_set_timing(self._h, 1 if on else 0))

    def stats(self):
        s = Stats()
        check(lib().frt_renderer_stats(self._h, C.byref(s)))
        return {"rays_closest": s.rays_closest, "rays_any": s.rays_any, "frames": s.frames,
                "ms_stage": list(s.ms_stage), "launches": list(s.launches),
                "rays_stage": [[int(s.rays_stage[i][0]), int(s.rays_stage[i][1])] for i in range(4)], "halo_overflow": int(s.halo_overflow),
                "ms_merge": s.ms_merge, "queue_overflow": int(s.queue_overflow), "queue_capacity": int(s.queue_capacity), "queue_bytes": int(s.queue_bytes),
                "speculated_frames": int(s.speculated_frames), "discarded_speculations": int(s.discarded_speculations)}


def _stats_dict(s):
    return {"rays_closest": s.rays_closest, "rays_any": s.rays_any, "frames": s.frames,
            "ms_stage": list(s.ms_stage), "launches": list(s.launches),
            "rays_stage": [[int(s.rays_stage[i][0]), int(s.rays_stage[i][1])] for i in range(4)], "halo_overflow": int(s.halo_overflow),
            "ms_merge": s.ms_merge, "queue_overflow": int(s.queue_overflow), "queue_capacity": int(s.queue_capacity), "queue_bytes": int(s.queue_bytes),
            "speculated_frames": int(s.speculated_frames), "discarded_speculations": int(s.discarded_speculations)}


class MultiRenderer:
    """Renderer::new / render (renderer.rs:206, :349) for several GPUs of one node through frt_multi_renderer_*: ONE process, one call per
    frame; strips, halo copies and the gather are inside libfrt.so. `devices`: HIP ordinals, repeats allowed (several strips on one GPU)."""

    def __init__(self, scene, width, height, devices, max_depth=8, motion_halo=0, flags=0, queue_capacity=0):
        o = RenderOpts()
        o.max_depth, o.flags, o.motion_halo_rows, o.queue_capacity = max_depth, flags, motion_halo, queue_capacity
        dev = (C.c_int32 * len(devices))(*devices)
        self.width, self.height, self.ndev = width, height, len(devices)
        self._scene = scene
        self._destroy = lib().frt_multi_renderer_destroy
        self._h = lib().frt_multi_renderer_create(scene._h, width, height, len(devices), dev, C.byref(o))
        if not self._h:
            raise FrtError("multi renderer creation failed: " + lib().frt_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    @property
    def frame_count(self):
        return int(lib().frt_multi_renderer_frame_count(self._h))

    def render(self, camera_uniform):
        check(lib().frt_multi_renderer_render(self._h, C.byref(camera_uniform)))

    def sync(self):
        check(lib().frt_multi_renderer_sync(self._h))

    def reset(self):
        check(lib().frt_multi_renderer_reset(self._h))

    def clear(self):
        """Back to the state right after creation on every strip; the way out of the failed state (a strip's step failed mid-frame)."""
        check(lib().frt_multi_renderer_clear(self._h))

    def set_jitter(self, jitter):
        check(lib().frt_multi_renderer_set_jitter(self._h, float(jitter[0]), float(jitter[1])))

    def inject_failure(self, strip, step):
        """Testing: the next render call fails on `strip` in step 0 (T-merge half) or 1 (spatial + post half)."""
        check(lib().frt_multi_renderer_inject_failure(self._h, strip, step))

    def peer_access(self):
        out = (C.c_uint32 * 2)()
        check(lib().frt_multi_renderer_peer_access(self._h, out))
        return {"neighbour_pairs_on_different_devices": int(out[0]), "pairs_with_peer_access": int(out[1])}

    def gather(self, buf, index, device, dst_ptr, stream=None):
        """Device-side gather of every strip's rows of `buf`[index] into the full-frame device buffer at `dst_ptr` on HIP device `device`."""
        check(lib().frt_multi_renderer_gather(self._h, buf, index, device, C.c_void_p(dst_ptr), C.c_void_p(stream) if stream else None))

    def boundaries(self):
        out = (C.c_uint32 * (self.ndev + 1))()
        check(lib().frt_multi_renderer_boundaries(self._h, out))
        return list(out)

    def read_buffer(self, buf, index=0):
        out = np.zeros((self.height, self.width, BUF_BPP[buf]), np.uint8)
        check(lib().frt_multi_renderer_read_buffer(self._h, buf, index, out.ctypes.data))
        return out

    def read_display(self):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        check(lib().frt_multi_renderer_read_display(self._h, out.ctypes.data))
        return out

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        check(lib().frt_multi_renderer_read_accum(self._h, out.ctypes.data))
        return out

    def stats(self):
        s = Stats()
        check(lib().frt_multi_renderer_stats(self._h, C.byref(s)))
        return _stats_dict(s)
