"""CameraController mirror (src/camera.rs): controller state (:17-56), build_uniform (:207-256) with the projection-shear jitter and
the previous view-projection, get_halton_jitter (:182-205). All arithmetic is in libfrt.so (frt_camera_build_uniform)."""
import ctypes as C
import math
import numpy as np
from ._lib import lib, check, CameraUniform


class CameraController:
    def __init__(self, position=(0.0, 0.0, 3.0), yaw=math.radians(-90.0), pitch=0.0):
        """CameraController::new (camera.rs:38-56): the initial pose of the benchmark; prev_view_proj starts as IDENTITY (= None here)."""
        self.position = [float(position[0]), float(position[1]), float(position[2])]
        self.yaw, self.pitch = float(np.float32(yaw)), float(np.float32(pitch))
        self.prev_view_proj = None
        self.unjittered_view_proj = None     # second element of the last build_uniform result

    @staticmethod
    def get_halton_jitter(index, width, height, scale=0.0):
        """camera.rs:182-205. scale = the literal 0 the reference multiplies the Halton offsets by (:202-203)."""
        out = (C.c_float * 2)()
        lib().frt_camera_halton_jitter(int(index), int(width), int(height), float(scale), out)
        return float(out[0]), float(out[1])

    def build_uniform(self, aspect, frame_count, num_lights, jitter=(0.0, 0.0)):
        """camera.rs:207-256 -> CameraUniform; the unjittered view-projection (the tuple's second element) is kept in
        self.unjittered_view_proj; commit_frame() makes it the next frame's prev_view_proj as state.rs:172 does."""
        cu = CameraUniform()
        pos = (C.c_float * 3)(*self.position)
        jit = (C.c_float * 2)(float(jitter[0]), float(jitter[1]))
        unj = (C.c_float * 16)()
        prev = None
        if self.prev_view_proj is not None:
            prev = (C.c_float * 16)(*[float(v) for v in self.prev_view_proj])
        check(lib().frt_camera_build_uniform(pos, self.yaw, self.pitch, prev, float(aspect), int(frame_count), int(num_lights), jit, C.byref(cu), unj))
        self.unjittered_view_proj = list(unj)
        return cu

    def commit_frame(self):
        """state.rs:172: camera_controller.prev_view_proj = unjittered view_proj of the frame just built."""
        self.prev_view_proj = self.unjittered_view_proj
