"""CameraController mirror (src/camera.rs): the benchmark uses the fixed initial pose (:40-42) and zero jitter (:202-203)."""
import ctypes as C
from ._lib import lib, CameraUniform


class CameraController:
    def build_uniform(self, aspect, frame_count, num_lights, jitter=(0.0, 0.0)):
        """camera.rs:207-256. jitter is accepted for signature parity; the reference multiplies it by 0."""
        cu = CameraUniform()
        lib().frt_camera_default(float(aspect), int(frame_count), int(num_lights), C.byref(cu))
        return cu
