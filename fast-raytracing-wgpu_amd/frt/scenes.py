"""src/scene/scenes.rs factories (built inside libfrt.so)."""
import numpy as np
from ._lib import lib, FrtError
from .scene import SceneBuilder


def _wrap(h):
    if not h:
        raise FrtError("scene factory failed: " + lib().frt_last_error().decode())
    return SceneBuilder(handle=h)


def create_cornell_box():    # scenes.rs:9-130
    return _wrap(lib().frt_scene_create_cornell_box())


def create_restir_scene():   # scenes.rs:133-223
    return _wrap(lib().frt_scene_create_restir_scene())


def create_gltf_scene(path, model_transform_colmajor, light_transform_colmajor):   # scenes.rs:246-322
    """Floor + 15-intensity quad light + the model, built. Raises FrtError with the loader's message when the model cannot be
    loaded (the reference logs the error and renders an empty scene instead)."""
    mt = np.ascontiguousarray(model_transform_colmajor, np.float32).reshape(16)
    lt = np.ascontiguousarray(light_transform_colmajor, np.float32).reshape(16)
    return _wrap(lib().frt_scene_create_gltf_scene(str(path).encode(), mt.ctypes.data, lt.ctypes.data))
