"""src/scene/scenes.rs factories (built inside libfrt.so)."""
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
