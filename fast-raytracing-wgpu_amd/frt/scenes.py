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


# ---- the glTF showcase scenes of scenes.rs:324-520 (assets are not shipped with either repo: pass the model's path) -------------
def _T(x, y, z):
    m = np.eye(4, dtype=np.float32); m[3, :3] = (x, y, z); return m            # column-major storage: m[c, r]


def _S(s):
    m = np.eye(4, dtype=np.float32) * np.float32(s); m[3, 3] = 1.0; return m


def _RX(a):
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    m = np.eye(4, dtype=np.float32); m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, s, -s, c; return m


def _RY(a):
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    m = np.eye(4, dtype=np.float32); m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, -s, s, c; return m


def _mul(*ms):
    """glam's A * B * ... on column-major matrices stored as m[c, r]: (A * B)[c] = sum_k A[k] * B[c][k]."""
    out = ms[0]
    for b in ms[1:]:
        out = (b.astype(np.float32) @ out.astype(np.float32)).astype(np.float32)
    return out


_LIGHT_ABOVE = lambda: _mul(_T(0.0, 5.0, 0.0), _RX(np.pi), _S(1.0))              # scenes.rs:336-338, :350-352, :368-370


def create_avocado_scene(path="assets/models/Avocado.glb"):                     # scenes.rs:324-339
    return create_gltf_scene(path, _mul(_T(0, 0, 0), _S(20.0)), _LIGHT_ABOVE())


def create_damaged_helmet_scene(path="assets/models/DamagedHelmet.glb"):        # scenes.rs:341-354
    return create_gltf_scene(path, _mul(_T(0, 0, 0), _RX(np.pi / 2.0), _S(1.0)), _LIGHT_ABOVE())


def create_multi_material_model_scene(path="assets/models/AliciaSolid.vrm"):    # scenes.rs:356-372
    return create_gltf_scene(path, _mul(_T(0, 0, 0), _S(0.5), _RY(np.pi)), _LIGHT_ABOVE())


def create_chocolate_truffle_scene(path="assets/models/gift_wrapped_chocolate_3d_model.glb", fallback="assets/models/Avocado.glb"):
    """scenes.rs:374-520: dark glossy table, the model with its materials re-tuned by brightness, three sphere lights. As in the
    reference, a model that cannot be loaded falls back to the avocado scene."""
    from . import geometry
    from .loader import load_gltf
    from .scene import material_new
    try:
        model = load_gltf(path)
    except FrtError:
        return create_avocado_scene(fallback)
    for i in range(model.counts()["materials"]):                                # :392-410
        m = model.material(i)
        r, g, b = (np.float32(v) for v in m.base_color[:3])
        brightness = r * np.float32(0.299) + g * np.float32(0.587) + b * np.float32(0.114)
        if brightness < 0.25:
            m.roughness, m.metallic = 0.02, 0.0
        else:
            m.roughness = 0.25
        model.set_material(i, m)
    b = SceneBuilder()
    plane_id = b.add_mesh(geometry.create_plane())
    b.add_mesh(geometry.create_plane())                                         # light_mesh_id: registered, never instanced (:421)
    mesh_ids = b.add_gltf_meshes(model)
    sphere_id = b.add_mesh(geometry.create_sphere(4))
    floor = material_new([0.02, 0.02, 0.02, 1.0]); floor.roughness = 0.1
    floor.metallic, floor.roughness = 1.0, 0.8                                   # .roughness(0.1).metallic(0.8): metallic(x) sets metallic = 1, roughness = x
    mat_dark_floor = b.add_material(floor)
    mat_ids = b.add_gltf_materials(model)
    b.add_instance(plane_id, mat_dark_floor, _mul(_T(0.0, -0.01, 0.0), _S(50.0)).reshape(16))
    b.add_gltf_instances(model, mesh_ids, mat_ids, _mul(_T(0.0, 0.7, 0.0), _RY(0.5), _S(4.0)).reshape(16))
    b.register_sphere_light(sphere_id, _mul(_T(8.0, 4.0, 2.0), _S(2.0)).reshape(16), [1.0, 0.95, 0.8], 80.0)
    b.register_sphere_light(sphere_id, _mul(_T(-3.0, 2.0, -4.0), _S(2.0)).reshape(16), [1.0, 0.05, 0.01], 40.0)
    b.register_sphere_light(sphere_id, _mul(_T(-3.0, 1.0, 3.0), _S(1.0)).reshape(16), [0.01, 0.05, 0.2], 10.0)
    return b.build()
