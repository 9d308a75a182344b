"""ctypes binding of include/frt.h (libfrt.so). The library is the product; this module only declares its symbols.

Fails loudly when the shared library is missing: there is no Python or CPU fallback for any entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRT_LIB: an alternative build of the library (lib/libfrt_exp.so = the experiments build of `make experiments`; A/B builds under _ab/)
LIB_PATH = os.environ.get("FRT_LIB") or os.path.join(_HERE, "..", "lib", "libfrt.so")


class VertexAttr(C.Structure):      # src/geometry.rs:4-10
    _fields_ = [("normal", C.c_float * 2), ("uv", C.c_float * 2), ("tangent", C.c_float * 4)]


class Material(C.Structure):        # src/scene/material.rs:2-28
    _fields_ = [("base_color", C.c_float * 4), ("emissive_factor", C.c_float * 3), ("roughness", C.c_float),
                ("metallic", C.c_float), ("transmission", C.c_float), ("ior", C.c_float), ("light_index", C.c_int32),
                ("tex_info_0", C.c_uint32), ("tex_info_1", C.c_uint32), ("tex_info_2", C.c_uint32), ("pad_final", C.c_uint32)]


class Light(C.Structure):           # src/scene/light.rs:1-16
    _fields_ = [("position", C.c_float * 3), ("type_", C.c_uint32), ("u", C.c_float * 3), ("area", C.c_float),
                ("v", C.c_float * 3), ("pad", C.c_uint32), ("emission", C.c_float * 4)]


class CameraUniform(C.Structure):   # src/camera.rs:4-15
    _fields_ = [("view_proj", C.c_float * 16), ("view_inverse", C.c_float * 16), ("proj_inverse", C.c_float * 16),
                ("view_pos", C.c_float * 4), ("prev_view_proj", C.c_float * 16),
                ("frame_count", C.c_uint32), ("num_lights", C.c_uint32), ("padding", C.c_uint32 * 2)]


class RenderOpts(C.Structure):
    _fields_ = [("max_depth", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("row_begin", C.c_uint32), ("row_end", C.c_uint32), ("device_arena", C.c_void_p),
                ("arena_bytes", C.c_uint64), ("flags", C.c_uint32), ("motion_halo_rows", C.c_uint32), ("queue_capacity", C.c_uint32),
                ("cut_depths", C.c_uint32 * 4)]


class Stats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("frames", C.c_uint64),
                ("ms_stage", C.c_double * 4), ("launches", C.c_uint64 * 4), ("rays_stage", (C.c_uint64 * 2) * 4), ("halo_overflow", C.c_uint64),
                ("ms_merge", C.c_double), ("queue_overflow", C.c_uint64), ("queue_capacity", C.c_uint64),
                ("speculated_frames", C.c_uint64), ("discarded_speculations", C.c_uint64), ("queue_bytes", C.c_uint64)]


assert C.sizeof(VertexAttr) == 32 and C.sizeof(Material) == 64 and C.sizeof(Light) == 64 and C.sizeof(CameraUniform) == 288

FLAG_TIMING = 1
FLAG_COMPACTION = 2
FLAG_USE_STREAM = 4
FLAG_PIPELINE = 8
FLAG_THIRD_GSET = 16
FLAG_WALK_WIDE = 32
FLAG_WALK_WIDE_HBM = 64
FLAG_WG_TRACE = 128
FLAG_OVERLAP_POST = FLAG_PIPELINE      # round-1 name
PHASE_GBUFFER, PHASE_TEMPORAL, PHASE_SPATIAL, PHASE_POST, PHASE_ALL = 1, 2, 4, 8, 15
PHASE_SPATIAL_INNER, PHASE_SPATIAL_EDGE = 16, 32
BUF_GPOS, BUF_GNORMAL, BUF_GALBEDO, BUF_GMOTION, BUF_RESERVOIR, BUF_RAW, BUF_DISPLAY, BUF_ACCUM, BUF_CANDIDATE = range(9)
BUF_BPP = {BUF_GPOS: 16, BUF_GNORMAL: 16, BUF_GALBEDO: 4, BUF_GMOTION: 8, BUF_RESERVOIR: 32, BUF_RAW: 8, BUF_DISPLAY: 4, BUF_ACCUM: 16, BUF_CANDIDATE: 16}

# every symbol include/frt.h declares: (restype, argtypes)
_P = C.c_void_p
_U32 = C.c_uint32
_FP = C.POINTER(C.c_float)
SYMBOLS = {
    "frt_last_error": (C.c_char_p, []),
    "frt_device_count": (C.c_int, []),
    "frt_geometry_create": (C.c_int, [C.c_int, _U32, C.POINTER(_U32), C.POINTER(_U32), _P, _P, _P]),
    "frt_encode_octahedral_normal": (None, [_P, _P]),
    "frt_material_default": (None, [_P, C.POINTER(Material)]),
    "frt_scene_create": (_P, []),
    "frt_scene_destroy": (None, [_P]),
    "frt_scene_add_mesh": (C.c_int, [_P, _P, _U32, _P, _P, _U32]),
    "frt_scene_add_material": (C.c_int, [_P, C.POINTER(Material)]),
    "frt_scene_add_instance": (C.c_int, [_P, _U32, _U32, _P]),
    "frt_scene_add_light": (C.c_int, [_P, C.POINTER(Light)]),
    "frt_scene_register_quad_light": (C.c_int, [_P, _U32, _P, _P, C.c_float]),
    "frt_scene_register_sphere_light": (C.c_int, [_P, _U32, _P, _P, C.c_float]),
    "frt_scene_add_texture": (C.c_int, [_P, C.c_int, _P]),
    "frt_scene_build": (C.c_int, [_P]),
    "frt_scene_create_cornell_box": (_P, []),
    "frt_scene_create_restir_scene": (_P, []),
    "frt_model_load": (_P, [C.c_char_p]),
    "frt_model_destroy": (None, [_P]),
    "frt_model_counts": (C.c_int, [_P, _P]),
    "frt_model_geometry_counts": (C.c_int, [_P, _U32, C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U32)]),
    "frt_model_geometry_get": (C.c_int, [_P, _U32, _P, _P, _P]),
    "frt_model_material_get": (C.c_int, [_P, _U32, C.POINTER(Material)]),
    "frt_model_material_set": (C.c_int, [_P, _U32, C.POINTER(Material)]),
    "frt_model_image_get": (C.c_int, [_P, _U32, _P]),
    "frt_model_warning": (C.c_char_p, [_P, _U32]),
    "frt_scene_add_gltf_materials": (C.c_int, [_P, _P, _P]),
    "frt_scene_add_gltf_meshes": (C.c_int, [_P, _P, _P]),
    "frt_scene_add_gltf_instances": (C.c_int, [_P, _P, _P, _U32, _P, _U32, _P]),
    "frt_scene_create_gltf_scene": (_P, [C.c_char_p, _P, _P]),
    "frt_scene_counts": (C.c_int, [_P, _P]),
    "frt_scene_get": (C.c_int, [_P, C.c_int, _P]),
    "frt_scene_bvh_stats": (C.c_int, [_P, _P]),
    "frt_scene_tree_stats": (C.c_int, [_P, _P]),
    "frt_camera_default": (None, [C.c_float, _U32, _U32, C.POINTER(CameraUniform)]),
    "frt_camera_build_uniform": (C.c_int, [_P, C.c_float, C.c_float, _P, C.c_float, _U32, _U32, _P, C.POINTER(CameraUniform), _P]),
    "frt_camera_halton_jitter": (None, [_U32, _U32, _U32, C.c_float, _P]),
    "frt_renderer_arena_bytes": (C.c_uint64, [_U32, _U32]),
    "frt_renderer_create": (_P, [_P, _U32, _U32, C.POINTER(RenderOpts)]),
    "frt_renderer_destroy": (None, [_P]),
    "frt_renderer_render": (C.c_int, [_P, C.POINTER(CameraUniform)]),
    "frt_renderer_render_phases": (C.c_int, [_P, C.POINTER(CameraUniform), C.c_int]),
    "frt_renderer_end_frame": (C.c_int, [_P]),
    "frt_renderer_set_jitter": (C.c_int, [_P, C.c_float, C.c_float]),
    "frt_renderer_render_jittered": (C.c_int, [_P, C.POINTER(CameraUniform), C.c_float, C.c_float]),
    "frt_renderer_sync": (C.c_int, [_P]),
    "frt_renderer_fence": (C.c_int, [_P]),
    "frt_renderer_order_edge_stream": (C.c_int, [_P]),
    "frt_renderer_stream": (_P, [_P, C.c_int]),
    "frt_renderer_frame_count": (_U32, [_P]),
    "frt_renderer_reset": (C.c_int, [_P]),
    "frt_renderer_clear": (C.c_int, [_P]),
    "frt_renderer_read_display": (C.c_int, [_P, _P]),
    "frt_renderer_read_accum": (C.c_int, [_P, _P]),
    "frt_renderer_read_buffer": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "frt_renderer_read_rows": (C.c_int, [_P, C.c_int, C.c_int, _U32, _U32, _P]),
    "frt_renderer_write_rows": (C.c_int, [_P, C.c_int, C.c_int, _U32, _U32, _P]),
    "frt_renderer_buffer_info": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(_U32)]),
    "frt_renderer_phase_rows": (C.c_int, [_P, _P]),
    "frt_renderer_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "frt_renderer_set_timing": (C.c_int, [_P, C.c_int]),
    "frt_multi_renderer_create": (_P, [_P, _U32, _U32, _U32, _P, C.POINTER(RenderOpts)]),
    "frt_multi_renderer_destroy": (None, [_P]),
    "frt_multi_renderer_render": (C.c_int, [_P, C.POINTER(CameraUniform)]),
    "frt_multi_renderer_sync": (C.c_int, [_P]),
    "frt_multi_renderer_frame_count": (_U32, [_P]),
    "frt_multi_renderer_reset": (C.c_int, [_P]),
    "frt_multi_renderer_clear": (C.c_int, [_P]),
    "frt_multi_renderer_set_jitter": (C.c_int, [_P, C.c_float, C.c_float]),
    "frt_multi_renderer_gather": (C.c_int, [_P, C.c_int, C.c_int, C.c_int32, _P, _P]),
    "frt_multi_renderer_peer_access": (C.c_int, [_P, _P]),
    "frt_multi_renderer_inject_failure": (C.c_int, [_P, _U32, C.c_int]),
    "frt_multi_renderer_read_display": (C.c_int, [_P, _P]),
    "frt_multi_renderer_read_accum": (C.c_int, [_P, _P]),
    "frt_multi_renderer_read_buffer": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "frt_multi_renderer_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "frt_multi_renderer_boundaries": (C.c_int, [_P, _P]),
}

_lib = None


class FrtError(RuntimeError):
    pass


def lib():
    """Load libfrt.so (built by `make -C fast-raytracing-wgpu_amd` / __graft_entry__.build()). No fallback."""
    global _lib
    if _lib is None:
        path = os.path.abspath(LIB_PATH)
        if not os.path.exists(path):
            raise FrtError(f"{path} is missing: build it with __graft_entry__.build(); there is no CPU fallback")
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)      # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise FrtError(f"libfrt error {rc}: {lib().frt_last_error().decode()}")
    return rc
