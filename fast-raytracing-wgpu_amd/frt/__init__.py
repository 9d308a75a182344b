"""Host-side mirror of the reference's Scene / Camera / Renderer surface over the C ABI of libfrt.so.

Reference (Rust)                      here
  scene::builder::SceneBuilder    ->  frt.SceneBuilder
  scene::scenes::create_*         ->  frt.scenes.create_cornell_box / create_restir_scene
  geometry::create_*_blas         ->  frt.geometry.create_plane / cube / sphere / crystal
  camera::CameraController        ->  frt.CameraController (build_uniform)
  scene::loader::load_gltf        ->  frt.loader.load_gltf (+ SceneBuilder.add_gltf_*; scenes.create_gltf_scene)
  renderer::Renderer              ->  frt.Renderer (render / frame_count / reset)
All arithmetic happens in libfrt.so (HIP); nothing here computes pixels.
"""
from ._lib import FrtError, lib, Material, Light, VertexAttr, CameraUniform, RenderOpts, Stats  # noqa: F401
from ._lib import (FLAG_TIMING, FLAG_COMPACTION, FLAG_USE_STREAM, FLAG_OVERLAP_POST, FLAG_PIPELINE, FLAG_THIRD_GSET, FLAG_WALK_WIDE, FLAG_WALK_WIDE_HBM, FLAG_WG_TRACE, PHASE_GBUFFER, PHASE_TEMPORAL, PHASE_SPATIAL, PHASE_POST, PHASE_ALL,  # noqa: F401
                   PHASE_SPATIAL_INNER, PHASE_SPATIAL_EDGE, BUF_CANDIDATE,
                   BUF_GPOS, BUF_GNORMAL, BUF_GALBEDO, BUF_GMOTION, BUF_RESERVOIR, BUF_RAW, BUF_DISPLAY, BUF_ACCUM, BUF_BPP)
from . import geometry, scenes, loader  # noqa: F401
from .scene import SceneBuilder, material_new  # noqa: F401
from .camera import CameraController  # noqa: F401
from .renderer import Renderer, MultiRenderer  # noqa: F401
