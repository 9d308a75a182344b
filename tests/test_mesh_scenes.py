"""BASELINE.json configs[3] / [4] as parity cases: large synthetic meshes (deep BVH, scene beyond L2) built through the public
builder API of both libraries. CPU leg: host scene parity + hostcheck; GPU leg: kernels vs oracle."""
import numpy as np
import pytest
from test_hostcheck_parity import compare_all
import _scenes


def test_bumpy_sphere_scene_host_parity(frt, orc, hostcheck):
    fs, os_ = _scenes.bumpy_sphere_in_box(frt, orc, subdiv=4)        # 5,120-triangle version keeps the CPU suite short
    assert fs.counts()["tris"] == 5120 + 12
    for k in ("tris", "tri_instance", "materials", "lights", "instances"):
        assert fs.get(k).tobytes() == os_.get(k).tobytes(), k
    st = fs.bvh_stats()
    assert st["depth"] <= 30 and st["max_leaf"] <= 2
    W, H = 64, 48
    ro = os_.renderer(W, H, 8, True, 8); rh = hostcheck.renderer(fs, W, H, 8, 8)
    rb = os_.renderer(W, H, 8, False, 8)                              # brute force over all triangles
    for f in range(2):
        cam = frt.CameraController().build_uniform(W / H, f, 1)
        ro.render(cam); rh.render(cam); rb.render(cam)
        compare_all(rh.read, ro.read, f, "bumpy sphere")
        compare_all(ro.read, rb.read, f, "bumpy sphere, BVH vs brute force")


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["bumpy82k", "colonnade250k"])
def test_mesh_scenes_on_gpu(frt, orc, which):
    if frt.lib().frt_device_count() < 1:
        pytest.fail("no HIP device")
    if which == "bumpy82k":
        fs, os_ = _scenes.bumpy_sphere_in_box(frt, orc, subdiv=6)     # configs[3]: ~82k triangles, 8 bounces
        depth, W, H, frames = 8, 160, 90, 3
        assert fs.counts()["tris"] == 81920 + 12
    else:
        fs, os_ = _scenes.colonnade(frt, orc)                         # configs[4]: ~250k triangles, 16 bounces
        depth, W, H, frames = 16, 160, 90, 2
        assert fs.counts()["tris"] > 245000
    assert fs.bvh_stats()["depth"] <= 30
    r = frt.Renderer(fs, W, H, max_depth=depth)
    ro = os_.renderer(W, H, depth, True, 16)
    for f in range(frames):
        cam = frt.CameraController().build_uniform(W / H, f, fs.num_lights)
        r.render(cam); ro.render(cam)
        compare_all(r.read_buffer, ro.read, f, which)
    st, so = r.stats(), ro.stats()["total"]
    assert (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"])
    acc = r.read_accum()
    assert acc[..., :3].mean() > 0.01 and not np.isnan(acc).any()
