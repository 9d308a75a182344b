"""Cut depths of the continuation queues on the deep-tree workloads (configs[3] / [4] stand-ins): ms per frame per cut list."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, frt
from _oracle import Oracle
import _scenes
orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))
def run(name, scene, W, H, depth, nl, cuts, frames=20, warm=6):
    cams = [frt.CameraController().build_uniform(W / H, f, nl) for f in range(frames)]
    r = frt.Renderer(scene, W, H, max_depth=depth, flags=frt.FLAG_PIPELINE, cuts=cuts)
    for f in range(warm): r.render(cams[f])
    r.sync(); t0 = time.perf_counter()
    for f in range(warm, frames): r.render(cams[f])
    r.sync(); t1 = time.perf_counter()
    print(f"{name} cuts {cuts}: {(t1 - t0) / (frames - warm) * 1e3:.3f} ms  overflow {r.stats()['queue_overflow']}", flush=True)
    del r
col, _ = _scenes.colonnade(frt, orc)
for cuts in (None, [3], [3, 5], [3, 4, 6], [3, 4, 6, 9], [2, 3, 4], [3, 4, 5, 7], [4, 6], [3, 6, 10]):
    run("colonnade 4K d16", col, 3840, 2160, 16, 1, cuts)
blob, _ = _scenes.bumpy_sphere_in_box(frt, orc)
for cuts in (None, [3], [3, 5], [3, 4, 6], [2, 3, 4]):
    run("blob 1080p d8", blob, 1920, 1080, 8, 1, cuts, frames=40, warm=8)
