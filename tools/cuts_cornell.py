"""Cut depths of the continuation queues on the headline workload (Cornell Box, 1920x1080, MAX_DEPTH 8, two streams): ms per frame per cut list,
best of 3 x 64 frames. (tools/cuts_big.py: the same for the deep-tree workloads.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
import torch, frt  # noqa: F401
W, H = 1920, 1080
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(72)]
for cuts in (None, [3, 4], [3], [3, 5], [2, 4], [2, 3], [3, 4, 5], [3, 4, 6], [4, 5], [2, 3, 4], [3, 4], None):
    r = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE, cuts=cuts)
    best = 1e9
    for rnd in range(3):
        for f in range(8): r.render(cams[f])
        r.sync(); t0 = time.perf_counter()
        for f in range(8, 72): r.render(cams[f])
        r.sync(); best = min(best, (time.perf_counter() - t0) / 64 * 1e3)
    print(f"cuts {cuts}: {best:.3f} ms  overflow {r.stats()['queue_overflow']}", flush=True)
    del r
