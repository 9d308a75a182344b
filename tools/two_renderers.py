"""How much would MORE concurrency between frames buy? Two independent renderers (two streams each) of the 1080p Cornell Box enqueued side by side on one
GPU: their frames have no dependency on each other, so the chip sees four streams of kernels. Amortised ms/frame of the pair vs one renderer alone =
the upper bound of any schedule that overlaps more of a frame sequence's stages (a second ahead stream, two frames of speculation).
    python tools/two_renderers.py [n_renderers]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
import torch  # noqa: F401
import frt
W, H = 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(72)]
for count in (1, n, 1, n):
    rs = [frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE) for _ in range(count)]
    best = None
    for rnd in range(3):
        for r in rs: r.clear()
        for f in range(8):
            for r in rs: r.render(cams[f])
        for r in rs: r.sync()
        t0 = time.perf_counter()
        for f in range(8, 72):
            for r in rs: r.render(cams[f])
        for r in rs: r.sync()
        t = (time.perf_counter() - t0) / 64 * 1e3
        best = t if best is None else min(best, t)
    print(f"{count} renderer(s): {best:.3f} ms per frame of each = {best / count:.3f} ms per frame amortised", flush=True)
    del rs
