"""Slowest of N equal strips of the 3840x2160 frame (BASELINE.json configs[2]), two streams, no exchange: the strong-scaling bound at 4K."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fast-raytracing-wgpu_amd"))
import frt
W, H = 3840, 2160
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(40)]
for world in (1, 2, 4, 8):
    worst = 0
    for rank in range(world):
        rb, re = H * rank // world, H * (rank + 1) // world
        r = frt.Renderer(scene, W, H, rows=(rb, re) if world > 1 else None, flags=frt.FLAG_PIPELINE)
        for f in range(6): r.render(cams[f])
        r.sync(); t0 = time.perf_counter()
        for f in range(6, 30): r.render(cams[f])
        r.sync(); dt = (time.perf_counter() - t0) / 24 * 1e3
        worst = max(worst, dt); del r
    print(f"4K world {world}: slowest strip {worst:.3f} ms", flush=True)
