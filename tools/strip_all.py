"""Per-frame time of every strip renderer of an N-way split of the 1080p frame, two streams, no per-stage events, no exchange: best of 3 x 64 frames.
The slowest strip bounds the strong scaling of N GPUs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fast-raytracing-wgpu_amd"))
import frt
W, H = (3840, 2160) if "4k" in sys.argv[1:] else (1920, 1080)
EXTRA = int(os.environ.get("FRT_EXTRA_FLAGS", "0"))      # e.g. 32 = FLAG_WALK_WIDE, 64 = FLAG_WALK_WIDE_HBM
print(f"{W}x{H}, extra flags {EXTRA}", flush=True)
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(80)]
for world in (1, 2, 4, 8):
    ts = []
    for rank in range(world):
        rb, re = H * rank // world, H * (rank + 1) // world
        r = frt.Renderer(scene, W, H, rows=(rb, re) if world > 1 else None, flags=frt.FLAG_PIPELINE | EXTRA)
        best = 1e9
        for rnd in range(3):
            for f in range(8): r.render(cams[f])
            r.sync(); t0 = time.perf_counter()
            for f in range(8, 72): r.render(cams[f])
            r.sync(); best = min(best, (time.perf_counter() - t0) / 64 * 1e3)
        ts.append(best); del r
    print(f"world {world}: " + " ".join(f"{t:.3f}" for t in ts) + f"  slowest {max(ts):.3f}", flush=True)
