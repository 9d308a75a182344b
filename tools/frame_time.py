"""ms/frame and per-stage ms of the default renderer at 1920x1080 (best of 3 rounds of 64 frames). FRT_CUTS / FRT_BVH_LEAF in the
environment are read by the library; an optional argument names an alternative libfrt.so (A/B builds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
import torch  # noqa: F401  (initialises the HIP runtime the way bench.py does)
import frt._lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import frt
W, H = 1920, 1080
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(72)]
flags = int(os.environ.get('FRT_FLAGS', frt.FLAG_TIMING | frt.FLAG_PIPELINE))      # experiment knob: e.g. FRT_FLAGS=1 = timing only, one stream
r = frt.Renderer(scene, W, H, flags=flags)
best = None
for rnd in range(3):
    r.clear()
    for f in range(8): r.render(cams[f])
    r.sync(); s0 = r.stats(); t0 = time.perf_counter()
    for f in range(8, 72): r.render(cams[f])
    r.sync(); t1 = time.perf_counter(); s1 = r.stats()
    ms = [(a - b) / 64 for a, b in zip(s1["ms_stage"], s0["ms_stage"])]
    t = (t1 - t0) / 64 * 1e3
    if best is None or t < best[0]: best = (t, ms)
st = r.stats()
print(f"speculated {st['speculated_frames']} discarded {st['discarded_speculations']} overflow {st['queue_overflow']} cap {st['queue_capacity']} merge_ms {st['ms_merge'] / max(st['frames'], 1):.4f}")
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'default'} flags={flags} FRT_CUTS={os.environ.get('FRT_CUTS', '-')} {best[0]:.3f} ms/frame stages " + " ".join(f"{m:.3f}" for m in best[1]), flush=True)
