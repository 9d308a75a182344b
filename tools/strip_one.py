"""One strip renderer of an N-way split (default: strip 4 of 8 of the 1080p frame), two streams, no exchange, 72 frames: the program
tools/strip_timeline.sh runs under rocprofv3 --kernel-trace.   python tools/strip_one.py [rank] [world] [4k]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fast-raytracing-wgpu_amd"))
import frt
args = [a for a in sys.argv[1:] if a != "4k"]
W, H = (3840, 2160) if "4k" in sys.argv[1:] else (1920, 1080)
rank = int(args[0]) if args else 4
world = int(args[1]) if len(args) > 1 else 8
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(80)]
rb, re = H * rank // world, H * (rank + 1) // world
r = frt.Renderer(scene, W, H, rows=(rb, re) if world > 1 else None, flags=frt.FLAG_PIPELINE | int(os.environ.get("FRT_EXTRA_FLAGS", "0")))
for f in range(8): r.render(cams[f])
r.sync(); t0 = time.perf_counter()
for f in range(8, 72): r.render(cams[f])
r.sync()
print(f"strip {rank} of {world}, rows {rb}..{re} of {W}x{H}: {(time.perf_counter() - t0) / 64 * 1e3:.3f} ms/frame", flush=True)
