"""Frame time when the display buffer (8.3 MB RGBA8) is read back to the host after every frame, as the Rust shim of INTEGRATION.md does to present it."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fast-raytracing-wgpu_amd"))
import torch, frt
W, H = 1920, 1080
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(80)]
r = frt.Renderer(scene, W, H, flags=frt.FLAG_OVERLAP_POST)
for f in range(8): r.render(cams[f])
r.sync(); s0 = r.stats(); t0 = time.perf_counter()
for f in range(8, 72):
    r.render(cams[f]); d = r.read_display()
t1 = time.perf_counter(); s1 = r.stats()
rays = s1["rays_closest"] + s1["rays_any"] - s0["rays_closest"] - s0["rays_any"]
print(f"render + read_display every frame: {(t1 - t0) / 64 * 1e3:.3f} ms/frame, {rays / (t1 - t0) / 1e6:.0f} Mrays/s")
