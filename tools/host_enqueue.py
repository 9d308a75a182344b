"""Host time of one frame's enqueue (no GPU wait): a single renderer vs frt_multi_renderer with N strips on one GPU. If the multi-device
renderer's enqueue takes longer than a strip's GPU frame (0.36 ms for 1/8 of a 1080p frame), the host thread is the limit, not the GPUs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fast-raytracing-wgpu_amd"))
import frt
W, H = 1920, 1080
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(80)]
def run(r, sync):
    for f in range(8): r.render(cams[f])
    sync(); t0 = time.perf_counter()
    for f in range(8, 72): r.render(cams[f])
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    return (t1 - t0) / 64 * 1e3, (t2 - t0) / 64 * 1e3
r = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
print("single renderer: enqueue %.3f ms/frame, total %.3f" % run(r, r.sync)); del r
for n in (2, 4, 8):
    m = frt.MultiRenderer(scene, W, H, [0] * n)
    print(f"multi renderer, {n} strips on one GPU: enqueue %.3f ms/frame, total %.3f" % run(m, m.sync)); del m
