import os, sys, time
sys.path.insert(0, "fast-raytracing-wgpu_amd")
import frt
W, H = 1920, 1080
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(40)]
for cuts in ("2", "3", "4", "3,4", "2,3", "3,5"):
    os.environ["FRT_CUTS"] = cuts
    for world in (2, 4, 8):
        worst = 0
        for rank in range(world):
            rb, re = H * rank // world, H * (rank + 1) // world
            r = frt.Renderer(scene, W, H, rows=(rb, re), flags=frt.FLAG_TIMING | frt.FLAG_PIPELINE)
            for f in range(8): r.render(cams[f])
            r.sync(); t0 = time.perf_counter()
            for f in range(8, 40): r.render(cams[f])
            r.sync(); dt = (time.perf_counter() - t0) / 32 * 1e3
            worst = max(worst, dt); del r
        print(f"cuts={cuts} world {world}: slowest strip {worst:.3f} ms", flush=True)
