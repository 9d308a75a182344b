"""Per-frame time of ONE strip renderer (no exchange) for several world sizes: the latency floor of strong scaling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
import frt
W, H = 1920, 1080
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(40)]
for cuts, extra in (("0", frt.FLAG_OVERLAP_POST), ("3", frt.FLAG_OVERLAP_POST), ("default", frt.FLAG_OVERLAP_POST)):
    if cuts == "default": os.environ.pop("FRT_CUTS", None)      # the renderer's own rule
    else: os.environ["FRT_CUTS"] = cuts                         # read by frt_renderer_create
    for world in (1, 2, 4, 8):
        worst = 0
        for rank in range(world):
            rb, re = H * rank // world, H * (rank + 1) // world
            r = frt.Renderer(scene, W, H, rows=(rb, re) if world > 1 else None, flags=frt.FLAG_TIMING | extra)
            for f in range(8):
                r.render(cams[f])
            r.sync(); s0 = r.stats(); t0 = time.perf_counter()
            for f in range(8, 40):
                r.render(cams[f])
            r.sync(); dt = (time.perf_counter() - t0) / 32 * 1e3; s1 = r.stats()
            ms = [(a - b) / 32 for a, b in zip(s1["ms_stage"], s0["ms_stage"])]
            worst = max(worst, dt)
            if rank == 0:
                print(f"cuts={cuts} world {world} rank {rank}: {dt:.3f} ms/frame stages " + " ".join(f"{m:.3f}" for m in ms), flush=True)
            del r
        print(f"cuts={cuts} world {world}: slowest strip {worst:.3f} ms -> ideal-exchange speedup bound")
