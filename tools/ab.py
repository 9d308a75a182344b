"""A/B timing of renderer variants in ONE process (interleaved rounds, §5.4 rule 24 of the CDNA guide) at 1920x1080.
Edit `rs` to compare flags (frt.FLAG_COMPACTION, frt.FLAG_OVERLAP_POST) or environment knobs (FRT_CUTS, FRT_BVH_LEAF) set before
a Renderer / scene is created. The sweeps quoted in HISTORY.md §6 were produced with this script."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
import frt
W, H = 1920, 1080
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(40)]
rs = {"default": frt.Renderer(scene, W, H, flags=frt.FLAG_TIMING)}
for rnd in range(2):
    for k, r in rs.items():
        r.clear()
        for f in range(8): r.render(cams[f])
        r.sync(); s0 = r.stats(); t0 = time.perf_counter()
        for f in range(8, 40): r.render(cams[f])
        r.sync(); t1 = time.perf_counter(); s1 = r.stats()
        rays = s1["rays_closest"] + s1["rays_any"] - s0["rays_closest"] - s0["rays_any"]
        ms = [(a - b) / 32 for a, b in zip(s1["ms_stage"], s0["ms_stage"])]
        print(f"round {rnd} {k:8s} {(t1 - t0) / 32 * 1e3:.3f} ms/frame {rays / (t1 - t0) / 1e6:.0f} Mrays/s stages " + " ".join(f"{m:.3f}" for m in ms), flush=True)
