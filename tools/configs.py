"""One-GPU throughput of every BASELINE.json workload (configs[1..4]; [3] and [4] are the synthetic stand-ins of tests/_scenes.py)
plus the 100-light ReSTIR scene: ms/frame, Mrays/s, rays/frame, per-stage ms. Only configs[1] is a bench line (bench.py); the
others are parity-test cases and are timed here for DESIGN.md §7. The oracle is loaded only because tests/_scenes.py builds each
mesh scene through both builders; nothing is rendered with it."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, frt
from _oracle import Oracle
import _scenes
orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))


def run(name, scene, W, H, depth, nlights, frames=40, warm=8):
    cams = [frt.CameraController().build_uniform(W / H, f, nlights) for f in range(frames)]
    r = frt.Renderer(scene, W, H, max_depth=depth, flags=frt.FLAG_TIMING | frt.FLAG_OVERLAP_POST | int(os.environ.get("FRT_EXTRA_FLAGS", "0")))      # e.g. 32 = FLAG_WALK_WIDE
    for f in range(warm): r.render(cams[f])
    r.sync(); s0 = r.stats(); t0 = time.perf_counter()
    for f in range(warm, frames): r.render(cams[f])
    r.sync(); t1 = time.perf_counter(); s1 = r.stats()
    n = frames - warm
    rays = s1["rays_closest"] + s1["rays_any"] - s0["rays_closest"] - s0["rays_any"]
    ms = [(a - b) / n for a, b in zip(s1["ms_stage"], s0["ms_stage"])]
    c = scene.counts()
    out = {"extra_flags": int(os.environ.get("FRT_EXTRA_FLAGS", "0")), "workload": name, "triangles": c["tris"], "lights": c["lights"], "width": W, "height": H, "max_depth": depth,
           "ms_per_frame": (t1 - t0) / n * 1e3, "Mrays_per_s": rays / (t1 - t0) / 1e6, "Mrays_per_frame": rays / n / 1e6,
           "stage_ms": dict(zip(("gbuffer", "temporal", "spatial", "post"), ms))}
    print(json.dumps(out), flush=True)
    del r


only = set(sys.argv[1:])      # e.g. `configs.py 3 4`: those workloads only ("r" = the ReSTIR scene)
if not only or "1" in only or "2" in only:
    cornell = frt.scenes.create_cornell_box()
    if not only or "1" in only: run("configs[1] Cornell 1080p d8", cornell, 1920, 1080, 8, 2)
    if not only or "2" in only: run("configs[2] Cornell 2160p d8 (one GPU)", cornell, 3840, 2160, 8, 2)
if not only or "r" in only:
    restir = frt.scenes.create_restir_scene()
    run("ReSTIR 100-light scene 1080p d8", restir, 1920, 1080, 8, restir.counts()["lights"])
if not only or "3" in only:
    fb, _ = _scenes.bumpy_sphere_in_box(frt, orc)
    run("configs[3] stand-in: 82k-triangle blob in the box 1080p d8", fb, 1920, 1080, 8, 1)
if not only or "4" in only:
    fb, _ = _scenes.colonnade(frt, orc)
    run("configs[4] stand-in: 250k-triangle colonnade 2160p d16 (one GPU)", fb, 3840, 2160, 16, 1, frames=24)
