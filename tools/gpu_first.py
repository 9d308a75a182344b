"""First-contact GPU script: smoke, mid-size parity of every buffer, 1080p timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import frt
from _oracle import Oracle

g.smoke()
orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))
scene = frt.scenes.create_cornell_box()
osc = orc.cornell(); osc.set_bvh(scene.get("bvh2_nodes"), scene.get("bvh2_tri_index"))
W = H = 256
r = frt.Renderer(scene, W, H); ro = osc.renderer(W, H, 8, True, 16)
names = {0: 'gpos', 1: 'gnormal', 2: 'galbedo', 3: 'gmotion', 4: 'reservoir', 5: 'raw', 6: 'display', 7: 'accum'}
bad = 0
for f in range(6):
    cam = frt.CameraController().build_uniform(W / H, f, 2)
    r.render(cam); ro.render(cam)
    for b in range(8):
        for idx in ((0, 1) if b in (0, 1, 2, 4, 7) else (0,)):
            a = r.read_buffer(b, idx); c = ro.read(b, idx)
            if a.tobytes() != c.tobytes():
                d = (a.view(np.uint32) != c.view(np.uint32)); bad += 1
                print(f'frame {f} {names[b]}[{idx}] DIFF words={d.sum()}')
print('256x256 parity mismatching buffers:', bad, 'rays', r.stats()['rays_closest'], r.stats()['rays_any'], ro.stats()['total'])
del r
W, H = 1920, 1080
r = frt.Renderer(scene, W, H, flags=frt.FLAG_TIMING)
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(24)]
for f in range(4): r.render(cams[f])
r.sync(); s0 = r.stats()
t0 = time.time()
for f in range(4, 24): r.render(cams[f])
r.sync(); t1 = time.time(); s1 = r.stats()
rays = (s1['rays_closest'] + s1['rays_any']) - (s0['rays_closest'] + s0['rays_any'])
print(f"1080p: {(t1 - t0) / 20 * 1e3:.3f} ms/frame, {rays / 20 / 1e6:.2f} Mrays/frame, {rays / (t1 - t0) / 1e6:.1f} Mrays/s")
print('stage ms/frame', [(a - b) / 20 for a, b in zip(s1['ms_stage'], s0['ms_stage'])])
