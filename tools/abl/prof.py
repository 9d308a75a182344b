import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
import torch
import frt._lib as L
if len(sys.argv) > 1 and sys.argv[1] != "default": L.LIB_PATH = os.path.join(ROOT, "tools", "abl", sys.argv[1])
import frt
W, H = 1920, 1080
rows = None
if len(sys.argv) > 2: rows = (0, int(sys.argv[2]))
scene = frt.scenes.create_cornell_box()
cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(24)]
r = frt.Renderer(scene, W, H, flags=0, rows=rows)
for f in range(24): r.render(cams[f])
r.sync()
