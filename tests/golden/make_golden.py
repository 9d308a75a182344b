"""Generates tests/golden/cornell_128.npz with the CPU oracle.

PARITY UNPINNED: the reference ships no tests, fixtures or golden images, and cannot be built or run in this pipeline
(no rustc / Vulkan ray-query device). These vectors therefore pin THIS BUILD's oracle (and through it the HIP path) against
regressions; they are not outputs of the reference. Contents: 128x128 Cornell Box, MAX_DEPTH in {1, 8},
frames {0, 1, 7}: G-buffer (frame 0), spatial reservoirs, radiance (f16 bits), accumulation (f32), display; ray counts.
Usage: python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle import Oracle  # noqa: E402

W = H = 128
FRAMES = (0, 1, 7)


def main():
    orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))
    out = {}
    for depth in (1, 8):
        scene = orc.cornell()
        r = scene.renderer(W, H, depth, False, 8)          # brute-force loop over the 1,320 triangles: no BVH involved
        for f in range(max(FRAMES) + 1):
            r.render(orc.camera(1.0, f, 2))
            if f in FRAMES:
                cur = f % 2
                k = f"d{depth}_f{f}_"
                if f == 0 and depth == 8:
                    out["gpos"] = r.read(0, cur).view(np.float32)
                    out["gnormal"] = r.read(1, cur).view(np.float32)
                    out["galbedo"] = r.read(2, cur)
                out[k + "reservoir"] = r.read(4, 1).view(np.uint32)
                out[k + "raw"] = r.read(5, 0).view(np.uint16)
                out[k + "accum"] = r.read(7, cur).view(np.float32)
                out[k + "display"] = r.read(6, 0)
        st = r.stats()["total"]
        out[f"d{depth}_rays"] = np.array([st["closest"], st["any"]], np.uint64)
    np.savez_compressed(os.path.join(HERE, "cornell_128.npz"), **out)
    print("wrote", os.path.join(HERE, "cornell_128.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
