"""One experimental kernel family of lib/libfrt_exp.so (`make experiments`) against the oracle, every buffer of every frame. Run by
tests/test_experiments.py with FRT_LIB pointing at the experiments build and the family's knob in the environment."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import frt
    from _oracle import Oracle
    flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    W, H, N = 160, 96, 4
    fs = frt.scenes.create_cornell_box()
    orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    r = frt.Renderer(fs, W, H, flags=flags)
    ro = os_.renderer(W, H, 8, True, 16)
    bad = []
    for f in range(N):
        cam = frt.CameraController().build_uniform(W / H, f, 2)
        r.render(cam); ro.render(cam)
        cur = f % 2
        for name, (b, idx) in {"gpos": (0, cur), "gnormal": (1, cur), "galbedo": (2, cur), "res0": (4, 0), "res1": (4, 1), "raw": (5, 0), "display": (6, 0), "accum": (7, cur)}.items():
            if r.read_buffer(b, idx).tobytes() != ro.read(b, idx).tobytes():
                bad.append((f, name))
    st, so = r.stats(), ro.stats()["total"]
    print(json.dumps({"ok": not bad and (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"]), "bad": bad[:6], "lib": frt._lib.LIB_PATH}))


if __name__ == "__main__":
    main()
