"""One experimental kernel family of lib/libfrt_exp.so (`make experiments`) against the oracle, every buffer of every frame. Run by
tests/test_experiments.py with FRT_LIB pointing at the experiments build and the family's knob in the environment.
    _exp_worker.py <flags> [scene W H depth frames]      scene: cornell | restir | blob5k (oracle: brute force, nothing of the product's) | bumpy82k | colonnade250k
    _exp_worker.py equal <flags,flags,...>               1920x1080 Cornell Box, 4 frames: the renderers with these flag sets give the same image and ray counts"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def equal(flag_sets):
    import numpy as np
    import frt
    W, H, N = 1920, 1080, 4
    fs = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(N)]
    out = []
    for fl in flag_sets:
        r = frt.Renderer(fs, W, H, flags=frt.FLAG_PIPELINE | fl)
        for c in cams:
            r.render(c)
        st = r.stats()
        out.append((r.read_accum(), r.read_buffer(frt.BUF_RESERVOIR, 1), st["rays_closest"], st["rays_any"]))
        del r
    ok = all(np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1]) and o[2:] == out[0][2:] for o in out[1:])
    print(json.dumps({"ok": bool(ok), "lib": frt._lib.LIB_PATH, "rays": [int(out[0][2]), int(out[0][3])]}))


def main():
    import numpy as np
    import frt
    from _oracle import Oracle
    import _scenes
    if len(sys.argv) > 1 and sys.argv[1] == "equal":
        return equal([int(x) for x in sys.argv[2].split(",")])
    flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    which = sys.argv[2] if len(sys.argv) > 2 else "cornell"
    W, H, depth, N = (int(a) for a in sys.argv[3:7]) if len(sys.argv) > 6 else (160, 96, 8, 4)
    orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))
    use_bvh = True
    if which == "blob5k":
        fs, os_ = _scenes.bumpy_sphere_in_box(frt, orc, subdiv=4, share_bvh=False); use_bvh = False
    elif which == "bumpy82k":
        fs, os_ = _scenes.bumpy_sphere_in_box(frt, orc, subdiv=6)
    elif which == "colonnade250k":
        fs, os_ = _scenes.colonnade(frt, orc)
    else:
        fs = frt.scenes.create_cornell_box() if which == "cornell" else frt.scenes.create_restir_scene()
        os_ = orc.cornell() if which == "cornell" else orc.restir_scene()
        os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    r = frt.Renderer(fs, W, H, max_depth=depth, flags=flags)
    ro = os_.renderer(W, H, depth, use_bvh, 16)
    bad = []
    for f in range(N):
        cam = frt.CameraController().build_uniform(W / H, f, fs.num_lights)
        r.render(cam); ro.render(cam)
        cur = f % 2
        for name, (b, idx) in {"gpos": (0, cur), "gnormal": (1, cur), "galbedo": (2, cur), "res0": (4, 0), "res1": (4, 1), "raw": (5, 0), "display": (6, 0), "accum": (7, cur)}.items():
            if r.read_buffer(b, idx).tobytes() != ro.read(b, idx).tobytes():
                bad.append((f, name))
    st, so = r.stats(), ro.stats()["total"]
    print(json.dumps({"ok": not bad and (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"]), "bad": bad[:6], "lib": frt._lib.LIB_PATH,
                      "tree": fs.tree_stats()}))


if __name__ == "__main__":
    main()
