#!/usr/bin/env python3
"""Render N frames with libfrt.so on the GPU and write the display buffer (post_processed_texture, state.rs:226-278) as a PNG.
usage: python tools/render_png.py [--scene cornell|restir] [--size 1280x720] [--frames 64] [--out out.png]"""
import argparse, os, struct, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
import frt


def write_png(path, rgba):
    h, w, _ = rgba.shape
    raw = b"".join(b"\x00" + rgba[y, :, :3].tobytes() for y in range(h))
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell"); ap.add_argument("--size", default="1280x720")   # main.rs:106-122 default --scale
    ap.add_argument("--frames", type=int, default=64); ap.add_argument("--out", default="gpurun_out/render.png")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    scene = frt.scenes.create_cornell_box() if a.scene == "cornell" else frt.scenes.create_restir_scene()
    r = frt.Renderer(scene, w, h)
    cam = frt.CameraController()
    for _ in range(a.frames):
        r.render(cam.build_uniform(r.aspect_ratio(), r.frame_count, scene.num_lights))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    write_png(a.out, r.read_display())
    st = r.stats()
    print(f"{a.out}: {w}x{h}, {a.frames} frames, {st['rays_closest'] + st['rays_any']} rays")


if __name__ == "__main__":
    main()
