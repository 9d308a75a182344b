"""Randomised parity sweep on the GPU box: libfrt.so (HIP kernels through the C ABI) against the oracle on image sizes, depths, schedules, queue
capacities and camera sequences nobody picked by hand. Test infrastructure (the oracle is the checker); not part of the product, not a pytest module.

    python tools/fuzz_parity.py [--seed S] [--seconds T] [--cases N] [--only K]

Two kinds of case, drawn at random:
  A  one renderer vs the oracle: every buffer of every frame bit for bit (tests/test_hostcheck_parity.py::compare_all) + the exact ray counters;
     scene (Cornell Box / ReSTIR scene / 5k-triangle blob with the brute-force oracle / tests/_scenes.py::random_scene), W x H (ragged tiles included), MAX_DEPTH, flags
     (plain / two streams / three G-buffer sets), continuation-queue capacity (tiny ones overflow: paths finish in place), cut depths, camera
     (static, moving with random steps, starts and stops, frame counter restarted while it moves as state.rs:152 does, Halton-jittered).
  B  strips vs the whole image: frt_multi_renderer (every strip on device 0: peer copies become device copies, the orderings are the real ones) and
     host-exchanged strip renderers against one renderer, static and moving camera (motion halo), bit for bit + ray totals.
Every failure prints the case's parameters (re-run it alone with --seed S --only K). Exit code = number of failing cases."""
import argparse
import os
import sys
import time
import traceback
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (one HIP runtime in the process, as in tests/conftest.py)
import frt  # noqa: E402
from _oracle import Oracle  # noqa: E402
import _scenes  # noqa: E402
from test_hostcheck_parity import compare_all  # noqa: E402


def camera_sequence(rng, aspect, nl, frames):
    """-> (list of CameraUniform, set of frames before which the host restarts the frame counter, description)"""
    mode = rng.choice(["static", "moving", "start_stop", "restart"], p=[0.35, 0.3, 0.2, 0.15])
    static = [frt.CameraController().build_uniform(aspect, f, nl) for f in range(frames)]
    if mode == "static":
        return static, set(), mode
    step = tuple(float(v) for v in rng.uniform(-0.04, 0.04, 3))
    yaw = float(rng.uniform(-0.02, 0.02))
    moving = _scenes.moving_camera_uniforms(frt, aspect, nl, frames, step=step, yaw_step=yaw)
    desc = f"{mode} step {tuple(round(s, 4) for s in step)} yaw {yaw:.4f}"
    if mode == "moving":
        return moving, set(), desc
    if mode == "start_stop":       # static -> moving -> the last moving pose held
        a = int(rng.integers(1, max(2, frames - 1)))
        b = int(rng.integers(a, frames))
        seq = static[:a] + moving[a:b + 1] + [moving[b]] * (frames - b - 1)
        return seq[:frames], set(), desc + f" [{a},{b}]"
    restarts = set(int(f) for f in range(1, frames) if rng.random() < 0.5)
    return moving, restarts, desc + f" restarts {sorted(restarts)}"


def case_a(rng, orc, k):
    which = rng.choice(["cornell", "restir", "blob", "random"], p=[0.4, 0.2, 0.1, 0.3])
    brute = which == "blob" or (which != "random" and rng.random() < 0.1)
    W, H = int(rng.integers(8, 300)), int(rng.integers(8, 200))
    if brute:
        W, H = min(W, 96), min(H, 64)
    depth = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 8, 12, 16]))
    frames = int(rng.integers(2, 6))
    flags = int(rng.choice([0, frt.FLAG_PIPELINE, frt.FLAG_PIPELINE | frt.FLAG_THIRD_GSET]))
    qcap = int(rng.choice([0, 0, 0, 64, 1000, W * H // 10 + 1]))
    cuts = [None, None, None, [], [1], [2, 3], [3], [1, 2, 3, 4], [5, 7]][int(rng.integers(0, 9))]
    desc = f"A#{k} {which}{' brute' if brute else ''} {W}x{H} depth {depth} frames {frames} flags {flags} qcap {qcap} cuts {cuts}"
    if which == "cornell":
        fs, os_ = frt.scenes.create_cornell_box(), orc.cornell()
    elif which == "restir":
        fs, os_ = frt.scenes.create_restir_scene(), orc.restir_scene()
    elif which == "random":      # a room of randomly placed / rotated / mirrored shapes in random diffuse, glossy, metal, glass, textured materials, 1 - 3 lights
        sseed = int(rng.integers(9, 10 ** 6))
        fs, os_, _ = _scenes.random_scene(frt, orc, sseed)
        desc += f" scene seed {sseed}"
    else:
        fs, os_ = _scenes.bumpy_sphere_in_box(frt, orc, subdiv=4, share_bvh=False)
    if not brute and which != "random":
        os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    jscale = float(rng.choice([1.0, 5.0, 23.0])) if rng.random() < 0.12 else 0.0
    r = frt.Renderer(fs, W, H, max_depth=depth, flags=flags, queue_capacity=qcap, cuts=cuts)
    ro = os_.renderer(W, H, depth, not brute, 16)
    if jscale:       # the jitter plumbing the reference multiplies by zero (camera.rs:202-203): sheared projection, bilinear post taps
        ctl, cams, restarts, jits = frt.CameraController(), [], set(), []
        for f in range(frames):
            jits.append(ctl.get_halton_jitter(f, W, H, jscale))
            cams.append(ctl.build_uniform(W / H, f, fs.num_lights, jits[-1])); ctl.commit_frame()
        desc += f" camera jittered x{jscale}"
    else:
        cams, restarts, cdesc = camera_sequence(rng, W / H, fs.num_lights, frames)
        jits = None
        desc += " camera " + cdesc
    for f, cam in enumerate(cams):
        if f in restarts:
            r.reset(); ro.restart_counter()
        if restarts:
            cam.frame_count = r.frame_count
        if jits:
            ro.set_jitter(jits[f]); ro.render(cam); r.render(cam, jitter=jits[f])
        else:
            r.render(cam); ro.render(cam)
        compare_all(r.read_buffer, ro.read, f, desc)
        assert r.frame_count == ro.frame_count, desc
    st, so = r.stats(), ro.stats()["total"]
    assert (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"]), desc + " ray counters"
    return desc + (f" (overflow {st['queue_overflow']})" if st["queue_overflow"] else "")


def case_b(rng, k):
    from frt.dist import StripPlan, exchange_halos_host, check_halo
    which = rng.choice(["cornell", "restir"], p=[0.75, 0.25])
    fs = frt.scenes.create_cornell_box() if which == "cornell" else frt.scenes.create_restir_scene()
    n = int(rng.integers(2, 5))
    moving = rng.random() < 0.5
    K = int(rng.integers(3, 9)) if moving else 0
    need = max(12, K + 1)
    W = int(rng.integers(16, 260))
    H = int(rng.integers(n * need + 2, n * need + 120))
    depth = int(rng.choice([1, 3, 8, 8, 16]))
    frames = int(rng.integers(2, 6))
    flags = int(rng.choice([0, frt.FLAG_PIPELINE, frt.FLAG_PIPELINE | frt.FLAG_THIRD_GSET]))
    form = rng.choice(["multi", "host"])
    desc = f"B#{k} {which} {W}x{H} {n} strips ({form}) depth {depth} frames {frames} flags {flags} motion halo {K}"
    if moving:       # small steps: the reprojection must stay inside the motion halo (check_halo / halo_overflow says when it does not)
        step = tuple(float(v) for v in rng.uniform(-0.01, 0.01, 3))
        cams = _scenes.moving_camera_uniforms(frt, W / H, fs.num_lights, frames, step=step, yaw_step=float(rng.uniform(-0.004, 0.004)))
        desc += f" step {tuple(round(s, 4) for s in step)}"
    else:
        cams = [frt.CameraController().build_uniform(W / H, f, fs.num_lights) for f in range(frames)]
    whole = frt.Renderer(fs, W, H, max_depth=depth)
    for c in cams: whole.render(c)
    want, wd = whole.read_accum(), whole.read_display()
    wres = whole.read_buffer(frt.BUF_RESERVOIR, 1)
    wtot = whole.stats()["rays_closest"] + whole.stats()["rays_any"]
    if form == "multi":
        m = frt.MultiRenderer(fs, W, H, [0] * n, max_depth=depth, motion_halo=K, flags=flags)
        desc += f" bounds {m.boundaries()}"
        for c in cams: m.render(c)
        st = m.stats()
        if st["halo_overflow"]:
            return desc + " SKIPPED: the camera outran the motion halo (detected)"
        assert np.array_equal(m.read_accum(), want), desc + " accum"
        assert np.array_equal(m.read_display(), wd), desc + " display"
        assert np.array_equal(m.read_buffer(frt.BUF_RESERVOIR, 1), wres), desc + " reservoirs"
        assert st["rays_closest"] + st["rays_any"] == wtot, desc + " ray totals"
        return desc
    # host-exchanged strips on random boundaries
    while True:
        cutsy = sorted(int(v) for v in rng.integers(need, H - need + 1, n - 1))
        b = [0] + cutsy + [H]
        if min(y1 - y0 for y0, y1 in zip(b, b[1:])) >= need:
            break
    desc += f" bounds {b}"
    plans = [StripPlan(H, n, i, boundaries=b, motion_halo=K) for i in range(n)]
    strips = [frt.Renderer(fs, W, H, max_depth=depth, rows=(p.row_begin, p.row_end), motion_halo=K, flags=flags) for p in plans]
    for f, cam in enumerate(cams):
        exchange_halos_host(strips, plans, f, when="pre")
        for s in strips: s.render_phases(cam, frt.PHASE_GBUFFER | frt.PHASE_TEMPORAL)
        for s in strips: s.render_phases(cam, frt.PHASE_SPATIAL_INNER)
        exchange_halos_host(strips, plans, f, when="mid")
        for s in strips: s.render_phases(cam, frt.PHASE_SPATIAL_EDGE)
        exchange_halos_host(strips, plans, f, when="post")
        for s in strips: s.render_phases(cam, frt.PHASE_POST); s.end_frame()
    if sum(s.stats()["halo_overflow"] for s in strips):
        return desc + " SKIPPED: the camera outran the motion halo (detected)"
    for s, p in zip(strips, plans):
        assert np.array_equal(s.read_accum()[p.row_begin:p.row_end], want[p.row_begin:p.row_end]), desc + f" accum, strip {p.rank}"
        assert np.array_equal(s.read_display()[p.row_begin:p.row_end], wd[p.row_begin:p.row_end]), desc + f" display, strip {p.rank}"
        check_halo(s)
    assert sum(s.stats()["rays_closest"] + s.stats()["rays_any"] for s in strips) == wtot, desc + " ray totals"
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--cases", type=int, default=10 ** 6)
    ap.add_argument("--only", type=int, default=-1)
    a = ap.parse_args()
    if frt.lib().frt_device_count() < 1:
        sys.exit("no HIP device visible")
    orc = Oracle(os.path.join(ROOT, "oracle", "_build", "liborc.so"))
    t0, k, fails, skipped = time.time(), 0, [], 0
    while k < a.cases and (time.time() - t0 < a.seconds or a.only >= 0):
        rng = np.random.default_rng([a.seed, k])       # every case has its own stream: --only K reproduces case K exactly
        if a.only >= 0 and k != a.only:
            k += 1
            if k > a.only: break
            continue
        try:
            d = case_a(rng, orc, k) if rng.random() < 0.6 else case_b(rng, k)
            skipped += "SKIPPED" in d
            print(f"ok   {d}", flush=True)
        except Exception as e:       # noqa: BLE001
            fails.append(k)
            print(f"FAIL case {k} (seed {a.seed}): {e}", flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
        k += 1
    print(f"fuzz_parity: seed {a.seed}: {k} cases in {time.time() - t0:.0f} s, {len(fails)} failed {fails}, {skipped} skipped", flush=True)
    sys.exit(min(len(fails), 100))


if __name__ == "__main__":
    main()
