"""frt_multi_renderer (include/frt.h): N GPUs behind ONE render call, through the C ABI alone (no torch.distributed, no Python frame loop).
On the one-GPU test box every logical device is ordinal 0: `ndev` strip renderers share the card, the halo rows move by device-to-device
copies on the streams and events the library sets up, and the gathered image must equal a single renderer's bit for bit."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(frt):
    if frt.lib().frt_device_count() < 1:
        pytest.skip("no HIP device")
    return frt


BUFS = ("BUF_RESERVOIR0", "BUF_RESERVOIR1", "BUF_RAW", "BUF_DISPLAY", "BUF_ACCUM0", "BUF_ACCUM1")


def _read_all(frt, r):
    out = {}
    for name, (buf, idx) in {"res0": (frt.BUF_RESERVOIR, 0), "res1": (frt.BUF_RESERVOIR, 1), "raw": (frt.BUF_RAW, 0), "display": (frt.BUF_DISPLAY, 0),
                             "acc0": (frt.BUF_ACCUM, 0), "acc1": (frt.BUF_ACCUM, 1)}.items():
        out[name] = r.read_buffer(buf, idx)
    return out


@pytest.mark.parametrize("ndev,W,H,frames", [(4, 320, 200, 6), (3, 160, 96, 5), (1, 96, 64, 3), (8, 480, 270, 4)])
def test_multi_equals_single_static_camera(gpu, ndev, W, H, frames):
    frt = gpu
    scene = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, scene.num_lights) for f in range(frames)]
    one = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
    multi = frt.MultiRenderer(scene, W, H, [0] * ndev)
    b = multi.boundaries()
    assert b[0] == 0 and b[-1] == H and all(y1 - y0 >= 12 for y0, y1 in zip(b, b[1:])) or ndev == 1
    for f in range(frames):
        one.render(cams[f]); multi.render(cams[f])
        assert multi.frame_count == one.frame_count == f + 1
        got, want = _read_all(frt, multi), _read_all(frt, one)       # (reads sync: also a frame boundary with nothing in flight)
        for k in got:
            assert got[k].tobytes() == want[k].tobytes(), f"frame {f}: {k} differs ({ndev} strips {b})"
    s1, sm = one.stats(), multi.stats()
    assert (sm["rays_closest"], sm["rays_any"]) == (s1["rays_closest"], s1["rays_any"])
    assert sm["halo_overflow"] == 0
    assert np.array_equal(multi.read_accum(), one.read_accum()) and np.array_equal(multi.read_display(), one.read_display())


def test_multi_without_reads_between_frames(gpu):
    """The asynchronous path: 12 frames enqueued back to back (speculated next frames, copies and kernels in flight together), one read at the end."""
    frt = gpu
    W, H, N = 640, 360, 12
    scene = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, scene.num_lights) for f in range(N)]
    one = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
    multi = frt.MultiRenderer(scene, W, H, [0, 0, 0, 0])
    for f in range(N):
        one.render(cams[f]); multi.render(cams[f])
    got, want = _read_all(frt, multi), _read_all(frt, one)
    for k in got:
        assert got[k].tobytes() == want[k].tobytes(), k
    assert multi.stats()["speculated_frames"] > 0


def test_multi_moving_camera(gpu):
    """SURVEY §8f-2 through the multi-device handle: reprojection and history fetches cross strip boundaries; motion_halo rows of the
    previous frame's spatial reservoirs ("pre") and accumulation ("post") are exchanged around every strip."""
    frt = gpu
    import _scenes
    W, H, N, K = 320, 192, 6, 8
    scene = frt.scenes.create_cornell_box()
    cams = _scenes.moving_camera_uniforms(frt, W / H, scene.num_lights, N)
    one = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
    multi = frt.MultiRenderer(scene, W, H, [0, 0, 0], motion_halo=K)
    for f in range(N):
        one.render(cams[f]); multi.render(cams[f])
    got, want = _read_all(frt, multi), _read_all(frt, one)
    for k in got:
        assert got[k].tobytes() == want[k].tobytes(), k
    assert multi.stats()["halo_overflow"] == 0


def test_multi_reset_and_errors(gpu):
    frt = gpu
    W, H = 160, 96
    scene = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, scene.num_lights) for f in range(3)]
    one = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
    multi = frt.MultiRenderer(scene, W, H, [0, 0])
    for r in (one, multi):
        r.render(cams[0]); r.render(cams[1]); r.reset()
        assert r.frame_count == 0
        r.render(cams[0]); r.render(cams[1])         # frame_count restarts, buffers keep their contents (state.rs:152)
    assert np.array_equal(multi.read_accum(), one.read_accum())
    with pytest.raises(frt.FrtError):
        frt.MultiRenderer(scene, W, H, [0, 99])          # device ordinal out of range
    with pytest.raises(frt.FrtError):
        frt.MultiRenderer(scene, 64, 40, [0] * 8)       # strips thinner than the halo


def test_multi_reset_every_frame_like_a_moving_reference_camera(gpu):
    """The reference's host loop with a moving camera: frame_count = 0 before EVERY frame the camera moves (state.rs:152; INTEGRATION.md section 4:
    frt_multi_renderer_reset with motion_halo_rows = K). T-merge still reprojects into the previous frame's spatial reservoirs up to K rows
    outside the strip (restir.wgsl:846-900 uses frame_count for the seed only), so the "pre" rows must travel although frame_count says 0
    (they were gated on frame_count and the strip-boundary pixels read stale halo rows)."""
    frt = gpu
    import _scenes
    W, H, N, K = 320, 192, 7, 8
    scene = frt.scenes.create_cornell_box()
    cams = _scenes.moving_camera_uniforms(frt, W / H, scene.num_lights, N)
    one = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
    multi = frt.MultiRenderer(scene, W, H, [0, 0, 0], motion_halo=K)
    for f in range(N):
        if f not in (0, 5):          # "camera moved" (the counter runs on over frames 4 -> 5: both forms in one sequence)
            one.reset(); multi.reset()
        cams[f].frame_count = one.frame_count
        assert multi.frame_count == one.frame_count
        one.render(cams[f]); multi.render(cams[f])
        got, want = _read_all(frt, multi), _read_all(frt, one)
        for k in got:
            assert got[k].tobytes() == want[k].tobytes(), f"frame {f}: {k} differs"
    assert multi.stats()["halo_overflow"] == 0


@pytest.mark.parametrize("strip,step", [(1, 0), (2, 1), (0, 1)])
def test_multi_strip_failure_latches_until_clear(gpu, strip, step):
    """A strip whose step fails leaves the other strips with a half-enqueued frame (their T-merge is in flight, `frame_open`): the handle must refuse
    further frames (FRT_ERR_STATE) instead of re-running step A on them, and frt_multi_renderer_clear must bring every strip back to the state right
    after create — the frames rendered afterwards equal a fresh single renderer's bit for bit."""
    frt = gpu
    W, H = 240, 160
    scene = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, scene.num_lights) for f in range(4)]
    multi = frt.MultiRenderer(scene, W, H, [0, 0, 0])
    multi.render(cams[0]); multi.render(cams[1])
    multi.inject_failure(strip, step)
    with pytest.raises(frt.FrtError, match="injected failure"):
        multi.render(cams[2])
    assert multi.frame_count == 2                      # the failed frame did not count
    for call in (lambda: multi.render(cams[2]), multi.reset, multi.read_display):
        with pytest.raises(frt.FrtError, match="frt_multi_renderer_clear"):
            call()
    multi.clear()
    assert multi.frame_count == 0
    one = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
    for f in range(4):
        one.render(cams[f]); multi.render(cams[f])
    got, want = _read_all(frt, multi), _read_all(frt, one)
    for k in got:
        assert got[k].tobytes() == want[k].tobytes(), k
    s1, sm = one.stats(), multi.stats()
    assert (sm["rays_closest"], sm["rays_any"], sm["frames"]) == (s1["rays_closest"], s1["rays_any"], 4)


def test_multi_gather_on_the_device(gpu):
    """frt_multi_renderer_gather: the strips' rows land in ONE device buffer (peer copies on the strips' copy streams, no host staging), ordered
    behind the frames enqueued so far and in front of the caller's stream; the next frame's writers wait for the copies."""
    frt = gpu
    import torch
    W, H, N = 320, 200, 5
    scene = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, scene.num_lights) for f in range(N + 1)]
    one = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
    multi = frt.MultiRenderer(scene, W, H, [0, 0, 0, 0])
    disp = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    side = torch.cuda.Stream()
    for f in range(N):
        one.render(cams[f]); multi.render(cams[f])
    multi.gather(frt.BUF_DISPLAY, 0, 0, disp.data_ptr(), stream=side.cuda_stream)      # asynchronous: `side` is ordered behind the copies
    multi.gather(frt.BUF_ACCUM, (N - 1) & 1, 0, acc.data_ptr(), stream=side.cuda_stream)
    multi.render(cams[N]); one.render(cams[N])            # the next frame is enqueued while the rows may still be travelling
    side.synchronize()
    want = frt.Renderer(scene, W, H, flags=frt.FLAG_PIPELINE)
    for f in range(N):
        want.render(cams[f])
    assert np.array_equal(disp.cpu().numpy(), want.read_display())
    assert np.array_equal(acc.cpu().numpy(), want.read_accum())
    assert np.array_equal(multi.read_display(), one.read_display()) and np.array_equal(multi.read_accum(), one.read_accum())
    pa = multi.peer_access()
    assert pa == {"neighbour_pairs_on_different_devices": 0, "pairs_with_peer_access": 0}      # every strip on ordinal 0 here
    with pytest.raises(frt.FrtError):
        multi.gather(frt.BUF_DISPLAY, 0, 99, disp.data_ptr())


def test_multi_jitter_is_refused_for_strips(gpu):
    frt = gpu
    scene = frt.scenes.create_cornell_box()
    multi = frt.MultiRenderer(scene, 160, 96, [0, 0])
    multi.set_jitter((0.0, 0.0))
    with pytest.raises(frt.FrtError, match="strips"):
        multi.set_jitter((0.25, -0.25))
    single = frt.MultiRenderer(scene, 160, 96, [0])
    single.set_jitter((0.25, -0.25))          # one strip = a whole-frame renderer: jitter is its business
    cam = frt.CameraController().build_uniform(160 / 96, 0, scene.num_lights)
    single.render(cam)
    ref = frt.Renderer(scene, 160, 96, flags=frt.FLAG_PIPELINE)
    ref.set_jitter((0.25, -0.25)); ref.render(cam)
    assert np.array_equal(single.read_display(), ref.read_display())


def test_bench_native_rehearsal_on_one_gpu(gpu):
    """`bench.py --gpus 3 --native`: ONE process, three strip renderers through frt_multi_renderer (all on this box's one GPU), one JSON line."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FRT_BENCH_ONE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--native", "--steps", "6", "--warmup", "2", "--no-4k"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 3 and res["config"]["native"] and res["value"] > 0
    assert 17.0e6 < res["config"]["rays_per_frame"] < 18.5e6          # the same frame as one renderer traces (17.76 M rays)
