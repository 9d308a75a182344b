"""T3 / T4 on a real MI355X: the HIP kernels, called through the C ABI (libfrt.so), against the oracle.
Bar (BASELINE.json north_star): accumulated image within 1e-4 L_inf of the oracle; this build expects — and asserts —
bit equality of every intermediate buffer. Tolerance for the headline comparison is written below as TOL."""
import os
import numpy as np
import pytest
from test_hostcheck_parity import compare_all
from test_golden import check_against_golden, GOLD

pytestmark = pytest.mark.gpu
TOL = 1e-4    # per-pixel L_inf on the accumulation buffer (north_star)


def _strip_frame(frt, strips, plans, cam, f, serial=None):
    """One frame of several strip renderers living in this process, rows exchanged through the host: the three exchanges of
    frt/dist.py in their places, the spatial stage issued as interior rows, then edge rows (the overlap form bench.py uses)."""
    from frt.dist import exchange_halos_host
    exchange_halos_host(strips, plans, f, when="pre", serial=serial)
    for s in strips: s.render_phases(cam, frt.PHASE_GBUFFER | frt.PHASE_TEMPORAL)
    for s in strips: s.render_phases(cam, frt.PHASE_SPATIAL_INNER)
    exchange_halos_host(strips, plans, f, when="mid")
    for s in strips: s.render_phases(cam, frt.PHASE_SPATIAL_EDGE)
    exchange_halos_host(strips, plans, f, when="post")
    for s in strips: s.render_phases(cam, frt.PHASE_POST); s.end_frame()


@pytest.fixture(scope="module")
def gpu(frt):
    if frt.lib().frt_device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need an MI355X (the product has no CPU path)")
    return frt


@pytest.mark.parametrize("which,W,H,depth,frames", [("cornell", 128, 128, 8, 8), ("cornell", 128, 128, 1, 2), ("cornell", 200, 120, 8, 4),
                                                    ("cornell", 37, 19, 8, 3), ("cornell", 96, 64, 16, 3), ("restir", 160, 96, 8, 4)])
def test_kernels_match_oracle_every_buffer(gpu, orc, which, W, H, depth, frames):
    frt = gpu
    fs = frt.scenes.create_cornell_box() if which == "cornell" else frt.scenes.create_restir_scene()
    os_ = orc.cornell() if which == "cornell" else orc.restir_scene()
    os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    r = frt.Renderer(fs, W, H, max_depth=depth)
    ro = os_.renderer(W, H, depth, True, 16)
    for f in range(frames):
        cam = frt.CameraController().build_uniform(W / H, f, fs.num_lights)
        r.render(cam); ro.render(cam)
        compare_all(r.read_buffer, ro.read, f, f"{which} {W}x{H} depth {depth}")
    assert r.frame_count == frames == ro.frame_count
    got = r.read_accum(); want = ro.read(7, (frames - 1) % 2).view(np.float32)
    assert np.abs(got - want).max() <= TOL
    st, so = r.stats(), ro.stats()["total"]
    assert (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"])      # device ray counters are exact


@pytest.mark.parametrize("which,W,H,depth,frames", [("cornell", 128, 128, 8, 3), ("blob5k", 64, 48, 8, 2)])
def test_kernels_match_the_brute_force_oracle(gpu, orc, which, W, H, depth, frames):
    """The one comparison in which NO product data feeds the checker (parity caveat): the oracle is never handed the product's BVH —
    every one of its rays is a scalar loop over all triangles of its own scene (BASELINE.json configs[0]'s "scalar loop over the triangles") —
    and the HIP path, which walks its own tree, must still agree on every buffer of every frame and on the exact ray counts. Hits are defined
    by the triangle test alone (frt_trace.hpp: hit semantics), so any tree may only prune."""
    frt = gpu
    import _scenes
    if which == "cornell":
        fs, os_ = frt.scenes.create_cornell_box(), orc.cornell()
    else:
        fs, os_ = _scenes.bumpy_sphere_in_box(frt, orc, subdiv=4, share_bvh=False)
    r = frt.Renderer(fs, W, H, max_depth=depth, flags=frt.FLAG_PIPELINE)
    ro = os_.renderer(W, H, depth, False, 16)          # use_bvh = False: brute force
    for f in range(frames):
        cam = frt.CameraController().build_uniform(W / H, f, fs.num_lights)
        r.render(cam); ro.render(cam)
        compare_all(r.read_buffer, ro.read, f, f"{which} {W}x{H} depth {depth}, brute-force oracle")
    st, so = r.stats(), ro.stats()["total"]
    assert (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"])


@pytest.mark.parametrize("depth", [1, 8])
def test_kernels_reproduce_golden(gpu, depth):
    frt = gpu
    gold = np.load(GOLD)
    fs = frt.scenes.create_cornell_box()
    r = frt.Renderer(fs, 128, 128, max_depth=depth)
    check_against_golden(lambda f: r.render(frt.CameraController().build_uniform(1.0, f, 2)), r.read_buffer, depth, gold)
    st = r.stats()
    assert [st["rays_closest"], st["rays_any"]] == gold[f"d{depth}_rays"].tolist()


def test_full_size_properties(gpu, orc):
    """BASELINE.json configs[1] (1920x1080, 8 bounces): size-independent properties + a band checked against the oracle."""
    frt = gpu
    W, H, N = 1920, 1080, 6
    fs = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(N)]
    r = frt.Renderer(fs, W, H)
    for c in cams:
        r.render(c)
    acc = r.read_accum(); disp = r.read_display(); st = r.stats()
    pos = r.read_buffer(frt.BUF_GPOS, (N - 1) % 2).view(np.float32)
    res = r.read_buffer(frt.BUF_RESERVOIR, 1)
    # determinism: a second renderer reproduces every bit (seeds depend only on pixel index and frame, restir.wgsl:797-798)
    r2 = frt.Renderer(fs, W, H)
    for c in cams:
        r2.render(c)
    assert r2.read_accum().tobytes() == acc.tobytes() and r2.read_display().tobytes() == disp.tobytes()
    assert r2.stats()["rays_closest"] == st["rays_closest"] and r2.stats()["rays_any"] == st["rays_any"]
    # background pixels: zero reservoir and zero radiance (restir_spatial.wgsl:874-884); post may bleed a little neighbour colour in
    bg = pos[..., 3] < 0
    raw = r.read_buffer(frt.BUF_RAW, 0)
    assert bg.any() and not res[bg].any() and not raw[bg].any() and acc[bg][:, :3].max() < 0.05
    # light quad pixels: trace_path returns exactly the emission (restir.wgsl:543-552) -> p_hat = 10, W ~ 1 in the spatial reservoir,
    # radiance ~ (10,10,10); the accumulated value is only near 10 (the bilateral filter mixes in ceiling pixels at the quad's rim)
    light = pos[..., 3] == 6.0
    rl = res[light].view(np.float32)
    assert light.sum() > 1000 and np.all(rl[:, 7] == 10.0) and np.all(np.abs(rl[:, 3] - 1.0) < 0.1)
    assert np.abs(raw[light].view(np.float16)[:, :3].astype(np.float32) - 10.0).max() < 1.0 and acc[light][:, :3].min() > 5.0
    assert np.all(acc[..., 3] == 1.0) and np.all(acc[..., :3] >= 0) and not np.isnan(acc).any()
    # ray budget: <= 36 rays per pixel per frame at MAX_DEPTH 8, and at least the primary ray
    rays = st["rays_closest"] + st["rays_any"]
    assert W * H * N <= rays <= 36 * W * H * N
    # oracle on two horizontal bands (pixels depend only on rows within +-12 of themselves per frame; N frames -> N*12 rows): rows 500-520
    # cross the back wall and the tall box, rows 900-930 the glass crystal with its sphere light, the mirror-like box and the checkered floor — the expensive
    # and quirky branches (delta glass, refraction, the roughness-0.01 box's overflowing GGX term, textured albedo)
    del r2
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    for y0, y1 in ((500, 520), (900, 930)):
        halo = 12 * N + 2
        ro = os_.renderer(W, H, 8, True, 16)
        for f in range(N):
            cam = cams[f]
            a, b = y0 - halo, y1 + halo
            ro.render_phases(cam, 1, a - 14, b + 14); ro.render_phases(cam, 2, a - 12, b + 12)
            ro.render_phases(cam, 4, a - 2, b + 2); ro.render_phases(cam, 8, a, b); ro.end_frame()
            halo -= 12
        want = ro.read(7, (N - 1) % 2).view(np.float32)[y0:y1]
        assert np.abs(acc[y0:y1] - want).max() <= TOL and np.array_equal(acc[y0:y1], want), (y0, y1)
        mats = set(np.unique(pos[y0:y1, :, 3]).astype(int).tolist())
        if y0 == 900:
            assert {3, 4, 5} <= mats, mats       # the checkered floor (material 3), the roughness-0.01 metal box (4) and the glass crystal (5) are in the band


def test_strips_equal_whole_image(gpu):
    """T4: two strip renderers with the halo exchange (here through the host) reproduce the single-renderer image bit for bit."""
    frt = gpu
    from frt.dist import StripPlan, exchange_halos_host
    W, H, N = 192, 128, 5
    fs = frt.scenes.create_cornell_box()
    whole = frt.Renderer(fs, W, H)
    plans = [StripPlan(H, 2, k) for k in range(2)]
    strips = [frt.Renderer(fs, W, H, rows=(p.row_begin, p.row_end)) for p in plans]
    for f in range(N):
        cam = frt.CameraController().build_uniform(W / H, f, 2)
        whole.render(cam)
        _strip_frame(frt, strips, plans, cam, f)
    want = whole.read_accum(); wd = whole.read_display()
    for s, p in zip(strips, plans):
        got = s.read_accum()
        assert np.array_equal(got[p.row_begin:p.row_end], want[p.row_begin:p.row_end])
        assert np.array_equal(s.read_display()[p.row_begin:p.row_end], wd[p.row_begin:p.row_end])
    tot = sum(s.stats()["rays_closest"] + s.stats()["rays_any"] for s in strips)
    assert tot == whole.stats()["rays_closest"] + whole.stats()["rays_any"]      # halo rows are not double counted


def test_external_arena_and_stream(gpu):
    """Caller-owned device memory (torch tensor) and stream (torch's current stream) give the same image."""
    frt = gpu
    import torch
    W, H = 128, 96
    fs = frt.scenes.create_cornell_box()
    nbytes = frt.Renderer.arena_bytes(W, H)
    arena = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda:0")
    off = (-arena.data_ptr()) % 256
    r1 = frt.Renderer(fs, W, H, arena=arena.data_ptr() + off, arena_bytes=nbytes, stream=torch.cuda.current_stream().cuda_stream)
    r2 = frt.Renderer(fs, W, H)
    for f in range(3):
        cam = frt.CameraController().build_uniform(W / H, f, 2)
        r1.render(cam); r2.render(cam)
    torch.cuda.synchronize()
    assert np.array_equal(r1.read_accum(), r2.read_accum())
    p, bpp = r1.buffer_info(frt.BUF_ACCUM, 0)
    assert bpp == 16 and arena.data_ptr() + off <= p < arena.data_ptr() + off + nbytes
    # the arena tensor aliases the renderer's buffers: read the accumulation slot straight from torch
    o = p - arena.data_ptr()
    t = arena[o:o + W * H * 16].view(torch.float32).reshape(H, W, 4).cpu().numpy()
    assert np.array_equal(t, r1.read_buffer(frt.BUF_ACCUM, 0).view(np.float32))


def test_reset_and_clear(gpu):
    frt = gpu
    fs = frt.scenes.create_cornell_box()
    r = frt.Renderer(fs, 64, 64)
    cams = [frt.CameraController().build_uniform(1.0, f, 2) for f in range(3)]
    for c in cams:
        r.render(c)
    a = r.read_accum()
    r.reset()
    assert r.frame_count == 0                       # state.rs:152: only the counter restarts
    r.clear()
    for c in cams:
        r.render(c)
    assert np.array_equal(r.read_accum(), a) and r.stats()["frames"] == 3


def test_two_rank_strips_through_torch_distributed(gpu, tmp_path):
    """bench.py's N > 1 code path (StripPlan + ArenaRows + exchange_halos over torch.distributed) with two ranks sharing the one
    GPU of the test box; transport is gloo with host staging here, RCCL on the 8-GPU node. Image and ray totals must match."""
    from test_dist_gloo import run_ranks
    res = run_ranks("gpu", 2, tmp_path, ("--H", "96", "--W", "160", "--frames", "5"))
    assert res["ok"], res
    assert res["rays_all_ranks"] == res["oracle_rays"]
    # work-balanced (unequal) strips from the probe render give the same image
    res = run_ranks("gpu", 2, tmp_path, ("--H", "128", "--W", "160", "--frames", "4", "--balanced", "1"))
    assert res["ok"] and res["rays_all_ranks"] == res["oracle_rays"], res
    assert res["bounds"][1] != 64, res
    # moving camera: motion halo of 6 rows, two exchanges per frame
    res = run_ranks("gpu", 2, tmp_path, ("--H", "96", "--W", "160", "--frames", "5", "--moving", "6"))
    assert res["ok"] and res["rays_all_ranks"] == res["oracle_rays"], res
    # bench.py's configuration: 1080p, two-stream schedule (half-frame strips are cut; the next frame's G-buffer + T-trace run ahead)
    res = run_ranks("gpu", 2, tmp_path, ("--H", "1080", "--W", "1920", "--frames", "4", "--flags", "8"))
    assert res["ok"] and res["rays_all_ranks"] == res["oracle_rays"], res
    # the same with a MOVING camera (the "pre" rows — previous spatial reservoirs — and the accumulation rows must be complete when
    # they are read for the exchange, whatever runs on the renderer's other streams): motion halo of 8 rows, speculation dropped every frame
    res = run_ranks("gpu", 2, tmp_path, ("--H", "1080", "--W", "1920", "--frames", "4", "--flags", "8", "--moving", "8"))
    assert res["ok"] and res["rays_all_ranks"] == res["oracle_rays"], res


def test_moving_camera_on_gpu(gpu, orc):
    frt = gpu
    import _scenes
    fs = frt.scenes.create_cornell_box(); os_ = orc.cornell()
    os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    W, H = 192, 128
    cams = _scenes.moving_camera_uniforms(frt, W / H, 2, 6)
    r = frt.Renderer(fs, W, H); ro = os_.renderer(W, H, 8, True, 16)
    for f, cam in enumerate(cams):
        r.render(cam); ro.render(cam)
        compare_all(r.read_buffer, ro.read, f, "moving camera")
    assert np.abs(r.read_buffer(frt.BUF_GMOTION, 0).view(np.float32)).max() > 1e-3


def test_post_overlap_flag_gives_identical_frames(gpu, orc):
    """FRT_FLAG_OVERLAP_POST: post(f) on a second stream concurrently with G-buffer + temporal of frame f+1 — same pixels, also for
    strips with the halo exchange in between."""
    frt = gpu
    from frt.dist import StripPlan, exchange_halos_host
    W, H, N = 256, 160, 7
    fs = frt.scenes.create_cornell_box()
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    ro = os_.renderer(W, H, 8, True, 16)
    r = frt.Renderer(fs, W, H, flags=frt.FLAG_OVERLAP_POST)
    plans = [StripPlan(H, 2, k) for k in range(2)]
    strips = [frt.Renderer(fs, W, H, rows=(p.row_begin, p.row_end), flags=frt.FLAG_OVERLAP_POST) for p in plans]
    for f in range(N):
        cam = frt.CameraController().build_uniform(W / H, f, 2)
        ro.render(cam); r.render(cam)                      # no sync between frames: frame f+1's G-buffer + T-trace really run beside spatial(f)
        _strip_frame(frt, strips, plans, cam, f)
    last = (N - 1) % 2
    for b, idx in ((0, last), (1, last), (2, last), (4, 0), (4, 1), (5, 0), (6, 0), (7, last), (7, last ^ 1)):
        assert r.read_buffer(b, idx).tobytes() == ro.read(b, idx).tobytes(), (b, idx)
    assert r.read_buffer(frt.BUF_GMOTION, last).tobytes() == ro.read(3, 0).tobytes()
    want = ro.read(7, last)
    for s, p in zip(strips, plans):
        assert s.read_buffer(7, last)[p.row_begin:p.row_end].tobytes() == want[p.row_begin:p.row_end].tobytes()


@pytest.mark.parametrize("cuts", [[], [3], [2, 3, 5, 6], [1, 2, 3, 4], [5, 7], [1], [2, 6], [5, 3], [3, 3, 9], [70000, 2]])
def test_every_cut_configuration_is_bit_identical(gpu, orc, cuts):
    """The continuation-queue protocol (pixel kernel -> park after the roulette -> continue kernels, one counter per segment) must
    not depend on where or how often paths are cut: frt_render_opts.cut_depths ([] = never cut; lists that are not ascending or hold
    out-of-range depths are reduced to their ascending valid entries and must not change pixels either)."""
    frt = gpu
    W, H, depth = 160, 96, 8
    fs = frt.scenes.create_cornell_box()
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    r = frt.Renderer(fs, W, H, max_depth=depth, cuts=cuts)
    ro = os_.renderer(W, H, depth, True, 16)
    for f in range(4):
        cam = frt.CameraController().build_uniform(W / H, f, 2)
        r.render(cam); ro.render(cam)
        compare_all(r.read_buffer, ro.read, f, f"cuts {cuts}")
    st, so = r.stats(), ro.stats()["total"]
    assert (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"])


def test_moving_camera_strips_equal_whole_image(gpu):
    """SURVEY §8f-2 on the GPU: three strip renderers with a motion halo + the two exchanges reproduce the single-renderer frames of a
    moving camera bit for bit, with and without the side-stream schedule; without the halo the reads are detected (halo_overflow)."""
    frt = gpu
    import _scenes
    from frt.dist import StripPlan, exchange_halos_host, check_halo
    W, H, N, K = 192, 144, 6, 6
    fs = frt.scenes.create_cornell_box()
    cams = _scenes.moving_camera_uniforms(frt, W / H, 2, N)
    whole = frt.Renderer(fs, W, H)
    for c in cams: whole.render(c)
    want = whole.read_accum(); wd = whole.read_display()
    for flags, halo in ((0, K), (frt.FLAG_OVERLAP_POST, K), (0, 0)):
        plans = [StripPlan(H, 3, k, motion_halo=halo) for k in range(3)]
        strips = [frt.Renderer(fs, W, H, rows=(p.row_begin, p.row_end), motion_halo=halo, flags=flags) for p in plans]
        for f, cam in enumerate(cams):
            _strip_frame(frt, strips, plans, cam, f)
        same = all(np.array_equal(s.read_accum()[p.row_begin:p.row_end], want[p.row_begin:p.row_end]) and
                   np.array_equal(s.read_display()[p.row_begin:p.row_end], wd[p.row_begin:p.row_end]) for s, p in zip(strips, plans))
        if halo:
            assert same, (flags, halo)
            for s in strips: check_halo(s)
            tot = sum(s.stats()["rays_closest"] + s.stats()["rays_any"] for s in strips)
            assert tot == whole.stats()["rays_closest"] + whole.stats()["rays_any"]
        else:
            assert sum(s.stats()["halo_overflow"] for s in strips) > 0       # a static-camera plan under a moving camera is detected
            with pytest.raises(RuntimeError):
                for s in strips: check_halo(s)
    assert whole.stats()["halo_overflow"] == 0


def test_moving_camera_strips_with_the_counter_reset_every_frame(gpu):
    """The reference's host loop while the camera moves (state.rs:152: frame_count = 0 every such frame) over strips: the "pre" rows — the previous
    frame's spatial reservoirs, which T-merge reprojects into whatever frame_count says — are gated on the frames rendered since creation
    (StripPlan.transfers(..., serial)), not on frame_count."""
    frt = gpu
    import _scenes
    from frt.dist import StripPlan, check_halo
    W, H, N, K = 192, 144, 6, 6
    fs = frt.scenes.create_cornell_box()
    cams = _scenes.moving_camera_uniforms(frt, W / H, 2, N)
    whole = frt.Renderer(fs, W, H)
    plans = [StripPlan(H, 3, k, motion_halo=K) for k in range(3)]
    strips = [frt.Renderer(fs, W, H, rows=(p.row_begin, p.row_end), motion_halo=K, flags=frt.FLAG_PIPELINE) for p in plans]
    for f, cam in enumerate(cams):
        if f > 0:
            whole.reset()
            for s in strips: s.reset()
        cam.frame_count = 0
        whole.render(cam)
        _strip_frame(frt, strips, plans, cam, 0, serial=f)
        want, wd = whole.read_accum(), whole.read_display()
        for s, p in zip(strips, plans):
            assert np.array_equal(s.read_accum()[p.row_begin:p.row_end], want[p.row_begin:p.row_end]), (f, p.rank)
            assert np.array_equal(s.read_display()[p.row_begin:p.row_end], wd[p.row_begin:p.row_end]), (f, p.rank)
    for s in strips: check_halo(s)


@pytest.mark.parametrize("flags", [0, 8], ids=["plain", "side-stream"])
def test_reset_every_frame_like_a_moving_reference_camera(gpu, orc, flags):
    """state.rs:152 sets frame_count = 0 on every frame in which the camera moved, so the ping-pong slots stop alternating. The
    side-stream schedule must survive that (reset orders the main stream behind the in-flight tail)."""
    frt = gpu
    import _scenes
    W, H = 160, 96
    fs = frt.scenes.create_cornell_box()
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    r = frt.Renderer(fs, W, H, flags=flags)
    ro = os_.renderer(W, H, 8, True, 16)
    cams = _scenes.moving_camera_uniforms(frt, W / H, 2, 7)
    for f, cam in enumerate(cams):
        if f in (2, 3, 4, 6):          # "camera moved": restart the counter, as the reference does
            r.reset(); ro.restart_counter()
        cam.frame_count = r.frame_count
        r.render(cam); ro.render(cam)
        compare_all(r.read_buffer, ro.read, f, f"reset sequence, flags {flags}")
        assert r.frame_count == ro.frame_count


def test_config2_4k_eight_strips_equal_whole(gpu):
    """BASELINE.json configs[2]: 3840x2160 cut into 8 strips of 270 rows (here on one GPU, rows exchanged through the host). Strips
    reproduce the whole-frame renderer bit for bit; the strips are cut once, the whole frame twice + two-stream schedule."""
    frt = gpu
    from frt.dist import StripPlan, exchange_halos_host
    W, H, N = 3840, 2160, 3
    fs = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(N)]
    whole = frt.Renderer(fs, W, H, flags=frt.FLAG_OVERLAP_POST)
    for c in cams: whole.render(c)
    want = whole.read_accum(); st = whole.stats(); total = st["rays_closest"] + st["rays_any"]
    # footprint at 4K: per-pixel arena + continuation queues (two cuts: a 22-word and a 30-word buffer per cut)
    footprint = frt.Renderer.arena_bytes(W, H) + st["queue_bytes"]
    assert st["queue_bytes"] >= st["queue_capacity"] * (22 * 0.6 + 30) * 4
    assert st["queue_overflow"] == 0 and footprint < 2.5e9, (footprint, st)
    del whole
    plans = [StripPlan(H, 8, k) for k in range(8)]
    assert all(p.row_end - p.row_begin == 270 for p in plans)
    strips = [frt.Renderer(fs, W, H, rows=(p.row_begin, p.row_end)) for p in plans]
    for f, cam in enumerate(cams):
        _strip_frame(frt, strips, plans, cam, f)
    for s, p in zip(strips, plans):
        assert np.array_equal(s.read_rows(frt.BUF_ACCUM, (N - 1) % 2, p.row_begin, p.row_end).view(np.float32).reshape(-1, W, 4), want[p.row_begin:p.row_end])
    assert sum(s.stats()["rays_closest"] + s.stats()["rays_any"] for s in strips) == total


@pytest.mark.parametrize("three_sets", [False, True], ids=["two-sets", "three-sets"])
def test_pipeline_speculation_is_adopted_for_a_static_camera_and_dropped_for_a_moving_one(gpu, orc, three_sets):
    """FRT_FLAG_PIPELINE: G-buffer + T-trace of frame f+1 run ahead under a speculated camera. Static camera: adopted from the third
    frame on (one frame to see the camera, one to see that it did not move), every buffer and the ray counts identical to the oracle.
    Moving camera: never speculated (nothing to drop). A camera that stops / starts moving: wrong guesses are dropped, same pixels."""
    frt = gpu
    import _scenes
    # three sets: what strip renderers get — a third G-buffer set, the ahead stream ordered behind the PREVIOUS frame's T-merge; here on a
    # whole frame (FRT_FLAG_THIRD_GSET), so that frames enqueued back to back can be compared with the oracle
    pipe = frt.FLAG_PIPELINE | (frt.FLAG_THIRD_GSET if three_sets else 0)
    W, H, N = 160, 96, 7
    fs = frt.scenes.create_cornell_box()
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    r = frt.Renderer(fs, W, H, flags=pipe); ro = os_.renderer(W, H, 8, True, 16)
    for f in range(N):
        cam = frt.CameraController().build_uniform(W / H, f, 2)
        r.render(cam); ro.render(cam)
    compare_all(r.read_buffer, ro.read, N - 1, "pipeline, static camera")
    st, so = r.stats(), ro.stats()["total"]
    assert (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"])
    assert st["speculated_frames"] == N - 2 and st["discarded_speculations"] == 0, st
    # static -> moving -> static: the guess made during the last static frame is wrong and must be dropped
    r = frt.Renderer(fs, W, H, flags=pipe); ro = os_.renderer(W, H, 8, True, 16)
    moving = _scenes.moving_camera_uniforms(frt, W / H, 2, 12)
    seq = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(4)] + moving[4:8]
    still = moving[7]
    for f in range(8, 12):
        c = frt.CameraUniform.from_buffer_copy(bytes(still)); c.frame_count = f; c.prev_view_proj[:] = list(still.view_proj); seq.append(c)
    for f, cam in enumerate(seq):
        r.render(cam); ro.render(cam)
        if f in (3, 4, 5, 8, 11):
            compare_all(r.read_buffer, ro.read, f, "pipeline, camera starts and stops moving")
    st, so = r.stats(), ro.stats()["total"]
    assert (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"])
    assert st["discarded_speculations"] in (1, 2) and st["speculated_frames"] >= 3, st     # (one guess per frame run ahead)


def test_three_gbuffer_sets_at_full_size_equal_two(gpu):
    """1920x1080, 16 frames enqueued back to back: a renderer with a third G-buffer set (ahead stream a whole frame ahead of the chain,
    what strips run) against the default two sets: accumulation, both reservoir buffers, G-buffer and ray counts identical."""
    frt = gpu
    W, H, N = 1920, 1080, 16
    fs = frt.scenes.create_cornell_box()
    cams = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(N)]
    a = frt.Renderer(fs, W, H, flags=frt.FLAG_PIPELINE)
    b = frt.Renderer(fs, W, H, flags=frt.FLAG_PIPELINE | frt.FLAG_THIRD_GSET)
    for c in cams:
        a.render(c); b.render(c)
    last = (N - 1) % 2
    for buf, idx in ((7, last), (4, 0), (4, 1), (0, last), (1, last), (2, last), (5, 0)):
        assert a.read_buffer(buf, idx).tobytes() == b.read_buffer(buf, idx).tobytes(), (buf, idx)
    sa, sb = a.stats(), b.stats()
    assert (sa["rays_closest"], sa["rays_any"]) == (sb["rays_closest"], sb["rays_any"]) and sb["speculated_frames"] == N - 2


def test_strips_with_a_camera_that_starts_and_stops(gpu, orc):
    """Strip renderers under the pipeline own a third G-buffer set, so the ahead stream is ordered behind the PREVIOUS frame's T-merge and
    T-trace(f+1) may start before T-merge(f). Static -> moving -> static camera over three strips (middle strip: two edge launches), no
    host synchronisation beyond the exchanges: adopted and dropped speculations, the hand-over between a T-trace on the main stream and the
    next one on the ahead stream, pending ray counts of a dropped frame. Accumulation rows and the summed ray counts equal the oracle."""
    frt = gpu
    import _scenes
    from frt.dist import StripPlan
    W, H, K = 192, 144, 8
    fs = frt.scenes.create_cornell_box()
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    ro = os_.renderer(W, H, 8, True, 16)
    plans = [StripPlan(H, 3, k, motion_halo=K) for k in range(3)]
    strips = [frt.Renderer(fs, W, H, rows=(p.row_begin, p.row_end), flags=frt.FLAG_PIPELINE, motion_halo=K) for p in plans]
    moving = _scenes.moving_camera_uniforms(frt, W / H, 2, 12)
    seq = [frt.CameraController().build_uniform(W / H, f, 2) for f in range(5)] + moving[5:8]
    still = moving[7]
    for f in range(8, 13):
        c = frt.CameraUniform.from_buffer_copy(bytes(still)); c.frame_count = f; c.prev_view_proj[:] = list(still.view_proj); seq.append(c)
    for f, cam in enumerate(seq):
        ro.render(cam)
        _strip_frame(frt, strips, plans, cam, f)
        if f in (2, 4, 5, 7, 8, 10, 12):
            want = ro.read(7, f % 2)
            for s, p in zip(strips, plans):
                assert s.read_buffer(7, f % 2)[p.row_begin:p.row_end].tobytes() == want[p.row_begin:p.row_end].tobytes(), (f, p.rank)
    sts = [s.stats() for s in strips]
    so = ro.stats()["total"]
    assert sum(st["rays_closest"] for st in sts) == so["closest"] and sum(st["rays_any"] for st in sts) == so["any"]
    assert all(st["speculated_frames"] >= 4 and st["discarded_speculations"] >= 1 and st["halo_overflow"] == 0 for st in sts), sts


def test_queue_overflow_finishes_paths_in_place(gpu, orc):
    """A continuation queue far too small for the paths that reach the cut: the overflowing lanes keep their paths and finish them in
    place — same pixels, same ray counts — and the overflow is counted; a renderer with the default capacity grows it after an overflow."""
    frt = gpu
    W, H = 160, 96
    fs = frt.scenes.create_cornell_box()
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    for flags in (0, frt.FLAG_PIPELINE):
        ro = os_.renderer(W, H, 8, True, 16)
        r = frt.Renderer(fs, W, H, flags=flags, queue_capacity=100, cuts=[2, 4])       # (a small image would be cut once, at 3, by default)
        for f in range(4):
            cam = frt.CameraController().build_uniform(W / H, f, 2)
            r.render(cam); ro.render(cam)
            compare_all(r.read_buffer, ro.read, f, f"queue capacity 100, flags {flags}")
        st, so = r.stats(), ro.stats()["total"]
        assert (st["rays_closest"], st["rays_any"]) == (so["closest"], so["any"])
        assert st["queue_overflow"] > 1000 and st["queue_capacity"] == 100, st


def test_jittered_frames_on_gpu(gpu, orc):
    """SURVEY §8f-2 jitter: frt_renderer_render_jittered with non-zero Halton jitter (projection shear + bilinear post taps) against
    the oracle, every buffer of every frame."""
    frt = gpu
    W, H = 160, 96
    fs = frt.scenes.create_cornell_box()
    os_ = orc.cornell(); os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    for flags, scale in ((0, 1.0), (frt.FLAG_PIPELINE, 23.0)):
        r = frt.Renderer(fs, W, H, flags=flags); ro = os_.renderer(W, H, 8, True, 16)
        ctl = frt.CameraController()
        for f in range(4):
            jit = ctl.get_halton_jitter(f, W, H, scale)
            cam = ctl.build_uniform(W / H, f, 2, jit); ctl.commit_frame()
            ro.set_jitter(jit); ro.render(cam)
            r.render(cam, jitter=jit)
            compare_all(r.read_buffer, ro.read, f, f"jitter x{scale}, flags {flags}")
    strip = frt.Renderer(fs, W, H, rows=(0, 48))
    strip.set_jitter((0.01, 0.0))
    with pytest.raises(frt.FrtError):
        strip.render(frt.CameraController().build_uniform(W / H, 0, 2))


def test_rccl_loopback_rehearsal_on_one_gpu(gpu):
    """The real "nccl" (= RCCL) backend with a world of one rank: bench.py's N > 1 frame loop with every halo transfer a device-to-device
    send-to-self of arena rows (tests/_nccl_selftest.py). An 8-GPU node is not available to the tests; this is as much of the RCCL code
    path as one GPU can execute, so that the first multi-GPU run is not its first execution."""
    import json, subprocess, sys
    from test_dist_gloo import _free_port
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_nccl_selftest.py"), str(_free_port())],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["nccl"], res
    # a strip too thin for interior rows: its edge launches are ordered behind the transfer on the edge stream
    assert res["thin"]["rows_arrive"] and res["thin"]["async_equals_stepwise"], res
    # ... and with RCCL called directly on the renderer's streams (frt.rccl: one grouped launch in the edge stream, no torch stream, no events)
    for k in ("interior", "thin"):
        assert res[k]["direct_rows_arrive"] and res[k]["direct_equals_stepwise"], res


def test_bench_self_launch_rehearsal_on_one_gpu(gpu):
    """`python bench.py --gpus 2` from a bare shell (no RANK / WORLD_SIZE): bench.py starts its two ranks itself; both share this box's one
    GPU over gloo (FRT_BENCH_ONE_GPU=1). One JSON line that names both ranks, their rows and devices."""
    import json, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FRT_BENCH_ONE_GPU"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--no-4k"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and p.stdout.strip().splitlines() == lines      # the JSON line and nothing else (gloo's chatter goes to stderr)
    res = json.loads(lines[0])
    cfg = res["config"]
    assert res["n_gpus"] == 2 and cfg["process_group_world_size"] == 2 and cfg["allreduce_of_ones"] == 2 and cfg["self_launched"]
    assert [r["rank"] for r in cfg["ranks"]] == [0, 1] and cfg["ranks"][0]["rows"][1] == cfg["ranks"][1]["rows"][0]
    assert res["value"] > 0 and res["scaling"] == "strong"


def test_bench_line_contract_at_one_gpu(gpu):
    """The line the round driver parses: `python bench.py --gpus 1 --steps K --warmup W` prints exactly one line on stdout, a JSON object with the
    contract's keys, a roofline block (achieved / peak / frac / traffic from the committed counters when they match the kernel sources) and a CPU
    baseline measured in the same run."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--cpu-frames", "1", "--no-4k"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = p.stdout.strip().splitlines()
    assert len(out) == 1, p.stdout[:2000]
    res = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["steps"] == 8 and res["warmup"] == 2 and res["unit"] == "Mrays/s" and res["dtype"] == "f32" and res["vs_baseline"] is None
    assert res["value"] > 1000 and 0.5 < res["ms_per_step"] < 20 and res["higher_is_better"] is True and "workload" in res["config"]
    assert abs(res["value"] - res["config"]["rays_per_frame"] / res["ms_per_step"] / 1e3) < 1e-6 * res["value"]      # Mrays/s = rays per frame / ms per frame
    roof = res["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "logical_GBs", "stages"):
        assert k in roof, k
    # the counters are numbers in the record (a stale profile blanked them), they come from a profile of THIS device code, and
    # frac is what the line says it is: VALU wave-instructions x 64 lanes / frame time / peak
    for k in ("achieved", "frac", "traffic"):
        assert isinstance(roof[k], float) and roof[k] > 0, (k, roof[k])
    assert roof["pmc_source"]["stale"] is False, roof["pmc_source"]
    assert roof["hbm_actual"]["frac"] > 0 and roof["valu_issue"]["frac_at_measured_clock"] > 0
    want = roof["valu_issue"]["wave_insts_per_frame"] * 64.0 / (res["ms_per_step"] * 1e-3) / 1e12 / roof["peak"]
    assert abs(roof["frac"] - want) <= 1e-6 * want
    cpu = res["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu and cpu["unit"] == "Mrays/s"
