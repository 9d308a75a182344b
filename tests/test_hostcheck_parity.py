"""T3 (CPU leg): the product's stage functions, instantiated on the host by tests/hostcheck, against the oracle — every
buffer of every frame, bit for bit. The -m gpu tests repeat this through the real kernels."""
import numpy as np
import pytest

NAMES = {0: "gpos", 1: "gnormal", 2: "galbedo", 3: "gmotion", 4: "reservoir", 5: "raw", 6: "display", 7: "accum"}


def _bits_equal(a, b):
    return a.tobytes() == b.tobytes()


def compare_all(got_reader, want_reader, frame, ctx=""):
    for b in range(8):
        for idx in ((0, 1) if b in (0, 1, 2, 4, 7) else (0,)):
            g, w = got_reader(b, idx), want_reader(b, idx)
            if not _bits_equal(g, w):
                d = np.argwhere((g.view(np.uint32) != w.view(np.uint32)).any(axis=2))
                raise AssertionError(f"{ctx} frame {frame} {NAMES[b]}[{idx}]: {len(d)} pixels differ, first at {tuple(d[0])}")


@pytest.mark.parametrize("sm", [0, 1, 2, 3, 5, 1001, 1003, 1008, 2001, 2003],
                         ids=["straight", "state_machine", "cut2", "cut3", "cut5", "split_cut1", "split_cut3", "split_uncut", "stream_cut1", "stream_cut3"])
@pytest.mark.parametrize("which,size,depth,frames,bvh", [("cornell", 64, 8, 4, True), ("cornell", 128, 1, 2, False),
                                                         ("cornell", 48, 16, 2, True), ("restir", 48, 8, 3, True)])
def test_stage_functions_match_oracle(frt, orc, hostcheck, which, size, depth, frames, bvh, sm):
    fs = frt.scenes.create_cornell_box() if which == "cornell" else frt.scenes.create_restir_scene()
    os_ = orc.cornell() if which == "cornell" else orc.restir_scene()
    os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    W, H = size, size * 3 // 4
    ro = os_.renderer(W, H, depth, bvh, 8)       # bvh=False: BASELINE.json configs[0] (scalar loop over all triangles)
    rh = hostcheck.renderer(fs, W, H, depth, 8, state_machine=sm)   # frt_mono.hpp (default kernels) or frt_path.hpp (compacting kernels)
    for f in range(frames):
        cam = frt.CameraController().build_uniform(W / H, f, fs.num_lights)
        ro.render(cam); rh.render(cam)
        compare_all(rh.read, ro.read, f, which)
    st = ro.stats()["total"]
    assert rh.rays() == (st["closest"], st["any"])
    assert st["closest"] + st["any"] <= (4 + 2 * (2 * depth - 1)) * W * H * frames    # SURVEY §8(a) ray budget: 36 at MAX_DEPTH 8


def test_moving_camera_matches_oracle(frt, orc, hostcheck):
    """Moving camera: motion vectors != 0, temporal reprojection lands on other pixels, post takes the TAA clamp branch
    (gbuffer.wgsl:230-242, restir.wgsl:846-900, post.wgsl:187-266). Same CameraUniform bytes to both sides."""
    import _scenes
    fs = frt.scenes.create_cornell_box(); os_ = orc.cornell()
    os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    W, H = 96, 64
    cams = _scenes.moving_camera_uniforms(frt, W / H, 2, 5)
    ro = os_.renderer(W, H, 8, True, 8); rh = hostcheck.renderer(fs, W, H, 8, 8)
    for f, cam in enumerate(cams):
        ro.render(cam); rh.render(cam)
        compare_all(rh.read, ro.read, f, "moving camera")
    mot = ro.read(3, 0).view(np.float32)
    assert np.abs(mot).max() > 1e-3                       # the motion path really is exercised
    res_t = ro.read(4, 0).view(np.uint32)
    assert (res_t[..., 2] > 1).any()                      # and temporal reuse still merges some reprojected reservoirs


@pytest.mark.parametrize("scale", [1.0, 23.0], ids=["halton", "beyond_a_pixel"])
def test_jittered_frames_match_oracle(frt, orc, hostcheck, scale):
    """SURVEY §8f-2 jitter plumbing: Halton jitter (camera.rs:182-205 with the literal 0 of :202-203 replaced by `scale`) shears the
    projection (camera.rs:224-228) and reaches post as PostParams.jitter (renderer.rs:361-379), where the radiance / albedo taps
    become bilinear samples at uv + unjitter_offset (post.wgsl:72-78, :97-109, :152-158). scale 23: offsets beyond a pixel, so the
    Repeat addressing at the image border is exercised too."""
    fs = frt.scenes.create_cornell_box(); os_ = orc.cornell()
    os_.set_bvh(fs.get("bvh2_nodes"), fs.get("bvh2_tri_index"))
    W, H = 80, 56
    ro = os_.renderer(W, H, 8, True, 8); rh = hostcheck.renderer(fs, W, H, 8, 8)
    ctl = frt.CameraController()
    plain = None
    for f in range(4):
        jit = ctl.get_halton_jitter(f, W, H, scale)
        assert jit == orc.halton_jitter(f, W, H, scale) and jit != (0.0, 0.0)
        cam = ctl.build_uniform(W / H, f, 2, jit); ctl.commit_frame()
        ro.set_jitter(jit); rh.set_jitter(jit)
        ro.render(cam); rh.render(cam)
        compare_all(rh.read, ro.read, f, f"jitter x{scale}")
    # and it is not a no-op: the same frames without PostParams.jitter accumulate to a different image
    ro2 = os_.renderer(W, H, 8, True, 8); ctl2 = frt.CameraController()
    for f in range(4):
        cam = ctl2.build_uniform(W / H, f, 2, ctl2.get_halton_jitter(f, W, H, scale)); ctl2.commit_frame()
        ro2.render(cam)
    assert ro2.read(7, 1).tobytes() != ro.read(7, 1).tobytes()
    assert ctl.get_halton_jitter(3, W, H) == (0.0, 0.0) or ctl.get_halton_jitter(3, W, H) == (-0.0, 0.0)     # the shipped reference: x 0
