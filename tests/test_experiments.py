"""The kernel designs that were measured and not kept (csrc/experiments/, HISTORY.md section 6; round 4: profiles/r4_experiments) live in lib/libfrt_exp.so, outside the
product library. They stay correct: each family renders the Cornell Box bit for bit like the oracle. The PRODUCT library ignores their
environment knobs and rejects FRT_FLAG_COMPACTION."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fast-raytracing-wgpu_amd")
EXP = os.path.join(PKG, "lib", "libfrt_exp.so")


def _run(env_extra, flags=0, lib=EXP, args=()):
    env = dict(os.environ, FRT_LIB=lib, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_exp_worker.py"), str(flags), *[str(a) for a in args]], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("name,env,flags", [
    ("compaction", {}, 2), ("resident", {"FRT_RESIDENT": "1"}, 0), ("resident+sweep", {"FRT_RESIDENT": "1", "FRT_TILE_ORDER": "1"}, 8),
    ("refill", {"FRT_REFILL": "1"}, 0), ("stream", {"FRT_STREAM": "1", "FRT_CUTS": "1"}, 0), ("wavefront", {"FRT_WAVEFRONT": "1"}, 0),
    ("cuts 2,5", {"FRT_CUTS": "2,5"}, 8), ("leaf 4", {"FRT_BVH_LEAF": "4"}, 0)])
def test_experimental_kernels_match_the_oracle(frt, name, env, flags):
    if frt.lib().frt_device_count() < 1:
        pytest.fail("no HIP device")
    if not os.path.exists(EXP):
        subprocess.run(["make", "-C", PKG, "experiments"], check=True, stdout=subprocess.DEVNULL)
    res = _run(env, flags)
    assert res["ok"] and res["lib"].endswith("libfrt_exp.so"), (name, res)


# Round 4's two walks (profiles/r4_experiments/wide8.md, collective_walks.md): FRT_FLAG_WALK_WIDE 32 / _HBM 64 — the 8-wide tree with 16-bit grid boxes,
# hit-mask stack words, octant order, in LDS or from HBM; FRT_FLAG_WG_TRACE 128 — a workgroup's rays re-dealt to dense, direction-sorted waves before every
# walk, paths run in workgroup-uniform loops over the pulled-apart forms. Every buffer of every frame and the exact ray counts against the oracle
# (blob5k: the oracle's brute-force loop, nothing of the product's); 8 = the two-stream schedule.
@pytest.mark.gpu
@pytest.mark.parametrize("name,flags,args", [
    ("wide LDS cornell", 8 | 32, ("cornell", 128, 128, 8, 4)), ("wide HBM cornell d16", 8 | 64, ("cornell", 96, 64, 16, 3)), ("wide ragged", 8 | 32, ("cornell", 37, 19, 8, 3)),
    ("wide restir", 8 | 32, ("restir", 160, 96, 8, 3)), ("wide blob5k brute force", 8 | 32, ("blob5k", 64, 48, 8, 2)),
    ("wide bumpy82k", 32, ("bumpy82k", 160, 90, 8, 3)), ("wide colonnade250k", 32, ("colonnade250k", 160, 90, 16, 2)),
    ("collective cornell", 8 | 128, ("cornell", 128, 128, 8, 4)), ("collective ragged", 8 | 128, ("cornell", 37, 19, 8, 3)), ("collective d16", 8 | 128, ("cornell", 96, 64, 16, 3)),
    ("collective restir", 8 | 128, ("restir", 160, 96, 8, 3)), ("collective blob5k brute force", 8 | 128, ("blob5k", 64, 48, 8, 2)),
    ("collective bumpy82k (voting walk)", 128, ("bumpy82k", 160, 90, 8, 3))])
def test_round4_walks_match_the_oracle(frt, name, flags, args):
    if frt.lib().frt_device_count() < 1:
        pytest.fail("no HIP device")
    res = _run({}, flags, args=args)
    assert res["ok"] and res["lib"].endswith("libfrt_exp.so"), (name, res)
    if flags & 96:
        assert 0 < res["tree"]["wide8_nodes"] <= 65536 and res["tree"]["wide8_stack_need"] <= 8, res["tree"]


@pytest.mark.gpu
def test_round4_walks_equal_the_quad_walk_at_full_size(frt):
    """1920x1080, MAX_DEPTH 8, 4 frames: the quad walk, the 8-wide walk (tree in LDS / from HBM) and the collective walks give the same image bit for bit
    and the same ray counts — hits do not depend on the tree or on the lane that walks a ray (frt_trace.hpp)."""
    if frt.lib().frt_device_count() < 1:
        pytest.fail("no HIP device")
    env = dict(os.environ, FRT_LIB=EXP)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_exp_worker.py"), "equal", "0,32,64,128"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["rays"][0] > 0, res


def test_product_library_has_no_experiment_code(frt):
    """No experiment kernel is in libfrt.so's code object, no FRT_* knob string is in the binary, FRT_FLAG_COMPACTION is refused."""
    blob = open(os.path.join(PKG, "lib", "libfrt.so"), "rb").read()
    for name in (b"compact_kernel", b"bounce_kernel", b"stream_kernel", b"wf_trace_kernel", b"wf_shade_kernel", b"resident_pixel_kernel", b"resident_continue_kernel",
                 b"pixel_kernel_wg", b"continue_kernel_wg", b"pixel_kernelILi1ELi2", b"pixel_kernelILi1ELi3", b"gbuffer_kernelILi2"):
        assert name not in blob, name
    for knob in (b"FRT_CUTS", b"FRT_RESIDENT", b"FRT_REFILL", b"FRT_STREAM", b"FRT_WAVEFRONT", b"FRT_TILE_ORDER", b"FRT_QUEUE_CAP", b"FRT_SPEC_DEPTH",
                 b"FRT_AHEAD_PRIO", b"FRT_BVH_LEAF", b"FRT_FORCE_EXTRAS", b"FRT_NO_EXTRAS", b"FRT_CONT_GRID", b"FRT_WG_PARK", b"FRT_DEBUG_QUEUES"):
        assert knob not in blob, knob
    if frt.lib().frt_device_count() > 0:
        for fl in (frt.FLAG_COMPACTION, frt.FLAG_WALK_WIDE, frt.FLAG_WALK_WIDE_HBM, frt.FLAG_WG_TRACE):
            with pytest.raises(frt.FrtError, match="experiment"):
                frt.Renderer(frt.scenes.create_cornell_box(), 32, 32, flags=fl)
