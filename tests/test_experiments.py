"""The kernel designs that were measured and not kept (csrc/experiments/) live in lib/libfrt_exp.so, outside the
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


def _run(env_extra, flags=0, lib=EXP):
    env = dict(os.environ, FRT_LIB=lib, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_exp_worker.py"), str(flags)], capture_output=True, text=True, timeout=600, env=env)
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


def test_product_library_has_no_experiment_code(frt):
    """No experiment kernel is in libfrt.so's code object, no FRT_* knob string is in the binary, FRT_FLAG_COMPACTION is refused."""
    blob = open(os.path.join(PKG, "lib", "libfrt.so"), "rb").read()
    for name in (b"compact_kernel", b"bounce_kernel", b"stream_kernel", b"wf_trace_kernel", b"wf_shade_kernel", b"resident_pixel_kernel", b"resident_continue_kernel"):
        assert name not in blob, name
    for knob in (b"FRT_CUTS", b"FRT_RESIDENT", b"FRT_REFILL", b"FRT_STREAM", b"FRT_WAVEFRONT", b"FRT_TILE_ORDER", b"FRT_QUEUE_CAP", b"FRT_SPEC_DEPTH",
                 b"FRT_AHEAD_PRIO", b"FRT_BVH_LEAF", b"FRT_FORCE_EXTRAS", b"FRT_NO_EXTRAS", b"FRT_CONT_GRID", b"FRT_WG_PARK", b"FRT_DEBUG_QUEUES"):
        assert knob not in blob, knob
    if frt.lib().frt_device_count() > 0:
        with pytest.raises(frt.FrtError, match="experiment"):
            frt.Renderer(frt.scenes.create_cornell_box(), 32, 32, flags=frt.FLAG_COMPACTION)
