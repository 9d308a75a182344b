"""The drop-in boundary from plain C: tests/cabi/frt_cabi_example.c includes include/frt.h, links libfrt.so and nothing else, and drives scene,
renderer and the multi-device renderer the way the reference's State does. Compiles on any machine (the header is valid C99); runs on the GPU box."""
import json
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "frt_cabi_example.c")
OUT = os.path.join(ROOT, "tests", "cabi", "_build", "frt_cabi_example")
LIBDIR = os.path.join(ROOT, "fast-raytracing-wgpu_amd", "lib")


def _build(frt):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-L", LIBDIR, "-lfrt",
                    f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", OUT], check=True)


def test_header_is_valid_c99_and_links(frt):
    _build(frt)
    assert os.path.exists(OUT)


@pytest.mark.gpu
def test_c_host_renders_and_multi_equals_single(frt):
    if frt.lib().frt_device_count() < 1:
        pytest.fail("no HIP device")
    _build(frt)
    p = subprocess.run([OUT], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["multi_equals_single"] and res["frames"] == 4 and res["tris"] == 1320 and res["display_sum"] > 0
