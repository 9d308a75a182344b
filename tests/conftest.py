import os
import sys
import subprocess
import ctypes as C
import numpy as np
import pytest

try:                      # load torch's HIP runtime before libfrt.so so that both share one runtime in this process
    import torch  # noqa: F401
except Exception:         # torch is only needed by the multi-rank tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure(path, make_dir):
    # build artefacts are normally produced by __graft_entry__.build(); build on demand for a bare checkout with a toolchain
    if not os.path.exists(path):
        subprocess.run(["make", "-C", make_dir], check=True, stdout=subprocess.DEVNULL)
    return path


@pytest.fixture(scope="session")
def orc():
    from _oracle import Oracle
    return Oracle(_ensure(os.path.join(ROOT, "oracle", "_build", "liborc.so"), os.path.join(ROOT, "oracle")))


@pytest.fixture(scope="session")
def frt():
    _ensure(os.path.join(ROOT, "fast-raytracing-wgpu_amd", "lib", "libfrt.so"), os.path.join(ROOT, "fast-raytracing-wgpu_amd"))
    import frt as _frt
    _frt.lib()
    return _frt


@pytest.fixture(scope="session")
def hostcheck(frt):
    from _hostcheck import HostCheck
    return HostCheck(_ensure(os.path.join(ROOT, "tests", "hostcheck", "_build", "libfrt_hostcheck.so"), os.path.join(ROOT, "tests", "hostcheck")))
