import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(path, target=None):
    cmd = ["make", "-s", "-C", path] + ([target] if target else [])
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.fixture(scope="session", autouse=True)
def _build_test_infrastructure():
    """Builds the CPU oracle and the wave emulator (both test infrastructure).  libcasim.so itself is
    built by __graft_entry__.build(); if hipcc is around and the .so is missing, build it too."""
    _make(os.path.join(ROOT, "oracle"))
    _make(os.path.join(ROOT, "tests", "emu"))
    lib = os.path.join(ROOT, "kubernetes_autoscaler_amd", "libcasim.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        _make(os.path.join(ROOT, "kubernetes_autoscaler_amd", "csrc"))
    yield
