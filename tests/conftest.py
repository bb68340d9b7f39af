import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # test modules import kubernetes_autoscaler_amd (which refuses to load without libcasim.so) while they are collected:
    # a fresh checkout has no built artefacts (*.so is git-ignored), so build before collection.  One process at a time
    # (pytest-xdist configures every worker).
    import fcntl
    with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            _build_all()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _make(path, target=None):
    cmd = ["make", "-s", "-C", path] + ([target] if target else [])
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def _build_all():
    """Builds the CPU oracle and the wave emulator (both test infrastructure).  libcasim.so itself is
    built by __graft_entry__.build(); if hipcc is around and the .so is missing, build it too."""
    _make(os.path.join(ROOT, "oracle"))
    _make(os.path.join(ROOT, "tests", "emu"))
    lib = os.path.join(ROOT, "kubernetes_autoscaler_amd", "libcasim.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        _make(os.path.join(ROOT, "kubernetes_autoscaler_amd", "csrc"))
    native = os.path.join(ROOT, "tools", "casim_native")
    if not os.path.exists(native) and os.path.exists(lib):
        _make(os.path.join(ROOT, "tools"))
