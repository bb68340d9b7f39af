import os
import subprocess
import sys

import pytest

# the host's setting for a streamed batch's internal streams (one hardware queue each); libcasim does not edit the environment by itself
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_tier_in_parallel(config):
    """`python -m pytest tests -m "not gpu"` (the CPU tier: ~9 000 oracle / emulator / host-logic cases, 25 minutes on one core) restarts
    itself under pytest-xdist with up to 8 workers when nobody chose a worker count: ~3.5 minutes.  Only that invocation: the GPU tier
    must stay one process (its tests time kernels and probe hardware queues).  CASIM_PYTEST_SERIAL=1 keeps the plain run."""
    if os.environ.get("CASIM_PYTEST_SERIAL") or os.environ.get("_CASIM_PYTEST_REEXEC") or hasattr(config, "workerinput"):
        return
    opt = config.option
    if getattr(opt, "numprocesses", None) or (getattr(opt, "markexpr", "") or "").strip() != "not gpu":
        return
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False) or len(config.invocation_params.args) == 0:
        return
    if not config.pluginmanager.hasplugin("xdist"):   # (not installed, or switched off with -p no:xdist)
        return
    n = min(8, os.cpu_count() or 1)
    if n < 2:
        return
    capman = config.pluginmanager.getplugin("capturemanager")
    if capman is not None:
        capman.stop_global_capturing()   # (fd-level capture has replaced stdout / stderr: give them back before the new image starts)
    os.environ["_CASIM_PYTEST_REEXEC"] = "1"
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, [sys.executable, "-m", "pytest", *[str(a) for a in config.invocation_params.args], "-n", str(n)])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _cpu_tier_in_parallel(config)
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
