"""The library's process-wide host pool (csrc/casim_pipeline.h: HostPool; round 6): the parts of a streamed enter -> return call are tasks of it, and
every parallel host loop (staging copies, the request passes of ProblemT::init) cuts its work over it FROM INSIDE such a task.  Tasks are handed out
in index order and the caller works along, so a task may wait for the task in front of it (the parts' upload turn, the list bases of the fetch) and
nested use cannot deadlock — whatever the number of workers.  The self-test lives in the emulator's library (same header, host code only)."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from harness import emu_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nested_loops_and_turn_order_from_several_callers():
    L = emu_lib()
    L.emu_pool_selftest.restype = C.c_int32
    L.emu_pool_selftest.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    assert L.emu_pool_selftest(20, 3, 8) == 0
    assert L.emu_pool_workers() >= 0


@pytest.mark.parametrize("workers", ["0", "1", "2", "32"])
def test_any_number_of_workers(workers):
    """CASIM_POOL_THREADS is read when the pool is created (first use in a process): one process per setting.  0 = no pool at all (every run() is the
    caller's own loop, in index order: the turn order still holds), 1 = fewer workers than waiting tasks"""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from harness import emu_lib; L = emu_lib(); "
            "L.emu_pool_selftest.argtypes = [C.c_int32] * 3; r = L.emu_pool_selftest(10, 2, 6); w = L.emu_pool_workers(); print(r, w); sys.exit(1 if r else 0)") % (
        ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, CASIM_POOL_THREADS=workers)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    r, w = out.stdout.split()
    assert int(r) == 0 and int(w) <= max(int(workers), 0)
