"""Runs of adjacent identical singleton PEGs merged into one table row on the host (csrc/casim_pipeline.h, SingletonRuns; the
reference's BenchmarkRunOnceScaleUp is 10 000 of them): results — PEG order, placed per PEG, nodes, limiter grants, lastIndex —
equal the oracle, which estimates every singleton on its own, and equal the unmerged run.  CPU: product kernels under the wave
emulator, both node stores, device-derived and explicit schedulable lists."""
import ctypes as C

import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from harness import GroupSpec, Scenario, alloc_results, assert_matches_oracle, emu_lib, encode, finish_results, run_emu, run_oracle


def _scenario(w, device_csr):
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes,
                    device_csr=device_csr)


def _run(enc, generic=False, no_merge=False):
    L = emu_lib()
    pegs, groups = enc.pegs, enc.groups
    ng, G = groups.n_groups, pegs.n_pegs
    nnz_cap = G * ng if not groups.peg_offsets else groups.peg_offsets[ng]
    st, arrs = alloc_results(ng, nnz_cap)
    opts = _abi.Options(force_generic_packer=int(generic), no_singleton_merge=int(no_merge))
    nnz = C.c_int32(0)
    off = np.zeros(ng + 1, np.int32)
    rc = L.emu_estimate_batch(C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), 0, C.byref(nnz), off.ctypes.data_as(_abi.i32p),
                              (C.c_int32 * 8)(), -1, 0, None, np.zeros(max(ng, 1), np.uint8).ctypes.data_as(_abi.u8p), np.zeros(10, np.int64).ctypes.data_as(_abi.i64p))
    assert rc == 0, (rc, L.emu_last_error())
    return finish_results(arrs, ng, int(nnz.value), off)


@pytest.mark.parametrize("device_csr", [False, True])
@pytest.mark.parametrize("seed", range(120))
def test_merged_runs_equal_the_oracle_and_the_unmerged_run(seed, device_csr):
    w = workloads.fuzz_singleton_runs(seed)
    sc = _scenario(w, device_csr)
    enc = encode(sc)
    want = run_oracle(sc)
    for generic in (False, True):
        res = _run(enc, generic=generic)
        assert_matches_oracle(res, want, f"runs {seed} generic={generic}")
        plain = _run(enc, generic=generic, no_merge=True)
        assert_matches_oracle(plain, want, f"unmerged {seed} generic={generic}")
        assert list(res.offsets) == list(plain.offsets) and list(res.order) == list(plain.order) and list(res.placed) == list(plain.placed)
    enc.close()


def test_the_merge_really_happens():
    """R1-shaped: 600 identical singletons against one group -> the device sorts and packs ONE row (emulated steps drop from 600 to 1)"""
    import time
    w = workloads.config_r1(nodes=12, pods_per_node=50)
    sc = _scenario(w, True)
    enc = encode(sc)
    t0 = time.time(); merged = _run(enc); t1 = time.time(); plain = _run(enc, no_merge=True); t2 = time.time()
    assert_matches_oracle(merged, run_oracle(sc), "R1 small")
    assert list(merged.order) == list(plain.order) == list(range(600)) and list(merged.placed) == list(plain.placed)
    assert (t1 - t0) * 3 < (t2 - t1), (t1 - t0, t2 - t1)
    enc.close()
