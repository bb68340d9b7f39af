"""casim_options.winners_only (SURVEY 8e "per-PEG placed[] arrays only travel for the winning NG"): the call returns
order / placed of the WINNING group of every simulation only, compacted on the device; scalars, offsets and the expander's answer are
unchanged.  Product kernels under the wave emulator, one part and cut into parts (casim_streams.h), both packers, a validity mask that
leaves simulations without an option, every expander kind."""
import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import winner_offsets
from harness import GroupSpec, Scenario, encode_batch, run_emu_streams, run_emu_tables

SCALARS = ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum")


def _batch(n, seed0=5200, max_groups=5, max_pegs=14):
    scs = []
    for k in range(n):
        w = workloads.fuzz(seed0 + k, max_groups=max_groups, max_pegs=max_pegs)
        scs.append(Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True))
    return encode_batch(scs)


def _check(full, fexp, got, gexp, what):
    for f in SCALARS:
        assert np.array_equal(getattr(full, f), getattr(got, f)), (what, f)
    assert list(gexp["best"]) == list(fexp["best"]) and list(gexp["packed"]) == list(fexp["packed"]), what
    w = winner_offsets(got.offsets, gexp["best"])
    assert w[-1] <= len(got.order)
    for s, b in enumerate(gexp["best"]):
        b = int(b)
        if b < 0:
            assert w[s + 1] == w[s]
            continue
        a, e = int(full.offsets[b]), int(full.offsets[b + 1])
        assert list(got.order[w[s]:w[s + 1]]) == list(full.order[a:e]), (what, "order of simulation", s)
        assert list(got.placed[w[s]:w[s + 1]]) == list(full.placed[a:e]), (what, "placed of simulation", s)
    return int(w[-1])


@pytest.mark.parametrize("generic", [False, True], ids=["register-packer", "int64-packer"])
@pytest.mark.parametrize("kind", [_abi.EXPANDER_LEAST_NODES, _abi.EXPANDER_MOST_PODS, _abi.EXPANDER_LEAST_WASTE])
def test_winners_only_is_the_full_answer_restricted_to_the_winners(kind, generic):
    enc, ts, _ = _batch(23)
    rng = np.random.default_rng(kind + 7)
    valid = (rng.random(ts.n_groups) < 0.55).astype(np.uint8)     # some simulations end up without any option
    full, fexp = run_emu_tables(ts, kinds=[kind], valid=valid, generic=generic)
    got, gexp = run_emu_tables(ts, kinds=[kind], valid=valid, generic=generic, winners_only=True)
    n = _check(full, fexp, got, gexp, "one part")
    assert 0 < n < int(full.offsets[-1])
    assert (np.asarray(gexp["best"]) < 0).any() and (np.asarray(gexp["best"]) >= 0).any()
    for k in (2, 5):
        got, gexp, parts = run_emu_streams(ts, k, kinds=[kind], valid=valid, generic=generic, winners_only=True)
        assert parts == k
        assert _check(full, fexp, got, gexp, f"{k} parts") == n
    enc.close()


def test_winners_only_with_one_simulation_and_the_fused_front_kernel():
    w = workloads.config_c2(3)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)
    enc, ts, _ = encode_batch([sc])
    full, fexp = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES])
    got, gexp = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES], winners_only=True)
    n = _check(full, fexp, got, gexp, "C2, one simulation")
    b = int(gexp["best"][0])
    assert n == int(full.offsets[b + 1] - full.offsets[b]) > 0
    enc.close()


def test_winners_only_without_a_query_is_refused():
    import ctypes as C
    from harness import emu_lib
    from kubernetes_autoscaler_amd.engine import alloc_results
    enc, ts, _ = _batch(3)
    pegs, groups = ts.structs()
    L = emu_lib()
    st, arrs = alloc_results(groups.n_groups, int((ts.peg_hi - ts.peg_lo).sum()))
    opts = _abi.Options(winners_only=1)
    nnz = C.c_int32(0)
    off = np.zeros(groups.n_groups + 1, np.int32)
    run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES])   # (binds emu_estimate_batch_query)
    rc = L.emu_estimate_batch_query(C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), 0, C.byref(nnz), off.ctypes.data_as(_abi.i32p), None)
    assert rc == _abi.ERR_INVALID
    enc.close()
