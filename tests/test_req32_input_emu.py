"""casim_pegs.req32 + req_unit (ABI 10): the requests handed over as 32-bit multiples of a per-lane unit, `req` NULL.  The int64 table the kernels
read is rebuilt on the device (req32 * unit), the packer's int32 table is the caller's own when no node-group amount forces a finer scale.
Every result field equal to the same batch with int64 requests: all three packers, streamed parts, winners only, chained groups, a scale finer
than the unit, lanes without a request, requests the packer cannot narrow, bad units.  CPU: product code under the wave emulator."""
import ctypes as C

import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, PodEquivalenceGroup
from kubernetes_autoscaler_amd.tables import TableSet
from harness import GroupSpec, Scenario, emu_lib, encode_batch, run_emu_streams, run_emu_tables

FIELDS = ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum", "order", "placed")
GiB, MiB = 1 << 30, 1 << 20


def _scenario(seed, **kw):
    w = workloads.fuzz(seed, max_groups=5, max_pegs=14, **kw)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True, lanes=w.lanes)


def _same(a, b, what):
    (ra, ea), (rb, eb) = a, b
    for f in FIELDS:
        assert list(getattr(ra, f)) == list(getattr(rb, f)), f"{what}: {f}"
    if ea is not None:
        assert list(ea["best"]) == list(eb["best"]) and list(ea["packed"]) == list(eb["packed"]) and np.array_equal(ea["keys"], eb["keys"]), f"{what}: expander"


@pytest.mark.parametrize("seed", range(40))
def test_narrowed_requests_give_the_same_batch(seed):
    scs = [_scenario(4000 + 10 * seed + k) for k in range(1 + seed % 4)]
    if len({sc.lanes for sc in scs}) > 1:
        scs = scs[:1]
    enc, ts, _ = encode_batch(scs)
    kinds = [[_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_LEAST_WASTE], [_abi.EXPANDER_MOST_PODS, _abi.EXPANDER_LEAST_WASTE]][seed % 3]
    for generic in (0, 1, 2):   # register packer on int32 lanes, LDS store, register packer on int64 lanes
        for chain in (False, True):
            wide = run_emu_tables(ts, kinds=kinds, generic=generic, chain=chain)
            narrow = run_emu_tables(ts, kinds=kinds, generic=generic, chain=chain, narrow_requests=True)
            _same(narrow, wide, f"seed {seed} packer {generic} chain {chain}")
    enc.close()


@pytest.mark.parametrize("seed", range(10))
def test_narrowed_requests_through_streamed_parts(seed):
    scs = [_scenario(4600 + 10 * seed + k) for k in range(5)]
    if len({sc.lanes for sc in scs}) > 1:
        scs = [scs[0]] * 5
    enc, ts, _ = encode_batch(scs)
    for wo in (False, True):
        a, ea, pa = run_emu_streams(ts, 3, kinds=[_abi.EXPANDER_LEAST_NODES], winners_only=wo)
        b, eb, pb = run_emu_streams(ts, 3, kinds=[_abi.EXPANDER_LEAST_NODES], winners_only=wo, narrow_requests=True)
        assert pa == pb == 3
        for f in ("node_count", "pods_scheduled", "last_index_out", "placed", "order"):
            assert list(getattr(a, f)) == list(getattr(b, f)), (f, wo)
        assert list(ea["packed"]) == list(eb["packed"])
    enc.close()


def _table_set(requests, allocs, pods_per_peg=3, max_nodes=6):
    """one simulation: a PEG per request pair, a node group per allocatable pair"""
    pegs = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}-{j}", labels={"app": f"a{i}"}, requests={"cpu": c, "memory": m}, controller_uid=f"rs{i}") for j in range(pods_per_peg)])
            for i, (c, m) in enumerate(requests)]
    groups = []
    for i, (c, m) in enumerate(allocs):
        cap = {"cpu": c, "memory": m, "pods": 110}
        groups.append(GroupSpec(NodeInfo(Node(name=f"t{i}", labels={}, taints=[], capacity=dict(cap), allocatable=dict(cap))), max_nodes, 0, None))
    enc, ts, _ = encode_batch([Scenario(pegs=pegs, groups=groups, device_csr=True)])
    return enc, ts


def test_a_node_group_amount_that_forces_a_finer_scale_than_the_unit():
    # requests are multiples of 1000 m / 1 GiB (the caller's units), one template offers 2500 m and 2.5 GiB: the packer's scale is 500 m / 0.5 GiB
    enc, ts = _table_set([(1000, 1 * GiB), (2000, 2 * GiB), (3000, 1 * GiB)], [(2500, 5 * GiB // 2), (4000, 8 * GiB), (7500, 3 * GiB + GiB // 2)])
    r32, unit = ts.narrowed_requests()
    assert list(unit) == [1000, GiB]
    for generic in (0, 1, 2):
        _same(run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_WASTE], generic=generic, narrow_requests=True), run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_WASTE], generic=generic),
              f"finer scale, packer {generic}")
    enc.close()


def test_lanes_without_a_request_and_pods_without_any():
    enc, ts = _table_set([(0, 1 * GiB), (0, 3 * GiB), (0, 0)], [(4000, 8 * GiB), (1000, 2 * GiB + 12345)])
    r32, unit = ts.narrowed_requests()
    assert unit[0] == 1 and (r32[:, 0] == 0).all()
    _same(run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES], narrow_requests=True), run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES]), "zero lane")
    enc.close()


def test_amounts_the_packer_cannot_narrow_take_the_int64_lanes():
    # co-prime byte amounts beyond 2^31 on the node side: the request table narrows for the LINK (its own unit), the packer stays on int64 lanes
    enc, ts = _table_set([(100, 3 * GiB), (250, 5 * GiB)], [(4000, 64 * GiB + 1), (8000, 128 * GiB + 7)])
    _same(run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_WASTE], narrow_requests=True), run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_WASTE]), "int64 lanes")
    enc.close()


def _raw_call(ts, mutate):
    L = emu_lib()
    from harness import alloc_results
    pegs, groups = ts.structs(narrow_requests=True)
    keep = mutate(pegs)
    ng = groups.n_groups
    st, arrs = alloc_results(ng, int((ts.peg_hi - ts.peg_lo).sum()), 0)
    nnz = C.c_int32(0)
    off = np.zeros(ng + 1, np.int32)
    opts = _abi.Options()
    rc = L.emu_estimate_batch_query(C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), 0, C.byref(nnz), off.ctypes.data_as(_abi.i32p), None)
    del keep
    return rc


def test_bad_units_and_missing_columns_are_rejected():
    enc, ts = _table_set([(100, GiB)], [(4000, 8 * GiB)])
    run_emu_tables(ts)   # (binds the entry point)

    def zero_unit(p):
        u = np.array([0, 1], np.int64)
        p.req_unit = u.ctypes.data_as(_abi.i64p)
        return u
    assert _raw_call(ts, zero_unit) == _abi.ERR_INVALID

    def no_unit(p):
        p.req_unit = None
    assert _raw_call(ts, no_unit) == _abi.ERR_INVALID   # (req32 without its unit and without req: no request table at all)
    enc.close()


def test_k_sched_entry_points_need_the_int64_table():
    from harness import RemovalCase, removal_encode
    from kubernetes_autoscaler_amd.workloads import removal_scale
    w = removal_scale(20, pods_per_node=3, frac_candidates=0.5, seed=1)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    enc, pod_class, off = removal_encode(case)
    from kubernetes_autoscaler_amd.engine import alloc_removal_results, make_removal_candidates
    L = emu_lib()
    L.emu_simulate_node_removals.restype = C.c_int32
    L.emu_simulate_node_removals.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.RemovalCandidates), C.c_int64, C.POINTER(_abi.RemovalResults)]
    st, keep = make_removal_candidates(case.candidates, off, pod_class, None, None, True, 0, 0, None, None, None, None)
    res, packed = alloc_removal_results(st)
    pegs = _abi.Pegs()
    C.memmove(C.byref(pegs), C.byref(enc.pegs), C.sizeof(_abi.Pegs))
    pegs.req = None
    assert L.emu_simulate_node_removals(C.byref(pegs), C.byref(enc.groups), C.byref(st), 0, C.byref(res)) == _abi.ERR_INVALID
    enc.close()
