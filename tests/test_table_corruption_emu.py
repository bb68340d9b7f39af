"""Tables that lie: one index-like entry of an otherwise valid batch is overwritten (a candidate range past the PEG table, a list offset that
runs backwards, a PEG id nobody has, a negative pod count, a simulation offset past the groups ...) and the batch is handed to the host
orchestration of casim_estimate_batch (csrc/casim_pipeline.h, csrc/casim_streams.h) under the emulator.  The library sits inside the
autoscaler's process: it must answer with an error code or with some result — never read or write outside the arrays it was given.
A plain run of this file catches crashes; tests/tools/sanitize_cpu.sh runs it with the kernels and the pipeline under AddressSanitizer,
where one element past a numpy array ends the process with a report."""
import copy
import ctypes as C

import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import alloc_results
from harness import GroupSpec, Scenario, emu_lib, encode, encode_batch
from kubernetes_autoscaler_amd.tables import TableSet

EXTREMES = (-1, -(2 ** 31), 2 ** 31 - 1, 10 ** 6)


def _call(ts, n_streams=0, chain=False):
    """rc of the batch call on `ts`; result arrays sized for the worst the dimensions allow (the lists of a lying table may be longer than honest ones)."""
    L = emu_lib()
    L.emu_estimate_batch_query.restype = C.c_int32
    L.emu_estimate_batch_query.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.Options), C.POINTER(_abi.Results),
                                           C.c_int64, _abi.i32p, _abi.i32p, C.POINTER(_abi.OptionQuery)]
    L.emu_estimate_batch_streams.restype = C.c_int32
    L.emu_estimate_batch_streams.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.Options), C.POINTER(_abi.Results),
                                             _abi.i32p, _abi.i32p, C.POINTER(_abi.OptionQuery), _abi.i32p]
    pegs, groups = ts.structs()
    ng = groups.n_groups
    st, arrs = alloc_results(ng, max(pegs.n_pegs, 1) * max(ng, 1) + 64)
    S = max(groups.n_sims, 1)
    ks = (C.c_int32 * 1)(_abi.EXPANDER_LEAST_NODES)
    exp = dict(best=np.full(S, -1, np.int32), n_best=np.zeros(S, np.int32), best_set=np.zeros(max(ng, 1), np.uint8), keys=np.zeros((S, 10), np.int64),
               packed=np.zeros(S, np.int64))
    q = _abi.OptionQuery(kinds=ks, n_kinds=1, per_sim=1, best_out=exp["best"].ctypes.data_as(_abi.i32p), n_best_out=exp["n_best"].ctypes.data_as(_abi.i32p),
                         best_set_out=exp["best_set"].ctypes.data_as(_abi.u8p), key_out=exp["keys"].ctypes.data_as(_abi.i64p),
                         packed_out=exp["packed"].ctypes.data_as(_abi.i64p))
    opts = _abi.Options(n_streams=int(n_streams), chain_last_index=int(chain))
    nnz, parts = C.c_int32(0), C.c_int32(0)
    off = np.zeros(ng + 1, np.int32)
    if n_streams:
        return L.emu_estimate_batch_streams(C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), C.byref(nnz), off.ctypes.data_as(_abi.i32p), C.byref(q),
                                            C.byref(parts))
    return L.emu_estimate_batch_query(C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), 0, C.byref(nnz), off.ctypes.data_as(_abi.i32p), C.byref(q))


def _targets(ts):
    """(name, array) of every index-like column of the set"""
    out = [("pegs.count", ts.pegs["count"])]
    for k in ("max_nodes", "existing_nodes", "last_index", "allowed_pods", "init_pods"):
        if ts.groups.get(k) is not None:
            out.append(("groups." + k, ts.groups[k]))
    for k in ("peg_lo", "peg_hi", "peg_offsets", "peg_index", "sim_offsets", "global_id"):
        a = getattr(ts, k)
        if a is not None and a.size:
            out.append((k, a))
    return out


def _batch(seed, n):
    scs = []
    for i in range(n):
        w = workloads.fuzz(91000 + 17 * seed + i, max_groups=4, max_pegs=10, rich=(seed % 2 == 0))
        scs.append(Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True))
    return encode_batch(scs)


@pytest.mark.parametrize("seed", range(12))
def test_one_lying_entry_in_a_batch_with_candidate_ranges(seed):
    enc, ts, _ = _batch(seed, 2 + seed % 3)
    assert _call(ts) == 0 and _call(ts, n_streams=2) == 0   # (the honest tables)
    rng = np.random.default_rng(seed)
    G, NG = ts.n_pegs, ts.n_groups
    for name, _ in _targets(ts):
        for _ in range(4):
            bad = copy.deepcopy(ts)
            arr = dict(_targets(bad))[name]
            flat = arr.reshape(-1)
            i = int(rng.integers(0, flat.size))
            flat[i] = int(rng.choice(EXTREMES + (G, G + 1, NG, NG + 1, int(flat[i]) + 1, int(flat[i]) - 1)))
            for streams, chain in ((0, False), (2, False), (0, True)):
                rc = _call(bad, n_streams=streams, chain=chain)
                assert isinstance(rc, int), (name, i)   # (whatever it says: it came back)
    enc.close()


@pytest.mark.parametrize("seed", range(8))
def test_one_lying_entry_in_a_call_with_explicit_lists(seed):
    w = workloads.fuzz(93000 + seed, max_groups=4, max_pegs=10, rich=True)
    rng = np.random.default_rng(100 + seed)
    groups = []
    for g in w.groups:
        ids = sorted(rng.choice(len(w.pegs), size=int(rng.integers(1, len(w.pegs) + 1)), replace=False).tolist()) if w.pegs else []
        groups.append(GroupSpec(g.template, g.max_nodes, g.last_index, ids))
    enc = encode(Scenario(pegs=w.pegs, groups=groups, existing=w.existing, lanes=w.lanes))
    ts = TableSet.from_encoder(enc)
    assert _call(ts) == 0
    G, NG = ts.n_pegs, ts.n_groups
    for name, _ in _targets(ts):
        for _ in range(6):
            bad = copy.deepcopy(ts)
            flat = dict(_targets(bad))[name].reshape(-1)
            # (peg_index is peg_offsets[NG] entries long BY DEFINITION: the last offset is the one entry nobody can check, it stays)
            i = int(rng.integers(0, flat.size - (1 if name == "peg_offsets" else 0)))
            flat[i] = int(rng.choice(EXTREMES + (G, G + 1, NG, NG + 1, int(flat[i]) + 1, int(flat[i]) - 1)))
            rc = _call(bad)
            assert isinstance(rc, int), (name, i)
    enc.close()


def test_dimensions_that_lie():
    """more PEGs / groups / words / lanes announced than any sane table has, or negative ones: refused before anything is read"""
    enc, ts, _ = _batch(3, 2)
    pegs, groups = ts.structs()
    L = emu_lib()
    for field, values in (("n_res", (-1, 0, 9, 1 << 20)), ("w_taint", (-1, 65, 1 << 20)), ("w_label", (-1, 1 << 20)), ("w_excl", (-1, 1 << 20)), ("w_zone", (-1, 1 << 20)),
                          ("n_pegs", (-1,))):
        for v in values:
            p2 = _abi.Pegs.from_buffer_copy(pegs)
            setattr(p2, field, v)
            st, arrs = alloc_results(groups.n_groups, ts.n_pegs * groups.n_groups + 64)
            nnz = C.c_int32(0)
            off = np.zeros(groups.n_groups + 1, np.int32)
            opts = _abi.Options()
            rc = L.emu_estimate_batch_query(C.byref(p2), C.byref(groups), C.byref(opts), C.byref(st), 0, C.byref(nnz), off.ctypes.data_as(_abi.i32p), None)
            assert rc != 0, (field, v)
    for v in (-1,):
        g2 = _abi.Groups.from_buffer_copy(groups)
        g2.n_groups = v
        st, arrs = alloc_results(groups.n_groups, ts.n_pegs * groups.n_groups + 64)
        nnz = C.c_int32(0)
        off = np.zeros(groups.n_groups + 1, np.int32)
        opts = _abi.Options()
        assert L.emu_estimate_batch_query(C.byref(pegs), C.byref(g2), C.byref(opts), C.byref(st), 0, C.byref(nnz), off.ctypes.data_as(_abi.i32p), None) != 0
    enc.close()


# ---- the callers either side of the path: casim_try_schedule_pods, casim_simulate_node_removals -------------------------------------
def _sched_lib():
    L = emu_lib()
    L.emu_try_schedule_pods.restype = C.c_int32
    L.emu_try_schedule_pods.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.PodSequence), C.c_int64, _abi.i32p, _abi.i32p, _abi.i32p, _abi.i32p]
    L.emu_simulate_node_removals.restype = C.c_int32
    L.emu_simulate_node_removals.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.RemovalCandidates), C.c_int64, C.POINTER(_abi.RemovalResults)]
    return L


@pytest.mark.parametrize("seed", range(10))
def test_one_lying_entry_in_a_pod_sequence(seed):
    from harness import SchedCase, sched_encode, similar_keys
    from kubernetes_autoscaler_amd.engine import make_pod_sequence
    gen = workloads.fuzz_pending_domains if seed % 2 else workloads.fuzz_pending
    w = gen(95000 + seed)
    case = SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)
    enc, pod_class = sched_encode(case)
    L = _sched_lib()
    rng = np.random.default_rng(seed)
    n_nodes, n_classes, P = enc.groups.n_groups, enc.pegs.n_pegs, len(case.pods)

    def run(pc, hints, last_index):
        seq, keep = make_pod_sequence(pc, hints, case.acceptable, case.break_on_failure, 0, enc.rules, similar_keys(case.pods))
        seq.last_index = int(last_index)
        out = np.full(max(P, 1), -1, np.int32)
        li, ns = C.c_int32(0), C.c_int32(0)
        return L.emu_try_schedule_pods(C.byref(enc.pegs), C.byref(enc.groups), C.byref(seq), 0, out.ctypes.data_as(_abi.i32p), C.byref(li), C.byref(ns), None)
    hints = None if case.hints is None else np.asarray(case.hints, np.int32)
    assert run(pod_class, hints, case.last_index) >= 0
    lies = EXTREMES + (n_nodes, n_nodes + 1, n_classes, n_classes + 1)
    for _ in range(10):
        if P:
            pc = np.array(pod_class, np.int32); pc[int(rng.integers(0, P))] = int(rng.choice(lies))
            assert isinstance(run(pc, hints, case.last_index), int)
            if hints is not None:
                h = hints.copy(); h[int(rng.integers(0, P))] = int(rng.choice(lies))
                assert isinstance(run(pod_class, h, case.last_index), int)
        assert isinstance(run(pod_class, hints, int(rng.choice(lies))), int)
    enc.close()


@pytest.mark.parametrize("seed", range(10))
def test_one_lying_entry_in_a_removal_loop(seed):
    from harness import RemovalCase, removal_encode
    from kubernetes_autoscaler_amd.engine import alloc_removal_results, make_removal_candidates
    gen = workloads.fuzz_removals_domains if seed % 2 else workloads.fuzz_removals
    r = gen(96000 + seed)
    case = RemovalCase(nodes=r.nodes, candidates=r.candidates, destination=r.destination, hints=r.hints, persist=r.persist, max_removable=r.max_removable,
                       last_index=r.last_index)
    enc, pod_class, off = removal_encode(case)
    L = _sched_lib()
    rng = np.random.default_rng(seed)
    n_nodes, n_classes, K, total = enc.groups.n_groups, enc.pegs.n_pegs, len(case.candidates), len(pod_class)
    hints = case.flat_hints()

    def run(cands, offsets, classes, h, last_index=case.last_index, max_removable=case.max_removable, ext_capacity=None):
        st, keep = make_removal_candidates(cands, offsets, classes, h, case.destination, case.persist, 0, 0, case.flat_sticky(), ext_capacity, enc.rules)
        st.last_index = int(last_index); st.max_removable = int(max_removable)
        # the result arrays follow the HONEST sizes: a lying scalar must not make the library write past them either
        honest, _ = make_removal_candidates(case.candidates, off, pod_class, hints, case.destination, case.persist, 0, 0, case.flat_sticky(),
                                            max(int(st.ext_capacity), 0) if int(st.ext_capacity) < 10 ** 6 else None, enc.rules)
        res, packed = alloc_removal_results(honest)
        return L.emu_simulate_node_removals(C.byref(enc.pegs), C.byref(enc.groups), C.byref(st), 0, C.byref(res))
    assert run(case.candidates, off, pod_class, hints) >= 0
    lies = EXTREMES + (n_nodes, n_nodes + 1, n_classes, n_classes + 1, total, total + 1)
    for _ in range(8):
        if K:
            c = np.array(case.candidates, np.int32); c[int(rng.integers(0, K))] = int(rng.choice(lies))
            assert isinstance(run(c, off, pod_class, hints), int)
            o = np.array(off, np.int32); o[int(rng.integers(0, K))] = int(rng.choice(lies))     # (not the last one: the lists are pod_offsets[K] long by definition)
            assert isinstance(run(case.candidates, o, pod_class, hints), int)
        if total:
            pc = np.array(pod_class, np.int32); pc[int(rng.integers(0, total))] = int(rng.choice(lies))
            assert isinstance(run(case.candidates, off, pc, hints), int)
            if hints is not None:
                h = np.array(hints, np.int32); h[int(rng.integers(0, total))] = int(rng.choice(lies))
                assert isinstance(run(case.candidates, off, pod_class, h), int)
        assert isinstance(run(case.candidates, off, pod_class, hints, last_index=int(rng.choice(lies))), int)
        assert isinstance(run(case.candidates, off, pod_class, hints, max_removable=int(rng.choice(lies))), int)
        assert isinstance(run(case.candidates, off, pod_class, hints, ext_capacity=int(rng.choice((-1, 0, 1, 2)))), int)
    enc.close()
