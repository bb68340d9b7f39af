"""Shared test harness: runs one scale-up scenario through
  (a) the CPU oracle (pod by pod, object level),
  (b) the product encoder + the product kernels under the wave emulator (CPU tests), or
  (c) the product encoder + libcasim on a real MI355X (-m gpu tests),
and compares them bit for bit."""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from kubernetes_autoscaler_amd import _abi
from kubernetes_autoscaler_amd.encoder import Encoder
from kubernetes_autoscaler_amd.engine import BatchResult, alloc_results, finish_results
from kubernetes_autoscaler_amd.objects import Node, NodeInfo, PodEquivalenceGroup, build_test_node
from oracle_driver import OracleEstimate, OracleScenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIB = os.environ.get("CASIM_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "libcasim_emu.so")   # override: sanitizer builds (tests/tools/sanitize_cpu.sh)


@dataclass
class GroupSpec:
    template: NodeInfo
    max_nodes: int = 0
    last_index: int = 0
    pegs: Optional[Sequence[int]] = None   # indices into Scenario.pegs; None = all / device-side feasibility


@dataclass
class Scenario:
    pegs: List[PodEquivalenceGroup]
    groups: List[GroupSpec]
    existing: List[NodeInfo] = field(default_factory=list)   # nodes already in the cluster snapshot
    lanes: Sequence[str] = ("cpu", "memory")
    fastpath: bool = False
    device_csr: bool = False    # let the engine derive the schedulable subsets (feasibility kernel)


# ---------------------------------------------------------------------------------------------
def run_oracle(sc: Scenario, list_shuffle_seed: int = 0, chain: bool = False):
    """Per group: OracleEstimate plus the list of global PEG ids it was given.  list_shuffle_seed != 0: every
    scheduling attempt sees the node list in a fresh random order (Go map iteration).
    chain: lastIndex carried from one group's Estimate to the next (one snapshot, one plugin runner: plugin_runner.go:138)."""
    s = OracleScenario(lanes=sc.lanes, list_shuffle_seed=list_shuffle_seed)
    for info in sc.existing:
        s.add_existing(info)
    out = []
    carried = None
    for g in sc.groups:
        tmpl = s.node(g.template)
        if g.pegs is None and sc.device_csr:
            ids = [i for i, pg in enumerate(sc.pegs) if pg.exemplar() is not None and s.check_predicates(tmpl, pg.exemplar())[0]]
        else:
            ids = list(range(len(sc.pegs))) if g.pegs is None else list(g.pegs)
        est = s.estimate(tmpl, [sc.pegs[i] for i in ids], max_nodes=g.max_nodes, last_index=g.last_index if (carried is None or not chain) else carried,
                         fastpath=sc.fastpath, node_pods_cap=0)
        carried = est.last_index_out
        out.append((est, ids))
    s.close()
    return out


def encode(sc: Scenario, named_lanes: bool = False) -> Encoder:
    """named_lanes: the Go shim's call sequence — three positional lanes, every other resource by NAME (casim_enc_pod_set_request /
    casim_enc_group_set_allocatable, ABI 9)"""
    enc = Encoder(lanes=sc.lanes, named_lanes=named_lanes)
    for pg in sc.pegs:
        enc.add_peg(pg)
    for info in sc.existing:
        for p in info.pods:
            enc.add_existing_pod(p, info.node.labels)
    for g in sc.groups:
        pegs = None
        if not sc.device_csr:
            pegs = list(range(len(sc.pegs))) if g.pegs is None else list(g.pegs)
        enc.add_group(g.template, max_nodes=g.max_nodes, existing_nodes=len(sc.existing), last_index=g.last_index, pegs=pegs)
    enc.finalize()
    return enc


_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        L = C.CDLL(EMU_LIB)
        L.emu_estimate_batch.restype = C.c_int32
        L.emu_estimate_batch.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.Options), C.POINTER(_abi.Results),
                                         C.c_int64, _abi.i32p, _abi.i32p, _abi.i32p, C.c_int32, C.c_int32, _abi.i32p, _abi.u8p, _abi.i64p]
        L.emu_feasibility.restype = C.c_int32
        L.emu_feasibility.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), _abi.u64p]
        L.emu_last_error.restype = C.c_char_p
        _emu = L
    return _emu


def run_emu(enc: Encoder, fastpath=False, lds_budget=0, kinds=None, group_id_base=0, generic=False, front=True, chain=False):
    """Product kernels under the wave emulator.  Returns (BatchResult, best) where best is
    None or (best_index, n_best, best_set, key)."""
    L = emu_lib()
    pegs, groups = enc.pegs, enc.groups
    ng, G = groups.n_groups, pegs.n_pegs
    nnz_cap = G * ng if not groups.peg_offsets else groups.peg_offsets[ng]
    st, arrs = alloc_results(ng, nnz_cap)
    opts = _abi.Options(fastpath=int(fastpath), force_generic_packer=int(generic), no_front_kernel=int(not front), chain_last_index=int(chain))
    nnz = C.c_int32(0)
    off = np.zeros(ng + 1, np.int32)
    best = (C.c_int32 * 2)(-1, 0)
    bset = np.zeros(max(ng, 1), np.uint8)
    key = np.zeros(10, np.int64)
    ks = (C.c_int32 * 8)(*(kinds or []))
    rc = L.emu_estimate_batch(C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), int(lds_budget), C.byref(nnz),
                              off.ctypes.data_as(_abi.i32p), ks, len(kinds) if kinds is not None else -1, group_id_base,
                              best if kinds is not None else None, bset.ctypes.data_as(_abi.u8p), key.ctypes.data_as(_abi.i64p))
    assert rc == 0, (rc, L.emu_last_error())
    res = finish_results(arrs, ng, int(nnz.value), off)
    return res, ((best[0], best[1], bset[:ng].copy(), key.copy()) if kinds is not None else None)


def run_emu_feasibility(enc: Encoder) -> np.ndarray:
    L = emu_lib()
    wg = (enc.pegs.n_pegs + 63) // 64
    bits = np.zeros((max(enc.groups.n_groups, 1), max(wg, 1)), np.uint64)
    rc = L.emu_feasibility(C.byref(enc.pegs), C.byref(enc.groups), bits.ctypes.data_as(_abi.u64p))
    assert rc == 0, (rc, L.emu_last_error())
    return bits[:enc.groups.n_groups, :wg]


def run_gpu(enc: Encoder, ctx, fastpath=False, kinds=None, group_id_base=0, generic=False, chain=False):
    from kubernetes_autoscaler_amd.engine import Problem
    with Problem(ctx, enc.pegs, enc.groups, fastpath, generic, chain_last_index=chain) as p:
        p.run()
        res = p.fetch()
        best = p.best_option(kinds, group_id_base) if kinds is not None else None
    return res, best


# ---------------------------------------------------------------------------------------------
def assert_matches_oracle(res: BatchResult, oracle, what=""):
    """Bit-exact comparison of one batch against the per-group oracle results."""
    assert len(oracle) == len(res.node_count), what
    for i, (est, ids) in enumerate(oracle):
        tag = f"{what} group {i}"
        order, placed = res.group(i)
        assert int(res.status[i]) == 0, tag
        assert list(order) == [ids[k] for k in est.order], f"{tag}: PEG order"
        assert list(placed) == list(est.placed), f"{tag}: placed per PEG\n got {list(placed)}\n want {list(est.placed)}"
        got = (int(res.node_count[i]), int(res.pods_scheduled[i]), int(res.nodes_added[i]), int(res.limiter_nodes[i]),
               int(res.last_index_out[i]), int(res.req_cpu_sum[i]), int(res.req_mem_sum[i]))
        want = (est.node_count, est.pods_scheduled, est.nodes_added, est.limiter_nodes, est.last_index_out, est.req_cpu_sum,
                est.req_mem_sum)
        assert got == want, f"{tag}: (nodes, pods, added, limiter, lastIndex, cpu, mem) got {got} want {want}"


def dummy_existing(n: int) -> List[NodeInfo]:
    return [NodeInfo(build_test_node(f"existing-{i}", 100, 100 * 1024 * 1024, pods=10)) for i in range(n)]


# ---------------------------------------------------------------------------------------------
# filter-out-schedulable: TrySchedulePods on oracle / emulator / GPU
# ---------------------------------------------------------------------------------------------
@dataclass
class SchedCase:
    nodes: List[NodeInfo]                       # the cluster snapshot, list order
    pods: list                                  # pending pods, processing order
    hints: Optional[Sequence[int]] = None       # node index per pod or -1
    acceptable: Optional[Sequence[int]] = None  # per node
    break_on_failure: bool = False
    last_index: int = 0
    lanes: Sequence[str] = ("cpu", "memory")


def similar_keys(pods):
    """dense controller ids for SimilarPodsScheduling: -1 = no controller or a DaemonSet pod"""
    keys = {}
    return [(-1 if not p.controller_uid or p.daemonset else keys.setdefault(p.controller_uid, len(keys))) for p in pods]


def sched_oracle(case: SchedCase):
    """(node_out, last_index, n_scheduled) from the object-level oracle; SimilarPods keyed by controller_uid."""
    s = OracleScenario(lanes=case.lanes)
    for info in case.nodes:
        s.add_existing(info)
    sk = similar_keys(case.pods)
    # the oracle's SimilarPodsScheduling entry is a pod spec id: equal specs (+ labels) must share one id, like the
    # reference's PodSpecSemanticallyEqual / DeepEqual(labels) match (similar_pods.go:48-50)
    canon = {}
    pods = [canon.setdefault(p.spec_key(), p) for p in case.pods]
    out = s.try_schedule_pods(pods, case.hints, sk, case.acceptable, case.break_on_failure, case.last_index)
    s.close()
    return out


def sched_encode(case: SchedCase):
    from kubernetes_autoscaler_amd.scheduling import encode_pending_pods
    return encode_pending_pods(case.nodes, case.pods, case.lanes)


def sched_emu(case: SchedCase, lds_budget=0):
    """Product encoder + K_sched under the wave emulator: (status, node_out, last_index, n_scheduled, info)."""
    from kubernetes_autoscaler_amd.engine import make_pod_sequence
    L = emu_lib()
    if not hasattr(L, "_sched_ready"):
        L.emu_try_schedule_pods.restype = C.c_int32
        L.emu_try_schedule_pods.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.PodSequence), C.c_int64,
                                            _abi.i32p, _abi.i32p, _abi.i32p, _abi.i32p]
        L._sched_ready = True
    enc, pod_class = sched_encode(case)
    seq, keep = make_pod_sequence(pod_class, case.hints, case.acceptable, case.break_on_failure, case.last_index, enc.rules,
                                  similar_keys(case.pods))
    node_out = np.full(max(len(case.pods), 1), -1, np.int32)
    li, ns = C.c_int32(0), C.c_int32(0)
    info = (C.c_int32 * 2)(0, 0)
    rc = L.emu_try_schedule_pods(C.byref(enc.pegs), C.byref(enc.groups), C.byref(seq), int(lds_budget),
                                 node_out.ctypes.data_as(_abi.i32p), C.byref(li), C.byref(ns), info)
    assert rc >= 0, (rc, L.emu_last_error())
    del keep
    enc.close()
    return rc, node_out[:len(case.pods)].copy(), li.value, ns.value, (info[0], info[1])


def sched_gpu(case: SchedCase, ctx):
    enc, pod_class = sched_encode(case)
    rc, node_out, li, ns = ctx.try_schedule_pods(enc.pegs, enc.groups, pod_class, case.hints, case.acceptable,
                                                 case.break_on_failure, case.last_index, rules=enc.rules,
                                                 similar_key=similar_keys(case.pods))
    enc.close()
    return rc, node_out.copy(), li, ns


def assert_sched_matches(got, want, what=""):
    rc, node_out, li, ns = got[0], got[1], got[2], got[3]
    w_out, w_li, w_ns = want
    assert rc == 0, f"{what}: status {rc}"
    assert list(node_out) == list(w_out), f"{what}: node per pod\n got {list(node_out)}\n want {list(w_out)}"
    assert (li, ns) == (w_li, w_ns), f"{what}: (lastIndex, scheduled) got {(li, ns)} want {(w_li, w_ns)}"


# ---------------------------------------------------------------------------------------------
# scale-down removal simulation
# ---------------------------------------------------------------------------------------------
@dataclass
class RemovalCase:
    nodes: List[NodeInfo]                          # the cluster snapshot, list order; pods = everything running there
    candidates: List[int]                          # node indices, planner order
    destination: Optional[Sequence[int]] = None    # per node
    hints: Optional[dict] = None                   # id(pod) -> node index
    persist: bool = True
    max_removable: int = 0
    last_index: int = 0
    sticky: Optional[set] = None                   # id(pod) of pods the host must re-examine before a second move
    ext_capacity: Optional[int] = None             # None = default (2 * pods + 64); 0 = stop at any arrival
    atomic: Optional[Sequence[int]] = None         # per candidate: node of an atomically scaled group (not counted toward the limit)
    lanes: Sequence[str] = ("cpu", "memory")

    def pod_lists(self):
        return [[p for p in self.nodes[c].pods if not p.daemonset] for c in self.candidates]

    def flat_hints(self):
        if not self.hints:
            return None
        return [self.hints.get(id(p), -1) for lst in self.pod_lists() for p in lst]

    def flat_sticky(self):
        if not self.sticky:
            return None
        return [1 if id(p) in self.sticky else 0 for lst in self.pod_lists() for p in lst]


def removal_oracle(case: RemovalCase):
    s = OracleScenario(lanes=case.lanes)
    for info in case.nodes:
        s.add_existing(info)
    out = s.simulate_node_removals(case.candidates, case.pod_lists(), case.flat_hints(), case.destination, case.persist,
                                   case.max_removable, case.flat_sticky(), case.ext_capacity, case.last_index, case.atomic)
    s.close()
    return out


def removal_encode(case: RemovalCase):
    """Classes of the pods to move + one node record per snapshot node (all its pods preloaded)."""
    enc = Encoder(lanes=case.lanes, explicit_self_exclusion=True)
    class_of, pod_class, off = {}, [], [0]
    for lst in case.pod_lists():
        for p in lst:
            k = p.spec_key()
            if k not in class_of:
                class_of[k] = enc.add_peg(PodEquivalenceGroup(pods=[p]))
            pod_class.append(class_of[k])
        off.append(len(pod_class))
    for info in case.nodes:
        enc.add_group(info, pegs=[])
    enc.finalize()
    return enc, np.array(pod_class, np.int32), np.array(off, np.int32)


class EmuContext:
    """Stands in for engine.Context in CPU tests of the host mirrors: same method, product kernels under the emulator."""

    def __init__(self, lds_budget=0):
        self.lds_budget = lds_budget

    def try_schedule_pods(self, classes, nodes, pod_class, hint_node=None, node_acceptable=None, break_on_failure=False,
                          last_index=0, time_iters=0, rules=None, similar_key=None):
        from kubernetes_autoscaler_amd.engine import make_pod_sequence
        L = emu_lib()
        if not hasattr(L, "_sched_ready"):
            L.emu_try_schedule_pods.restype = C.c_int32
            L.emu_try_schedule_pods.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.PodSequence), C.c_int64,
                                                _abi.i32p, _abi.i32p, _abi.i32p, _abi.i32p]
            L._sched_ready = True
        seq, keep = make_pod_sequence(pod_class, hint_node, node_acceptable, break_on_failure, last_index, rules, similar_key)
        node_out = np.full(max(seq.n_pods, 1), -1, np.int32)
        li, ns = C.c_int32(0), C.c_int32(0)
        info = (C.c_int32 * 2)(0, 0)
        rc = L.emu_try_schedule_pods(C.byref(classes), C.byref(nodes), C.byref(seq), int(self.lds_budget),
                                     node_out.ctypes.data_as(_abi.i32p), C.byref(li), C.byref(ns), info)
        assert rc >= 0, (rc, L.emu_last_error())
        del keep
        return rc, node_out[:seq.n_pods], li.value, ns.value

    def simulate_node_removals(self, classes, nodes, cand_node, pod_offsets, pod_class, hint_node=None, destination=None,
                               persist=True, max_removable=0, last_index=0, pod_sticky=None, ext_capacity=None, rules=None,
                               cand_atomic=None):
        from kubernetes_autoscaler_amd.engine import alloc_removal_results, finish_removal_results, make_removal_candidates
        L = emu_lib()
        if not hasattr(L, "_removal_ready"):
            L.emu_simulate_node_removals.restype = C.c_int32
            L.emu_simulate_node_removals.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.RemovalCandidates),
                                                     C.c_int64, C.POINTER(_abi.RemovalResults)]
            L._removal_ready = True
        st, keep = make_removal_candidates(cand_node, pod_offsets, pod_class, hint_node, destination, persist, max_removable, last_index,
                                           pod_sticky, ext_capacity, rules, cand_atomic)
        res, packed = alloc_removal_results(st)
        rc = L.emu_simulate_node_removals(C.byref(classes), C.byref(nodes), C.byref(st), int(self.lds_budget), C.byref(res))
        assert rc >= 0, (rc, L.emu_last_error())
        del keep
        return finish_removal_results(rc, st, res, packed)


def removal_device(case: RemovalCase, ctx):
    """ctx: engine.Context (MI355X) or EmuContext.  Returns engine.RemovalResult."""
    enc, pod_class, off = removal_encode(case)
    out = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pod_class, case.flat_hints(), case.destination,
                                     persist=case.persist, max_removable=case.max_removable, last_index=case.last_index,
                                     pod_sticky=case.flat_sticky(), ext_capacity=case.ext_capacity, rules=enc.rules,
                                     cand_atomic=case.atomic)
    enc.close()
    return out


def assert_removal_matches(got, want, what=""):
    assert got.status == 0, f"{what}: status {got.status}"
    assert got.n_processed == want["n_processed"], f"{what}: candidates processed got {got.n_processed} want {want['n_processed']}"
    assert list(got.removable) == list(want["removable"]), f"{what}: removable\n got {list(got.removable)}\n want {list(want['removable'])}"
    assert list(got.node_out) == list(want["node_out"]), f"{what}: destinations\n got {list(got.node_out)}\n want {list(want['node_out'])}"
    ext = list(zip(got.ext_candidate.tolist(), got.ext_pod.tolist(), got.ext_node.tolist()))
    assert ext == want["ext"], f"{what}: pods listed again (candidate, pod, node)\n got {ext}\n want {want['ext']}"
    assert got.last_index == want["last_index"], f"{what}: lastIndex got {got.last_index} want {want['last_index']}"


# ---------------------------------------------------------------------------------------------
# Estimate on the whole snapshot: groups whose PEGs carry domain rules
# ---------------------------------------------------------------------------------------------
def cluster_encode(sc: Scenario, group_index: int = 0):
    """Per-node tables for one group of a scenario (product helper estimator.encode_cluster_estimate)."""
    from kubernetes_autoscaler_amd.estimator import encode_cluster_estimate
    g = sc.groups[group_index]
    ids = list(range(len(sc.pegs))) if g.pegs is None else list(g.pegs)
    return encode_cluster_estimate(sc.lanes, [sc.pegs[i] for i in ids], sc.existing, g.template, g.max_nodes), ids


def cluster_estimate_emu(sc: Scenario, group_index: int = 0, lds_budget=0):
    """K_est under the wave emulator for one group: (status, result dict, PEG ids)."""
    from kubernetes_autoscaler_amd.engine import finish_cluster_estimate, make_cluster_estimate
    L = emu_lib()
    if not hasattr(L, "_cluster_ready"):
        L.emu_estimate_on_cluster.restype = C.c_int32
        L.emu_estimate_on_cluster.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.ClusterEstimate), C.c_int64,
                                              C.POINTER(_abi.ClusterEstimateResult)]
        L._cluster_ready = True
    enc, ids = cluster_encode(sc, group_index)
    g = sc.groups[group_index]
    params, res, arrs = make_cluster_estimate(enc.pegs, len(sc.existing), g.max_nodes, g.last_index, enc.rules, enc.port_block)
    rc = L.emu_estimate_on_cluster(C.byref(enc.pegs), C.byref(enc.groups), C.byref(params), int(lds_budget), C.byref(res))
    assert rc >= 0, (rc, L.emu_last_error())
    out = finish_cluster_estimate(res, arrs, enc.pegs.n_pegs)
    enc.close()
    return rc, out, ids


def cluster_estimate_gpu(sc: Scenario, ctx, group_index: int = 0):
    enc, ids = cluster_encode(sc, group_index)
    g = sc.groups[group_index]
    rc, out = ctx.estimate_on_cluster(enc.pegs, enc.groups, len(sc.existing), g.max_nodes, g.last_index, enc.rules, enc.port_block)
    enc.close()
    return rc, out, ids


def assert_cluster_estimate_matches(got, est: OracleEstimate, oracle_ids, what=""):
    rc, out, ids = got
    assert rc == 0, f"{what}: status {rc}"
    assert ids == oracle_ids, what
    assert list(out["order"]) == list(est.order), f"{what}: PEG order got {list(out['order'])} want {list(est.order)}"
    assert list(out["placed"]) == list(est.placed), f"{what}: placed per PEG\n got {list(out['placed'])}\n want {list(est.placed)}"
    g = (out["node_count"], out["pods_scheduled"], out["nodes_added"], out["limiter_nodes"], out["last_index_out"], out["req_cpu_sum"], out["req_mem_sum"])
    w = (est.node_count, est.pods_scheduled, est.nodes_added, est.limiter_nodes, est.last_index_out, est.req_cpu_sum, est.req_mem_sum)
    assert g == w, f"{what}: (nodes, pods, added, limiter, lastIndex, cpu, mem) got {g} want {w}"


# ---------------------------------------------------------------------------------------------
# batches of independent simulations (casim_groups.peg_lo / peg_hi / sim_offsets)
# ---------------------------------------------------------------------------------------------
def encode_batch(scenarios: Sequence[Scenario]):
    """Several scenarios in ONE encoder (shared dictionaries), each group restricted to its own simulation's PEGs on the
    device.  Returns (encoder, TableSet, [(peg base, group base)])."""
    from kubernetes_autoscaler_amd.tables import TableSet
    enc = Encoder(lanes=scenarios[0].lanes)
    bases, lo, hi, so = [], [], [], [0]
    pb = gb = 0
    for sc in scenarios:
        # (nodes already in the cluster only shift list positions here — the rotation origin of a2 — : their pods, if any, would be shared by the batch)
        assert sc.device_csr and not any(info.pods for info in sc.existing)
        for pg in sc.pegs:
            enc.add_peg(pg)
        for g in sc.groups:
            enc.add_group(g.template, max_nodes=g.max_nodes, existing_nodes=len(sc.existing), last_index=g.last_index, pegs=None)
            lo.append(pb); hi.append(pb + len(sc.pegs))
        bases.append((pb, gb))
        pb += len(sc.pegs); gb += len(sc.groups)
        so.append(gb)
    enc.finalize()
    ts = TableSet.from_encoder(enc)
    ts.peg_lo, ts.peg_hi = np.array(lo, np.int32), np.array(hi, np.int32)
    ts.sim_offsets = np.array(so, np.int32)
    ts.global_id = np.concatenate([np.arange(len(sc.groups), dtype=np.int32) for sc in scenarios])
    return enc, ts, bases


def run_emu_tables(ts, kinds=None, per_sim=True, valid=None, fastpath=False, lds_budget=0, node_pods_capacity=0, generic=False, front=True,
                   winners_only=False, chain=False, narrow_requests=False):
    """Product kernels under the wave emulator on a TableSet.  Returns (BatchResult, expander dict or None)."""
    L = emu_lib()
    if not hasattr(L, "_query_bound"):
        L.emu_estimate_batch_query.restype = C.c_int32
        L.emu_estimate_batch_query.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.Options), C.POINTER(_abi.Results),
                                               C.c_int64, _abi.i32p, _abi.i32p, C.POINTER(_abi.OptionQuery)]
        L._query_bound = True
    pegs, groups = ts.structs(narrow_requests=narrow_requests)
    ng = groups.n_groups
    if ts.peg_offsets is not None:
        nnz_cap = int(ts.peg_offsets[ng])
    elif ts.peg_lo is not None:
        nnz_cap = int((ts.peg_hi - ts.peg_lo).sum())
    else:
        nnz_cap = pegs.n_pegs * ng
    st, arrs = alloc_results(ng, nnz_cap, node_pods_capacity)
    opts = _abi.Options(fastpath=int(fastpath), node_pods=int(node_pods_capacity > 0), force_generic_packer=int(generic), no_front_kernel=int(not front),
                        winners_only=int(winners_only), chain_last_index=int(chain))
    nnz = C.c_int32(0)
    off = np.zeros(ng + 1, np.int32)
    q = exp = None
    if kinds is not None:
        S = ts.n_sims if (per_sim and ts.n_sims) else 1
        ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
        exp = dict(best=np.full(S, -1, np.int32), n_best=np.zeros(S, np.int32), best_set=np.zeros(max(ng, 1), np.uint8),
                   keys=np.zeros((S, 10), np.int64), packed=np.zeros(S, np.int64))
        q = _abi.OptionQuery(kinds=ks, n_kinds=len(kinds), per_sim=int(per_sim), best_out=exp["best"].ctypes.data_as(_abi.i32p),
                             n_best_out=exp["n_best"].ctypes.data_as(_abi.i32p), best_set_out=exp["best_set"].ctypes.data_as(_abi.u8p),
                             key_out=exp["keys"].ctypes.data_as(_abi.i64p), packed_out=exp["packed"].ctypes.data_as(_abi.i64p))
        if valid is not None:
            v = np.ascontiguousarray(valid, np.uint8)
            q.valid = v.ctypes.data_as(_abi.u8p)
    rc = L.emu_estimate_batch_query(C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), int(lds_budget), C.byref(nnz),
                                    off.ctypes.data_as(_abi.i32p), C.byref(q) if q is not None else None)
    assert rc == 0, (rc, L.emu_last_error())
    return finish_results(arrs, ng, int(nnz.value), off), exp


def run_emu_streams(ts, n_streams, kinds=None, valid=None, generic=False, group_id_base=0, winners_only=False, chain=False, narrow_requests=False):
    """The batch cut into sub-batches the way casim_options.n_streams does it (csrc/casim_streams.h).  Returns (BatchResult, expander
    dict or None, parts).  The emulator runs the parts one after the other, or — CASIM_EMU_THREADS=1 — as tasks of the host pool, the way
    the product runs them on the device; a caller that did not choose gets BOTH, compared array by array."""
    args = (ts, n_streams, kinds, valid, generic, group_id_base, winners_only, chain, narrow_requests)
    if os.environ.get("CASIM_EMU_THREADS") is not None:
        return _run_emu_streams(*args)
    first = _run_emu_streams(*args)
    os.environ["CASIM_EMU_THREADS"] = "1"
    try:
        second = _run_emu_streams(*args)
    finally:
        del os.environ["CASIM_EMU_THREADS"]
    (ra, ea, pa), (rb, eb, pb) = first, second
    assert pa == pb, ("parts", pa, pb)
    for f in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum", "offsets", "order", "placed"):
        assert np.array_equal(np.asarray(getattr(ra, f)), np.asarray(getattr(rb, f))), ("parts on the pool differ from parts in turn", f)
    if ea is not None:
        for k in ea:
            assert np.array_equal(ea[k], eb[k]), ("parts on the pool differ from parts in turn", k)
    return first


def _run_emu_streams(ts, n_streams, kinds, valid, generic, group_id_base, winners_only, chain, narrow_requests):
    L = emu_lib()
    if not hasattr(L, "_streams_bound"):
        L.emu_estimate_batch_streams.restype = C.c_int32
        L.emu_estimate_batch_streams.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.Options), C.POINTER(_abi.Results),
                                                 _abi.i32p, _abi.i32p, C.POINTER(_abi.OptionQuery), _abi.i32p]
        L._streams_bound = True
    pegs, groups = ts.structs(narrow_requests=narrow_requests)
    ng = groups.n_groups
    nnz_cap = int((ts.peg_hi - ts.peg_lo).sum()) if ts.peg_lo is not None else (int(ts.peg_offsets[ng]) if ts.peg_offsets is not None else pegs.n_pegs * ng)
    st, arrs = alloc_results(ng, nnz_cap)
    opts = _abi.Options(force_generic_packer=int(generic), n_streams=int(n_streams), winners_only=int(winners_only), chain_last_index=int(chain))
    nnz, parts = C.c_int32(0), C.c_int32(0)
    off = np.zeros(ng + 1, np.int32)
    q = exp = None
    if kinds is not None:
        S = ts.n_sims if ts.n_sims else 1
        ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
        exp = dict(best=np.full(S, -1, np.int32), n_best=np.zeros(S, np.int32), best_set=np.zeros(max(ng, 1), np.uint8),
                   keys=np.zeros((S, 10), np.int64), packed=np.zeros(S, np.int64))
        q = _abi.OptionQuery(kinds=ks, n_kinds=len(kinds), per_sim=1, group_id_base=int(group_id_base), best_out=exp["best"].ctypes.data_as(_abi.i32p),
                             n_best_out=exp["n_best"].ctypes.data_as(_abi.i32p), best_set_out=exp["best_set"].ctypes.data_as(_abi.u8p),
                             key_out=exp["keys"].ctypes.data_as(_abi.i64p), packed_out=exp["packed"].ctypes.data_as(_abi.i64p))
        if valid is not None:
            v = np.ascontiguousarray(valid, np.uint8)
            q.valid = v.ctypes.data_as(_abi.u8p)
    rc = L.emu_estimate_batch_streams(C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), C.byref(nnz), off.ctypes.data_as(_abi.i32p),
                                      C.byref(q) if q is not None else None, C.byref(parts))
    assert rc == 0, (rc, L.emu_last_error())
    return finish_results(arrs, ng, int(nnz.value), off), exp, int(parts.value)


def run_gpu_tables(ts, ctx, kinds=None, per_sim=True, valid=None, fastpath=False, n_streams=0, generic=False, chain=False, narrow_requests=False):
    from kubernetes_autoscaler_amd.engine import Problem
    pegs, groups = ts.structs(narrow_requests=narrow_requests)
    with Problem(ctx, pegs, groups, fastpath, generic, n_streams=n_streams, chain_last_index=chain) as p:
        p.run()
        res = p.fetch()
        exp = p.best_option_sims(kinds, per_sim=per_sim, valid=valid, n_sims=ts.n_sims) if kinds is not None else None
    return res, exp


def run_emu_multi(ts, n_devices, kinds=None, valid=None, use_hook=True, group_id_base=0, expect_rc=0):
    """The batch over n emulated devices (casim_multi.h).  Returns (BatchResult, expander dict or None, info)."""
    L = emu_lib()
    if not hasattr(L, "_multi_bound"):
        L.emu_estimate_batch_multi.restype = C.c_int32
        L.emu_estimate_batch_multi.argtypes = [C.c_int32, C.c_int32, C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.Options),
                                               C.POINTER(_abi.Results), _abi.i32p, C.POINTER(_abi.OptionQuery), _abi.i32p]
        L._multi_bound = True
    pegs, groups = ts.structs()
    ng = groups.n_groups
    if ts.peg_offsets is not None:
        nnz_cap = int(ts.peg_offsets[ng])
    elif ts.peg_lo is not None:
        nnz_cap = int((ts.peg_hi - ts.peg_lo).sum())
    else:
        nnz_cap = pegs.n_pegs * ng
    st, arrs = alloc_results(ng, nnz_cap)
    off = np.zeros(ng + 1, np.int32)
    opts = _abi.Options()
    q = exp = None
    if kinds is not None:
        S = ts.n_sims if ts.n_sims else 1
        ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
        exp = dict(best=np.full(S, -1, np.int32), n_best=np.zeros(S, np.int32), keys=np.zeros((S, 10), np.int64), packed=np.zeros(S, np.int64))
        q = _abi.OptionQuery(kinds=ks, n_kinds=len(kinds), per_sim=1, group_id_base=int(group_id_base), best_out=exp["best"].ctypes.data_as(_abi.i32p),
                             n_best_out=exp["n_best"].ctypes.data_as(_abi.i32p), key_out=exp["keys"].ctypes.data_as(_abi.i64p),
                             packed_out=exp["packed"].ctypes.data_as(_abi.i64p))
        if valid is not None:
            v = np.ascontiguousarray(valid, np.uint8)
            q.valid = v.ctypes.data_as(_abi.u8p)
    info = (C.c_int32 * (1 + n_devices))()
    rc = L.emu_estimate_batch_multi(n_devices, int(use_hook), C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st),
                                    off.ctypes.data_as(_abi.i32p), C.byref(q) if q is not None else None, info)
    assert rc == expect_rc, (rc, L.emu_last_error())
    if rc != 0:
        return None, None, list(info)
    return finish_results(arrs, ng, int(off[ng]), off), exp, list(info)


# ---------------------------------------------------------------------------------------------
# resident cluster (casim_cluster_*): emulator counterpart of engine.ResidentCluster
# ---------------------------------------------------------------------------------------------
class EmuCluster:
    def __init__(self, classes, nodes, lds_budget=0):
        L = self.L = emu_lib()
        if not hasattr(L, "_cluster_bound"):
            L.emu_cluster_create.restype = C.c_void_p
            L.emu_cluster_create.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.c_int64]
            L.emu_cluster_destroy.argtypes = [C.c_void_p]
            L.emu_cluster_update_nodes.argtypes = [C.c_void_p, C.c_int32, _abi.i32p, C.POINTER(_abi.Groups)]
            L.emu_cluster_try_schedule_pods.argtypes = [C.c_void_p, C.POINTER(_abi.PodSequence), C.c_int32, _abi.i32p, _abi.i32p, _abi.i32p]
            L.emu_cluster_simulate_node_removals.argtypes = [C.c_void_p, C.POINTER(_abi.RemovalCandidates), C.POINTER(_abi.RemovalResults)]
            L.emu_cluster_fetch_nodes.argtypes = [C.c_void_p, _abi.i64p, _abi.i32p, _abi.u64p]
            L.emu_cluster_stats.argtypes = [C.c_void_p, _abi.i64p]
            L.emu_cluster_forget_commits.argtypes = [C.c_void_p]
            L._cluster_bound = True
        self.n_nodes, self.n_res, self.w_excl = nodes.n_groups, classes.n_res, classes.w_excl
        self._h = L.emu_cluster_create(C.byref(classes), C.byref(nodes), int(lds_budget))
        assert self._h, L.emu_last_error()

    def close(self):
        if self._h:
            self.L.emu_cluster_destroy(self._h)
            self._h = None

    def try_schedule_pods(self, pod_class, hint_node=None, node_acceptable=None, break_on_failure=False, last_index=0, commit=True,
                          rules=None, similar_key=None):
        from kubernetes_autoscaler_amd.engine import make_pod_sequence
        seq, keep = make_pod_sequence(pod_class, hint_node, node_acceptable, break_on_failure, last_index, rules, similar_key)
        node_out = np.full(max(seq.n_pods, 1), -1, np.int32)
        li, ns = C.c_int32(0), C.c_int32(0)
        rc = self.L.emu_cluster_try_schedule_pods(self._h, C.byref(seq), int(bool(commit)), node_out.ctypes.data_as(_abi.i32p), C.byref(li), C.byref(ns))
        assert rc >= 0, (rc, self.L.emu_last_error())
        del keep
        return rc, node_out[:seq.n_pods], li.value, ns.value

    def simulate_node_removals(self, cand_node, pod_offsets, pod_class, hint_node=None, destination=None, persist=True, max_removable=0,
                               last_index=0, pod_sticky=None, ext_capacity=None, rules=None, cand_atomic=None):
        from kubernetes_autoscaler_amd.engine import alloc_removal_results, finish_removal_results, make_removal_candidates
        st, keep = make_removal_candidates(cand_node, pod_offsets, pod_class, hint_node, destination, persist, max_removable, last_index,
                                           pod_sticky, ext_capacity, rules, cand_atomic)
        res, arrs = alloc_removal_results(st)
        rc = self.L.emu_cluster_simulate_node_removals(self._h, C.byref(st), C.byref(res))
        assert rc >= 0, (rc, self.L.emu_last_error())
        del keep
        return finish_removal_results(rc, st, res, arrs)

    def update_nodes(self, node_index, rows):
        idx = np.ascontiguousarray(node_index, np.int32)
        rc = self.L.emu_cluster_update_nodes(self._h, int(idx.shape[0]), idx.ctypes.data_as(_abi.i32p), C.byref(rows))
        assert rc == 0, (rc, self.L.emu_last_error())

    def fetch_nodes(self):
        req = np.zeros((max(self.n_nodes, 1), self.n_res), np.int64)
        pods = np.zeros(max(self.n_nodes, 1), np.int32)
        excl = np.zeros((max(self.n_nodes, 1), max(self.w_excl, 1)), np.uint64)
        self.L.emu_cluster_fetch_nodes(self._h, req.ctypes.data_as(_abi.i64p), pods.ctypes.data_as(_abi.i32p), excl.ctypes.data_as(_abi.u64p))
        return req[:self.n_nodes], pods[:self.n_nodes], excl[:self.n_nodes, :self.w_excl]

    def forget_commits(self):
        assert self.L.emu_cluster_forget_commits(self._h) == 0

    def stats(self):
        out = (C.c_int64 * 4)()
        self.L.emu_cluster_stats(self._h, out)
        return {"full_uploads": out[0], "delta_rows": out[1], "commits": out[2], "nodes": out[3]}


def resident_iteration(make_cluster, w, n_candidates=4, with_rules=False):
    """One RunOnce-shaped sequence on a resident cluster vs the oracle, which threads ONE snapshot through it the way the
    reference does: filter-out-schedulable (committed) -> the same pods again (nothing may fit twice the same way; reverted)
    -> the planner's removal loop over the emptiest nodes, whose pod lists include what filter-out-schedulable just placed.
    `w` is a workloads.PendingWorkload; with_rules: every call carries the encoder's domain rules (PodTopologySpread, zone
    anti-affinity, pod affinity) — counters from BEFORE the commit, which the cluster has to bring up to date itself.
    Returns what was compared, for the caller's bookkeeping."""
    from kubernetes_autoscaler_amd.objects import PodEquivalenceGroup as PEG
    nodes, pods = w.nodes, w.pods
    # ---- one encoder session for the iteration: classes = pending specs + the specs of every running pod
    enc = Encoder(explicit_self_exclusion=True)
    class_of = {}

    def cls(p):
        k = p.spec_key()
        if k not in class_of:
            class_of[k] = enc.add_peg(PEG(pods=[p]))
        return class_of[k]
    pod_class = np.array([cls(p) for p in pods], np.int32)
    for info in nodes:
        for p in info.pods:
            cls(p)
    for info in nodes:
        enc.add_group(info, pegs=[])
    enc.finalize()
    cl = make_cluster(enc.pegs, enc.groups)
    rules = enc.rules if with_rules else None
    sk = similar_keys(pods) if with_rules else None
    # ---- oracle: one snapshot for the whole sequence
    s = OracleScenario()
    for info in nodes:
        s.add_existing(info)
    canon = {}
    opods = [canon.setdefault(p.spec_key(), p) for p in pods]
    want1 = s.try_schedule_pods(opods, w.hints, sk, w.acceptable, w.break_on_failure, w.last_index)
    # ---- 1. filter-out-schedulable, committed
    before = cl.fetch_nodes()
    rc, out1, li1, ns1 = cl.try_schedule_pods(pod_class, w.hints, w.acceptable, w.break_on_failure, w.last_index, commit=True, rules=rules, similar_key=sk)
    assert rc == 0
    assert list(out1) == list(want1[0]) and li1 == want1[1] and ns1 == want1[2], "committed pass"
    after = cl.fetch_nodes()
    placed_on = [[] for _ in nodes]
    for i, m in enumerate(out1):
        if m >= 0:
            placed_on[int(m)].append(pods[i])
    for m, info in enumerate(nodes):   # the image holds exactly what the oracle's snapshot holds now
        add = [0] * enc.pegs.n_res
        for p in placed_on[m]:
            add[0] += p.requests.get("cpu", 0); add[1] += p.requests.get("memory", 0)
        assert list(after[0][m] - before[0][m]) == add and int(after[1][m] - before[1][m]) == len(placed_on[m]), ("image", m)
    # ---- 2. the same pods once more, reverted: forks from the COMMITTED image (the oracle commits, so give it a throw-away copy)
    s2 = OracleScenario()
    for m, info in enumerate(nodes):
        from kubernetes_autoscaler_amd.objects import NodeInfo
        s2.add_existing(NodeInfo(info.node, list(info.pods) + placed_on[m]))
    want2 = s2.try_schedule_pods(opods, w.hints, sk, w.acceptable, w.break_on_failure, li1)
    s2.close()
    rc, out2, li2, ns2 = cl.try_schedule_pods(pod_class, w.hints, w.acceptable, w.break_on_failure, li1, commit=False, rules=rules, similar_key=sk)
    assert list(out2) == list(want2[0]) and li2 == want2[1] and ns2 == want2[2], "reverted pass"
    again = cl.fetch_nodes()
    assert all((a == b).all() for a, b in zip(after, again)), "a reverted pass must not touch the image"
    # ---- 3. removal loop on the committed image: the emptiest nodes, their pods = running + just placed (arrival order)
    load = [len(info.pods) + len(placed_on[m]) for m, info in enumerate(nodes)]
    cands = sorted(range(len(nodes)), key=lambda m: (load[m], m))[:n_candidates]
    lists = [[p for p in nodes[c].pods if not p.daemonset] + placed_on[c] for c in cands]
    want3 = s.simulate_node_removals(cands, lists, None, None, True, 0, None, None, li1, None)
    flat = np.array([class_of[p.spec_key()] for lst in lists for p in lst], np.int32)
    off = np.cumsum([0] + [len(x) for x in lists]).astype(np.int32)
    got3 = cl.simulate_node_removals(cands, off, flat, last_index=li1, rules=rules)
    assert_removal_matches(got3, want3, "removals on the committed image")
    final = cl.fetch_nodes()
    assert all((a == b).all() for a, b in zip(after, final)), "the removal loop must not touch the image"
    st = cl.stats()
    s.close(); cl.close(); enc.close()
    return {"scheduled": int(ns1), "removable": int((got3.removable == 1).sum()), "stats": st}


def mixed_list_simulations():
    """Four simulations whose PEG lists are 40, 300, 700 and 120 long (three node groups each, limits 4 / 7 / none): lists
    beyond the orderer's one-wave networks, several record chunks per group in the packer, most PEGs behind a dry limiter."""
    from kubernetes_autoscaler_amd.objects import GiB, MiB, Node, NodeInfo, Pod, PodEquivalenceGroup

    def tmpl(cpu, mem, pods=110):
        cap = {"cpu": cpu, "memory": mem, "pods": pods}
        return NodeInfo(Node(name=f"t{cpu}", labels={}, allocatable=dict(cap), capacity=dict(cap)))

    def sim(n_pegs, salt):
        pegs = [PodEquivalenceGroup(pods=[Pod(name="p", requests={"cpu": 50 + ((i * 131 + salt) % 1500), "memory": (1 + (i * 7 + salt) % 40) * 100 * MiB})] * (1 + (i + salt) % 5))
                for i in range(n_pegs)]
        return Scenario(pegs=pegs, groups=[GroupSpec(tmpl(4000, 16 * GiB), 4), GroupSpec(tmpl(16000, 64 * GiB), 7), GroupSpec(tmpl(2000, 4 * GiB), 0)],
                        device_csr=True)
    return [sim(40, 1), sim(300, 2), sim(700, 3), sim(120, 4)]
