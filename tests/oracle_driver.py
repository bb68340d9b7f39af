"""ctypes driver of the CPU oracle (oracle/libcasim_oracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("CASIM_ORACLE_LIB") or os.path.join(ROOT, "oracle", "libcasim_oracle.so")   # override: sanitizer builds (tests/tools/sanitize_cpu.sh)
MAX_RES = 8

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)
f64p = C.POINTER(C.c_double)
cstr = C.c_char_p
cstrp = C.POINTER(C.c_char_p)


class EstimateResult(C.Structure):
    _fields_ = [
        ("node_count", C.c_int32), ("pods_scheduled", C.c_int32), ("nodes_added", C.c_int32),
        ("limiter_nodes", C.c_int32), ("last_index_out", C.c_int32), ("internal_error", C.c_int32),
        ("req_cpu_sum", C.c_int64), ("req_mem_sum", C.c_int64), ("filter_runs", C.c_int64),
        ("order", i32p), ("placed", i32p), ("node_pods", i32p), ("node_pods_cap", C.c_int32),
    ]


class Limiter(C.Structure):
    _fields_ = [("max_nodes", C.c_int), ("nodes", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            import subprocess
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        L = C.CDLL(LIB)
        P = C.c_void_p
        sigs = {
            "orc_new": (P, [C.c_int]), "orc_free": (None, [P]),
            "orc_pod": (C.c_int, [P, cstr, i64p]), "orc_pod_label": (C.c_int, [P, C.c_int, cstr, cstr]),
            "orc_pod_toleration": (C.c_int, [P, C.c_int, cstr, cstr, cstr, cstr]),
            "orc_pod_node_selector": (C.c_int, [P, C.c_int, cstr, cstr]),
            "orc_pod_node_affinity_req": (C.c_int, [P, C.c_int, cstr, cstr, cstrp, C.c_int]),
            "orc_pod_node_affinity_term": (C.c_int, [P, C.c_int]),
            "orc_namespace_label": (C.c_int, [P, cstr, cstr, cstr]),
            "orc_term_namespace_selector": (C.c_int, [P, C.c_int, C.c_int]),
            "orc_aff_term_namespace_selector": (C.c_int, [P, C.c_int, C.c_int]),
            "orc_term_namespace_requirement": (C.c_int, [P, C.c_int, C.c_int, cstr, cstr, cstrp, C.c_int]),
            "orc_pod_node_term_req": (C.c_int, [P, C.c_int, C.c_int, C.c_int, cstr, cstr, cstrp, C.c_int]),
            "orc_pod_host_port": (C.c_int, [P, C.c_int, cstr, cstr, C.c_int]),
            "orc_pod_anti_affinity_term": (C.c_int, [P, C.c_int, cstr, cstrp, C.c_int]),
            "orc_term_requirement": (C.c_int, [P, C.c_int, C.c_int, cstr, cstr, cstrp, C.c_int]),
            "orc_pod_affinity_term": (C.c_int, [P, C.c_int, cstr, cstrp, C.c_int]),
            "orc_aff_term_requirement": (C.c_int, [P, C.c_int, C.c_int, cstr, cstr, cstrp, C.c_int]),
            "orc_aff_term_namespace_requirement": (C.c_int, [P, C.c_int, C.c_int, cstr, cstr, cstrp, C.c_int]),
            "orc_pod_fastpath_requests": (C.c_int, [P, C.c_int, C.c_double, C.c_double]),
            "orc_pod_has_topology_spread": (C.c_int, [P, C.c_int, C.c_int]),
            "orc_pod_spread_constraint": (C.c_int, [P, C.c_int, C.c_int, cstr, C.c_int]),
            "orc_spread_requirement": (C.c_int, [P, C.c_int, C.c_int, cstr, cstr, cstrp, C.c_int]),
            "orc_spread_taints_policy_honor": (C.c_int, [P, C.c_int, C.c_int, C.c_int]),
            "orc_spread_affinity_policy_ignore": (C.c_int, [P, C.c_int, C.c_int, C.c_int]),
            "orc_node": (C.c_int, [P, cstr, i64p, C.c_int, C.c_int64, C.c_int64, C.c_int]),
            "orc_node_fastpath_capacity": (C.c_int, [P, C.c_int, C.c_double, C.c_double]),
            "orc_node_label": (C.c_int, [P, C.c_int, cstr, cstr]),
            "orc_node_taint": (C.c_int, [P, C.c_int, cstr, cstr, cstr]),
            "orc_node_add_pod": (C.c_int, [P, C.c_int, C.c_int]),
            "orc_snapshot_add": (C.c_int, [P, C.c_int]),
            "orc_set_taint_comparison_ops": (None, [P, C.c_int]),
            "orc_set_list_shuffle": (None, [P, C.c_uint64]),
            "orc_estimate": (C.c_int, [P, C.c_int, C.c_int, i32p, i32p, C.c_int, C.c_int, C.c_int, C.POINTER(EstimateResult)]),
            "orc_last_fit_reasons": (C.c_uint, []),
            "orc_scale_up_simulation": (C.c_int, [P, C.c_int, i32p, C.c_int, i32p, i32p, i32p, i32p, C.POINTER(EstimateResult), i32p, i32p,
                                                  i32p, i32p, i64p]),
            "orc_scale_up_simulation_chained": (C.c_int, [P, C.c_int, i32p, C.c_int, i32p, i32p, i32p, i32p, C.POINTER(EstimateResult), i32p, i32p,
                                                          i32p, i32p, i64p]),
            "orc_check_predicates": (C.c_int, [P, C.c_int, C.c_int, cstrp, cstrp]),
            "orc_run_filters_on_snapshot_node": (C.c_int, [P, C.c_int, C.c_int, cstrp, cstrp]),
            "orc_run_filters_until_passing": (C.c_int, [P, C.c_int, C.POINTER(C.c_int)]),
            "orc_run_filters_until_passing_ordered": (C.c_int, [P, C.c_int, i32p, C.c_int, u8p, i32p, C.POINTER(C.c_int)]),
            "orc_try_schedule_pods": (C.c_int, [P, C.c_int, i32p, i32p, i32p, u8p, C.c_int, C.POINTER(C.c_int), i32p]),
            "orc_snapshot_size": (C.c_int, [P]),
            "orc_simulate_node_removals": (C.c_int, [P, C.c_int, i32p, i32p, i32p, i32p, u8p, u8p, u8p, C.c_int, C.c_int, C.c_int,
                                                    C.POINTER(C.c_int), u8p, i32p, i32p, i32p, i32p, C.POINTER(C.c_int), i32p,
                                                    C.POINTER(C.c_int)]),
            "orc_get_min_limit": (C.c_int64, [C.c_int64, C.c_int64]),
            "orc_sng_capacity_limit": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "orc_cluster_capacity_limit": (C.c_int, [C.c_int, C.c_int, C.c_int]),
            "orc_limiter_start": (None, [C.POINTER(Limiter), C.c_int, C.POINTER(C.c_int)]),
            "orc_limiter_permission": (C.c_int, [C.POINTER(Limiter)]),
            "orc_last_index_at": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
            "orc_pod_score": (C.c_double, [C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
            "orc_order": (None, [C.c_int, i64p, i64p, u8p, C.c_int64, C.c_int64, i32p]),
            "orc_best_fastpath_peg": (C.c_int, [C.c_int, i32p, f64p, f64p, u8p, u8p, u8p, C.c_double, C.c_double]),
            "orc_least_nodes": (C.c_int, [C.c_int, i32p, u8p]),
            "orc_most_pods": (C.c_int, [C.c_int, i32p, u8p]),
            "orc_least_waste": (C.c_int, [C.c_int, i32p, i64p, i64p, i64p, i64p, u8p, u8p]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _b(s):
    return (s or "").encode("utf-8")


def _strs(values):
    arr = (C.c_char_p * max(len(values), 1))()
    for i, v in enumerate(values):
        arr[i] = _b(v)
    return arr


@dataclass
class OracleEstimate:
    node_count: int
    pods_scheduled: int
    nodes_added: int
    limiter_nodes: int
    last_index_out: int
    req_cpu_sum: int
    req_mem_sum: int
    filter_runs: int
    order: np.ndarray      # input PEG index processed k-th
    placed: np.ndarray     # pods scheduled of that PEG
    node_pods: np.ndarray  # pods per simulated node


class OracleScenario:
    """A cluster snapshot + pod specs + node templates inside the oracle."""

    def __init__(self, lanes: Sequence[str] = ("cpu", "memory"), taint_comparison_ops: bool = False, list_shuffle_seed: int = 0):
        self.L = lib()
        self.lanes = tuple(lanes)
        self.h = self.L.orc_new(len(self.lanes))
        assert self.h
        if taint_comparison_ops:
            self.L.orc_set_taint_comparison_ops(self.h, 1)
        if list_shuffle_seed:
            self.L.orc_set_list_shuffle(self.h, list_shuffle_seed)
        self._pod_ids: Dict[int, int] = {}
        self._keep: List[object] = []
        from kubernetes_autoscaler_amd import objects
        for ns, labels in objects.NAMESPACE_LISTER.items():
            self.L.orc_namespace_label(self.h, _b(ns), None, None)
            for k, v in labels.items():
                self.L.orc_namespace_label(self.h, _b(ns), _b(k), _b(v))

    def close(self):
        if self.h:
            self.L.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _vec(self, res):
        v = (C.c_int64 * MAX_RES)()
        for i, name in enumerate(self.lanes):
            v[i] = int(res.get(name, 0))
        return v

    def pod(self, pod) -> int:
        key = id(pod)
        if key in self._pod_ids:
            return self._pod_ids[key]
        self._keep.append(pod)
        L, h = self.L, self.h
        p = L.orc_pod(h, _b(pod.namespace), self._vec(pod.requests))
        assert p >= 0
        for k, v in pod.labels.items():
            L.orc_pod_label(h, p, _b(k), _b(v))
        for t in pod.tolerations:
            L.orc_pod_toleration(h, p, _b(t.key), _b(t.operator), _b(t.value), _b(t.effect))
        for k, v in pod.node_selector.items():
            L.orc_pod_node_selector(h, p, _b(k), _b(v))
        for r in pod.node_affinity:
            L.orc_pod_node_affinity_req(h, p, _b(r.key), _b(r.operator), _strs(r.values), len(r.values))
        from kubernetes_autoscaler_amd.objects import NodeSelectorTerm
        for term in (pod.node_affinity_terms or [NodeSelectorTerm()] if pod.node_affinity_terms is not None else []):
            t = L.orc_pod_node_affinity_term(h, p)
            assert t >= 0
            for is_field, reqs in ((0, term.match_expressions), (1, term.match_fields)):
                for r in reqs:
                    assert L.orc_pod_node_term_req(h, p, t, is_field, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)) == 0
        for hp in pod.host_ports:
            L.orc_pod_host_port(h, p, _b(hp.host_ip), _b(hp.protocol), int(hp.host_port))
        for term in pod.anti_affinity:
            t = L.orc_pod_anti_affinity_term(h, p, _b(term.topology_key), _strs(term.namespaces), len(term.namespaces))
            for r in term.requirements():
                L.orc_term_requirement(h, p, t, _b(r.key), _b(r.operator), _strs(r.values), len(r.values))
            if term.namespace_selector is not None:
                assert L.orc_term_namespace_selector(h, p, t) == 0
                for r in term.namespace_selector:
                    assert L.orc_term_namespace_requirement(h, p, t, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)) == 0
        for term in getattr(pod, "affinity", ()):
            t = L.orc_pod_affinity_term(h, p, _b(term.topology_key), _strs(term.namespaces), len(term.namespaces))
            for r in term.requirements():
                assert L.orc_aff_term_requirement(h, p, t, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)) == 0
            if term.namespace_selector is not None:
                assert L.orc_aff_term_namespace_selector(h, p, t) == 0
                for r in term.namespace_selector:
                    assert L.orc_aff_term_namespace_requirement(h, p, t, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)) == 0
        if pod.has_containers:
            cpu, mem = pod.fastpath_requests()
            L.orc_pod_fastpath_requests(h, p, cpu, mem)
        if pod.topology_spread:
            L.orc_pod_has_topology_spread(h, p, 1)
        for sc in pod.spread_constraints:
            c = L.orc_pod_spread_constraint(h, p, int(sc.max_skew), _b(sc.topology_key), int(sc.min_domains))
            for k, v in sc.effective_match_labels(pod.labels).items():
                L.orc_spread_requirement(h, p, c, _b(k), _b("In"), _strs([v]), 1)
            if sc.node_taints_policy == "Honor":
                L.orc_spread_taints_policy_honor(h, p, c, 1)
            if sc.node_affinity_policy == "Ignore":
                L.orc_spread_affinity_policy_ignore(h, p, c, 1)
        self._pod_ids[key] = p
        return p

    def node(self, info) -> int:
        """info: NodeInfo (node + preloaded pods)"""
        L, h = self.L, self.h
        nd = info.node
        n = L.orc_node(h, _b(nd.name), self._vec(nd.allocatable), nd.allowed_pods(), int(nd.capacity.get("cpu", 0)),
                       int(nd.capacity.get("memory", 0)), int(nd.unschedulable))
        assert n >= 0
        for k, v in nd.labels.items():
            L.orc_node_label(h, n, _b(k), _b(v))
        for t in nd.taints:
            L.orc_node_taint(h, n, _b(t.key), _b(t.value), _b(t.effect))
        for p in info.pods:
            L.orc_node_add_pod(h, n, self.pod(p))
        return n

    def add_existing(self, info) -> int:
        """AddNodeInfo into the cluster snapshot (list order = insertion order)."""
        return self.L.orc_snapshot_add(self.h, self.node(info))

    def estimate(self, template_node: int, pegs, max_nodes: int = 0, last_index: int = 0, fastpath: bool = False,
                 node_pods_cap: int = 0) -> OracleEstimate:
        n = len(pegs)
        pod_ids = (C.c_int32 * max(n, 1))()
        counts = (C.c_int32 * max(n, 1))()
        from kubernetes_autoscaler_amd.objects import Pod
        for i, g in enumerate(pegs):
            ex = g.exemplar()
            if ex is None:
                ex = Pod(name="<empty>")
                self._keep.append(ex)
            pod_ids[i] = self.pod(ex)
            counts[i] = len(g.pods)
        order = np.zeros(max(n, 1), np.int32)
        placed = np.zeros(max(n, 1), np.int32)
        cap = node_pods_cap or 1
        node_pods = np.zeros(cap, np.int32)
        res = EstimateResult(order=order.ctypes.data_as(i32p), placed=placed.ctypes.data_as(i32p),
                             node_pods=node_pods.ctypes.data_as(i32p) if node_pods_cap else None, node_pods_cap=node_pods_cap)
        rc = self.L.orc_estimate(self.h, template_node, n, pod_ids, counts, int(max_nodes), int(last_index), int(fastpath), C.byref(res))
        assert rc == 0, rc
        return OracleEstimate(res.node_count, res.pods_scheduled, res.nodes_added, res.limiter_nodes, res.last_index_out,
                              res.req_cpu_sum, res.req_mem_sum, res.filter_runs, order[:n].copy(), placed[:n].copy(),
                              node_pods[:min(res.nodes_added, node_pods_cap)].copy())

    def prepare_simulation(self, template_nodes, pegs, max_nodes, last_index, chain=False):
        """Argument block of orc_scale_up_simulation, built once; returns run() -> ([(OracleEstimate, schedulable PEG ids)], filter runs).
        run() is ONE native call: the whole node-group loop (SchedulablePodGroups + Estimate per group)."""
        from kubernetes_autoscaler_amd.objects import Pod
        ng, n = len(template_nodes), len(pegs)
        tn = np.ascontiguousarray(template_nodes, np.int32)
        pod_ids, counts = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        for i, g in enumerate(pegs):
            ex = g.exemplar()
            if ex is None:
                ex = Pod(name="<empty>")
                self._keep.append(ex)
            pod_ids[i] = self.pod(ex)
            counts[i] = len(g.pods)
        mx, li = np.ascontiguousarray(max_nodes, np.int32), np.ascontiguousarray(last_index, np.int32)
        res = (EstimateResult * max(ng, 1))()
        nsched = np.zeros(max(ng, 1), np.int32)
        sched, order, placed = (np.zeros(max(ng * n, 1), np.int32) for _ in range(3))
        runs = C.c_int64(0)
        p = lambda a: a.ctypes.data_as(i32p)

        def run(collect=True):
            rc = (self.L.orc_scale_up_simulation_chained if chain else self.L.orc_scale_up_simulation)(self.h, ng, p(tn), n, p(pod_ids), p(counts), p(mx), p(li), res, p(nsched), p(sched),
                                                p(order), p(placed), C.byref(runs))
            assert rc == 0, rc
            if not collect:
                return None, runs.value
            out = []
            for i in range(ng):
                k, r = int(nsched[i]), res[i]
                out.append((OracleEstimate(r.node_count, r.pods_scheduled, r.nodes_added, r.limiter_nodes, r.last_index_out, r.req_cpu_sum,
                                           r.req_mem_sum, r.filter_runs, order[i * n:i * n + k].copy(), placed[i * n:i * n + k].copy(),
                                           np.zeros(0, np.int32)), [int(x) for x in sched[i * n:i * n + k]]))
            return out, runs.value
        return run

    def check_predicates(self, template_node: int, pod):
        plug, reason = C.c_char_p(), C.c_char_p()
        ok = self.L.orc_check_predicates(self.h, template_node, self.pod(pod), C.byref(plug), C.byref(reason))
        return bool(ok), (plug.value or b"").decode(), (reason.value or b"").decode()

    def check_predicates_code(self, template_node: int, pod, lanes):
        """CheckPredicates as the uint16 code of casim_feasibility_reasons (plugin id + NodeResourcesFit reasons)."""
        ok, plugin, reason = self.check_predicates(template_node, pod)
        if ok:
            return 0
        ids = {"NodeUnschedulable": 2, "TaintToleration": 3, "NodePorts": 5, "NodeResourcesFit": 6, "PodTopologySpread": 7, "InterPodAffinity": 8}
        if plugin == "NodeAffinity":
            return 1 if reason == "PreFilter filtered the Node out" else 4
        code = ids[plugin]
        if plugin == "NodeResourcesFit":
            m = self.L.orc_last_fit_reasons()
            code |= (0x10 if m & 1 else 0) | ((m >> 1) << 5)
        return code

    def run_filters_on_node(self, snapshot_index: int, pod):
        plug, reason = C.c_char_p(), C.c_char_p()
        ok = self.L.orc_run_filters_on_snapshot_node(self.h, snapshot_index, self.pod(pod), C.byref(plug), C.byref(reason))
        return bool(ok), (plug.value or b"").decode(), (reason.value or b"").decode()

    def run_filters_until_passing(self, pod, last_index: int = 0):
        li = C.c_int(last_index)
        idx = self.L.orc_run_filters_until_passing(self.h, self.pod(pod), C.byref(li))
        return idx, li.value

    def run_filters_until_passing_ordered(self, pod, order, acceptable):
        """RunFiltersUntilPassingNode under a NodeOrderMapping given as the list of snapshot indices it yields (shorter than the
        snapshot = the mapping answers -1 from there on); acceptable[idx] = IsNodeAcceptable.  Returns (index found or -1, visited)."""
        order = np.ascontiguousarray(order, np.int32)
        acc = np.ascontiguousarray(acceptable, np.uint8)
        visited = np.full(max(len(acc), 1), -1, np.int32)
        nv = C.c_int(0)
        idx = self.L.orc_run_filters_until_passing_ordered(self.h, self.pod(pod), order.ctypes.data_as(i32p), len(order), acc.ctypes.data_as(u8p),
                                                           visited.ctypes.data_as(i32p), C.byref(nv))
        return idx, [int(x) for x in visited[:nv.value]]

    def try_schedule_pods(self, pods, hints=None, similar_keys=None, acceptable=None, break_on_failure: bool = False,
                          last_index: int = 0):
        """HintingSimulator.TrySchedulePods against the snapshot built with add_existing(); pods are committed.
        Returns (node_out[P], last_index, n_scheduled)."""
        n = len(pods)
        ids = np.array([self.pod(p) for p in pods], np.int32) if n else np.zeros(1, np.int32)
        hn = np.full(max(n, 1), -1, np.int32) if hints is None else np.ascontiguousarray(hints, np.int32)
        sk = np.full(max(n, 1), -1, np.int32) if similar_keys is None else np.ascontiguousarray(similar_keys, np.int32)
        acc = None if acceptable is None else np.ascontiguousarray(acceptable, np.uint8)
        out = np.full(max(n, 1), -1, np.int32)
        li = C.c_int(last_index)
        import time
        t0 = time.perf_counter()
        ns = self.L.orc_try_schedule_pods(self.h, n, ids.ctypes.data_as(i32p), hn.ctypes.data_as(i32p), sk.ctypes.data_as(i32p),
                                          acc.ctypes.data_as(u8p) if acc is not None else None, int(break_on_failure), C.byref(li),
                                          out.ctypes.data_as(i32p))
        self.last_native_s = time.perf_counter() - t0   # (the native call alone: what bench.py's oracle_ms reports)
        assert ns >= 0, ns
        return out[:n].copy(), li.value, ns

    def simulate_node_removals(self, cand_node, pod_lists, hints=None, destination=None, persist=True, max_removable=0,
                               pod_sticky=None, ext_capacity=None, last_index=0, cand_atomic=None):
        """Planner loop around SimulateNodeRemoval on the snapshot built with add_existing().  pod_lists[k] = pods to move
        of candidate k (Pod objects).  Returns a dict: removable[K], node_out[total], ext (list of (candidate, pod, node)),
        final[total], last_index, n_processed."""
        K = len(cand_node)
        off = np.zeros(K + 1, np.int32)
        flat = []
        for k, lst in enumerate(pod_lists):
            flat.extend(self.pod(p) for p in lst)
            off[k + 1] = len(flat)
        total = len(flat)
        E = 2 * total + 64 if ext_capacity is None else int(ext_capacity)
        cn = np.ascontiguousarray(cand_node, np.int32) if K else np.zeros(1, np.int32)
        pods = np.array(flat, np.int32) if total else np.zeros(1, np.int32)
        hn = None if hints is None else np.ascontiguousarray(hints, np.int32)
        ds = None if destination is None else np.ascontiguousarray(destination, np.uint8)
        sk = None if pod_sticky is None else np.ascontiguousarray(pod_sticky, np.uint8)
        at = None if cand_atomic is None else np.ascontiguousarray(cand_atomic, np.uint8)
        removable = np.full(max(K, 1), 2, np.uint8)
        node_out = np.full(max(total, 1), -1, np.int32)
        final = np.full(max(total, 1), -1, np.int32)
        ec, ep, en = (np.full(max(E, 1) + total + 1, -1, np.int32) for _ in range(3))
        li, npr, ne = C.c_int(last_index), C.c_int(0), C.c_int(0)
        import time
        t0 = time.perf_counter()
        rc = self.L.orc_simulate_node_removals(self.h, K, cn.ctypes.data_as(i32p), off.ctypes.data_as(i32p), pods.ctypes.data_as(i32p),
                                               hn.ctypes.data_as(i32p) if hn is not None and hn.size else None,
                                               ds.ctypes.data_as(u8p) if ds is not None and ds.size else None,
                                               sk.ctypes.data_as(u8p) if sk is not None and sk.size else None,
                                               at.ctypes.data_as(u8p) if at is not None and at.size else None, int(persist),
                                               int(max_removable), E, C.byref(li), removable.ctypes.data_as(u8p),
                                               node_out.ctypes.data_as(i32p), ec.ctypes.data_as(i32p), ep.ctypes.data_as(i32p),
                                               en.ctypes.data_as(i32p), C.byref(ne), final.ctypes.data_as(i32p), C.byref(npr))
        self.last_native_s = time.perf_counter() - t0
        assert rc >= 0, rc
        n = ne.value
        return dict(removable=removable[:K].copy(), node_out=node_out[:total].copy(),
                    ext=list(zip(ec[:n].tolist(), ep[:n].tolist(), en[:n].tolist())), final=final[:total].copy(),
                    last_index=li.value, n_processed=npr.value)
