"""Python binding of the ENGINE half of include/casim.h: contexts, resident problems, results.

Every call goes through the C ABI into the HIP kernels of libcasim.so.  Nothing here computes a
result on the host; without an MI355X `Context()` raises NoDeviceError."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _abi
from ._ffi import CasimError, NoDeviceError, check, last_error, lib  # noqa: F401


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


@dataclass
class BatchResult:
    """Results of one batch: one entry per node group (= one Estimate() call)."""
    offsets: np.ndarray          # [NG+1] CSR offsets of order/placed
    node_count: np.ndarray       # len(newNodesWithPods)
    pods_scheduled: np.ndarray
    nodes_added: np.ndarray
    limiter_nodes: np.ndarray
    last_index_out: np.ndarray
    status: np.ndarray
    req_cpu_sum: np.ndarray
    req_mem_sum: np.ndarray
    order: np.ndarray            # [nnz] PEG id processed k-th within its group
    placed: np.ndarray           # [nnz] pods of that PEG that were scheduled (a prefix)
    node_pods: Optional[np.ndarray] = None          # pods per added node (Problem(..., node_pods=True)), compact
    node_pods_offsets: Optional[np.ndarray] = None  # [NG+1]
    winner_offsets: Optional[np.ndarray] = None     # winners_only calls: order / placed hold the winners' lists only, list s at [w[s], w[s+1])

    def nodes_with_pods(self, i: int, template_name: str):
        """newNodesWithPods of group i as the reference names them: '<template>-e-<j>' for every added node holding a pod
        (binpacking_estimator.go:326-342, :58-61)."""
        a, b = int(self.node_pods_offsets[i]), int(self.node_pods_offsets[i + 1])
        return [f"{template_name}-e-{j}" for j, n in enumerate(self.node_pods[a:b]) if n > 0]

    def group(self, i: int):
        a, b = int(self.offsets[i]), int(self.offsets[i + 1])
        return self.order[a:b], self.placed[a:b]


def alloc_results(n_groups: int, nnz: int, node_pods_capacity: int = 0, pinned_lists: bool = False):
    """pinned_lists: `order` / `placed` in page-locked memory (casim_host_alloc) — the D2H copy of a big batch then goes straight into
    them at link speed instead of through the runtime's bounce buffers."""
    ng = max(n_groups, 1)
    big = (lambda n: pinned_copy(np.zeros(n, np.int32))) if pinned_lists else (lambda n: np.zeros(n, np.int32))
    arrs = dict(
        node_count=np.zeros(ng, np.int32), pods_scheduled=np.zeros(ng, np.int32), nodes_added=np.zeros(ng, np.int32),
        limiter_nodes=np.zeros(ng, np.int32), last_index_out=np.zeros(ng, np.int32), status=np.zeros(ng, np.int32),
        req_cpu_sum=np.zeros(ng, np.int64), req_mem_sum=np.zeros(ng, np.int64),
        order=big(max(nnz, 1)), placed=big(max(nnz, 1)))
    st = _abi.Results(
        node_count=_ptr(arrs["node_count"], C.c_int32), pods_scheduled=_ptr(arrs["pods_scheduled"], C.c_int32),
        nodes_added=_ptr(arrs["nodes_added"], C.c_int32), limiter_nodes=_ptr(arrs["limiter_nodes"], C.c_int32),
        last_index_out=_ptr(arrs["last_index_out"], C.c_int32), status=_ptr(arrs["status"], C.c_int32),
        req_cpu_sum=_ptr(arrs["req_cpu_sum"], C.c_int64), req_mem_sum=_ptr(arrs["req_mem_sum"], C.c_int64),
        order=_ptr(arrs["order"], C.c_int32), placed=_ptr(arrs["placed"], C.c_int32))
    if node_pods_capacity > 0:
        arrs["node_pods"] = np.zeros(node_pods_capacity, np.int32)
        arrs["node_pods_offsets"] = np.zeros(ng + 1, np.int32)
        st.node_pods = _ptr(arrs["node_pods"], C.c_int32); st.node_pods_offsets = _ptr(arrs["node_pods_offsets"], C.c_int32)
        st.node_pods_capacity = node_pods_capacity
    return st, arrs


def finish_results(arrs, n_groups: int, nnz: int, offsets: np.ndarray) -> BatchResult:
    out = {k: (v[:nnz] if k in ("order", "placed") else v[:n_groups]) for k, v in arrs.items() if not k.startswith("node_pods")}
    if "node_pods" in arrs:
        out["node_pods_offsets"] = arrs["node_pods_offsets"][:n_groups + 1]
        out["node_pods"] = arrs["node_pods"][:int(arrs["node_pods_offsets"][n_groups])]
    return BatchResult(offsets=offsets, **out)


def make_pod_sequence(pod_class, hint_node=None, node_acceptable=None, break_on_failure: bool = False, last_index: int = 0,
                      rules=None, similar_key=None):
    """casim_pod_sequence over numpy arrays; returns (struct, arrays to keep alive)."""
    pc = np.ascontiguousarray(pod_class, np.int32)
    hn = None if hint_node is None else np.ascontiguousarray(hint_node, np.int32)
    na = None if node_acceptable is None else np.ascontiguousarray(node_acceptable, np.uint8)
    if hn is not None and hn.shape != pc.shape:
        raise ValueError("hint_node must have one entry per pod")
    seq = _abi.PodSequence(n_pods=int(pc.shape[0]), pod_class=_ptr(pc, C.c_int32) if pc.size else None,
                           hint_node=_ptr(hn, C.c_int32) if hn is not None and hn.size else None,
                           node_acceptable=_ptr(na, C.c_uint8) if na is not None and na.size else None,
                           break_on_failure=int(bool(break_on_failure)), last_index=int(last_index),
                           rules=C.pointer(rules) if rules is not None and rules.n_rules > 0 else None)
    sk = None if similar_key is None else np.ascontiguousarray(similar_key, np.int32)
    if sk is not None and sk.size:
        if sk.shape != pc.shape:
            raise ValueError("similar_key must have one entry per pod")
        seq.similar_key = _ptr(sk, C.c_int32)
    return seq, (pc, hn, na, rules, sk)


@dataclass
class RemovalResult:
    status: int                  # CASIM_OK or NG_UNSUPPORTED (delegate the whole loop)
    removable: np.ndarray        # [K] 1 removable / 0 no place / 2 not evaluated
    node_out: np.ndarray         # [total] destination of every listed pod in its own candidate's simulation
    ext_candidate: np.ndarray    # [n_ext] pods listed again by a later candidate: which candidate,
    ext_pod: np.ndarray          #         which pod (flat index),
    ext_node: np.ndarray         #         where it went
    last_index: int
    n_processed: int


def make_removal_candidates(cand_node, pod_offsets, pod_class, hint_node=None, destination=None, persist=True, max_removable=0,
                            last_index=0, pod_sticky=None, ext_capacity=None, rules=None, cand_atomic=None):
    """casim_removal_candidates over numpy arrays; returns (struct, arrays to keep alive)."""
    cn = np.ascontiguousarray(cand_node, np.int32)
    po = np.ascontiguousarray(pod_offsets, np.int32)
    pc = np.ascontiguousarray(pod_class, np.int32)
    hn = None if hint_node is None else np.ascontiguousarray(hint_node, np.int32)
    ds = None if destination is None else np.ascontiguousarray(destination, np.uint8)
    sk = None if pod_sticky is None else np.ascontiguousarray(pod_sticky, np.uint8)
    at = None if cand_atomic is None else np.ascontiguousarray(cand_atomic, np.uint8)
    if at is not None and at.shape[0] != cn.shape[0]:
        raise ValueError("cand_atomic must have one entry per candidate")
    if po.shape[0] != cn.shape[0] + 1:
        raise ValueError("pod_offsets must have one more entry than cand_node")
    total = int(po[-1]) if po.size else 0
    if ext_capacity is None:
        ext_capacity = 2 * total + 64   # a pod is listed again at most once per later candidate it lands on
    st = _abi.RemovalCandidates(n_candidates=int(cn.shape[0]), cand_node=_ptr(cn, C.c_int32) if cn.size else None,
                                pod_offsets=_ptr(po, C.c_int32), pod_class=_ptr(pc, C.c_int32) if pc.size else None,
                                hint_node=_ptr(hn, C.c_int32) if hn is not None and hn.size else None,
                                destination=_ptr(ds, C.c_uint8) if ds is not None and ds.size else None,
                                pod_sticky=_ptr(sk, C.c_uint8) if sk is not None and sk.size else None,
                                cand_atomic=_ptr(at, C.c_uint8) if at is not None and at.size else None,
                                persist=int(bool(persist)), max_removable=int(max_removable), last_index=int(last_index),
                                ext_capacity=int(ext_capacity),
                                rules=C.pointer(rules) if rules is not None and rules.n_rules > 0 else None)
    return st, (cn, po, pc, hn, ds, sk, at, rules)


def alloc_removal_results(st: "_abi.RemovalCandidates"):
    K = st.n_candidates
    total = int(st.pod_offsets[K]) if K >= 0 and st.pod_offsets else 0
    E = max(int(st.ext_capacity), 0)
    arrs = dict(removable=np.full(max(K, 1), 2, np.uint8), node_out=np.full(max(total, 1), -1, np.int32),
                ext_candidate=np.full(max(E, 1), -1, np.int32), ext_pod=np.full(max(E, 1), -1, np.int32),
                ext_node=np.full(max(E, 1), -1, np.int32))
    res = _abi.RemovalResults(removable=_ptr(arrs["removable"], C.c_uint8), node_out=_ptr(arrs["node_out"], C.c_int32),
                              ext_candidate=_ptr(arrs["ext_candidate"], C.c_int32), ext_pod=_ptr(arrs["ext_pod"], C.c_int32),
                              ext_node=_ptr(arrs["ext_node"], C.c_int32))
    return res, (arrs, K, total)


def finish_removal_results(rc, st, res, packed) -> RemovalResult:
    arrs, K, total = packed
    ne = int(res.n_ext)
    return RemovalResult(rc, arrs["removable"][:K].copy(), arrs["node_out"][:total].copy(), arrs["ext_candidate"][:ne].copy(),
                         arrs["ext_pod"][:ne].copy(), arrs["ext_node"][:ne].copy(), int(res.last_index), int(res.n_processed))


def make_cluster_estimate(classes, n_existing, max_nodes=0, last_index=0, rules=None, port_block=None):
    G = max(classes.n_pegs, 1)
    arrs = dict(order=np.zeros(G, np.int32), placed=np.zeros(G, np.int32))
    params = _abi.ClusterEstimate(n_existing=int(n_existing), max_nodes=int(max_nodes), last_index=int(last_index),
                                  rules=C.pointer(rules) if rules is not None and rules.n_rules > 0 else None, port_block=port_block)
    res = _abi.ClusterEstimateResult(order=_ptr(arrs["order"], C.c_int32), placed=_ptr(arrs["placed"], C.c_int32))
    return params, res, arrs


def finish_cluster_estimate(res, arrs, n_pegs):
    return dict(node_count=int(res.node_count), pods_scheduled=int(res.pods_scheduled), nodes_added=int(res.nodes_added),
                limiter_nodes=int(res.limiter_nodes), last_index_out=int(res.last_index_out), req_cpu_sum=int(res.req_cpu_sum),
                req_mem_sum=int(res.req_mem_sum), order=arrs["order"][:n_pegs].copy(), placed=arrs["placed"][:n_pegs].copy())


def device_count() -> int:
    return int(lib.casim_device_count())


class _HostBlock:
    """One casim_host_alloc block, freed with the last array that views it."""

    def __init__(self, nbytes: int):
        self.ptr = lib.casim_host_alloc(max(int(nbytes), 8))
        if not self.ptr:
            raise CasimError("casim_host_alloc failed")

    def __del__(self):
        if getattr(self, "ptr", None):
            lib.casim_host_free(self.ptr)
            self.ptr = None


def pinned_copy(a: np.ndarray) -> np.ndarray:
    """A copy of `a` in page-locked host memory (casim_host_alloc): a table column that lives there is copied to the device where it lies
    — the library skips its staging memcpy for columns of >= 1 MiB (include/casim.h)."""
    a = np.ascontiguousarray(a)
    blk = _HostBlock(a.nbytes)
    buf = (C.c_uint8 * max(a.nbytes, 8)).from_address(blk.ptr)
    buf._casim_block = blk          # (numpy keeps `buf` as the array's base: the block lives as long as a view of it does)
    out = np.frombuffer(buf, dtype=np.uint8, count=a.nbytes).view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


class Context:
    """casim_ctx: one HIP device + one stream.  `stream` may be a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream, so that torch events bracket the kernels."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._h = lib.casim_ctx_create(int(device), C.c_void_p(stream) if stream else None)
        if not self._h:
            msg = last_error()
            if device_count() == 0:
                raise NoDeviceError(_abi.ERR_NO_DEVICE, msg or "no HIP device visible")
            raise CasimError(_abi.ERR_HIP, msg)
        self.device = device

    def close(self):
        if self._h:
            lib.casim_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def pack_build_info(self) -> dict:
        """casim_pack_build_info: which build of the register packer the self-check left standing for this context's device."""
        out = (C.c_int32 * 4)()
        check(lib.casim_pack_build_info(self.device, out), "casim_pack_build_info")
        return {"build": {0: "auto", 1: "plain", 2: "option"}[out[0]], "batches_compared": out[1], "batches_differing": out[2], "forced_by_env": bool(out[3])}

    def copy_bandwidth_gbps(self, nbytes: int = 1 << 30, iters: int = 10) -> float:
        out = C.c_double(0)
        check(lib.casim_copy_bandwidth(self._h, nbytes, iters, C.byref(out)), "casim_copy_bandwidth")
        return out.value

    def stream_probe_gbps(self, nbytes: int = 1 << 30, lane_bytes: int = 4, iters: int = 5) -> float:
        out = C.c_double(0)
        check(lib.casim_stream_probe(self._h, nbytes, lane_bytes, iters, C.byref(out)), "casim_stream_probe")
        return out.value

    def feasibility_reasons(self, pegs: _abi.Pegs, groups: _abi.Groups, port_block=None) -> np.ndarray:
        """casim_feasibility_reasons: uint16 [NG][L] (L = n_pegs, or the longest candidate range of a batch): 0 = fits, else
        first failing Filter plugin (low 4 bits) + NodeResourcesFit reasons."""
        ng = groups.n_groups
        L = pegs.n_pegs if not groups.peg_lo else max((groups.peg_hi[i] - groups.peg_lo[i] for i in range(ng)), default=0)
        codes = np.zeros((max(ng, 1), max(L, 1)), np.uint16)
        check(lib.casim_feasibility_reasons(self._h, C.byref(pegs), C.byref(groups), port_block, codes.ctypes.data_as(C.POINTER(C.c_uint16))),
              "casim_feasibility_reasons")
        return codes[:ng, :L]

    def feasibility(self, pegs: _abi.Pegs, groups: _abi.Groups) -> np.ndarray:
        """bit-matrix [NG][ceil(G/64)]: PEG g passes every encoded Filter on a fresh node of group i."""
        wg = (pegs.n_pegs + 63) // 64
        bits = np.zeros((max(groups.n_groups, 1), max(wg, 1)), np.uint64)
        check(lib.casim_feasibility(self._h, C.byref(pegs), C.byref(groups), _ptr(bits, C.c_uint64)), "casim_feasibility")
        return bits[:groups.n_groups, :wg]


    def try_schedule_pods(self, classes: _abi.Pegs, nodes: _abi.Groups, pod_class, hint_node=None, node_acceptable=None,
                          break_on_failure: bool = False, last_index: int = 0, time_iters: int = 0, rules=None, similar_key=None):
        """HintingSimulator.TrySchedulePods on the device (casim_try_schedule_pods).
        Returns (status, node_out[P], last_index, n_scheduled); status NG_UNSUPPORTED => delegate to the Go path.
        With time_iters > 0 returns the HIP-event time in ms of one resident pass instead."""
        seq, keep = make_pod_sequence(pod_class, hint_node, node_acceptable, break_on_failure, last_index, rules, similar_key)
        if time_iters > 0:
            ms = C.c_float(0)
            rc = lib.casim_time_try_schedule_pods(self._h, C.byref(classes), C.byref(nodes), C.byref(seq), int(time_iters), C.byref(ms))
            if rc < 0:
                check(rc, "casim_time_try_schedule_pods")
            return rc, ms.value / time_iters
        node_out = np.full(max(seq.n_pods, 1), -1, np.int32)
        li, ns = C.c_int32(0), C.c_int32(0)
        rc = lib.casim_try_schedule_pods(self._h, C.byref(classes), C.byref(nodes), C.byref(seq), _ptr(node_out, C.c_int32),
                                         C.byref(li), C.byref(ns))
        if rc < 0:
            check(rc, "casim_try_schedule_pods")
        del keep
        return rc, node_out[:seq.n_pods], li.value, ns.value


    def estimate_on_cluster(self, classes: _abi.Pegs, nodes: _abi.Groups, n_existing: int, max_nodes: int = 0, last_index: int = 0,
                            rules=None, port_block=None):
        """BinpackingNodeEstimator.Estimate on the whole snapshot (casim_estimate_on_cluster): the path for node groups
        whose PEGs carry domain rules.  Returns (status, ClusterEstimateResult fields as a dict)."""
        params, res, arrs = make_cluster_estimate(classes, n_existing, max_nodes, last_index, rules, port_block)
        rc = lib.casim_estimate_on_cluster(self._h, C.byref(classes), C.byref(nodes), C.byref(params), C.byref(res))
        if rc < 0:
            check(rc, "casim_estimate_on_cluster")
        return rc, finish_cluster_estimate(res, arrs, classes.n_pegs)

    @staticmethod
    def last_removals_info():
        """casim_last_removals_info: which kernel this thread's last removal simulation ran as — {"lean": the one-wave kernel over per-class
        fit masks, "threads", "state_in_lds", "runs"}"""
        info = (C.c_int32 * 4)()
        check(lib.casim_last_removals_info(info), "casim_last_removals_info")
        return {"lean": bool(info[0]), "threads": int(info[1]), "state_in_lds": bool(info[2]), "runs": int(info[3])}

    @staticmethod
    def last_chain_info():
        """casim_last_chain_info: this thread's last run with chain_last_index — {"bound": passes the fixed point is bounded by, "passes":
        passes enqueued, "checks": times the host read a pass's marks back, "whole": the chain was enqueued whole}"""
        info = (C.c_int32 * 4)()
        check(lib.casim_last_chain_info(info), "casim_last_chain_info")
        return {"bound": int(info[0]), "passes": int(info[1]), "checks": int(info[2]), "whole": bool(info[3])}

    def simulate_node_removals(self, classes: _abi.Pegs, nodes: _abi.Groups, cand_node, pod_offsets, pod_class, hint_node=None,
                               destination=None, persist: bool = True, max_removable: int = 0, last_index: int = 0,
                               pod_sticky=None, ext_capacity: Optional[int] = None, time_iters: int = 0, rules=None,
                               cand_atomic=None):
        """Planner.categorizeNodes loop around SimulateNodeRemoval on the device (casim_simulate_node_removals).
        Returns a RemovalResult; with time_iters > 0 (status, HIP-event ms of one resident pass) instead."""
        st, keep = make_removal_candidates(cand_node, pod_offsets, pod_class, hint_node, destination, persist, max_removable, last_index,
                                           pod_sticky, ext_capacity, rules, cand_atomic)
        if time_iters > 0:
            ms = C.c_float(0)
            rc = lib.casim_time_node_removals(self._h, C.byref(classes), C.byref(nodes), C.byref(st), int(time_iters), C.byref(ms))
            if rc < 0:
                check(rc, "casim_time_node_removals")
            return rc, ms.value / time_iters
        res, arrs = alloc_removal_results(st)
        rc = lib.casim_simulate_node_removals(self._h, C.byref(classes), C.byref(nodes), C.byref(st), C.byref(res))
        if rc < 0:
            check(rc, "casim_simulate_node_removals")
        del keep
        return finish_removal_results(rc, st, res, arrs)


class ResidentCluster:
    """casim_cluster: the snapshot's node table resident in HBM for a RunOnce iteration.  try_schedule_pods(commit=True) is
    filter-out-schedulable adding its pods to the snapshot, simulate_node_removals runs on that committed image and never
    persists (the planner's Fork / Revert), update_nodes replaces single node records between calls."""

    def __init__(self, ctx: "Context", classes: _abi.Pegs, nodes: _abi.Groups):
        self.ctx = ctx
        self.n_nodes, self.n_res, self.w_excl = nodes.n_groups, classes.n_res, classes.w_excl
        self._h = lib.casim_cluster_create(ctx._h, C.byref(classes), C.byref(nodes))
        if not self._h:
            raise CasimError(_abi.ERR_INVALID, last_error())

    def close(self):
        if self._h:
            lib.casim_cluster_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def try_schedule_pods(self, pod_class, hint_node=None, node_acceptable=None, break_on_failure=False, last_index=0, commit=True,
                          rules=None, similar_key=None):
        seq, keep = make_pod_sequence(pod_class, hint_node, node_acceptable, break_on_failure, last_index, rules, similar_key)
        node_out = np.full(max(seq.n_pods, 1), -1, np.int32)
        li, ns = C.c_int32(0), C.c_int32(0)
        rc = lib.casim_cluster_try_schedule_pods(self._h, C.byref(seq), int(bool(commit)), _ptr(node_out, C.c_int32), C.byref(li), C.byref(ns))
        if rc < 0:
            check(rc, "casim_cluster_try_schedule_pods")
        del keep
        return rc, node_out[:seq.n_pods], li.value, ns.value

    def simulate_node_removals(self, cand_node, pod_offsets, pod_class, hint_node=None, destination=None, persist=True, max_removable=0,
                               last_index=0, pod_sticky=None, ext_capacity=None, rules=None, cand_atomic=None):
        st, keep = make_removal_candidates(cand_node, pod_offsets, pod_class, hint_node, destination, persist, max_removable, last_index,
                                           pod_sticky, ext_capacity, rules, cand_atomic)
        res, arrs = alloc_removal_results(st)
        rc = lib.casim_cluster_simulate_node_removals(self._h, C.byref(st), C.byref(res))
        if rc < 0:
            check(rc, "casim_cluster_simulate_node_removals")
        del keep
        return finish_removal_results(rc, st, res, arrs)

    def update_nodes(self, node_index, rows: _abi.Groups):
        idx = np.ascontiguousarray(node_index, np.int32)
        check(lib.casim_cluster_update_nodes(self._h, int(idx.shape[0]), _ptr(idx, C.c_int32), C.byref(rows)), "casim_cluster_update_nodes")

    def fetch_nodes(self):
        req = np.zeros((max(self.n_nodes, 1), self.n_res), np.int64)
        pods = np.zeros(max(self.n_nodes, 1), np.int32)
        excl = np.zeros((max(self.n_nodes, 1), max(self.w_excl, 1)), np.uint64)
        check(lib.casim_cluster_fetch_nodes(self._h, _ptr(req, C.c_int64), _ptr(pods, C.c_int32), _ptr(excl, C.c_uint64)), "casim_cluster_fetch_nodes")
        return req[:self.n_nodes], pods[:self.n_nodes], excl[:self.n_nodes, :self.w_excl]

    def forget_commits(self):
        """the caller's next rules come from a snapshot that already holds the committed pods (casim.h)"""
        check(lib.casim_cluster_forget_commits(self._h), "casim_cluster_forget_commits")

    def stats(self):
        out = (C.c_int64 * 4)()
        check(lib.casim_cluster_stats(self._h, out), "casim_cluster_stats")
        return {"full_uploads": out[0], "delta_rows": out[1], "commits": out[2], "nodes": out[3]}


class MultiContext:
    """casim_mctx: several devices behind one caller (one process, one host thread — the shape a Go estimator has).
    estimate_batch() block-partitions the node groups of every simulation over the devices, runs them concurrently and
    settles the expander with ONE all-reduce(min) over the per-simulation packed keys (RCCL over xGMI, or the host)."""

    def __init__(self, devices: Sequence[int], use_rccl: bool = True):
        arr = (C.c_int32 * len(devices))(*devices)
        self._h = lib.casim_mctx_create(arr, len(devices), int(bool(use_rccl)))
        if not self._h:
            msg = last_error()
            if device_count() == 0:
                raise NoDeviceError(_abi.ERR_NO_DEVICE, msg or "no HIP device visible")
            raise CasimError(_abi.ERR_HIP, msg)
        self.n_devices = len(devices)

    def close(self):
        if self._h:
            lib.casim_mctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def info(self):
        n, rccl, red = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        per = (C.c_int32 * max(self.n_devices, 1))()
        check(lib.casim_mctx_info(self._h, C.byref(n), C.byref(rccl), C.byref(red), per), "casim_mctx_info")
        return {"devices": n.value, "rccl": bool(rccl.value), "last_reduce_by_rccl": bool(red.value), "groups_per_device": list(per)[:n.value]}

    def estimate_batch(self, pegs: _abi.Pegs, groups: _abi.Groups, kinds: Optional[Sequence[int]] = None, valid=None, fastpath: bool = False,
                       nnz_cap: Optional[int] = None):
        """Returns (BatchResult, expander dict or None): results in the caller's group order."""
        ng = groups.n_groups
        if nnz_cap is None:
            if groups.peg_offsets:
                nnz_cap = int(groups.peg_offsets[ng]) if ng else 0
            elif groups.peg_lo:
                nnz_cap = int(sum(groups.peg_hi[i] - groups.peg_lo[i] for i in range(ng)))
            else:
                nnz_cap = pegs.n_pegs * ng
        st, arrs = alloc_results(ng, nnz_cap)
        opts = _abi.Options(fastpath=int(fastpath))
        off = np.zeros(ng + 1, np.int32)
        q = exp = None
        keep = []
        if kinds is not None:
            S = groups.n_sims if groups.n_sims > 0 else 1
            ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
            exp = dict(best=np.full(S, -1, np.int32), n_best=np.zeros(S, np.int32), packed=np.zeros(S, np.int64), keys=np.zeros((S, 10), np.int64))
            q = _abi.OptionQuery(kinds=ks, n_kinds=len(kinds), per_sim=1, best_out=_ptr(exp["best"], C.c_int32),
                                 n_best_out=_ptr(exp["n_best"], C.c_int32), packed_out=_ptr(exp["packed"], C.c_int64), key_out=_ptr(exp["keys"], C.c_int64))
            if valid is not None:
                v = np.ascontiguousarray(valid, np.uint8); keep.append(v)
                q.valid = _ptr(v, C.c_uint8)
        check(lib.casim_estimate_batch_multi(self._h, C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st), _ptr(off, C.c_int32),
                                             C.byref(q) if q is not None else None), "casim_estimate_batch_multi")
        return finish_results(arrs, ng, int(off[ng]), off), exp


class Problem:
    """casim_problem: a batch resident in HBM; run() enqueues feasibility -> order -> pack."""

    def __init__(self, ctx: Context, pegs: _abi.Pegs, groups: _abi.Groups, fastpath: bool = False, force_generic_packer: bool = False,
                 node_pods: bool = False, n_streams: int = 0, pack_build: int = 0, no_front_kernel: bool = False, chain_last_index: bool = False):
        """n_streams > 1: a batch of simulations runs as up to n_streams sub-batches on internal HIP streams of the context
        (casim_options.n_streams); results are identical, info()["parts"] tells whether the batch was cut.
        pack_build: _abi.PACK_BUILD_AUTO / _PLAIN / _OPTION (casim_options.pack_build).
        no_front_kernel: the four separate launches of a batch instead of the fused one (casim_options.no_front_kernel; info()["front_kernel"]).
        chain_last_index: the groups of one simulation as successive Estimate() calls on one snapshot — lastIndex handed from group to group
        (casim_options.chain_last_index, plugin_runner.go:138)."""
        self.ctx = ctx
        self.n_groups = groups.n_groups
        self.n_pegs = pegs.n_pegs
        # room for the per-node pod counts: every group's limiter bound, or its pods when unlimited
        self._node_pods_cap = 0
        if node_pods:
            total = int(sum(max(int(pegs.count[i]), 1) for i in range(pegs.n_pegs)))
            self._node_pods_cap = int(sum((int(groups.max_nodes[i]) if groups.max_nodes[i] > 0 else (0 if groups.max_nodes[i] < 0 else total))
                                          for i in range(groups.n_groups))) + 64
        opts = _abi.Options(fastpath=int(fastpath), force_generic_packer=int(force_generic_packer), node_pods=int(bool(node_pods)),
                            n_streams=int(n_streams), pack_build=int(pack_build), no_front_kernel=int(bool(no_front_kernel)),
                            chain_last_index=int(bool(chain_last_index)))
        self._h = lib.casim_problem_create(ctx._h, C.byref(pegs), C.byref(groups), C.byref(opts))
        if not self._h:
            raise CasimError(_abi.ERR_INVALID, last_error())

    def close(self):
        if self._h:
            lib.casim_problem_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def run(self):
        check(lib.casim_problem_run(self._h), "casim_problem_run")

    def info(self):
        out = (C.c_int32 * 8)()
        check(lib.casim_problem_info(self._h, out), "casim_problem_info")
        return {"fast_packer_slots_per_lane": out[0], "fast_packer_lanes": out[1], "generic_state_in_lds": bool(out[2]),
                "csr_on_device": bool(out[3]), "parts": int(out[4]), "forks": int(out[5]), "parked_streams": int(out[6]), "front_kernel": bool(out[7] & 1), "ranked_orderer": bool(out[7] & 2)}

    def set_group_result(self, ng: int, r: dict):
        """casim_problem_set_group_result: a group estimated by Context.estimate_on_cluster joins the expander reduce."""
        st = _abi.ClusterEstimateResult(node_count=r["node_count"], pods_scheduled=r["pods_scheduled"], nodes_added=r["nodes_added"],
                                        limiter_nodes=r["limiter_nodes"], last_index_out=r["last_index_out"], status=0,
                                        req_cpu_sum=r["req_cpu_sum"], req_mem_sum=r["req_mem_sum"])
        check(lib.casim_problem_set_group_result(self._h, int(ng), C.byref(st)), "casim_problem_set_group_result")

    def csr(self):
        nnz = C.c_int32(0)
        off = np.zeros(self.n_groups + 1, np.int32)
        check(lib.casim_problem_csr(self._h, C.byref(nnz), _ptr(off, C.c_int32)), "casim_problem_csr")
        return int(nnz.value), off

    def fetch(self) -> BatchResult:
        nnz, off = self.csr()
        st, arrs = alloc_results(self.n_groups, nnz, self._node_pods_cap)
        check(lib.casim_problem_fetch(self._h, C.byref(st)), "casim_problem_fetch")
        return finish_results(arrs, self.n_groups, nnz, off)

    def best_option(self, kinds: Sequence[int], group_id_base: int = 0, dev_key_ptr: Optional[int] = None):
        ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
        best, nbest = C.c_int32(-1), C.c_int32(0)
        bset = np.zeros(max(self.n_groups, 1), np.uint8)
        key = np.zeros(10, np.int64)
        check(lib.casim_best_option(self._h, ks, len(kinds), int(group_id_base), C.byref(best), C.byref(nbest),
                                    _ptr(bset, C.c_uint8), _ptr(key, C.c_int64),
                                    C.c_void_p(dev_key_ptr) if dev_key_ptr else None), "casim_best_option")
        return int(best.value), int(nbest.value), bset[:self.n_groups], key

    def best_option_sims(self, kinds: Sequence[int], per_sim: bool = True, valid=None, group_id_base: int = 0,
                         dev_packed_ptr: Optional[int] = None, fetch: bool = True, n_sims: Optional[int] = None, join_stream: Optional[int] = None):
        """casim_best_option_sims: the expander chain once per simulation of the batch (or once over every group), with an
        optional validity mask (all-or-nothing).  fetch=False only enqueues the kernel (the packed keys land in
        dev_packed_ptr for an RCCL all-reduce; join_stream = raw handle of the stream that shall wait for them instead of the
        context's stream: casim_option_query.join_stream).  Returns dict(best, n_best, best_set, keys, packed) when fetching."""
        ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
        S = (n_sims if n_sims else 1) if per_sim else 1
        q = _abi.OptionQuery(kinds=ks, n_kinds=len(kinds), group_id_base=int(group_id_base), per_sim=int(bool(per_sim)))
        keep = [ks]
        if valid is not None:
            v = np.ascontiguousarray(valid, np.uint8)
            if v.shape[0] != self.n_groups:
                raise ValueError("valid must have one entry per group")
            q.valid = _ptr(v, C.c_uint8); keep.append(v)
        if dev_packed_ptr:
            q.dev_packed_out = C.c_void_p(dev_packed_ptr)
        if join_stream:
            q.join_stream = C.c_void_p(join_stream)
        out = None
        if fetch:
            out = dict(best=np.full(S, -1, np.int32), n_best=np.zeros(S, np.int32), best_set=np.zeros(max(self.n_groups, 1), np.uint8),
                       keys=np.zeros((S, 10), np.int64), packed=np.zeros(S, np.int64))
            q.best_out = _ptr(out["best"], C.c_int32); q.n_best_out = _ptr(out["n_best"], C.c_int32)
            q.best_set_out = _ptr(out["best_set"], C.c_uint8); q.key_out = _ptr(out["keys"], C.c_int64)
            q.packed_out = _ptr(out["packed"], C.c_int64)
        check(lib.casim_best_option_sims(self._h, C.byref(q)), "casim_best_option_sims")
        del keep
        if out is not None:
            out["best_set"] = out["best_set"][:self.n_groups]
        return out

    def best_option_device(self, kinds: Sequence[int], group_id_base: int, dev_key_ptr: int):
        """Asynchronous form: only writes the 10-int64 key block into device memory (for the RCCL reduce)."""
        ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
        check(lib.casim_best_option(self._h, ks, len(kinds), int(group_id_base), None, None, None, None,
                                    C.c_void_p(dev_key_ptr)), "casim_best_option")

    def run_marked(self):
        """run() with HIP events recorded around the kernel classes; nothing waits (see marked_ms)."""
        check(lib.casim_problem_run_marked(self._h), "casim_problem_run_marked")

    def marked_ms(self):
        """Mean (total ms, per-kernel-class ms, runs) over the run_marked() calls since the last collection."""
        tot, n = C.c_float(0), C.c_int32(0)
        ks = (C.c_float * 3)()
        check(lib.casim_problem_marked_ms(self._h, C.byref(tot), ks, C.byref(n)), "casim_problem_marked_ms")
        return float(tot.value), {"feasibility_csr_ms": ks[0], "order_ms": ks[1], "pack_ms": ks[2]}, int(n.value)

    def time(self, iters: int = 10):
        tot = C.c_float(0)
        ks = (C.c_float * 3)()
        check(lib.casim_problem_time(self._h, int(iters), C.byref(tot), ks), "casim_problem_time")
        return float(tot.value), {"feasibility_csr_ms": ks[0], "order_ms": ks[1], "pack_ms": ks[2]}


    def time_feasibility(self, iters: int = 50):
        """casim_problem_time_feasibility: (ms per feasibility launch, {"stream", "lean", "narrow_masks", "unsched_on_spare_bit", "workgroups"})"""
        ms = C.c_float(0)
        info = (C.c_int32 * 4)()
        check(lib.casim_problem_time_feasibility(self._h, int(iters), C.byref(ms), info), "casim_problem_time_feasibility")
        return float(ms.value), {"stream": bool(info[0]), "lean": bool(info[1]), "narrow_masks": bool(info[2] & 1), "unsched_on_spare_bit": bool(info[2] & 2),
                                 "workgroups": int(info[3])}


def estimate_batch_timed(ctx: Context, pegs: _abi.Pegs, groups: _abi.Groups, kinds: Optional[Sequence[int]] = None, fastpath: bool = False,
                         nnz_cap: Optional[int] = None):
    """casim_estimate_batch_timed: one whole call (tables -> HBM -> kernels -> results) with its phase breakdown.
    Returns (results dict of arrays, phases dict in ms, expander dict or None)."""
    ng = groups.n_groups
    if nnz_cap is None:
        if groups.peg_offsets:
            nnz_cap = int(groups.peg_offsets[ng]) if ng else 0
        elif groups.peg_lo:
            nnz_cap = int(sum(groups.peg_hi[i] - groups.peg_lo[i] for i in range(ng)))
        else:
            nnz_cap = pegs.n_pegs * ng
    st, arrs = alloc_results(ng, nnz_cap)
    opts = _abi.Options(fastpath=int(fastpath))
    ph = (C.c_double * 8)()
    q = exp = None
    if kinds is not None:
        ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
        S = groups.n_sims if groups.n_sims > 0 else 1
        exp = dict(best=np.full(S, -1, np.int32), n_best=np.zeros(S, np.int32), packed=np.zeros(S, np.int64))
        q = _abi.OptionQuery(kinds=ks, n_kinds=len(kinds), per_sim=int(groups.n_sims > 0), best_out=_ptr(exp["best"], C.c_int32),
                             n_best_out=_ptr(exp["n_best"], C.c_int32), packed_out=_ptr(exp["packed"], C.c_int64))
    check(lib.casim_estimate_batch_timed(ctx._h, C.byref(pegs), C.byref(groups), C.byref(opts), C.byref(st),
                                         C.byref(q) if q is not None else None, ph), "casim_estimate_batch_timed")
    names = ("upload_ms", "feasibility_csr_ms", "order_ms", "pack_ms", "expander_ms", "fetch_ms", "wall_ms")
    return arrs, {k: ph[i] for i, k in enumerate(names)}, exp


def _nnz_cap(pegs, groups):
    ng = groups.n_groups
    if groups.peg_offsets:
        return int(groups.peg_offsets[ng]) if ng else 0
    if groups.peg_lo:
        lo = np.ctypeslib.as_array(groups.peg_lo, shape=(ng,)); hi = np.ctypeslib.as_array(groups.peg_hi, shape=(ng,))
        return int((hi.astype(np.int64) - lo).sum())
    return pegs.n_pegs * ng


class BatchCall:
    """casim_estimate_batch_query with its buffers kept between calls: upload + kernels + expander reduce per simulation + fetch,
    enter -> return in ONE call of the C ABI (SURVEY 8d's wall time).  n_streams > 1: the parts of the batch run end to end on
    the context's internal streams.  call() returns (BatchResult, expander dict or None)."""

    def __init__(self, ctx: Context, pegs: _abi.Pegs, groups: _abi.Groups, kinds: Optional[Sequence[int]] = None, fastpath: bool = False,
                 force_generic_packer: bool = False, n_streams: int = 0, winners_only: bool = False, pinned_results: bool = False,
                 chain_last_index: bool = False):
        """winners_only (casim_options.winners_only): order / placed come back for the winning group of every simulation only —
        call() then returns a BatchResult whose `order` / `placed` are those compact lists and whose `winner_offsets` [S + 1] says where
        simulation s's list sits (see winners_view)."""
        self.ctx, self.pegs, self.groups = ctx, pegs, groups
        ng = groups.n_groups
        self.winners_only = bool(winners_only)
        self.st, self.arrs = alloc_results(ng, _nnz_cap(pegs, groups), pinned_lists=pinned_results)
        self.opts = _abi.Options(fastpath=int(fastpath), force_generic_packer=int(force_generic_packer), n_streams=int(n_streams),
                                 winners_only=int(self.winners_only), chain_last_index=int(bool(chain_last_index)))
        self.off = np.zeros(ng + 1, np.int32)
        self.q = self.exp = None
        if kinds is not None:
            S = groups.n_sims if groups.n_sims > 0 else 1
            self._ks = (C.c_int32 * max(len(kinds), 1))(*kinds)
            self.exp = dict(best=np.full(S, -1, np.int32), n_best=np.zeros(S, np.int32), packed=np.zeros(S, np.int64))
            self.q = _abi.OptionQuery(kinds=self._ks, n_kinds=len(kinds), per_sim=int(groups.n_sims > 0), best_out=_ptr(self.exp["best"], C.c_int32),
                                      n_best_out=_ptr(self.exp["n_best"], C.c_int32), packed_out=_ptr(self.exp["packed"], C.c_int64))

    def call_raw(self):
        check(lib.casim_estimate_batch_query(self.ctx._h, C.byref(self.pegs), C.byref(self.groups), C.byref(self.opts), C.byref(self.st),
                                             _ptr(self.off, C.c_int32), C.byref(self.q) if self.q is not None else None), "casim_estimate_batch_query")

    def call(self):
        self.call_raw()
        ng = self.groups.n_groups
        res = finish_results(self.arrs, ng, int(self.off[ng]), self.off.copy())
        if self.winners_only:
            res.winner_offsets = winner_offsets(self.off, self.exp["best"])
        return res, self.exp


def winner_offsets(offsets, best) -> np.ndarray:
    """casim_options.winners_only: where simulation s's winning list sits inside the compact order / placed arrays — [S + 1] prefix sums of
    the winners' list lengths (0 for a simulation without an option), from the batch's full CSR offsets and the expander's best_out."""
    best = np.asarray(best, np.int64)
    off = np.asarray(offsets, np.int64)
    has = best >= 0
    lens = np.where(has, off[np.where(has, best, 0) + 1] - off[np.where(has, best, 0)], 0)
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def estimate_batch(ctx: Context, pegs: _abi.Pegs, groups: _abi.Groups, fastpath: bool = False) -> BatchResult:
    with Problem(ctx, pegs, groups, fastpath) as p:
        p.run()
        return p.fetch()


class PrefetchCache:
    """casim_prefetch_*: one batch over every candidate node group (fill), one lookup per Estimate()."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._h = lib.casim_prefetch_create(ctx._h)
        if not self._h:
            raise CasimError(_abi.ERR_INVALID, "casim_prefetch_create failed")

    def close(self):
        if self._h:
            lib.casim_prefetch_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def clear(self):
        lib.casim_prefetch_clear(self._h)

    def fill(self, pegs: _abi.Pegs, groups: _abi.Groups, group_keys, peg_keys, fastpath: bool = False, chain_last_index: bool = False):
        gk = np.ascontiguousarray(group_keys, np.uint64); pk = np.ascontiguousarray(peg_keys, np.uint64)
        if gk.shape[0] != groups.n_groups or pk.shape[0] != pegs.n_pegs:
            raise ValueError("one key per group and per PEG")
        opts = _abi.Options(fastpath=int(fastpath), chain_last_index=int(bool(chain_last_index)))
        rc = lib.casim_prefetch_fill(self._h, C.byref(pegs), C.byref(groups), C.byref(opts), gk.ctypes.data_as(_abi.u64p), pk.ctypes.data_as(_abi.u64p))
        if rc != 0:
            raise CasimError(rc, (lib.casim_prefetch_error(self._h) or b"").decode())

    def lookup(self, group_key: int, peg_keys, max_nodes: int, existing_nodes: int, last_index: int):
        """(hit, dict): hit -> node_count, pods_scheduled, ..., order (positions in the caller's PEG list), placed; miss -> {'miss_reason'}."""
        pk = np.ascontiguousarray(peg_keys, np.uint64)
        n = int(pk.shape[0])
        r = _abi.PrefetchResult()
        order = np.zeros(max(n, 1), np.int32); placed = np.zeros(max(n, 1), np.int32)
        rc = lib.casim_prefetch_lookup(self._h, C.c_uint64(int(group_key)), pk.ctypes.data_as(_abi.u64p), n, int(max_nodes), int(existing_nodes), int(last_index),
                                       C.byref(r), order.ctypes.data_as(_abi.i32p), placed.ctypes.data_as(_abi.i32p))
        if rc == _abi.PREFETCH_MISS:
            return False, {"miss_reason": int(r.miss_reason)}
        if rc != 0:
            raise CasimError(rc, "casim_prefetch_lookup")
        out = {f: int(getattr(r, f)) for f, _ in _abi.PrefetchResult._fields_}
        out["order"] = order[:n].copy(); out["placed"] = placed[:n].copy()
        return True, out

    def stats(self):
        out = (C.c_int64 * 8)()
        lib.casim_prefetch_stats(self._h, out)
        return {"fills": out[0], "groups_cached": out[1], "hits": out[2], "miss_group": out[3], "miss_pegs": out[4], "miss_limits": out[5],
                "miss_last_index": out[6]}


class StreamedBatch:
    """One casim context on `device` (on `stream`, a raw hipStream_t handle, when given) and one streamed casim_problem: a TableSet of
    independent simulations whose parts run on the context's internal streams (casim_options.n_streams, csrc/casim_streams.h)."""

    def __init__(self, device: int, tables, n_streams: int = 4, stream: Optional[int] = None, **problem_kw):
        self.tables = tables if tables.peg_lo is not None else tables.as_one_simulation()
        self.n_sims = self.tables.n_sims
        self.ctx = Context(device, stream=stream)
        self._structs = self.tables.structs()
        self.prob = Problem(self.ctx, *self._structs, n_streams=n_streams, **problem_kw)
        self.parts = self.prob.info()["parts"]

    def close(self):
        if self.prob is not None:
            self.prob.close(); self.ctx.close()
            self.prob = self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def run(self):
        self.prob.run()

    def best_option_sims(self, kinds: Sequence[int], dev_packed_ptr: Optional[int] = None, fetch: bool = True, join_stream: Optional[int] = None):
        return self.prob.best_option_sims(kinds, per_sim=True, fetch=fetch, dev_packed_ptr=dev_packed_ptr, n_sims=self.n_sims, join_stream=join_stream)

    def fetch(self) -> BatchResult:
        return self.prob.fetch()
