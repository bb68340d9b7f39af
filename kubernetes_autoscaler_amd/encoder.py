"""Python driver of the C++ host encoder (casim_enc_* in include/casim.h).

It walks pod / node-template objects exactly the way the cgo shim of INTEGRATION.md does and
returns flat tables (ctypes views owned by the C++ encoder)."""
import ctypes as C
import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

from . import _abi
from ._ffi import check, lib
from .objects import NodeInfo, NodeSelectorTerm, Pod, PodEquivalenceGroup, RES_CPU, RES_EPHEMERAL, RES_MEMORY

DEFAULT_LANES = (RES_CPU, RES_MEMORY)


def _b(s: str) -> bytes:
    return (s or "").encode("utf-8")


def _strs(values: Sequence[str]):
    arr = (C.c_char_p * max(len(values), 1))()
    for i, v in enumerate(values):
        arr[i] = _b(v)
    return arr


class Encoder:
    """One encoder instance == one scale-up loop's worth of PEGs and node groups."""

    def __init__(self, lanes: Sequence[str] = DEFAULT_LANES, enable_taint_comparison_ops: bool = False,
                 explicit_self_exclusion: bool = False, named_lanes: bool = False):
        """named_lanes: the call sequence of the Go shim (integration/go/gpubinpacking/encode.go) — three positional lanes (cpu, memory,
        ephemeral-storage) and EVERY other resource of a pod's requests / a node's allocatable handed over by name
        (casim_enc_pod_set_request / casim_enc_group_set_allocatable, ABI 9): lane numbers stay behind the ABI; a name that finds no
        lane delegates the pod instead of dropping the request.  `lanes` is ignored then (self.lanes is read back after finalize)."""
        self.named_lanes = bool(named_lanes)
        if self.named_lanes:
            lanes = (RES_CPU, RES_MEMORY, RES_EPHEMERAL)
        if len(lanes) < 2 or len(lanes) > _abi.MAX_RES or lanes[0] != RES_CPU or lanes[1] != RES_MEMORY:
            raise ValueError("lanes must start with ('cpu', 'memory') and have 2..8 entries")
        self.lanes = tuple(lanes)
        opts = _abi.EncoderOptions(n_res=len(lanes), enable_taint_comparison_ops=int(enable_taint_comparison_ops),
                                   explicit_self_exclusion=int(explicit_self_exclusion))
        self._h = lib.casim_enc_create(C.byref(opts))
        if not self._h:
            raise MemoryError("casim_enc_create failed")
        from . import objects
        for ns, labels in objects.NAMESPACE_LISTER.items():   # the framework handle's namespace lister, as of now
            check(lib.casim_enc_add_namespace(self._h, _b(ns)))
            for k, v in labels.items():
                check(lib.casim_enc_namespace_add_label(self._h, _b(ns), _b(k), _b(v)))
        self._spec_of: Dict[int, int] = {}      # id(pod object) -> spec id (PEGs repeat one pointer)
        self._keep: List[object] = []
        self.finalized = False
        self.n_pegs = 0
        self.n_groups = 0

    def close(self):
        if self._h:
            lib.casim_enc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- pods ---------------------------------------------------------------------------
    def _lane_vector(self, res: Dict[str, int]):
        vec = (C.c_int64 * _abi.MAX_RES)()
        for i, name in enumerate(self.lanes):
            vec[i] = int(res.get(name, 0))
        return vec

    def add_pod_spec(self, pod: Pod) -> int:
        key = id(pod)
        if key in self._spec_of:
            return self._spec_of[key]
        self._keep.append(pod)
        h = self._h
        s = lib.casim_enc_add_pod_spec(h, _b(pod.namespace), self._lane_vector(pod.requests))
        if s < 0:
            check(s, "casim_enc_add_pod_spec")
        self._named_requests(pod, s)
        for k, v in pod.labels.items():
            check(lib.casim_enc_pod_add_label(h, s, _b(k), _b(v)))
        for t in pod.tolerations:
            check(lib.casim_enc_pod_add_toleration(h, s, _b(t.key), _b(t.operator), _b(t.value), _b(t.effect)))
        for k, v in pod.node_selector.items():
            check(lib.casim_enc_pod_add_node_selector(h, s, _b(k), _b(v)))
        cpu, mem = pod.fastpath_requests()
        check(lib.casim_enc_pod_set_fastpath_requests(h, s, cpu, mem))
        self._pod_rest(pod, s)
        self._spec_of[key] = s
        return s

    def _named_requests(self, pod: Pod, s: int):
        if self.named_lanes:   # ScalarResources by name (fit.go:731-763); CASIM_ENC_DELEGATED (1) = no lane left, the pod is marked unsupported
            for name, v in pod.requests.items():
                if name not in (RES_CPU, RES_MEMORY, RES_EPHEMERAL):
                    rc = lib.casim_enc_pod_set_request(self._h, s, _b(name), int(v))
                    if rc < 0:
                        check(rc, "casim_enc_pod_set_request")

    def _pod_rest(self, pod: Pod, s: int, digest: bool = True):
        """What casim_enc_add_pods (ABI 11) has no column for: node affinity, host ports, (anti-)affinity terms, spread constraints, marks, the
        grouping digest — per-pod calls on spec record `s`, after its namespace / requests / labels / tolerations / nodeSelector / fastpath
        requests are in (pod by pod: add_pod_spec; in bulk: add_pegs)."""
        h = self._h
        unknown = [r for r, v in pod.requests.items() if r not in self.lanes and v] if not self.named_lanes else []
        for r in pod.node_affinity:
            check(lib.casim_enc_pod_add_node_affinity_req(h, s, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)))
        for term in (pod.node_affinity_terms or [NodeSelectorTerm()] if pod.node_affinity_terms is not None else []):
            t = lib.casim_enc_pod_add_node_affinity_term(h, s)
            if t < 0:
                check(t, "casim_enc_pod_add_node_affinity_term")
            for is_field, reqs in ((0, term.match_expressions), (1, term.match_fields)):
                for r in reqs:
                    check(lib.casim_enc_node_term_add_requirement(h, s, t, is_field, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)))
        for p in pod.host_ports:
            check(lib.casim_enc_pod_add_host_port(h, s, _b(p.host_ip), _b(p.protocol), int(p.host_port)))
        for term in pod.anti_affinity:
            t = lib.casim_enc_pod_add_anti_affinity_term(h, s, _b(term.topology_key), _strs(term.namespaces), len(term.namespaces))
            if t < 0:
                check(t, "casim_enc_pod_add_anti_affinity_term")
            for r in term.requirements():
                check(lib.casim_enc_term_add_requirement(h, s, t, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)))
            if term.namespace_selector is not None:
                check(lib.casim_enc_term_set_namespace_selector(h, s, t))
                for r in term.namespace_selector:
                    check(lib.casim_enc_term_add_namespace_requirement(h, s, t, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)))
        for term in pod.affinity:   # required pod affinity: domain rules of kind 2 in per-node mode, delegated in template mode
            t = lib.casim_enc_pod_add_affinity_term(h, s, _b(term.topology_key), _strs(term.namespaces), len(term.namespaces))
            if t < 0:
                check(t, "casim_enc_pod_add_affinity_term")
            if term.namespace_selector is not None:
                check(lib.casim_enc_aff_term_set_namespace_selector(h, s, t))
                for r in term.namespace_selector:
                    check(lib.casim_enc_aff_term_add_namespace_requirement(h, s, t, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)))
            for r in term.requirements():
                check(lib.casim_enc_aff_term_add_requirement(h, s, t, _b(r.key), _b(r.operator), _strs(r.values), len(r.values)))
        for sc in pod.spread_constraints:   # evaluated on the device in per-node mode, flagged UNSUPPORTED by finalize otherwise
            ci = lib.casim_enc_pod_add_spread_constraint(h, s, int(sc.max_skew), _b(sc.topology_key), int(sc.min_domains))
            if ci < 0:
                check(ci, "casim_enc_pod_add_spread_constraint")
            for k, v in sc.effective_match_labels(pod.labels).items():
                check(lib.casim_enc_spread_add_requirement(h, s, ci, _b(k), b"In", _strs([v]), 1))
            if sc.node_taints_policy == "Honor":
                check(lib.casim_enc_spread_set_taints_policy(h, s, ci, 1))
            elif sc.node_taints_policy != "Ignore":
                check(lib.casim_enc_pod_mark_unsupported(h, s, b"topologySpreadConstraints: nodeTaintsPolicy"))
            if sc.node_affinity_policy == "Ignore":
                check(lib.casim_enc_spread_set_affinity_policy(h, s, ci, 0))
            elif sc.node_affinity_policy != "Honor":
                check(lib.casim_enc_pod_mark_unsupported(h, s, b"topologySpreadConstraints: nodeAffinityPolicy"))
        if pod.topology_spread and not pod.spread_constraints:
            check(lib.casim_enc_pod_mark_unsupported(h, s, b"topologySpreadConstraints"))
        if pod.unsupported_reason:
            check(lib.casim_enc_pod_mark_unsupported(h, s, _b(pod.unsupported_reason)))
        if unknown:
            check(lib.casim_enc_pod_mark_unsupported(h, s, _b("resource not in lanes: " + ",".join(unknown))))
        if not digest:
            return
        # grouping only: the fields PodSpecSemanticallyEqual compares that the encoder has no call for
        extra = repr((pod.spec_extra, pod.has_containers, pod.topology_spread, tuple(tuple(c.match_label_keys) for c in pod.spread_constraints),
                      tuple(tuple(sorted(c.match_labels.items())) for c in pod.spread_constraints)))
        check(lib.casim_enc_pod_set_spec_extra(h, s, _b(extra)))

    def add_pegs(self, pegs: Sequence[PodEquivalenceGroup], digests: bool = True) -> List[int]:
        """The PEGs of a loop through casim_enc_add_pods (ABI 11): ONE crossing for every exemplar's namespace, requests, labels, tolerations,
        nodeSelector and fastpath requests, strings by index into an interned table; the rarer fields follow pod by pod (_pod_rest).  Builds the
        same records, PEG ids and tables as add_peg called in a loop (tests/test_bulk_pods.py); what integration/go/gpubinpacking/encode.go
        (session.pegs) does.  An exemplar that already has a spec record (two PEGs, one pod object) ends the run and takes add_peg.
        digests=False leaves out casim_enc_pod_set_spec_extra (read by casim_enc_group_pods only: the Go shim's Estimate path never sets it — its PEGs
        arrive grouped); keep the default when the same pod objects may go through group_pods later."""
        import numpy as np
        out: List[int] = []
        run: List[Tuple[PodEquivalenceGroup, Pod]] = []
        in_run = set()

        def flush():
            if not run:
                return
            strs: Dict[str, int] = {}
            table: List[bytes] = []

            def sid(x) -> int:
                if x is None:
                    return -1
                i = strs.get(x)
                if i is None:
                    i = strs[x] = len(table)
                    table.append(_b(x))
                return i
            n = len(run)
            R = len(self.lanes)
            ns = np.empty(n, np.int32); req = np.zeros((n, R), np.int64); fp = np.empty((n, 2), np.float64); cnt = np.empty(n, np.int32)
            loff, lk, lv = [0], [], []
            toff, tk, to, tv, te = [0], [], [], [], []
            soff, sk, sv = [0], [], []
            for i, (pg, ex) in enumerate(run):
                ns[i] = sid(ex.namespace)
                for r, name in enumerate(self.lanes):
                    req[i, r] = int(ex.requests.get(name, 0))
                fp[i] = ex.fastpath_requests()
                cnt[i] = len(pg.pods)
                for k, v in ex.labels.items():
                    lk.append(sid(k)); lv.append(sid(v))
                loff.append(len(lk))
                for t in ex.tolerations:
                    tk.append(sid(t.key)); to.append(sid(t.operator)); tv.append(sid(t.value)); te.append(sid(t.effect))
                toff.append(len(tk))
                for k, v in ex.node_selector.items():
                    sk.append(sid(k)); sv.append(sid(v))
                soff.append(len(sk))
            cols = [np.asarray(a, np.int32) for a in (loff, lk, lv, toff, tk, to, tv, te, soff, sk, sv)]
            keep = [ns, req, fp, cnt] + cols
            ptr = lambda a: a.ctypes.data_as(_abi.i32p)   # noqa: E731
            pc = _abi.PodColumns(n_pods=n, n_strings=len(table), strings=(C.c_char_p * max(len(table), 1))(*table), ns=ptr(ns),
                                 req=req.ctypes.data_as(_abi.i64p), fastpath_req=fp.ctypes.data_as(_abi.f64p), peg_count=ptr(cnt),
                                 label_off=ptr(cols[0]), label_key=ptr(cols[1]), label_val=ptr(cols[2]),
                                 tol_off=ptr(cols[3]), tol_key=ptr(cols[4]), tol_op=ptr(cols[5]), tol_value=ptr(cols[6]), tol_effect=ptr(cols[7]),
                                 sel_off=ptr(cols[8]), sel_key=ptr(cols[9]), sel_val=ptr(cols[10]))
            peg_ids = np.empty(n, np.int32)
            first = lib.casim_enc_add_pods(self._h, C.byref(pc), ptr(peg_ids))
            if first < 0:
                check(first, "casim_enc_add_pods")
            del keep
            for i, (pg, ex) in enumerate(run):
                self._keep.append(ex)
                self._named_requests(ex, first + i)
                self._pod_rest(ex, first + i, digest=digests)
                self._spec_of[id(ex)] = first + i
                out.append(int(peg_ids[i]))
            self.n_pegs += n
            run.clear(); in_run.clear()

        for pg in pegs:
            ex = pg.exemplar()
            if ex is None or id(ex) in self._spec_of or id(ex) in in_run:
                flush()
                out.append(self.add_peg(pg))
                continue
            run.append((pg, ex)); in_run.add(id(ex))
        flush()
        return out

    def add_peg(self, peg: PodEquivalenceGroup) -> int:
        ex = peg.exemplar()
        if ex is None:
            # an empty group still occupies a slot of the orderer (score 0, Exemplar() == nil)
            ex = Pod(name="<empty>")
        s = self.add_pod_spec(ex)
        g = lib.casim_enc_add_peg(self._h, s, len(peg.pods))
        if g < 0:
            check(g, "casim_enc_add_peg")
        self.n_pegs += 1
        return g

    def group_pods(self, pods: Sequence[Pod], share_specs: bool = False):
        """equivalence.BuildPodGroups behind the C ABI (casim_enc_group_pods, CA/core/scaleup/equivalence/groups.go:39-104):
        returns (group id per pod, number of groups).  Every pod OBJECT gets a spec record of its own unless share_specs (then pods
        with equal Pod.spec_key() share one, the way a shim that interns specs would call it): the native side compares content."""
        import numpy as np
        n = len(pods)
        if share_specs:
            seen: Dict[object, int] = {}
            spec = np.empty(n, np.int32)
            for i, p in enumerate(pods):
                k = p.spec_key()
                if k not in seen:
                    seen[k] = self.add_pod_spec(p)
                spec[i] = seen[k]
        else:
            spec = np.array([self.add_pod_spec(p) for p in pods], np.int32)
        uids = (C.c_char_p * max(n, 1))(*[_b(p.controller_uid) if p.controller_uid else None for p in pods])
        ds = np.array([1 if p.daemonset else 0 for p in pods], np.uint8)
        out = np.empty(n, np.int32)
        ng = C.c_int32(0)
        check(lib.casim_enc_group_pods(self._h, n, spec.ctypes.data_as(_abi.i32p), uids, ds.ctypes.data_as(_abi.u8p),
                                       out.ctypes.data_as(_abi.i32p), C.byref(ng)), "casim_enc_group_pods")
        self._last_group_specs = spec
        return out, int(ng.value)

    def add_grouped_pegs(self, pods: Sequence[Pod], group, n_groups: int):
        """One PEG per equivalence group (exemplar = first pod, count = size): casim_enc_add_grouped_pegs."""
        import numpy as np
        spec = np.array([self.add_pod_spec(p) for p in pods], np.int32)
        group = np.ascontiguousarray(group, np.int32)
        ids = np.empty(n_groups, np.int32)
        first = lib.casim_enc_add_grouped_pegs(self._h, len(pods), spec.ctypes.data_as(_abi.i32p), group.ctypes.data_as(_abi.i32p), n_groups,
                                               ids.ctypes.data_as(_abi.i32p))
        if first < 0:
            check(first, "casim_enc_add_grouped_pegs")
        self.n_pegs += n_groups
        return ids

    def add_resource_pegs(self, requests, counts, namespace: str = "default"):
        """Bulk form (casim_enc_add_resource_pegs): `requests` is an [n][R] integer array of lanes,
        `counts` an [n] array.  Returns the PEG ids (a contiguous range)."""
        import numpy as np
        req = np.ascontiguousarray(requests, dtype=np.int64)
        cnt = np.ascontiguousarray(counts, dtype=np.int32)
        if req.ndim != 2 or req.shape[1] != len(self.lanes) or cnt.shape[0] != req.shape[0]:
            raise ValueError("requests must be [n][len(lanes)] and counts [n]")
        n = int(req.shape[0])
        first = lib.casim_enc_add_resource_pegs(self._h, _b(namespace), n, req.ctypes.data_as(_abi.i64p),
                                                cnt.ctypes.data_as(_abi.i32p), None)
        if first < 0:
            check(first, "casim_enc_add_resource_pegs")
        self.n_pegs += n
        return range(first, first + n)

    # ---- node groups -----------------------------------------------------------------------
    def add_group(self, template: NodeInfo, max_nodes: int = 0, existing_nodes: int = 0, last_index: int = 0,
                  pegs: Optional[Sequence[int]] = None) -> int:
        node = template.node
        unknown = [r for r in node.allocatable if r not in self.lanes and r != "pods"]
        del unknown  # extra node resources nobody requests are irrelevant to the Filters
        g = lib.casim_enc_add_group(self._h, _b(node.name), self._lane_vector(node.allocatable), node.allowed_pods(),
                                    int(node.capacity.get(RES_CPU, 0)), int(node.capacity.get(RES_MEMORY, 0)),
                                    int(node.unschedulable))
        if g < 0:
            check(g, "casim_enc_add_group")
        if self.named_lanes:
            for name, v in node.allocatable.items():
                if name not in (RES_CPU, RES_MEMORY, RES_EPHEMERAL, "pods"):
                    rc = lib.casim_enc_group_set_allocatable(self._h, g, _b(name), int(v))
                    if rc < 0:
                        check(rc, "casim_enc_group_set_allocatable")
        for k, v in node.labels.items():
            check(lib.casim_enc_group_add_label(self._h, g, _b(k), _b(v)))
        for t in node.taints:
            check(lib.casim_enc_group_add_taint(self._h, g, _b(t.key), _b(t.value), _b(t.effect)))
        check(lib.casim_enc_group_set_limits(self._h, g, int(max_nodes), int(existing_nodes), int(last_index)))
        for p in template.pods:
            check(lib.casim_enc_group_add_preloaded_pod(self._h, g, self.add_pod_spec(p)))
        if pegs is not None:
            arr = (C.c_int32 * max(len(pegs), 1))(*pegs)
            check(lib.casim_enc_group_set_pegs(self._h, g, arr, len(pegs)))
        self.n_groups += 1
        return g

    def add_running_pods(self, pods_of_group: Sequence[Sequence[Pod]], groups: Optional[Sequence[int]] = None):
        """casim_enc_add_running_pods: the running pods of many nodes in ONE call (plain pods: namespace, requests, labels; anything with
        tolerations / selectors / ports / terms goes through add_pod_spec + casim_enc_group_add_preloaded_pod as before).
        pods_of_group[k] = pods running on group groups[k] (default: group k).  Returns the spec ids, one list per group."""
        strings, index = [], {}

        def sid(x: str) -> int:
            if x not in index:
                index[x] = len(strings); strings.append(x)
            return index[x]

        out = [[] for _ in pods_of_group]
        gidx, ns, req, off, lk, lv, plain = [], [], [], [0], [], [], []

        def flush():
            # (spec ids follow the order of the pods: a pod that needs the per-pod calls ends the batch collected so far)
            if not plain:
                return
            arr = lambda t, v: (t * max(len(v), 1))(*v)   # noqa: E731
            cs = (C.c_char_p * max(len(strings), 1))(*[_b(x) for x in strings])
            first = lib.casim_enc_add_running_pods(self._h, len(plain), arr(C.c_int32, gidx), arr(C.c_int32, ns), arr(C.c_int64, [int(x) for x in req]),
                                                   arr(C.c_int32, off), arr(C.c_int32, lk), arr(C.c_int32, lv), cs, len(strings))
            if first < 0:
                check(first, "casim_enc_add_running_pods")
            for i, (k, p) in enumerate(plain):
                self._spec_of[id(p)] = first + i
                out[k].append(first + i)
            del gidx[:], ns[:], req[:], lk[:], lv[:], plain[:]
            off[:] = [0]

        for k, pods in enumerate(pods_of_group):
            g = k if groups is None else int(groups[k])
            for p in pods:
                # plain = nothing but namespace, labels and requests on known lanes differs from a default Pod
                bare = dataclasses.replace(p, name="", uid="", controller_uid="", daemonset=False, priority=0)
                is_plain = bare == Pod(name="", namespace=p.namespace, labels=p.labels, requests=p.requests) and \
                    not [r for r, v in p.requests.items() if r not in self.lanes and v]
                if not is_plain or id(p) in self._spec_of:
                    flush()
                    s = self.add_pod_spec(p)
                    check(lib.casim_enc_group_add_preloaded_pod(self._h, g, s))
                    out[k].append(s)
                    continue
                self._keep.append(p)
                gidx.append(g); ns.append(sid(p.namespace)); req.extend(self._lane_vector(p.requests)[:len(self.lanes)])
                for a, b in p.labels.items():
                    lk.append(sid(a)); lv.append(sid(b))
                off.append(len(lk)); plain.append((k, p))
        flush()
        return out

    def add_existing_pod(self, pod: Pod, node_labels: Dict[str, str]):
        ks, vs = list(node_labels.keys()), list(node_labels.values())
        check(lib.casim_enc_add_existing_pod(self._h, self.add_pod_spec(pod), _strs(ks), _strs(vs), len(ks)))

    # ---- tables ----------------------------------------------------------------------------
    def finalize(self):
        check(lib.casim_enc_finalize(self._h), "casim_enc_finalize")
        self.finalized = True
        if self.named_lanes:   # which resource each lane of the tables stands for
            self.lanes = tuple((lib.casim_enc_lane_name(self._h, i) or b"").decode() for i in range(lib.casim_enc_lane_count(self._h)))
        self.pegs = _abi.Pegs()
        self.groups = _abi.Groups()
        check(lib.casim_enc_tables(self._h, C.byref(self.pegs), C.byref(self.groups)))
        self.rules = _abi.DomainRules()
        check(lib.casim_enc_domain_rules(self._h, C.byref(self.rules)))
        self.port_block = lib.casim_enc_port_block(self._h)
        return self.pegs, self.groups

    # ---- incremental re-encode (per-node mode; casim.h "incremental re-encode") ----------------------
    def begin_update(self):
        check(lib.casim_enc_begin_update(self._h), "casim_enc_begin_update")

    def reset_group(self, g: int, template: NodeInfo):
        """Describe node `g` again: casim_enc_group_reset + the same calls add_group makes for its labels, taints and running pods."""
        node = template.node
        check(lib.casim_enc_group_reset(self._h, int(g), self._lane_vector(node.allocatable), node.allowed_pods(), int(node.capacity.get(RES_CPU, 0)),
                                        int(node.capacity.get(RES_MEMORY, 0)), int(node.unschedulable)), "casim_enc_group_reset")
        if self.named_lanes:
            for name, v in node.allocatable.items():
                if name not in (RES_CPU, RES_MEMORY, RES_EPHEMERAL, "pods"):
                    rc = lib.casim_enc_group_set_allocatable(self._h, int(g), _b(name), int(v))
                    if rc < 0:
                        check(rc, "casim_enc_group_set_allocatable")
        for k, v in node.labels.items():
            check(lib.casim_enc_group_add_label(self._h, g, _b(k), _b(v)))
        for t in node.taints:
            check(lib.casim_enc_group_add_taint(self._h, g, _b(t.key), _b(t.value), _b(t.effect)))
        for p in template.pods:
            check(lib.casim_enc_group_add_preloaded_pod(self._h, g, self.add_pod_spec(p)))

    def set_peg_count(self, peg: int, count: int):
        check(lib.casim_enc_set_peg_count(self._h, int(peg), int(count)), "casim_enc_set_peg_count")

    def refinalize(self):
        """(True, changed node indices) after an incremental update, (False, None) when the update needs a full finalize — which
        this method then runs on the same encoder."""
        import numpy as np
        cap = max(self.n_groups, 1)
        changed = np.zeros(cap, np.int32)
        n = C.c_int32(0)
        rc = lib.casim_enc_refinalize(self._h, changed.ctypes.data_as(_abi.i32p), cap, C.byref(n))
        if rc == _abi.ENC_NEEDS_FULL:
            self.finalized = False
            self.finalize()
            return False, None
        check(rc, "casim_enc_refinalize")
        return True, changed[:n.value].copy()

    def group_rows(self, groups) -> _abi.Groups:
        import numpy as np
        idx = np.ascontiguousarray(groups, np.int32)
        rows = _abi.Groups()
        check(lib.casim_enc_group_rows(self._h, idx.ctypes.data_as(_abi.i32p), int(idx.shape[0]), C.byref(rows)), "casim_enc_group_rows")
        return rows

    def dict_sizes(self):
        out = (C.c_int32 * 4)()
        check(lib.casim_enc_dict_sizes(self._h, out))
        return {"taints": out[0], "label_requirements": out[1], "node_bits": out[2], "zone_bits": out[3]}
