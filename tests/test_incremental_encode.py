"""casim_enc_begin_update / _group_reset / _refinalize: the node table and the domain-rule counters after an incremental update of
a few nodes are IDENTICAL, column by column, to a full finalize of the same objects; updates that would change a dictionary say
CASIM_ENC_NEEDS_FULL and the full finalize that follows gives the same tables as a fresh encoder.  CPU only (host encoder)."""
import ctypes as C
import copy

import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.encoder import Encoder
from kubernetes_autoscaler_amd.objects import NodeInfo, PodEquivalenceGroup, Taint


def _encode(nodes, pods, extra_specs=()):
    """one encoder session in per-node mode: classes = pending specs + the specs of running pods, fixed order"""
    enc = Encoder(explicit_self_exclusion=True)
    class_of = {}
    for p in list(pods) + [q for info in nodes for q in info.pods] + list(extra_specs):
        k = p.spec_key()
        if k not in class_of:
            class_of[k] = enc.add_peg(PodEquivalenceGroup(pods=[p]))
    for info in nodes:
        enc.add_group(info, pegs=[])
    enc.finalize()
    return enc


def _columns(enc):
    """every column of the node table and of the domain rules, as bytes"""
    from kubernetes_autoscaler_amd._ffi import lib
    enc.pegs, enc.groups, enc.rules = _abi.Pegs(), _abi.Groups(), _abi.DomainRules()   # (fresh views: a full finalize re-allocates the columns)
    assert lib.casim_enc_tables(enc._h, C.byref(enc.pegs), C.byref(enc.groups)) == 0
    assert lib.casim_enc_domain_rules(enc._h, C.byref(enc.rules)) == 0
    g, p, r = enc.groups, enc.pegs, enc.rules
    NG, R = g.n_groups, p.n_res
    out = {}
    def col(name, ptr, n, dt):
        out[name] = b"" if (not ptr or n == 0) else np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=False).tobytes()
    col("alloc", g.alloc, NG * R, np.int64); col("init_req", g.init_req, NG * R, np.int64); col("allowed", g.allowed_pods, NG, np.int32)
    col("init_pods", g.init_pods, NG, np.int32); col("flags", g.flags, NG, np.uint32); col("taint", g.taint_mask, NG * p.w_taint, np.uint64)
    col("label", g.label_mask, NG * p.w_label, np.uint64); col("init_excl", g.init_excl, NG * p.w_excl, np.uint64)
    col("count", p.count, p.n_pegs, np.int32)
    out["dims"] = (p.n_pegs, NG, p.w_taint, p.w_label, p.w_excl, r.n_keys, r.n_rules, r.n_elig_rows)
    if r.n_rules > 0:
        tot = int(r.rule_offset[r.n_rules])
        col("node_domain", r.node_domain, r.n_keys * NG, np.int32); col("count_init", r.count_init, tot, np.int32)
        col("domain_exists", r.domain_exists, tot, np.uint8); col("domain_nodes", r.domain_nodes, tot, np.int32)
        col("node_contrib", r.node_contrib, r.n_rules * NG, np.int32); col("elig", r.elig_bits, r.n_elig_rows * ((NG + 63) // 64), np.uint64)
        col("rule_offset", r.rule_offset, r.n_rules + 1, np.int64); col("inc_rule", r.inc_rule, int(r.inc_off[r.n_classes]), np.int32)
    return out


def _churn(nodes, rng, frac=0.2):
    """pods move between nodes, a node changes a label value that already exists elsewhere, a node turns unschedulable"""
    new = [NodeInfo(copy.deepcopy(info.node), list(info.pods)) for info in nodes]
    changed = set()
    n = len(new)
    for _ in range(max(1, int(n * frac))):
        a, b = rng.randrange(n), rng.randrange(n)
        if new[a].pods:
            q = new[a].pods.pop(rng.randrange(len(new[a].pods)))
            new[b].pods.append(q)
            changed.update((a, b))
    c = rng.randrange(n)
    new[c].node.unschedulable = not new[c].node.unschedulable
    changed.add(c)
    zones = sorted({info.node.labels.get("topology.kubernetes.io/zone") for info in nodes if "topology.kubernetes.io/zone" in info.node.labels})
    if zones:
        d = rng.randrange(n)
        if "topology.kubernetes.io/zone" in new[d].node.labels:
            new[d].node.labels["topology.kubernetes.io/zone"] = zones[rng.randrange(len(zones))]
            changed.add(d)
    return new, sorted(changed)


@pytest.mark.parametrize("seed", range(80))
def test_incremental_update_equals_a_full_finalize(seed):
    import random
    from kubernetes_autoscaler_amd._ffi import lib
    w = workloads.fuzz_pending_domains(9000 + seed) if seed % 2 else workloads.fuzz_pending(9000 + seed)
    if len(w.nodes) < 2:
        pytest.skip("needs two nodes")
    a, b = _encode(w.nodes, w.pods), _encode(w.nodes, w.pods)      # two identical sessions
    new_nodes, changed = _churn(w.nodes, random.Random(seed))
    for enc in (a, b):
        enc.begin_update()
        for m in changed:
            enc.reset_group(m, new_nodes[m])
    ok, idx = a.refinalize()                                        # incremental (or its own fallback)
    assert lib.casim_enc_finalize(b._h) == 0                        # full, on the same objects
    got, want = _columns(a), _columns(b)
    assert got.keys() == want.keys()
    for k in want:
        assert got[k] == want[k], (k, ok, changed)
    if ok:
        assert sorted(int(x) for x in idx) == changed
        rows = a.group_rows(changed)
        assert rows.n_groups == len(changed)
        for j, m in enumerate(changed):
            assert rows.init_pods[j] == a.groups.init_pods[m] and rows.allowed_pods[j] == a.groups.allowed_pods[m]
        # a second session on top of the first: back to the original nodes
        a.begin_update(); b.begin_update()
        for enc in (a, b):
            for m in changed:
                enc.reset_group(m, w.nodes[m])
        ok2, _ = a.refinalize()
        assert lib.casim_enc_finalize(b._h) == 0
        got, want = _columns(a), _columns(b)
        for k in want:
            assert got[k] == want[k], ("second session", k, ok2)
    a.close(); b.close()


def test_a_changed_list_that_does_not_fit_is_refused_before_anything_is_recomputed():
    """with capacity < changed nodes the list used to be truncated and the dirty set cleared — rows the caller could never ship
    to casim_cluster_update_nodes.  Now: CASIM_ERR_INVALID, the needed capacity in n_changed_out, the session still open; the retry with
    room gives the tables of a full finalize."""
    import random
    from kubernetes_autoscaler_amd._ffi import lib
    w = workloads.fuzz_pending(9007)
    a, b = _encode(w.nodes, w.pods), _encode(w.nodes, w.pods)
    new_nodes, changed = _churn(w.nodes, random.Random(3))
    assert len(changed) >= 2
    for enc in (a, b):
        enc.begin_update()
        for m in changed:
            enc.reset_group(m, new_nodes[m])
    small = np.zeros(1, np.int32)
    n = C.c_int32(0)
    assert lib.casim_enc_refinalize(a._h, small.ctypes.data_as(_abi.i32p), 1, C.byref(n)) == _abi.ERR_INVALID
    assert n.value == len(changed)
    ok, idx = a.refinalize()                                        # the session is still open: same call with room
    assert lib.casim_enc_finalize(b._h) == 0
    got, want = _columns(a), _columns(b)
    for k in want:
        assert got[k] == want[k], k
    if ok:
        assert sorted(int(x) for x in idx) == changed
    a.close(); b.close()


def test_updates_that_touch_a_dictionary_ask_for_a_full_finalize():
    w = workloads.fuzz_pending_domains(9100)
    enc = _encode(w.nodes, w.pods)
    # a taint nobody had: the PEGs' toleration masks need a new bit
    info = NodeInfo(copy.deepcopy(w.nodes[0].node), list(w.nodes[0].pods))
    info.node.taints = list(info.node.taints) + [Taint("brand-new", "x", "NoSchedule")]
    enc.begin_update(); enc.reset_group(0, info)
    n = C.c_int32(0)
    from kubernetes_autoscaler_amd._ffi import lib
    assert lib.casim_enc_refinalize(enc._h, None, 0, C.byref(n)) == _abi.ENC_NEEDS_FULL
    enc.finalized = False
    enc.finalize()                                            # the session's fallback: everything described so far is still there
    assert enc.groups.n_groups == len(w.nodes) and enc.dict_sizes()["taints"] >= 1
    t = np.ctypeslib.as_array(enc.groups.taint_mask, shape=(enc.groups.n_groups * enc.pegs.w_taint,))
    assert t[:enc.pegs.w_taint].any()                          # node 0 carries the new taint's bit now
    # outside a session the encoder stays frozen
    assert lib.casim_enc_group_reset(enc._h, 0, enc._lane_vector(info.node.allocatable), 10, 0, 0, 0) == _abi.ERR_INVALID
    assert lib.casim_enc_group_add_label(enc._h, 0, b"a", b"b") == _abi.ERR_INVALID
    enc.close()


def _encode_per_pod_records(nodes, pods, copies):
    """like _encode, but every running pod is described through a spec record of its OWN (a deep copy: the binding caches by object
    identity) when `copies` is set — what a shim without a spec cache hands over at cluster scale"""
    enc = Encoder(explicit_self_exclusion=True)
    for p in pods:
        enc.add_peg(PodEquivalenceGroup(pods=[p]))
    for info in nodes:
        running = [copy.deepcopy(q) for q in info.pods] if copies else list(info.pods)
        enc.add_group(NodeInfo(info.node, running), pegs=[])
    enc.finalize()
    return enc


@pytest.mark.parametrize("seed", range(40))
def test_content_classes_of_running_pods_threads_and_record_sharing(seed, monkeypatch):
    """casim_enc_finalize puts the running pods' spec records into content classes (namespace + labels; terms / host ports: a class of
    its own) on up to four threads and evaluates selectors once per class.  The tables must not depend on the number of threads, and
    the domain-rule columns must not depend on whether equal running pods share one spec record or bring one each."""
    w = workloads.fuzz_pending_domains(9300 + seed) if seed % 4 else workloads.fuzz_pending(9300 + seed)
    monkeypatch.setenv("CASIM_HOST_THREADS", "1")
    one = _encode_per_pod_records(w.nodes, w.pods, copies=True)
    want = _columns(one)
    monkeypatch.setenv("CASIM_HOST_THREADS", "4"); monkeypatch.setenv("CASIM_HOST_GRAIN", "1")
    four = _encode_per_pod_records(w.nodes, w.pods, copies=True)
    got = _columns(four)
    assert got.keys() == want.keys()
    for k in want:
        assert got[k] == want[k], ("threads", k)
    monkeypatch.delenv("CASIM_HOST_GRAIN")
    shared = _encode_per_pod_records(w.nodes, w.pods, copies=False)
    sh = _columns(shared)
    for k in ("node_domain", "count_init", "domain_exists", "domain_nodes", "node_contrib", "elig", "rule_offset", "inc_rule", "alloc", "init_req",
              "init_pods", "flags", "taint", "label"):
        if k in want or k in sh:
            assert sh.get(k) == want.get(k), ("shared records", k)
    for enc in (one, four, shared):
        enc.close()


@pytest.mark.parametrize("seed", range(30))
def test_bulk_running_pods_give_the_same_tables(seed):
    """casim_enc_add_running_pods (the running pods of every node in ONE call; pods with tolerations / ports / terms still one by one)
    against the per-pod calls: every column of the node table and of the domain rules identical; bad indices add nothing."""
    import ctypes as C
    from kubernetes_autoscaler_amd._ffi import lib
    w = workloads.fuzz_pending_domains(9500 + seed) if seed % 3 else workloads.fuzz_pending(9500 + seed)
    one = _encode_per_pod_records(w.nodes, w.pods, copies=True)
    want = _columns(one)
    enc = Encoder(explicit_self_exclusion=True)
    for p in w.pods:
        enc.add_peg(PodEquivalenceGroup(pods=[p]))
    for info in w.nodes:
        enc.add_group(NodeInfo(info.node, []), pegs=[])
    before = lib.casim_enc_add_running_pods(enc._h, 0, None, None, None, None, None, None, None, 0)   # (0 pods: the next spec id)
    bad = (C.c_int32 * 1)(len(w.nodes) + 5)
    zero = (C.c_int32 * 2)(0, 0)
    strs = (C.c_char_p * 1)(b"default")
    req = (C.c_int64 * 8)(*([1] * 8))
    assert lib.casim_enc_add_running_pods(enc._h, 1, bad, zero, req, zero, None, None, strs, 1) < 0            # group out of range
    assert lib.casim_enc_add_running_pods(enc._h, 0, None, None, None, None, None, None, None, 0) == before  # nothing was added
    ids = enc.add_running_pods([[copy.deepcopy(q) for q in info.pods] for info in w.nodes])
    assert [len(x) for x in ids] == [len(info.pods) for info in w.nodes]
    enc.finalize()
    got = _columns(enc)
    assert got.keys() == want.keys()
    for k in want:
        assert got[k] == want[k], k
    one.close(); enc.close()
