"""casim_enc_add_pods (ABI 11, include/casim.h: casim_pod_columns): the pods of a loop in ONE crossing over an interned string table.

The claim is "the same records as the per-pod calls": so the tables (every column of casim_pegs / casim_groups, FNV-1a as tools/casim_native
hashes them) and the domain rules of an encoder fed through Encoder.add_pegs must equal those of one fed PEG by PEG — over the bench configs,
the estimate fuzz corpus (taints, tolerations with every operator, selectors, affinity terms, host ports, extended resources by lane and
by name), and a shared-exemplar case; a bad column adds nothing; the trace of the bulk call replays natively to the same tables."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

from kubernetes_autoscaler_amd import _abi, workloads  # noqa: E402
from kubernetes_autoscaler_amd._ffi import lib  # noqa: E402
from kubernetes_autoscaler_amd.encoder import Encoder  # noqa: E402
from kubernetes_autoscaler_amd.objects import Pod, PodEquivalenceGroup, Toleration  # noqa: E402
import native_trace as nt  # noqa: E402
from encoder_tables_corpus import rules_hash  # noqa: E402


def build(w, bulk, named_lanes=False):
    enc = Encoder(lanes=w.lanes, named_lanes=named_lanes)
    if bulk:
        ids = enc.add_pegs(w.pegs)
    else:
        ids = [enc.add_peg(pg) for pg in w.pegs]
    assert list(ids) == list(range(len(w.pegs)))
    for info in w.existing:
        for p in info.pods:
            enc.add_existing_pod(p, info.node.labels)
    for g in w.groups:
        enc.add_group(g.template, max_nodes=g.max_nodes, existing_nodes=len(w.existing), last_index=g.last_index,
                      pegs=list(g.pegs) if g.pegs is not None else None)
    enc.finalize()
    return enc


def same_tables(w, named_lanes=False):
    a, b = build(w, False, named_lanes), build(w, True, named_lanes)
    assert nt.tables_fnv(a.pegs, a.groups) == nt.tables_fnv(b.pegs, b.groups)
    assert rules_hash(a) == rules_hash(b)
    assert a.dict_sizes() == b.dict_sizes()
    assert a.lanes == b.lanes
    a.close(); b.close()


@pytest.mark.parametrize("name", ["config_c0", "config_c1", "config_c2", "config_c3", "config_c4", "config_r2", "config_retry_mix"])
def test_bulk_and_per_pod_calls_build_the_same_tables_on_the_bench_configs(name):
    same_tables(getattr(workloads, name)())


@pytest.mark.parametrize("base", range(0, 240, 40))
def test_bulk_and_per_pod_calls_build_the_same_tables_on_the_fuzz_corpus(base):
    for seed in range(base, base + 40):
        same_tables(workloads.fuzz(seed))
        same_tables(workloads.fuzz(seed), named_lanes=True)   # the Go shim's sequence: three positional lanes, the rest by name
    for seed in range(base // 4, base // 4 + 10):
        same_tables(workloads.fuzz_singleton_runs(seed))
        same_tables(workloads.fuzz_estimate_domains(seed))


def test_a_shared_exemplar_and_an_empty_group_take_the_per_pod_path():
    a = Pod(name="a", namespace="x", requests={"cpu": 100, "memory": 1 << 20}, labels={"app": "a"},
            tolerations=[Toleration(key="k", operator="Exists")])
    b = Pod(name="b", namespace="x", requests={"cpu": 200, "memory": 2 << 20}, node_selector={"pool": "p"})
    pegs = [PodEquivalenceGroup([a] * 3), PodEquivalenceGroup([b] * 2), PodEquivalenceGroup([a] * 4), PodEquivalenceGroup([]), PodEquivalenceGroup([b])]
    w = workloads.config_c0()
    w = workloads.Workload("shared", pegs, w.groups)
    same_tables(w)
    enc = Encoder()
    ids = enc.add_pegs(pegs)
    assert ids == [0, 1, 2, 3, 4] and enc.n_pegs == 5
    assert enc.add_pod_spec(a) == 0 and enc.add_pod_spec(b) == 1    # one record per pod object, as with add_peg


def _columns(n, n_strings, **over):
    keep = {}

    def arr(name, values, dtype=np.int32):
        keep[name] = np.ascontiguousarray(values, dtype)
        return keep[name].ctypes.data_as({np.int32: _abi.i32p, np.int64: _abi.i64p, np.float64: _abi.f64p}[dtype])
    table = [b"default", b"k", b"Exists", b"v", b"NoSchedule"][:n_strings]
    keep["strings"] = (C.c_char_p * max(len(table), 1))(*table)
    pc = _abi.PodColumns(n_pods=n, n_strings=n_strings, strings=keep["strings"], ns=arr("ns", over.get("ns", [0] * n)),
                         req=arr("req", [[100, 1 << 20]] * n, np.int64), peg_count=arr("cnt", over.get("cnt", [1] * n)),
                         label_off=arr("loff", over.get("loff", [0] * (n + 1))), label_key=arr("lk", over.get("lk", [0])), label_val=arr("lv", over.get("lv", [0])),
                         tol_off=arr("toff", over.get("toff", [0] * (n + 1))), tol_key=arr("tk", over.get("tk", [1])), tol_op=arr("to", over.get("to", [2])),
                         tol_value=arr("tv", over.get("tv", [-1])), tol_effect=arr("te", over.get("te", [4])))
    return pc, keep


@pytest.mark.parametrize("bad", [dict(ns=[0, 9]), dict(ns=[0, -2]), dict(cnt=[1, -2]), dict(loff=[0, 1, 0]), dict(loff=[1, 1, 1]),
                                 dict(loff=[0, 1, 1], lk=[7]), dict(toff=[0, 0, 1], to=[5]), dict(toff=[0, 0, 1], tk=[-2])])
def test_a_bad_column_adds_nothing(bad):
    enc = Encoder()
    before = lib.casim_enc_add_pod_spec(enc._h, b"default", (C.c_int64 * _abi.MAX_RES)(1, 1))
    pc, keep = _columns(2, 5, **bad)
    assert lib.casim_enc_add_pods(enc._h, C.byref(pc), None) == _abi.ERR_INVALID
    after = lib.casim_enc_add_pod_spec(enc._h, b"default", (C.c_int64 * _abi.MAX_RES)(1, 1))
    assert after == before + 1     # no record slipped in
    ok, keep2 = _columns(2, 5, toff=[0, 1, 1])
    peg_ids = (C.c_int32 * 2)()
    assert lib.casim_enc_add_pods(enc._h, C.byref(ok), peg_ids) == after + 1 and list(peg_ids) == [0, 1]
    enc.finalize()
    assert enc.pegs.n_pegs == 2
    del keep, keep2


def test_null_columns_and_the_derived_fastpath_requests():
    """NULL offset columns = no pod has any; NULL fastpath_req = milli * 1e-3 / bytes, as casim_enc_add_resource_pegs derives them."""
    enc = Encoder()
    keep = [np.array([0, 0], np.int32), np.array([[250, 1 << 20], [500, 2 << 20]], np.int64), np.array([3, -1], np.int32)]
    strings = (C.c_char_p * 1)(b"default")
    pc = _abi.PodColumns(n_pods=2, n_strings=1, strings=strings, ns=keep[0].ctypes.data_as(_abi.i32p), req=keep[1].ctypes.data_as(_abi.i64p),
                         peg_count=keep[2].ctypes.data_as(_abi.i32p))
    peg_ids = (C.c_int32 * 2)()
    assert lib.casim_enc_add_pods(enc._h, C.byref(pc), peg_ids) == 0 and list(peg_ids) == [0, -1]
    other = Encoder()
    other.add_resource_pegs(np.array([[250, 1 << 20]], np.int64), np.array([3], np.int32))
    w = workloads.config_c0()
    for e in (enc, other):
        e.add_group(w.groups[0].template, max_nodes=5)
        e.finalize()
    assert nt.tables_fnv(enc.pegs, enc.groups) == nt.tables_fnv(other.pegs, other.groups)


def test_not_after_finalize():
    enc = Encoder()
    enc.add_group(workloads.config_c0().groups[0].template, max_nodes=5)
    enc.finalize()
    pc, keep = _columns(1, 5)
    assert lib.casim_enc_add_pods(enc._h, C.byref(pc), None) == _abi.ERR_INVALID
    del keep


@pytest.mark.parametrize("name", ["config_c2", "config_c4"])
def test_the_bulk_trace_replays_natively_to_the_same_tables(name, tmp_path):
    """tools/casim_native (plain C++ over include/casim.h) replays the recorded casim_enc_add_pods line: same tables, a fraction of the calls."""
    if not os.path.exists(nt.NATIVE):
        nt.build()
    w = getattr(workloads, name)()
    per_pod, bulk = str(tmp_path / "per_pod.trace"), str(tmp_path / "bulk.trace")
    e1 = nt.trace_estimate(w, per_pod, bulk=False)
    e2 = nt.trace_estimate(w, bulk, bulk=True)
    want = nt.tables_fnv(e1.pegs, e1.groups)
    assert nt.tables_fnv(e2.pegs, e2.groups) == want
    _, a = nt.run_native(per_pod, repeat=1)
    _, b = nt.run_native(bulk, repeat=1)
    assert a["tables_fnv"] == want and b["tables_fnv"] == want
    # (what is left per pod here: the grouping digest the Python mirror sets on every spec — the Go shim does not —, and in C4 the two calls of
    # an anti-affinity term)
    assert b["enc_calls"] * (5 if name == "config_c2" else 2) < a["enc_calls"]


def test_the_go_shim_adds_the_pegs_of_a_loop_in_one_crossing():
    """integration/go/gpubinpacking: session.pegs builds a casim_pod_columns under a runtime.Pinner, then names the scalar requests and
    adds the rarer fields per pod; both callers (per-call Estimate, the prefetch fill) go through it and fall back on an error."""
    shim = os.path.join(ROOT, "integration", "go", "gpubinpacking")
    src = open(os.path.join(shim, "encode.go")).read()
    i = src.index("func (s *session) pegs(")
    body = src[i:src.index("\n}\n", i)]
    assert "C.casim_enc_add_pods(s.enc, &pc, &out[0])" in body and "var pc C.casim_pod_columns" in body
    assert "runtime.Pinner" in body and "pin.Unpin()" in body and body.count("pin.Pin(") >= 4
    assert "scalars(podutils.PodRequests(pod), func(name apiv1.ResourceName, v int64) { s.request(id, name, v) })" in body   # (request: no return code dropped)
    assert "s.podRest(pod, id)" in body and "ids[gi] = s.peg(g)" in body
    for column in ("ns", "req", "fastpath_req", "peg_count", "label_off", "label_key", "label_val", "tol_off", "tol_key", "tol_op", "tol_value",
                   "tol_effect", "sel_off", "sel_key", "sel_val", "strings", "n_pods", "n_strings"):
        assert f"pc.{column}" in body, column
    header = open(os.path.join(ROOT, "include", "casim.h")).read()
    h = header[header.index("typedef struct casim_pod_columns {"):header.index("} casim_pod_columns;")]
    import re
    declared = []
    for line in h.splitlines()[1:]:
        line = line.split("/*")[0]
        m = re.match(r"\s*(?:const\s+)?[\w\s]+?[\s\*]+(?:const\s*\*\s*)?([\w\s,]+);", line)
        if m:
            declared += [x.strip() for x in m.group(1).split(",")]
    assert declared == [f for f, _ in _abi.PodColumns._fields_]   # the ctypes mirror: same fields, same order
    pod = src[src.index("func (s *session) pod("):]
    pod = pod[:pod.index("\n}\n")]
    assert "s.podRest(pod, id)" in pod and "casim_enc_pod_add_host_port" not in pod
    assert "err := s.pegs(pegs)" in open(os.path.join(shim, "estimator.go")).read()
    assert "err := sess.pegs(pegs)" in open(os.path.join(shim, "prefetch.go")).read()


@pytest.mark.parametrize("make", [lambda: workloads.removal_scale(300, pods_per_node=12, frac_candidates=0.3, seed=4), lambda: workloads.runonce_scale_down(40),
                                  lambda: workloads.fuzz_removals(3), lambda: workloads.fuzz_removals_plain(5), lambda: workloads.fuzz_removals_domains(7)],
                         ids=["removal_scale", "runonce_scale_down", "fuzz_removals", "fuzz_removals_plain", "fuzz_removals_domains"])
def test_running_pods_in_one_crossing_trace_and_replay(make, tmp_path):
    """casim_enc_add_running_pods (what the Go shim's runningPods uses for a snapshot's plain pods) recorded and replayed by tools/casim_native: the
    same tables as the per-pod calls, from the Python mirror and from the plain-C++ replay, in a fraction of the crossings"""
    if not os.path.exists(nt.NATIVE):
        nt.build()
    w = make()
    out = {}
    for bulk in (False, True):
        path = str(tmp_path / f"r{int(bulk)}.trace")
        enc, _, _ = nt.trace_removals(w, path, iters=1, bulk=bulk)
        h = (nt.tables_fnv(enc.pegs, enc.groups), rules_hash(enc))
        enc.close()
        _, nat = nt.run_native(path, repeat=1)
        out[bulk] = (h, nat["tables_fnv"], nat["enc_calls"])
    assert out[False][0] == out[True][0] and out[False][1] == out[True][1] == out[True][0][0]
    assert out[True][2] <= out[False][2]
