"""SURVEY §8 f2, first half: pod equivalence grouping BEHIND THE C ABI (casim_enc_group_pods in libcasim's host encoder).

Reference: equivalence.BuildPodGroups / groupPodsBySchedulingProperties / match
(CA/core/scaleup/equivalence/groups.go:39-104), utils.PodSpecSemanticallyEqual (CA/utils/utils.go:63-119).
Pinned on the three known-answer tests the reference holds for it (groups_test.go: TestGroupSchedulablePodsForNode,
TestEquivalenceGroupSizeLimiting, TestEquivalenceGroupIgnoresDaemonSets, transcribed in tests/golden/reference_vectors.json);
the Python restatement in kubernetes_autoscaler_amd/equivalence.py is the checker for the fuzz families."""
import json
import os
import random

import numpy as np
import pytest

from kubernetes_autoscaler_amd.encoder import Encoder
from kubernetes_autoscaler_amd.equivalence import build_pod_groups, group_pods_by_scheduling_properties, group_pods_native
from kubernetes_autoscaler_amd.objects import (ContainerPort, PodAffinityTerm, PodEquivalenceGroup, Requirement, Toleration, build_test_pod)


def _gold():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["equivalence_groups"]


def _names(groups):
    return [[p.name for p in g] for g in groups]


@pytest.mark.parametrize("share", [False, True])
def test_group_schedulable_pods_for_node_native(share):
    G = _gold()["group_schedulable_pods_for_node"]
    pods = []
    for p in G["pods"]:
        pod = build_test_pod(p["name"], p["cpu"], p["mem"])
        pod.controller_uid = p.get("controller", "")
        pod.spec_extra = p.get("spec_extra", "")
        pods.append(pod)
    groups = group_pods_native(pods, share_specs=share)
    assert sorted(sorted(n) for n in _names(groups)) == sorted(sorted(g) for g in G["want_groups"])
    assert _names(groups) == _names(group_pods_by_scheduling_properties(pods))
    assert [len(g.pods) for g in build_pod_groups(pods)] == [len(g) for g in groups]   # BuildPodGroups goes through the library


@pytest.mark.parametrize("share", [False, True])
def test_equivalence_group_size_limiting_native(share):
    G = _gold()["size_limiting"]
    pods = []
    for i in range(G["n_pods"]):
        p = build_test_pod(f"p{i}", G["cpu"], G["mem"])
        p.controller_uid = G["controller"]
        p.labels = {"uniqueLabel": f"l{i}"}
        pods.append(p)
    assert [len(g) for g in group_pods_native(pods, share_specs=share)] == G["want_group_sizes"]
    # the 11th distinct spec is not remembered (groups.go:81-90): its twin opens another group, a twin of the 1st joins the 1st
    twin_last = build_test_pod("twin-last", G["cpu"], G["mem"]); twin_last.controller_uid = G["controller"]; twin_last.labels = {"uniqueLabel": "l10"}
    twin_first = build_test_pod("twin-first", G["cpu"], G["mem"]); twin_first.controller_uid = G["controller"]; twin_first.labels = {"uniqueLabel": "l0"}
    assert [len(g) for g in group_pods_native(pods + [twin_last, twin_first], share_specs=share)] == [2] + [1] * 11


def test_equivalence_group_ignores_daemonsets_native():
    G = _gold()["ignores_daemonsets"]
    pods = []
    for i in range(G["n_pods"]):
        p = build_test_pod(f"p{i + 1}", G["cpu"], G["mem"])
        p.controller_uid = G["controller"]
        p.daemonset = True
        pods.append(p)
    assert len(group_pods_native(pods)) == G["want_groups"]


def _random_pod(rng, i):
    """A pod drawn from a small pool of variations in every field match() looks at."""
    p = build_test_pod(f"p{i}", rng.choice([100, 200, 250]), rng.choice([1000, 2000]))
    p.namespace = rng.choice(["default", "default", "kube-system"])
    p.controller_uid = rng.choice(["", "rs-a", "rs-b", "rs-c", "job-d"])
    p.daemonset = rng.random() < 0.05
    if rng.random() < 0.6:
        p.labels = {"app": rng.choice(["web", "db"])}
        if rng.random() < 0.3:
            p.labels["tier"] = rng.choice(["a", "b"])
    if rng.random() < 0.3:
        p.tolerations = [Toleration(key="dedicated", operator="Equal", value=rng.choice(["x", "y"]), effect="NoSchedule")]
    if rng.random() < 0.3:
        p.node_selector = {"pool": rng.choice(["p1", "p2"])}
    if rng.random() < 0.2:
        p.node_affinity = [Requirement("zone", rng.choice(["In", "NotIn"]), [rng.choice(["z1", "z2"])])]
    if rng.random() < 0.15:
        p.host_ports = [ContainerPort(host_port=rng.choice([80, 8080]))]
    if rng.random() < 0.15:
        p.anti_affinity = [PodAffinityTerm(topology_key=rng.choice(["kubernetes.io/hostname", "zone"]), match_labels={"app": rng.choice(["web", "db"])})]
    if rng.random() < 0.1:
        p.spec_extra = rng.choice(["vol-1", "vol-2"])
    if rng.random() < 0.05:
        p.unsupported_reason = "volumes"
    return p


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_native_grouping_equals_the_restatement(seed):
    rng = random.Random(9000 + seed)
    pods = [_random_pod(rng, i) for i in range(rng.choice([0, 1, 17, 120, 400]))]
    want = _names(group_pods_by_scheduling_properties(pods))
    assert _names(group_pods_native(pods)) == want
    assert _names(group_pods_native(pods, share_specs=True)) == want


def test_field_sensitivity():
    """Every field the encoder knows separates two otherwise equal pods of one controller (a field that did not would merge pods the
    reference keeps apart); node-selector insertion order and label insertion order do not (Go maps)."""
    def base():
        p = build_test_pod("p", 100, 1000); p.controller_uid = "rs"; p.labels = {"a": "1", "b": "2"}; p.node_selector = {"k1": "v", "k2": "w"}
        return p
    def two(mutate):
        a, b = base(), base()
        mutate(b)
        return len(group_pods_native([a, b]))
    assert two(lambda p: None) == 1
    assert two(lambda p: setattr(p, "labels", {"b": "2", "a": "1"})) == 1
    assert two(lambda p: setattr(p, "node_selector", {"k2": "w", "k1": "v"})) == 1
    assert two(lambda p: setattr(p, "namespace", "other")) == 2
    assert two(lambda p: p.requests.update(cpu=101)) == 2
    assert two(lambda p: p.labels.update(c="3")) == 2
    assert two(lambda p: p.tolerations.append(Toleration(key="k", operator="Exists"))) == 2
    assert two(lambda p: p.node_selector.update(k3="x")) == 2
    assert two(lambda p: p.node_affinity.append(Requirement("z", "In", ["1"]))) == 2
    assert two(lambda p: p.host_ports.append(ContainerPort(host_port=80))) == 2
    assert two(lambda p: p.anti_affinity.append(PodAffinityTerm(topology_key="zone", match_labels={"a": "1"}))) == 2
    assert two(lambda p: p.affinity.append(PodAffinityTerm(topology_key="zone", match_labels={"a": "1"}))) == 2
    assert two(lambda p: setattr(p, "spec_extra", "pvc-7")) == 2
    assert two(lambda p: setattr(p, "unsupported_reason", "dra")) == 2


def test_grouped_pegs_feed_the_encoder():
    """casim_enc_add_grouped_pegs: one PEG per group, exemplar = first pod, count = size; the tables equal those built PEG by PEG."""
    rng = random.Random(5)
    pods = [_random_pod(rng, i) for i in range(200)]
    for p in pods:
        p.unsupported_reason = ""
    groups = group_pods_by_scheduling_properties(pods)
    a = Encoder()
    gid, n = a.group_pods(pods)
    ids = a.add_grouped_pegs(pods, gid, n)
    assert list(ids) == list(range(n))
    a.finalize()
    b = Encoder()
    for g in groups:
        b.add_peg(PodEquivalenceGroup(pods=g))
    b.finalize()
    assert n == len(groups)
    assert [int(a.pegs.count[i]) for i in range(n)] == [len(g) for g in groups]
    R = a.pegs.n_res
    assert [int(a.pegs.req[i]) for i in range(n * R)] == [int(b.pegs.req[i]) for i in range(n * R)]
    a.close(); b.close()


def test_an_empty_group_adds_no_peg_at_all():
    """casim_enc_add_grouped_pegs used to push the PEGs in front of the first empty group before it noticed"""
    from kubernetes_autoscaler_amd import _abi
    from kubernetes_autoscaler_amd._ffi import lib
    import ctypes as C
    rng = random.Random(9)
    pods = [_random_pod(rng, i) for i in range(6)]
    e = Encoder()
    spec = np.array([e.add_pod_spec(p) for p in pods], np.int32)
    gid = np.array([0, 0, 1, 3, 3, 3], np.int32)   # group 2 is empty
    ids = np.full(4, -7, np.int32)
    assert lib.casim_enc_add_grouped_pegs(e._h, 6, spec.ctypes.data_as(_abi.i32p), gid.ctypes.data_as(_abi.i32p), 4, ids.ctypes.data_as(_abi.i32p)) == _abi.ERR_INVALID
    assert list(ids) == [-7] * 4
    gid2 = np.array([0, 0, 1, 2, 2, 2], np.int32)
    assert lib.casim_enc_add_grouped_pegs(e._h, 6, spec.ctypes.data_as(_abi.i32p), gid2.ctypes.data_as(_abi.i32p), 3, ids.ctypes.data_as(_abi.i32p)) == 0   # first id 0: nothing was left behind
    assert list(ids[:3]) == [0, 1, 2]
    e.close()


def test_invalid_arguments_are_refused():
    from kubernetes_autoscaler_amd import _abi
    from kubernetes_autoscaler_amd._ffi import lib
    import ctypes as C
    e = Encoder()
    out = np.zeros(2, np.int32)
    spec = np.array([0, 5], np.int32)   # no such spec records
    ng = C.c_int32(0)
    assert lib.casim_enc_group_pods(e._h, 2, spec.ctypes.data_as(_abi.i32p), None, None, out.ctypes.data_as(_abi.i32p), C.byref(ng)) < 0
    assert lib.casim_enc_group_pods(e._h, 2, None, None, None, out.ctypes.data_as(_abi.i32p), C.byref(ng)) < 0
    assert lib.casim_enc_group_pods(e._h, 0, None, None, None, None, C.byref(ng)) == 0 and ng.value == 0
    e.close()
