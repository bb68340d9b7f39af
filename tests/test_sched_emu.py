"""SURVEY §8 row f1 (filter-out-schedulable): the product encoder + K_sched_static / K_sched under the wave
emulator against the object-level CPU oracle, bit for bit (node of every pending pod, lastIndex, count).
CPU only; the same cases run on the MI355X in test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

from harness import SchedCase, assert_sched_matches, sched_emu, sched_oracle
from kubernetes_autoscaler_amd import _abi
from kubernetes_autoscaler_amd.objects import (LABEL_ZONE, ContainerPort, NodeInfo, Pod, PodAffinityTerm, build_test_node,
                                               build_test_pod, with_host_port, with_pod_hostname_anti_affinity)
from kubernetes_autoscaler_amd.workloads import _node, filter_out_schedulable_benchmark, fuzz_pending, pending_scale
from test_oracle_golden import GOLD, golden_hinted_cases, golden_sched_case


def case_of(w) -> SchedCase:
    return SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure,
                     last_index=w.last_index)


def check(case: SchedCase, what="", lds_budgets=(0, 64)):
    want = sched_oracle(case)
    for lds in lds_budgets:  # 0 = node state in LDS, 64 B = forced HBM slab
        got = sched_emu(case, lds_budget=lds)
        assert not case.pods or not case.nodes or got[4][1] == (1 if lds == 0 else 0)
        assert_sched_matches(got, want, f"{what} lds={lds}")
    return want


@pytest.mark.parametrize("case", GOLD["try_schedule_pods"]["cases"], ids=lambda c: c["name"])
def test_golden_try_schedule_pods(case):
    sc, names = golden_sched_case(case)
    _, node_out, _, n_sched, _ = sched_emu(sc)
    got = {p.name: names[m] for p, m in zip(sc.pods, node_out) if m >= 0}
    assert got == case["want"] and n_sched == len(case["want"])
    check(sc, case["name"])


@pytest.mark.parametrize("case", GOLD["pod_schedules_on_hinted_node"]["cases"], ids=lambda c: c["name"])
def test_golden_pod_schedules_on_hinted_node(case):
    for sc, want in golden_hinted_cases(case):
        _, node_out, last_index, n_sched, _ = sched_emu(sc)
        assert list(node_out) == want and n_sched == len(want) and last_index == 0


@pytest.mark.parametrize("seed", range(400))
def test_fuzz(seed):
    w = fuzz_pending(seed)
    check(case_of(w), w.name)


def test_benchmark_small_shapes():
    # BenchmarkFilterOutSchedulable "nothing" / "small" / "medium": nothing fits, every pod stays pending
    for n, s, p in ((1, 30, 1000), (10, 300, 1000), (100, 3000, 1000)):
        w = filter_out_schedulable_benchmark(n, s, p)
        node_out, li, ns = check(case_of(w), w.name, lds_budgets=(0,))
        assert ns == 0 and li == 0 and (node_out == -1).all()


def test_packing_many_slots():
    # > 64 nodes (several slots per lane), runs longer than one round
    w = pending_scale(300, 2500, n_classes=12, seed=3)
    node_out, _, ns = check(case_of(w), w.name)
    assert ns > 1000


def test_round_robin_order_inside_a_run():
    # 3 nodes with capacities 3 / 1 / 2 for the pod, lastIndex = 1 -> rotated order n2, n0, n1 per round
    nodes = [NodeInfo(build_test_node("n0", 3000, 10**9)), NodeInfo(build_test_node("n1", 1000, 10**9)), NodeInfo(build_test_node("n2", 2000, 10**9))]
    pods = [build_test_pod(f"p{i}", 1000, 1) for i in range(8)]
    node_out, li, ns = check(SchedCase(nodes=nodes, pods=pods, last_index=1))
    assert list(node_out) == [2, 0, 1, 2, 0, 0, -1, -1] and ns == 6 and li == 0


def test_self_exclusion_across_runs():
    # a host-port class split into two runs by another class: the second run must see the first run's pods
    nodes = [NodeInfo(_node(f"n{i}", 4000, 10**9, 100)) for i in range(3)]
    a = [build_test_pod(f"a{i}", 100, 1, with_host_port(8080)) for i in range(4)]
    b = [build_test_pod("b0", 100, 1)]
    aa = [build_test_pod(f"h{i}", 100, 1, with_pod_hostname_anti_affinity({"app": "h"})) for i in range(5)]
    for p in aa:
        p.labels = {"app": "h"}
    pods = a[:2] + b + a[2:] + aa[:2] + b + aa[2:]
    node_out, _, ns = check(SchedCase(nodes=nodes, pods=pods))
    assert ns == 3 + 2 + 3


def test_unsupported_predicates_are_delegated():
    nodes = [NodeInfo(build_test_node("n0", 4000, 10**9))]
    spread = Pod(name="s", requests={"cpu": 1}, topology_spread=True)   # a constraint the shim could not describe
    assert sched_emu(SchedCase(nodes=nodes, pods=[build_test_pod("ok", 1, 1), spread]))[0] == _abi.NG_UNSUPPORTED
    # hostname anti-affinity against a node without the hostname label cannot be expressed by node bits
    named = NodeInfo(_node("named", 4000, 10**9, 100))
    bare = NodeInfo(build_test_node("bare", 4000, 10**9))  # BuildTestNode sets no labels at all
    h = build_test_pod("h", 1, 1, with_pod_hostname_anti_affinity({"app": "h"}))
    h.labels = {"app": "h"}
    assert sched_emu(SchedCase(nodes=[named, bare], pods=[h]))[0] == _abi.NG_UNSUPPORTED
    assert sched_emu(SchedCase(nodes=[named], pods=[h]))[0] == 0


# ---- domain rules: PodTopologySpread and anti-affinity on non-hostname keys ---------------------------------------
def _zoned(n, zones, cpu=4000):
    return [NodeInfo(_node(f"n{i}", cpu, 10**9, 110, {LABEL_ZONE: f"z{i % zones}"})) for i in range(n)]


def test_zone_anti_affinity_one_pod_per_zone():
    nodes = _zoned(6, 3)
    pods = [Pod(name=f"z{i}", labels={"app": "z"}, requests={"cpu": 1}, anti_affinity=[PodAffinityTerm(LABEL_ZONE, match_labels={"app": "z"})])
            for i in range(5)]
    node_out, li, ns = check(SchedCase(nodes=nodes, pods=pods))
    assert ns == 3 and list(node_out) == [1, 2, 3, -1, -1]   # lastIndex 0: the walk starts at n1; one pod per zone


def test_hostname_spread_fills_evenly():
    from kubernetes_autoscaler_amd.objects import TopologySpreadConstraint
    nodes = _zoned(4, 2)
    pods = [Pod(name=f"s{i}", labels={"app": "s"}, requests={"cpu": 1}, topology_spread=True,
                spread_constraints=[TopologySpreadConstraint(1, "kubernetes.io/hostname", 0, {"app": "s"})]) for i in range(9)]
    node_out, li, ns = check(SchedCase(nodes=nodes, pods=pods))
    assert ns == 9 and sorted(np.bincount(node_out, minlength=4)) == [2, 2, 2, 3]


def test_zone_spread_with_min_domains_and_partial_eligibility():
    from kubernetes_autoscaler_amd.objects import TopologySpreadConstraint
    nodes = _zoned(6, 2)
    for i, info in enumerate(nodes):
        info.node.labels["pool"] = "a" if i < 4 else "b"
    nodes[0].pods.append(Pod(name="r0", labels={"app": "s"}, requests={"cpu": 1}))
    # minDomains 3 > 2 zones: the global minimum counts as 0, so every zone takes at most maxSkew pods
    pods = [Pod(name=f"s{i}", labels={"app": "s"}, requests={"cpu": 1}, node_selector={"pool": "a"}, topology_spread=True,
                spread_constraints=[TopologySpreadConstraint(2, LABEL_ZONE, 3, {"app": "s"})]) for i in range(8)]
    node_out, li, ns = check(SchedCase(nodes=nodes, pods=pods))
    assert ns == 3   # z0 already holds one matching pod on an eligible node: 1 more there, 2 in z1


@pytest.mark.parametrize("seed", range(400))
def test_fuzz_domain_rules(seed):
    from kubernetes_autoscaler_amd.workloads import fuzz_pending_domains
    w = fuzz_pending_domains(seed)
    check(case_of(w), w.name)


def test_empty_inputs():
    nodes = [NodeInfo(build_test_node("n0", 1000, 1000))]
    rc, node_out, li, ns, _ = sched_emu(SchedCase(nodes=nodes, pods=[], last_index=5))
    assert (rc, len(node_out), li, ns) == (0, 0, 5, 0)
    rc, node_out, li, ns, _ = sched_emu(SchedCase(nodes=[], pods=[build_test_pod("p", 1, 1)], last_index=0))
    assert (rc, list(node_out), ns) == (0, [-1], 0)


def _pool_cluster(n_nodes, every, cpu=4000):
    nodes = []
    for i in range(n_nodes):
        nd = _node(f"n{i}", cpu if i % 3 else cpu // 2, 10**9, 110, {"pool": "a" if i % every == 0 else "b"})
        nodes.append(NodeInfo(nd))
    return nodes


def _pool_pods(prefix, n, cpu, pool=None):
    return [Pod(name=f"{prefix}{i}", labels={"app": prefix}, requests={"cpu": cpu, "memory": 1}, node_selector={"pool": pool} if pool else {},
                controller_uid=prefix) for i in range(n)]


@pytest.mark.parametrize("last_index", [0, 63, 1023, 1100, 2047, 2499])
def test_several_chunks_of_1024_nodes(last_index):
    """N > 1024: chunks of T = 1024 nodes, cyclic origin inside a chunk (wrap piece), rounds 2.. across chunks."""
    nodes = _pool_cluster(2500, 29)
    pods = (_pool_pods("x", 200, 500, "a")       # 87 eligible nodes: rounds with a partial last round
            + _pool_pods("y", 3, 1500)           # three single first-fits right after the origin
            + _pool_pods("x", 700, 500, "a")     # saturates pool a, the rest stays pending (memo afterwards)
            + _pool_pods("z", 1500, 1000)        # fewer pods than fitting nodes: stops early
            + _pool_pods("x", 5, 500, "a"))      # memo hit
    hints = [-1] * len(pods)
    hints[201] = 2400
    hints[950] = 7
    want = check(SchedCase(nodes=nodes, pods=pods, hints=hints, last_index=last_index), f"li={last_index}", lds_budgets=(0, 64))
    assert want[2] > 1900


def test_fuzz_large_clusters():
    for seed in range(6):
        w = pending_scale(1100 + 400 * seed, 1500, n_classes=10, seed=40 + seed)
        w.last_index = 37 * seed * seed
        check(case_of(w), w.name, lds_budgets=(0,) if seed % 2 else (64,))


# ---- the reference's own table: podlistprocessor/filter_out_schedulable_test.go TestFilterOutSchedulable -------------
@pytest.mark.parametrize("row", GOLD["filter_out_schedulable"]["cases"], ids=lambda r: r["name"])
def test_reference_filter_out_schedulable_table(row):
    from harness import EmuContext
    from kubernetes_autoscaler_amd.scheduling import FilterOutSchedulablePodListProcessor
    from test_oracle_golden import golden_filter_case
    nodes, cands, acceptable = golden_filter_case(row)
    if cands:
        check(SchedCase(nodes=nodes, pods=cands, acceptable=acceptable), row["name"])
    # and through the host mirror, unsorted input like the reference test hands it over
    node_filter = (lambda info: False) if row.get("node_filter") == "none" else None
    proc = FilterOutSchedulablePodListProcessor(EmuContext(0), node_filter)
    unsorted = sorted(cands, key=lambda p: [c["name"] for c in row["candidates"]].index(p.name))
    left = proc.process(nodes, unsorted)
    assert sorted(p.name for p in left) == sorted(row["unscheduled"])
    assert sorted(proc.scheduling_simulator.hints.old) == sorted(f"default/{n}" for n in row["scheduled"])


def test_reference_similar_pods_limiting_and_daemonsets():
    """similar_pods_test.go TestSimilarPodsSchedulingLimiting / ...IgnoreDaemonSets, observed where the mirror exposes the
    bookkeeping: 11 distinct unschedulable specs of one controller -> one overflowing controller; DaemonSet pods and
    pods without a controller are never recorded."""
    from harness import EmuContext
    from kubernetes_autoscaler_amd.scheduling import HintingSimulator
    nodes = [NodeInfo(build_test_node("n", 1000, 100000))]
    pods = []
    for i in range(11):
        p = build_test_pod(f"p{i}", 3000, 200000)
        p.controller_uid = "12345678-1234-1234-1234-123456789012"
        p.labels = {"uniqueLabel": f"l{i}"}
        pods.append(p)
    statuses, overflowing = HintingSimulator(EmuContext(0)).try_schedule_pods(nodes, pods)
    assert not statuses and overflowing == 1
    statuses, overflowing = HintingSimulator(EmuContext(0)).try_schedule_pods(nodes, pods[:10])
    assert not statuses and overflowing == 0
    for p in pods:
        p.daemonset = True
    assert HintingSimulator(EmuContext(0)).try_schedule_pods(nodes, pods) == ([], 0)
    for p in pods:
        p.daemonset, p.controller_uid = False, ""
    assert HintingSimulator(EmuContext(0)).try_schedule_pods(nodes, pods) == ([], 0)


def test_large_cluster_uses_the_wide_workgroup():
    # from 4096 nodes on the TrySchedulePods workgroup runs 512 threads (casim_sched.h init)
    w = pending_scale(4500, 3000, n_classes=12, seed=18)
    check(case_of(w), w.name, lds_budgets=(0,))


@pytest.mark.parametrize("row", GOLD["topology_spread_taint_scheduling"]["cases"], ids=lambda r: r["name"])
def test_reference_topology_spread_next_to_a_tainted_node(row):
    """simulator/cluster_scheduling_test.go TestTopologySpreadTaintScheduling: the tainted, emptied node still is a spread
    domain under the default policy (nothing fits); with nodeTaintsPolicy: Honor it is no domain member and node2 takes the pod."""
    from test_oracle_golden import golden_taint_spread_case
    sc = golden_taint_spread_case(row)
    want = check(sc, row["name"])
    assert (want[2] == 1) == row["schedulable"]


def honor_taints_variant(w, seed):
    """fuzz_pending_domains cluster with tainted nodes; some specs tolerate them, some spread constraints carry
    nodeTaintsPolicy: Honor (equal specs stay equal)."""
    import dataclasses
    import random
    from kubernetes_autoscaler_amd.objects import Taint, Toleration
    rng = random.Random(9000 + seed)
    for info in w.nodes:
        r = rng.random()
        if r < 0.25:
            info.node.taints.append(Taint("dedicated", "x", "NoSchedule"))
        elif r < 0.35:
            info.node.taints.append(Taint("maint", "y", "NoExecute"))
        elif r < 0.45:
            info.node.taints.append(Taint("soft", "z", "PreferNoSchedule"))   # never counts
    plan = {}
    for p in w.pods:
        k = p.spec_key()
        if k not in plan:
            tol = rng.choice([[], [], [Toleration("dedicated", "Equal", "x", "NoSchedule")], [Toleration("", "Exists", "", "")]])
            plan[k] = (tol, [dataclasses.replace(c, node_taints_policy=("Honor" if rng.random() < 0.6 else "Ignore"),
                                                 node_affinity_policy=("Ignore" if rng.random() < 0.4 else "Honor")) for c in p.spread_constraints])
        p.tolerations, p.spread_constraints = list(plan[k][0]), list(plan[k][1])
    return w


@pytest.mark.parametrize("seed", range(120))
def test_fuzz_node_taints_policy_honor(seed):
    from kubernetes_autoscaler_amd.workloads import fuzz_pending_domains
    w = honor_taints_variant(fuzz_pending_domains(3000 + seed), seed)
    check(case_of(w), w.name)


@pytest.mark.parametrize("policy,fits", [("Honor", True), ("Ignore", False)])
def test_node_affinity_policy_of_a_spread_constraint(policy, fits):
    """common.go:46-51: with nodeAffinityPolicy: Honor (default) only nodes matching the pod's nodeSelector are domain
    members — zone z1 does not exist for the pod and z0's two pods are the global minimum; with Ignore z1 is an empty
    domain and one more pod in z0 would skew by 3."""
    from kubernetes_autoscaler_amd.objects import TopologySpreadConstraint
    n0 = NodeInfo(_node("n0", 4000, 8 << 30, 110, {"pool": "a", LABEL_ZONE: "z0"}))
    n1 = NodeInfo(_node("n1", 4000, 8 << 30, 110, {"pool": "b", LABEL_ZONE: "z1"}))
    for i in range(2):
        n0.pods.append(Pod(name=f"run{i}", labels={"app": "x"}, requests={"cpu": 100}))
    pend = Pod(name="pend", labels={"app": "x"}, requests={"cpu": 100}, node_selector={"pool": "a"}, topology_spread=True,
               spread_constraints=[TopologySpreadConstraint(1, LABEL_ZONE, 0, {"app": "x"}, "Ignore", policy)])
    want = check(SchedCase(nodes=[n0, n1], pods=[pend]), f"nodeAffinityPolicy {policy}")
    assert (want[2] == 1) == fits and (list(want[0]) == [0]) == fits


def test_match_label_keys_of_a_spread_constraint():
    """common.go:96-107: matchLabelKeys adds `key == <the incoming pod's value>` to the selector — the pods of another
    revision (pod-template-hash) do not count.  Two zones; zone z0 holds two pods of revision A, z1 none; a pod of
    revision B spreads only against revision-B pods: both zones are at 0 and it lands on the first node; without the keys
    the two revision-A pods make z0 too heavy and it goes to z1."""
    from kubernetes_autoscaler_amd.objects import TopologySpreadConstraint
    def cluster():
        n0 = NodeInfo(_node("n0", 4000, 8 << 30, 110, {LABEL_ZONE: "z0"}))
        n1 = NodeInfo(_node("n1", 4000, 8 << 30, 110, {LABEL_ZONE: "z1"}))
        for i in range(2):
            n0.pods.append(Pod(name=f"run{i}", labels={"app": "x", "pod-template-hash": "A"}, requests={"cpu": 100}))
        return [n0, n1]
    def pend(keys):
        return Pod(name="pend", labels={"app": "x", "pod-template-hash": "B"}, requests={"cpu": 100}, topology_spread=True,
                   spread_constraints=[TopologySpreadConstraint(1, LABEL_ZONE, 0, {"app": "x"}, match_label_keys=keys)])
    with_keys = check(SchedCase(nodes=cluster(), pods=[pend(("pod-template-hash",))], last_index=1), "matchLabelKeys")
    without = check(SchedCase(nodes=cluster(), pods=[pend(())], last_index=1), "no matchLabelKeys")
    assert list(with_keys[0]) == [0] and list(without[0]) == [1]
