"""SURVEY §8 row f4 (scale-down removal simulation): product encoder + K_sched transactions under the wave emulator
against the object-level CPU oracle — removable flag per candidate, destination of every moved pod, lastIndex, and the
number of candidates the call could decide.  CPU only; the same cases run on the MI355X in test_gpu_sched.py."""
import numpy as np
import pytest

from harness import EmuContext, RemovalCase, assert_removal_matches, removal_device, removal_oracle
from kubernetes_autoscaler_amd.objects import NodeInfo, Pod, build_test_pod
from kubernetes_autoscaler_amd.scaledown import NO_PLACE_TO_MOVE_PODS, RemovalSimulator
from kubernetes_autoscaler_amd.workloads import _node, fuzz_removals, removal_scale


def case_of(w) -> RemovalCase:
    return RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                       max_removable=w.max_removable, last_index=w.last_index)


def check(case, what="", lds=(0, 64)):
    want = removal_oracle(case)
    for b in lds:
        assert_removal_matches(removal_device(case, EmuContext(b)), want, f"{what} lds={b}")
    return want


@pytest.mark.parametrize("seed", range(300))
def test_fuzz(seed):
    w = fuzz_removals(seed)
    check(case_of(w), w.name)


def test_simple_chain():
    # three nodes of 1000m: n0 {400}, n1 {400}, n2 {300}; candidates n0, n1, n2 in that order, everything persists
    nodes = [NodeInfo(_node(f"n{i}", 1000, 10**9, 10)) for i in range(3)]
    for i, cpu in enumerate((400, 400, 100)):
        nodes[i].pods.append(build_test_pod(f"p{i}", cpu, 1))
    # n0's pod goes to n1 (first node after lastIndex 0); n1 then lists it again after its own pod: both go to n2;
    # n2 finally lists all three and has nowhere to go (its first pod already fails: the others are not tried)
    w = check(RemovalCase(nodes=nodes, candidates=[0, 1, 2]))
    assert list(w["removable"]) == [1, 1, 0] and list(w["node_out"]) == [1, 2, -1] and w["n_processed"] == 3
    assert w["ext"] == [(1, 0, 2), (2, 1, -1), (2, 0, -1)] and list(w["final"]) == [2, 2, 2]
    # ext_capacity 0: the old protocol, stop in front of n1
    w = check(RemovalCase(nodes=nodes, candidates=[0, 1, 2], ext_capacity=0))
    assert list(w["removable"]) == [1, 2, 2] and w["n_processed"] == 1
    # a sticky pod hands the loop back exactly where it is listed again
    w = check(RemovalCase(nodes=nodes, candidates=[0, 1, 2], sticky={id(nodes[0].pods[0])}))
    assert list(w["removable"]) == [1, 2, 2] and w["n_processed"] == 1


def test_reverted_simulation_leaves_no_trace():
    # candidate n0 has two pods, only one fits elsewhere: reverted; candidate n2's pod then still sees n1 untouched
    nodes = [NodeInfo(_node(f"n{i}", 1000, 10**9, 10)) for i in range(3)]
    nodes[0].pods += [build_test_pod("a", 600, 1), build_test_pod("b", 600, 1)]
    nodes[1].pods += [build_test_pod("c", 300, 1)]
    nodes[2].pods += [build_test_pod("d", 700, 1)]
    w = check(RemovalCase(nodes=nodes, candidates=[0, 2], destination=[1, 1, 0]))
    assert list(w["removable"]) == [0, 1] and list(w["node_out"]) == [1, -1, 1] and w["n_processed"] == 2


def test_list_positions_shift_after_a_removal():
    # after n1 is removed the list is [n0, n2, n3]: lastIndex keeps counting positions, not node ids
    nodes = [NodeInfo(_node(f"n{i}", 1000, 10**9, 10)) for i in range(4)]
    nodes[1].pods.append(build_test_pod("x", 100, 1))
    nodes[3].pods.append(build_test_pod("y", 100, 1))
    for li in range(5):
        check(RemovalCase(nodes=nodes, candidates=[1, 3], last_index=li), f"li={li}")


def test_empty_nodes_are_removable():
    nodes = [NodeInfo(_node(f"n{i}", 1000, 10**9, 10)) for i in range(5)]
    w = check(RemovalCase(nodes=nodes, candidates=[4, 0, 2, 1, 3], last_index=2))
    assert list(w["removable"]) == [1] * 5 and w["n_processed"] == 5 and w["last_index"] == 2


def test_max_removable_and_not_persisting():
    w = removal_scale(40, pods_per_node=4, frac_candidates=0.5, seed=3)
    for persist in (True, False):
        for limit in (0, 1, 4):
            check(RemovalCase(nodes=w.nodes, candidates=w.candidates, persist=persist, max_removable=limit), f"persist={persist} limit={limit}")


def test_ext_table_overflow_hands_back():
    w = removal_scale(30, pods_per_node=3, frac_candidates=0.8, seed=9)
    full = check(RemovalCase(nodes=w.nodes, candidates=w.candidates))
    assert len(full["ext"]) > 4
    part = check(RemovalCase(nodes=w.nodes, candidates=w.candidates, ext_capacity=3))
    assert part["n_processed"] < full["n_processed"] and len(part["ext"]) <= 3


def test_large_cluster_several_chunks():
    w = removal_scale(1300, pods_per_node=3, frac_candidates=0.05, seed=5)
    want = check(case_of(w), w.name, lds=(0,))
    assert want["n_processed"] == len(w.candidates)


# ---- host mirror: the planner loop, re-submitting where a sticky pod was listed again ----------------------------
@pytest.mark.parametrize("mode", ["device-lists", "resubmit-at-arrivals", "sticky"])
@pytest.mark.parametrize("seed", range(40))
def test_mirror_equals_the_reference_loop(seed, mode):
    """RemovalSimulator.simulate_node_removals — one device call, or several with GetPodsToMove re-run in between —
    == the oracle walking the whole candidate list on the committed snapshot."""
    w = fuzz_removals(1000 + seed)
    nodes = [NodeInfo(info.node, list(info.pods)) for info in w.nodes]
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=None, persist=True,
                       max_removable=w.max_removable, last_index=w.last_index)
    want = removal_oracle(case)
    rem = want["removable"]
    sticky_ids = {id(p) for i, info in enumerate(w.nodes) for j, p in enumerate(info.pods) if (i + j) % 3 == 0} if mode == "sticky" else set()
    sim = RemovalSimulator(EmuContext(), nodes, persist_successful_simulations=True, is_sticky=lambda p: id(p) in sticky_ids,
                           ext_capacity=0 if mode == "resubmit-at-arrivals" else None)
    sim.last_index = w.last_index
    dest = {info.node.name: (w.destination is None or bool(w.destination[i])) for i, info in enumerate(w.nodes)}
    removable, unremovable, skipped = sim.simulate_node_removals([w.nodes[c].node.name for c in w.candidates], dest, w.max_removable or None)   # the fuzz rows keep the ABI's 0 = no limit
    want_removable = [w.nodes[c].node.name for k, c in enumerate(w.candidates) if rem[k] == 1]
    want_unremovable = [w.nodes[c].node.name for k, c in enumerate(w.candidates) if rem[k] == 0]
    assert [r.node.name for r in removable] == want_removable
    assert [u.node.name for u in unremovable] == want_unremovable and all(u.reason == NO_PLACE_TO_MOVE_PODS for u in unremovable)
    assert len(skipped) == len(w.candidates) - want["n_processed"]
    assert sim.last_index == want["last_index"]
    if mode == "device-lists":
        assert sim.device_calls <= 1
    # where every listed pod sits at the end
    where = {id(p): info.node.name for info in nodes for p in info.pods}
    flat = [p for lst in case.pod_lists() for p in lst]
    for p, f in zip(flat, want["final"]):
        assert where[id(p)] == w.nodes[f].node.name, (p.name, f)
    # pods_to_reschedule of a removable node = its own pods + the pods it listed again
    again = {}
    for k, e, m in want["ext"]:
        again.setdefault(k, []).append(flat[e].name)
    lists = case.pod_lists()
    ks = [k for k in range(len(w.candidates)) if rem[k] == 1]
    for r, k in zip(removable, ks):
        assert [p.name for p in r.pods_to_reschedule] == [p.name for p in lists[k]] + again.get(k, [])


def test_zone_anti_affinity_in_the_removal_loop():
    """Removing n0 takes its anti-affine pod out of zone z0's counter for the simulation; it can only land in a zone
    that holds no such pod."""
    from kubernetes_autoscaler_amd.objects import LABEL_ZONE, PodAffinityTerm
    nodes = [NodeInfo(_node(f"n{i}", 1000, 10**9, 10, {LABEL_ZONE: f"z{i % 2}"})) for i in range(4)]

    def z(name):
        return Pod(name=name, labels={"app": "z"}, requests={"cpu": 100}, anti_affinity=[PodAffinityTerm(LABEL_ZONE, match_labels={"app": "z"})])
    nodes[0].pods.append(z("a"))   # zone z0
    nodes[1].pods.append(z("b"))   # zone z1
    w = check(RemovalCase(nodes=nodes, candidates=[0, 1]))
    # a: z1 is taken by b, z0 is free once n0 is a ghost -> n2; then b: z0 now holds a -> only z1 -> n3
    assert list(w["removable"]) == [1, 1] and list(w["node_out"]) == [2, 3]


@pytest.mark.parametrize("seed", range(300))
def test_fuzz_domain_rules(seed):
    from kubernetes_autoscaler_amd.workloads import fuzz_removals_domains
    w = fuzz_removals_domains(seed)
    check(case_of(w), w.name)


# ---- the reference's own table: simulator/cluster_test.go TestSimulateNodeRemoval --------------------------------
def _golden_rows():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")) as f:
        return json.load(f)["simulate_node_removal"]["cases"]


@pytest.mark.parametrize("row", _golden_rows(), ids=lambda r: r["name"])
def test_reference_table_device(row):
    from test_oracle_golden import golden_removal_case
    case = golden_removal_case(row)
    if case is None:
        return   # NoNodeInfo is decided by the host mirror (test below)
    if row.get("device_delegates"):
        # nodeTaintsPolicy: Honor makes the ghost's ToBeDeleted taint a domain-membership input: outside the encoded subset
        assert removal_device(case, EmuContext(0)).status == 1   # CASIM_NG_UNSUPPORTED
        return
    want = check(case, row["name"])
    assert bool(want["removable"][0] == 1) == row["removable"]


@pytest.mark.parametrize("row", _golden_rows(), ids=lambda r: r["name"])
def test_reference_table_through_the_mirror(row):
    from test_oracle_golden import golden_removal_case
    from kubernetes_autoscaler_amd.scaledown import NO_NODE_INFO
    from kubernetes_autoscaler_amd.scheduling import UnsupportedPredicate
    case = golden_removal_case(row)
    nodes = case.nodes if case is not None else []
    sim = RemovalSimulator(EmuContext(0), list(nodes), persist_successful_simulations=False)
    dest = {info.node.name: True for info in nodes}
    if row.get("device_delegates"):
        with pytest.raises(UnsupportedPredicate):
            sim.simulate_node_removals([row["candidate"]], dest)
        return
    removable, unremovable, skipped = sim.simulate_node_removals([row["candidate"]], dest)
    assert not skipped
    if row["removable"]:
        assert [r.node.name for r in removable] == [row["candidate"]] and not unremovable
        assert [p.name for p in removable[0].pods_to_reschedule] == row.get("reschedule", [])
    else:
        assert not removable and [u.node.name for u in unremovable] == [row["candidate"]]
        assert unremovable[0].reason == (NO_NODE_INFO if row.get("no_node_info") else NO_PLACE_TO_MOVE_PODS)


# ---- the reference's planner table: core/scaledown/planner/planner_test.go TestUpdateClusterState ---------------
def _planner_rows():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")) as f:
        return json.load(f)["planner_update_cluster_state"]["cases"]


@pytest.mark.parametrize("lds", [0, 64])
@pytest.mark.parametrize("row", _planner_rows(), ids=lambda r: r["name"])
def test_reference_planner_table(row, lds):
    from test_oracle_golden import golden_planner_case
    from kubernetes_autoscaler_amd.scaledown import Planner
    infos, inject = golden_planner_case(row)
    names = [i.node.name for i in infos]
    planner = Planner(EmuContext(lds), infos)
    removable, unremovable, skipped = planner.update_cluster_state(names, row["eligible"], inject)
    assert not skipped
    assert [r.node.name for r in removable] == row["unneeded"]
    assert [u.node.name for u in unremovable] == [n for n in row["eligible"] if n not in row["unneeded"]]
    assert [i.node.name for i in infos] == [n for n in names if n not in row["unneeded"]]   # persisted removals left the snapshot


def _limit_rows():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")) as f:
        return json.load(f)["planner_unneeded_nodes_limit"]["cases"]


@pytest.mark.parametrize("row", _limit_rows(), ids=lambda r: r["name"])
def test_reference_unneeded_nodes_limit_table(row):
    """planner_test.go TestUpdateClusterStatUnneededNodesLimit through the Planner mirror."""
    from kubernetes_autoscaler_amd.objects import build_test_node
    from kubernetes_autoscaler_amd.scaledown import Planner
    infos = [NodeInfo(build_test_node(f"n{i}", 1000, 10)) for i in range(row["nodes"])]
    names = [i.node.name for i in infos]
    limit = Planner.unneeded_nodes_limit(row["previously_unneeded"], row["max_parallelism"], row["unneeded_time_s"], row["update_interval_s"])
    removable, unremovable, skipped = Planner(EmuContext(0), infos).update_cluster_state(
        names, names, (), limit, (lambda n: True) if row["atomic"] else None)
    assert len(removable) == row["want_unneeded"] and not unremovable
    assert skipped == names[row["want_unneeded"]:]


def test_a_limit_of_zero_skips_every_candidate():
    """planner.go:303: `len(removableList)-atomic >= unneededNodesLimit` holds before the first candidate when the limit
    is 0 (no parallelism, nothing previously unneeded); None is the mirror's spelling of "no limit"."""
    from kubernetes_autoscaler_amd.objects import build_test_node
    from kubernetes_autoscaler_amd.scaledown import Planner
    assert Planner.unneeded_nodes_limit(0, 0, 600.0, 10.0) == 0
    for limit, want in ((0, 0), (None, 5), (2, 2)):
        infos = [NodeInfo(build_test_node(f"n{i}", 1000, 10)) for i in range(5)]
        names = [i.node.name for i in infos]
        removable, unremovable, skipped = Planner(EmuContext(0), infos).update_cluster_state(names, names, (), limit)
        assert len(removable) == want and not unremovable and skipped == names[want:]


def test_atomic_candidates_do_not_count_toward_the_limit():
    # mixed: every third candidate is atomic; limit 4 -> the loop stops after the 4th counted removal
    nodes = [NodeInfo(_node(f"n{i}", 1000, 10**9, 10)) for i in range(20)]
    atomic = [1 if i % 3 == 0 else 0 for i in range(20)]
    w = check(RemovalCase(nodes=nodes, candidates=list(range(20)), max_removable=4, atomic=atomic))
    assert w["n_processed"] == 6 and list(w["removable"][:6]) == [1] * 6   # n0 atomic, n1, n2, n3 atomic, n4, n5


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_atomic_candidates(seed):
    import random
    w = fuzz_removals(7000 + seed)
    rng = random.Random(seed)
    case = case_of(w)
    case.max_removable = rng.randint(1, 3)
    case.atomic = [1 if rng.random() < 0.4 else 0 for _ in case.candidates]
    check(case, w.name)


@pytest.mark.parametrize("seed", range(200))
def test_fuzz_node_taints_policy_honor_in_the_removal_loop(seed):
    """Spread constraints with nodeTaintsPolicy: Honor inside removal transactions: the candidate's ghost carries the
    ToBeDeletedByClusterAutoscaler taint for the length of its simulation (cluster.go:240-252), so it is no member of such a rule's
    domains unless the pod tolerates it — it leaves on Fork, rejoins on Revert, stays out on Commit."""
    import dataclasses
    import random
    from kubernetes_autoscaler_amd.objects import Taint, Toleration
    from kubernetes_autoscaler_amd.workloads import fuzz_removals_domains
    w = fuzz_removals_domains(9000 + seed)
    rng = random.Random(77 + seed)
    for info in w.nodes:
        if rng.random() < 0.2:
            info.node.taints.append(Taint("dedicated", "x", rng.choice(["NoSchedule", "NoExecute", "PreferNoSchedule"])))
    plan = {}
    for info in w.nodes:
        for p in info.pods:
            k = p.spec_key()
            if k not in plan:
                tol = rng.choice([[], [], [Toleration("dedicated", "Equal", "x", "")], [Toleration("", "Exists", "", "")],
                                  [Toleration("ToBeDeletedByClusterAutoscaler", "Exists", "", "NoSchedule")]])
                plan[k] = (tol, [dataclasses.replace(c, node_taints_policy=("Honor" if rng.random() < 0.7 else "Ignore"),
                                                     node_affinity_policy=("Ignore" if rng.random() < 0.3 else "Honor")) for c in p.spread_constraints])
            p.tolerations, p.spread_constraints = list(plan[k][0]), list(plan[k][1])
    check(case_of(w), w.name)
