"""-m gpu: SURVEY §8 row f1 (filter-out-schedulable) on a real MI355X through the C ABI
(casim_try_schedule_pods), against the CPU oracle — bit-exact node per pending pod."""
import numpy as np
import pytest

import kubernetes_autoscaler_amd as kaa
from harness import SchedCase, assert_sched_matches, sched_gpu, sched_oracle
from kubernetes_autoscaler_amd.objects import NodeInfo, build_test_node, build_test_pod
from kubernetes_autoscaler_amd.scheduling import FilterOutSchedulablePodListProcessor, HintingSimulator, UnsupportedPredicate
from kubernetes_autoscaler_amd.workloads import _node, filter_out_schedulable_benchmark, fuzz_pending, pending_scale
from test_oracle_golden import GOLD, golden_hinted_cases, golden_sched_case
from test_sched_emu import case_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = kaa.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", GOLD["try_schedule_pods"]["cases"], ids=lambda c: c["name"])
def test_golden_try_schedule_pods(ctx, case):
    sc, names = golden_sched_case(case)
    got = sched_gpu(sc, ctx)
    assert {p.name: names[m] for p, m in zip(sc.pods, got[1]) if m >= 0} == case["want"]
    assert_sched_matches(got, sched_oracle(sc), case["name"])


@pytest.mark.parametrize("case", GOLD["pod_schedules_on_hinted_node"]["cases"], ids=lambda c: c["name"])
def test_golden_pod_schedules_on_hinted_node(ctx, case):
    for sc, want in golden_hinted_cases(case):
        rc, node_out, last_index, n_sched = sched_gpu(sc, ctx)
        assert rc == 0 and list(node_out) == want and n_sched == len(want) and last_index == 0


def test_fuzz(ctx):
    for seed in range(300):
        w = fuzz_pending(seed)
        sc = case_of(w)
        assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name)


def test_fuzz_domain_rules(ctx):
    """PodTopologySpread + anti-affinity on non-hostname keys (domain rules) on the MI355X."""
    from kubernetes_autoscaler_amd.workloads import fuzz_pending_domains
    for seed in range(300):
        w = fuzz_pending_domains(seed)
        sc = case_of(w)
        assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name)


@pytest.mark.parametrize("shape", [(1, 30, 1000), (10, 300, 1000), (100, 3000, 1000), (200, 200, 60000), (1000, 1000, 12000)],
                         ids=lambda s: f"{s[0]}n_{s[1]}s_{s[2]}p")
def test_benchmark_filter_out_schedulable_shapes(ctx, shape):
    """BenchmarkFilterOutSchedulable's grid at full size: every pending pod stays pending.  The oracle runs the
    pods x nodes Filters only for the shapes it finishes in seconds; the property holds for all."""
    w = filter_out_schedulable_benchmark(*shape)
    sc = case_of(w)
    rc, node_out, li, ns = sched_gpu(sc, ctx)
    assert rc == 0 and ns == 0 and li == 0 and (node_out == -1).all() and len(node_out) == shape[2]
    if shape[0] * shape[2] <= 200000:
        assert_sched_matches((rc, node_out, li, ns), sched_oracle(sc), w.name)


def test_packing_on_a_large_cluster(ctx):
    """4500 nodes: the TrySchedulePods workgroup runs 512 threads from 4096 nodes on."""
    w = pending_scale(4500, 12000, n_classes=24, seed=17)
    sc = case_of(w)
    assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name)


def test_packing_at_scale(ctx):
    """2000 heterogeneous nodes (state in LDS or HBM), 20000 pending pods that mostly fit."""
    w = pending_scale(2000, 20000, n_classes=32, seed=7)
    sc = case_of(w)
    got = sched_gpu(sc, ctx)
    want = sched_oracle(sc)
    assert want[2] > 10000
    assert_sched_matches(got, want, w.name)
    # size-independent properties: no node over-committed, pods of one spec placed in run order
    node_out = got[1]
    cpu = np.zeros(len(w.nodes), np.int64)
    for p, m in zip(w.pods, node_out):
        if m >= 0:
            cpu[m] += p.requests["cpu"]
    for i, info in enumerate(w.nodes):
        used = sum(q.requests.get("cpu", 0) for q in info.pods)
        assert cpu[i] == 0 or used + cpu[i] <= info.node.allocatable["cpu"]


def test_reference_filter_out_schedulable_table(ctx):
    """podlistprocessor/filter_out_schedulable_test.go TestFilterOutSchedulable rows (tests/golden)."""
    import json
    import os
    from harness import SchedCase, assert_sched_matches, sched_oracle
    from test_oracle_golden import golden_filter_case
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")) as f:
        rows = json.load(f)["filter_out_schedulable"]["cases"]
    for row in rows:
        nodes, cands, acceptable = golden_filter_case(row)
        node_filter = (lambda info: False) if row.get("node_filter") == "none" else None
        proc = FilterOutSchedulablePodListProcessor(ctx, node_filter)
        left = proc.process(nodes, list(cands))
        assert sorted(p.name for p in left) == sorted(row["unscheduled"]), row["name"]
        assert sorted(proc.scheduling_simulator.hints.old) == sorted(f"default/{n}" for n in row["scheduled"]), row["name"]


def test_host_mirror_processor_keeps_hints(ctx):
    """FilterOutSchedulablePodListProcessor.Process twice: the second loop iteration finds every pod on its hinted
    node (hints survive DropOldHints once)."""
    nodes = [NodeInfo(_node(f"n{i}", 2000, 8 << 30, 110)) for i in range(5)]
    pods = [build_test_pod(f"p{i}", 600, 1 << 20) for i in range(17)]
    for i, p in enumerate(pods):
        p.priority = i % 3
    proc = FilterOutSchedulablePodListProcessor(ctx)
    left = proc.process(nodes, list(pods))
    assert len(left) == 17 - 15  # 3 pods of 600m per 2000m node
    assert all(p.priority == 0 for p in left)  # highest priority first
    sim = proc.scheduling_simulator
    first = {k: v for k, v in sim.hints.old.items()}
    assert len(first) == 15
    statuses, _ = sim.try_schedule_pods(nodes, [p for p in pods if p not in left])
    assert {f"{s.pod.namespace}/{s.pod.name}": s.node_name for s in statuses} == first


def test_host_mirror_delegates_unsupported(ctx):
    from kubernetes_autoscaler_amd.objects import Pod
    sim = HintingSimulator(ctx)
    with pytest.raises(UnsupportedPredicate):
        sim.try_schedule_pods([NodeInfo(build_test_node("n", 1000, 1000))], [Pod(name="s", requests={"cpu": 1}, topology_spread=True)])


def test_schedulable_pod_groups_matrix(ctx):
    """BuildPodGroups on the host + the PEG x node-group matrix in one device call == CheckPredicates
    of every group exemplar on every template (oracle)."""
    from kubernetes_autoscaler_amd import workloads
    from kubernetes_autoscaler_amd.equivalence import build_pod_groups, schedulable_pod_groups
    from oracle_driver import OracleScenario
    for seed in range(40):
        w = workloads.fuzz(7000 + seed, max_groups=6, max_pegs=24)
        pods = []
        for i, pg in enumerate(w.pegs):
            for k in range(min(len(pg.pods), 3)):
                p = pg.pods[0]
                q = type(p)(**{**p.__dict__, "name": f"{p.name}-{k}", "controller_uid": f"ctrl{i % 5}"})
                pods.append(q)
        groups = build_pod_groups(pods)
        assert sum(len(g.pods) for g in groups) == len(pods)
        templates = {g.template.node.name: g.template for g in w.groups}
        if any(g.pods[0].anti_affinity and g.pods[0].anti_affinity[0].topology_key != "kubernetes.io/hostname" for g in groups):
            continue  # zone terms need the existing-cluster context of a full scenario
        ok = schedulable_pod_groups(ctx, groups, templates)
        s = OracleScenario()
        for i, (name, tmpl) in enumerate(templates.items()):
            t = s.node(tmpl)
            for j, g in enumerate(groups):
                assert bool(ok[i, j]) == s.check_predicates(t, g.pods[0])[0], (seed, name, j)
        s.close()


# ---- scale-down removal simulation -----------------------------------------------------------------
def test_removal_reference_table(ctx):
    """simulator/cluster_test.go TestSimulateNodeRemoval rows (tests/golden/reference_vectors.json)."""
    import json
    import os
    from harness import assert_removal_matches, removal_device, removal_oracle
    from test_oracle_golden import golden_removal_case
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")) as f:
        rows = json.load(f)["simulate_node_removal"]["cases"]
    seen = 0
    for row in rows:
        case = golden_removal_case(row)
        if case is None:
            continue
        got = removal_device(case, ctx)
        if row.get("device_delegates"):
            assert got.status == 1, row["name"]
            continue
        assert_removal_matches(got, removal_oracle(case), row["name"])
        assert bool(got.removable[0] == 1) == row["removable"], row["name"]
        seen += 1
    assert seen == 7      # incl. the nodeTaintsPolicy: Honor row (the ghost leaves the domains of such a rule)


def test_fuzz_node_taints_policy_honor(ctx):
    """Spread constraints with nodeTaintsPolicy: Honor next to tainted nodes (eligibility rows built by the encoder)."""
    from kubernetes_autoscaler_amd.workloads import fuzz_pending_domains
    from test_sched_emu import honor_taints_variant
    for seed in range(120):
        w = honor_taints_variant(fuzz_pending_domains(3000 + seed), seed)
        sc = case_of(w)
        assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name)


def test_reference_planner_table(ctx):
    """core/scaledown/planner/planner_test.go TestUpdateClusterState rows through the Planner mirror: injectPods
    (one TrySchedulePods call) then categorizeNodes (one removal call)."""
    import json
    import os
    from kubernetes_autoscaler_amd.scaledown import Planner
    from test_oracle_golden import golden_planner_case
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")) as f:
        rows = json.load(f)["planner_update_cluster_state"]["cases"]
    assert len(rows) == 25
    for row in rows:
        infos, inject = golden_planner_case(row)
        names = [i.node.name for i in infos]
        removable, unremovable, skipped = Planner(ctx, infos).update_cluster_state(names, row["eligible"], inject)
        assert not skipped and [r.node.name for r in removable] == row["unneeded"], row["name"]
        assert [u.node.name for u in unremovable] == [n for n in row["eligible"] if n not in row["unneeded"]], row["name"]


def test_reference_unneeded_nodes_limit_table(ctx):
    """planner_test.go TestUpdateClusterStatUnneededNodesLimit rows + random atomic flags on fuzz clusters."""
    import json
    import os
    import random
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.scaledown import Planner
    from kubernetes_autoscaler_amd.workloads import fuzz_removals
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")) as f:
        rows = json.load(f)["planner_unneeded_nodes_limit"]["cases"]
    for row in rows:
        infos = [NodeInfo(build_test_node(f"n{i}", 1000, 10)) for i in range(row["nodes"])]
        names = [i.node.name for i in infos]
        limit = Planner.unneeded_nodes_limit(row["previously_unneeded"], row["max_parallelism"], row["unneeded_time_s"], row["update_interval_s"])
        removable, unremovable, skipped = Planner(ctx, infos).update_cluster_state(names, names, (), limit, (lambda n: True) if row["atomic"] else None)
        assert len(removable) == row["want_unneeded"] and not unremovable and skipped == names[row["want_unneeded"]:], row["name"]
    for seed in range(40):
        w = fuzz_removals(7000 + seed)
        rng = random.Random(seed)
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=rng.randint(1, 3), last_index=w.last_index)
        case.atomic = [1 if rng.random() < 0.4 else 0 for _ in case.candidates]
        assert_removal_matches(removal_device(case, ctx), removal_oracle(case), w.name)


def test_removal_fuzz(ctx):
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import fuzz_removals
    for seed in range(250):
        w = fuzz_removals(seed)
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=w.max_removable, last_index=w.last_index)
        assert_removal_matches(removal_device(case, ctx), removal_oracle(case), w.name)


@pytest.mark.parametrize("n_nodes", [300, 2500])
def test_removal_at_scale(ctx, n_nodes):
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import removal_scale
    w = removal_scale(n_nodes, pods_per_node=10, frac_candidates=0.3, seed=11)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    got, want = removal_device(case, ctx), removal_oracle(case)
    assert_removal_matches(got, want, w.name)
    assert int((got.removable == 1).sum()) > n_nodes // 20 and len(want["ext"]) > 0
    # size-independent property: the pods of every removable node are all placed, on nodes that were not removed before
    # their move and are not the node itself
    off = np.cumsum([0] + [len(l) for l in case.pod_lists()])
    for k, c in enumerate(w.candidates):
        if got.removable[k] == 1:
            d = got.node_out[off[k]:off[k + 1]]
            assert (d >= 0).all() and (d != c).all()


def test_removal_mirror_planner_loop(ctx):
    from harness import RemovalCase, removal_oracle
    from kubernetes_autoscaler_amd.scaledown import RemovalSimulator
    from kubernetes_autoscaler_amd.workloads import fuzz_removals
    for seed in range(30):
        w = fuzz_removals(2000 + seed)
        nodes = [NodeInfo(info.node, list(info.pods)) for info in w.nodes]
        want = removal_oracle(RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, persist=True,
                                          max_removable=w.max_removable, last_index=w.last_index))
        sim = RemovalSimulator(ctx, nodes, persist_successful_simulations=True)
        sim.last_index = w.last_index
        dest = {info.node.name: (w.destination is None or bool(w.destination[i])) for i, info in enumerate(w.nodes)}
        removable, unremovable, skipped = sim.simulate_node_removals([w.nodes[c].node.name for c in w.candidates], dest, w.max_removable or None)   # the fuzz rows keep the ABI's 0 = no limit
        assert [r.node.name for r in removable] == [w.nodes[c].node.name for k, c in enumerate(w.candidates) if want["removable"][k] == 1]
        assert [u.node.name for u in unremovable] == [w.nodes[c].node.name for k, c in enumerate(w.candidates) if want["removable"][k] == 0]
        assert sim.last_index == want["last_index"] and sim.device_calls <= 1


def test_removal_fuzz_domain_rules(ctx):
    """The removal loop with PodTopologySpread / zone anti-affinity among the pods: counters lose the candidate's pods
    for the simulation, get them back on revert, lose the node's domain membership on commit."""
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import fuzz_removals_domains
    for seed in range(250):
        w = fuzz_removals_domains(seed)
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=w.max_removable, last_index=w.last_index)
        assert_removal_matches(removal_device(case, ctx), removal_oracle(case), w.name)


def test_domain_rules_at_scale(ctx):
    """2000 nodes in 3 zones, 5000 pending pods of 16 controller specs, half of them with a zone or hostname spread
    constraint (per-pod walks, block-wide minima over 2000 hostname domains) — and the removal loop on the same cluster."""
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.objects import LABEL_ZONE, TopologySpreadConstraint
    w = pending_scale(2000, 5000, n_classes=16, seed=21)
    for p in w.pods:
        c = int(p.labels["app"][1:])
        if c % 4 == 0:
            p.spread_constraints = [TopologySpreadConstraint(2, LABEL_ZONE, 0, dict(p.labels))]
        elif c % 4 == 1:
            p.spread_constraints = [TopologySpreadConstraint(1, "kubernetes.io/hostname", 0, dict(p.labels))]
        p.topology_spread = bool(p.spread_constraints)
    sc = case_of(w)
    got, want = sched_gpu(sc, ctx), sched_oracle(sc)
    assert want[2] > 2000
    assert_sched_matches(got, want, w.name)
    # removal loop: the placed spread pods now RUN on the cluster; the 200 emptiest nodes are candidates
    for p, m in zip(w.pods, got[1]):
        if m >= 0:
            w.nodes[m].pods.append(p)
    util = sorted(range(len(w.nodes)), key=lambda i: (sum(q.requests["cpu"] for q in w.nodes[i].pods), i))
    case = RemovalCase(nodes=w.nodes, candidates=util[:200])
    assert_removal_matches(removal_device(case, ctx), removal_oracle(case), "removals with spread pods")


# ---- required node affinity: several nodeSelectorTerms (ORed) + matchFields on metadata.name ------------------------
def test_reference_node_names_affinity_rows(ctx):
    """TestRunFiltersOnNode's two WithNodeNamesAffinity rows (plugin_runner_test.go:121-136) through the C ABI."""
    from kubernetes_autoscaler_amd.objects import with_node_names_affinity
    G = GOLD["run_filters_on_node"]
    nd = G["node"]
    rows = [c for c in G["cases"] if "affinity" in c["name"]]
    assert len(rows) == 2
    for case in rows:
        cpu, mem = G["pods"][case["test"]]
        pod = build_test_pod(case["test"], cpu, mem, with_node_names_affinity(*G["node_names_affinity"][case["test"]]))
        sc = SchedCase(nodes=[NodeInfo(build_test_node(nd["name"], nd["cpu"], nd["mem"]))], pods=[pod])
        got = sched_gpu(sc, ctx)
        assert got[0] == 0 and (got[1][0] == 0) == case["ok"]
        assert_sched_matches(got, sched_oracle(sc), case["name"])


def test_fuzz_node_affinity_terms(ctx):
    """The CPU suite's term fuzz (tests/test_node_affinity_terms_emu.py) on the MI355X: TrySchedulePods with and without
    domain rules, and the removal loop."""
    from harness import assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import add_random_node_affinity_terms, fuzz_pending_domains, fuzz_removals
    from test_removal_emu import case_of as removal_case_of
    ran = 0
    for seed in range(150):
        for gen in (fuzz_pending, fuzz_pending_domains):
            w = gen(seed, max_nodes=24, max_pods=60)
            if add_random_node_affinity_terms(seed, w.pods, w.nodes):
                sc = case_of(w)
                assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name)
                ran += 1
        w = fuzz_removals(seed)
        if add_random_node_affinity_terms(seed, [p for info in w.nodes for p in info.pods], w.nodes):
            case = removal_case_of(w)
            assert_removal_matches(removal_device(case, ctx), removal_oracle(case), w.name)
            ran += 1
    assert ran > 300


def test_fuzz_namespace_selectors(ctx):
    """namespaceSelector of anti-affinity terms, resolved by the encoder against the namespace lister
    (tests/test_namespace_selector_emu.py) on the MI355X: TrySchedulePods with and without domain rules, the removal loop."""
    from harness import assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.objects import namespaces
    from kubernetes_autoscaler_amd.workloads import add_random_namespace_selectors, fuzz_pending_domains, fuzz_removals
    from test_removal_emu import case_of as removal_case_of
    for seed in range(120):
        for gen, host_only in ((fuzz_pending, True), (fuzz_pending_domains, False)):
            w = gen(seed, max_nodes=24, max_pods=60)
            table = add_random_namespace_selectors(seed, list(w.pods) + [p for info in w.nodes for p in info.pods], hostname_only=host_only)
            with namespaces(table):
                sc = case_of(w)
                assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name)
        w = fuzz_removals(seed)
        table = add_random_namespace_selectors(seed, [p for info in w.nodes for p in info.pods], hostname_only=True)
        with namespaces(table):
            case = removal_case_of(w)
            assert_removal_matches(removal_device(case, ctx), removal_oracle(case), w.name)


def test_extended_resource_lanes_on_the_device(ctx):
    """More than two resource lanes select the `<.., 8>` instantiations of sched_kernel (148-160 vector registers, no scratch since the
    launch bound of 512): TrySchedulePods with and without domain rules and the removal loop, 4 and 8 lanes, against the oracle."""
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import fuzz_pending_domains, fuzz_removals
    from test_sched_lanes_emu import LANES4, LANES8, with_extra_resources
    for seed in range(80):
        lanes = LANES8 if seed % 2 else LANES4
        w = fuzz_pending(seed) if seed % 4 < 2 else fuzz_pending_domains(seed)
        nodes, pods = with_extra_resources(w.nodes, w.pods, lanes, seed)
        case = SchedCase(nodes=nodes, pods=pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index,
                         lanes=lanes)
        assert_sched_matches(sched_gpu(case, ctx), sched_oracle(case), f"{w.name} {len(lanes)} lanes")
    for seed in range(50):
        lanes = LANES8 if seed % 2 else LANES4
        w = fuzz_removals(seed)
        nodes, _ = with_extra_resources(w.nodes, [], lanes, seed)
        case = RemovalCase(nodes=nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist, max_removable=w.max_removable,
                           last_index=w.last_index, lanes=lanes)
        assert_removal_matches(removal_device(case, ctx), removal_oracle(case), f"{w.name} {len(lanes)} lanes")
