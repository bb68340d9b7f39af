"""CPU tests of the host-side mirror of the reference interfaces (limiter, thresholds, expander chain
configuration, object builders) against the reference's own unit-test vectors."""
import json
import os

import pytest

from kubernetes_autoscaler_amd import estimator as est
from kubernetes_autoscaler_amd import expander, objects, workloads

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


def test_min_limit():
    for base, target, want in GOLD["min_limit"]["cases"]:
        assert est.get_min_limit(base, target) == want


class DynamicThreshold:
    """dynamicThreshold of threshold_based_limiter_test.go:41-53."""
    def __init__(self, n): self.n = n
    def duration_limit(self, ng, ctx): return 0.0
    def node_limit(self, ng, ctx):
        self.n += 1
        return self.n


@pytest.mark.parametrize("case", GOLD["limiter"]["cases"], ids=lambda c: c["name"])
def test_threshold_based_limiter(case):
    if "dynamic_threshold_start" in case:
        ths = [DynamicThreshold(case["dynamic_threshold_start"])]
    else:
        ths = [est.StaticThreshold(n, 0.0) for n in case["thresholds"]]
    lim = est.ThresholdBasedEstimationLimiter(ths)
    lim.start_estimation([], None, None)
    for op in case["ops"]:
        if op == "reset":
            lim.end_estimation()
            lim.start_estimation([], None, None)
        else:
            assert lim.permission_to_add_node() == (op == "allow")
    assert lim.nodes == case["expect_nodes"]


def test_negative_duration_forbids_every_node():
    # "binpacking is stopped if at least one threshold has negative max duration limit" (:112-121)
    lim = est.ThresholdBasedEstimationLimiter([est.StaticThreshold(100, -1), est.StaticThreshold(10, 3600.0)])
    lim.start_estimation([], None, None)
    assert not lim.permission_to_add_node() and lim.device_max_nodes() == -1


@pytest.mark.parametrize("case", GOLD["sng_capacity_threshold"]["cases"], ids=lambda c: c["name"])
def test_sng_capacity_threshold(case):
    cur = est.NodeGroup("main-ng", case["current"][0], case["current"][1])
    ctx = est.EstimationContext(0, [est.NodeGroup(f"ng{i}", m, t) for i, (m, t) in enumerate(case["similar"])], 0)
    assert est.SngCapacityThreshold().node_limit(cur, ctx) == case["want"]
    assert est.SngCapacityThreshold().node_limit(cur, None) == 0


def test_cluster_capacity_threshold():
    for mx, cur, want in GOLD["cluster_capacity_threshold"]["cases"]:
        assert est.ClusterCapacityThreshold().node_limit(None, est.EstimationContext(mx, [], cur)) == want
    assert est.ClusterCapacityThreshold().duration_limit(None, None) == 0


def test_expander_chain_configuration():
    assert expander.kinds_of(["least-waste"]) == [1]
    assert expander.kinds_of(["most-pods", "least-nodes", "random"]) == [2, 0]
    with pytest.raises(ValueError):
        expander.kinds_of(["priority"])


def test_builders_match_the_reference_helpers():
    p = objects.build_test_pod("p", 350, 1000, objects.with_namespace("universe"), objects.with_labels({"app": "x"}),
                               objects.with_host_port(5555))
    assert (p.namespace, p.requests, p.host_ports[0].host_port) == ("universe", {"cpu": 350, "memory": 1000}, 5555)
    n = objects.make_node(1000, 5000, 10, "template", "zone-mars")
    assert n.allocatable == {"cpu": 1000, "memory": 5000 * objects.MiB, "pods": 10}
    assert n.labels == {"kubernetes.io/hostname": "template", "topology.kubernetes.io/zone": "zone-mars"}
    assert objects.build_test_node("n", 1000, 2000000).allocatable["pods"] == 100
    assert len(objects.make_pod_equivalence_group(p, 7).pods) == 7
    assert p.fastpath_requests() == (350 * 1e-3, 1000.0)


def test_bulk_resource_pegs_equal_object_path():
    """casim_enc_add_resource_pegs (one ABI crossing) must build the same tables as pod objects."""
    import numpy as np
    from kubernetes_autoscaler_amd import Encoder, workloads
    w = workloads.config_c1(seed_offset=3, n_pegs=40, pods_per_peg=7, cap=32)
    a = Encoder()
    ids = [a.add_peg(pg) for pg in w.pegs]
    a.add_group(w.groups[0].template, max_nodes=32, pegs=ids)
    a.finalize()
    b = Encoder()
    ids_b = b.add_resource_pegs(np.array(workloads.c1_pairs(3, 40), dtype=np.int64), np.full(40, 7, np.int32))
    b.add_group(w.groups[0].template, max_nodes=32, pegs=list(ids_b))
    b.finalize()
    assert list(ids_b) == ids
    for f in ("n_pegs", "n_res", "w_taint", "w_label", "w_excl", "w_zone"):
        assert getattr(a.pegs, f) == getattr(b.pegs, f)
    n = a.pegs.n_pegs
    assert [a.pegs.req[i] for i in range(2 * n)] == [b.pegs.req[i] for i in range(2 * n)]
    assert [a.pegs.count[i] for i in range(n)] == [b.pegs.count[i] for i in range(n)]
    assert [a.pegs.flags[i] for i in range(n)] == [b.pegs.flags[i] for i in range(n)]
    assert [a.pegs.fp_cpu[i] for i in range(n)] == [b.pegs.fp_cpu[i] for i in range(n)]
    assert [a.pegs.fp_mem[i] for i in range(n)] == [b.pegs.fp_mem[i] for i in range(n)]


# ---- equivalence groups (core/scaleup/equivalence/groups_test.go) ----------------------------
def _gold():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["equivalence_groups"]


def test_group_schedulable_pods_for_node():
    from kubernetes_autoscaler_amd.equivalence import build_pod_groups, group_pods_by_scheduling_properties
    from kubernetes_autoscaler_amd.objects import build_test_pod
    G = _gold()["group_schedulable_pods_for_node"]
    pods = []
    for p in G["pods"]:
        pod = build_test_pod(p["name"], p["cpu"], p["mem"])
        pod.controller_uid = p.get("controller", "")
        pod.spec_extra = p.get("spec_extra", "")
        pods.append(pod)
    groups = group_pods_by_scheduling_properties(pods)
    assert sorted(sorted(q.name for q in g) for g in groups) == sorted(sorted(g) for g in G["want_groups"])
    assert [len(g.pods) for g in build_pod_groups(pods)] == [len(g) for g in groups]


def test_equivalence_group_size_limiting():
    from kubernetes_autoscaler_amd.equivalence import group_pods_by_scheduling_properties
    from kubernetes_autoscaler_amd.objects import build_test_pod
    G = _gold()["size_limiting"]
    pods = []
    for i in range(G["n_pods"]):
        p = build_test_pod(f"p{i}", G["cpu"], G["mem"])
        p.controller_uid = G["controller"]
        p.labels = {"uniqueLabel": f"l{i}"}
        pods.append(p)
    assert [len(g) for g in group_pods_by_scheduling_properties(pods)] == G["want_group_sizes"]
    # the 11th distinct spec is not cached: a twin of it still opens its own group, a twin of the 1st joins it
    twin_last = build_test_pod("twin-last", G["cpu"], G["mem"]); twin_last.controller_uid = G["controller"]; twin_last.labels = {"uniqueLabel": "l10"}
    twin_first = build_test_pod("twin-first", G["cpu"], G["mem"]); twin_first.controller_uid = G["controller"]; twin_first.labels = {"uniqueLabel": "l0"}
    sizes = [len(g) for g in group_pods_by_scheduling_properties(pods + [twin_last, twin_first])]
    assert sizes == [2] + [1] * 11


def test_equivalence_group_ignores_daemonsets():
    from kubernetes_autoscaler_amd.equivalence import group_pods_by_scheduling_properties
    from kubernetes_autoscaler_amd.objects import build_test_pod
    G = _gold()["ignores_daemonsets"]
    pods = []
    for i in range(G["n_pods"]):
        p = build_test_pod(f"p{i + 1}", G["cpu"], G["mem"])
        p.controller_uid = G["controller"]
        p.daemonset = True
        pods.append(p)
    assert len(group_pods_by_scheduling_properties(pods)) == G["want_groups"]


def test_bench_multi_process_cpu_leg():
    """bench.py's cpu_baseline_all_cores: independent oracle processes started together (no GPU, no torch in the workers)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = bench.cpu_baseline_all_cores("C0", 1000, budget_s=0.3, max_procs=2)
    assert "error" not in out, out
    assert out["cores"] in (1, 2) and out["sims_per_s"] > 0 and out["unit"] == "checks/s"


def test_bench_single_thread_cpu_leg_uses_the_native_loop():
    """bench.py's cpu_baseline: orc_scale_up_simulation (one native call per simulation) == the per-call oracle path."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from harness import GroupSpec, Scenario, run_oracle
    make = lambda seed_offset=0: workloads.config_c2(seed_offset, n_groups=6, n_pegs=40, pods_per_peg=5, cap=8)
    w, s, run = bench.oracle_simulation(workloads, make, 3)
    got, _, runs = run()
    s.close()
    want = run_oracle(Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True))
    assert runs > 0 and len(got) == len(want)
    for (a, ia), (b, ib) in zip(got, want):
        assert ia == ib and list(a.order) == list(b.order) and list(a.placed) == list(b.placed)
        assert (a.node_count, a.pods_scheduled, a.nodes_added, a.limiter_nodes, a.last_index_out) == \
               (b.node_count, b.pods_scheduled, b.nodes_added, b.limiter_nodes, b.last_index_out)
    out = bench.cpu_baseline(workloads, make, range(2), 1000, budget_s=0.2)
    assert out["kind"] == "port" and out["cores"] == 1 and out["sims_per_s"] > 0


def test_hints_drop_old_property():
    """hints_test.go TestDropOld (testing/quick property): after DropOld the keys set since the previous DropOld survive
    exactly one more generation."""
    import random
    from kubernetes_autoscaler_amd.scheduling import Hints
    rng = random.Random(5)

    def incorrect(s, want, all_keys):
        return any(s.get(k) != k for k in want) or any(s.get(k) is not None for k in set(all_keys) - set(want))

    for _ in range(200):
        initial = [f"k{rng.randrange(40)}" for _ in range(rng.randrange(8))]
        final = [f"k{rng.randrange(40)}" for _ in range(rng.randrange(8))]
        all_keys = initial + final
        s = Hints()
        assert not incorrect(s, [], all_keys)
        for k in initial:
            s.set(k, k)
        assert not incorrect(s, initial, all_keys)
        s.drop_old()
        assert not incorrect(s, initial, all_keys)
        for k in final:
            s.set(k, k)
        assert not incorrect(s, all_keys, all_keys)
        s.drop_old()
        assert not incorrect(s, final, all_keys)


def test_pod_requests_follows_the_sidecar_formula():
    """resource.PodRequests (V/component-helpers/resource/helpers.go:151-281); the vendored module ships no tests, the
    expectations are worked by hand from the formula in its comments (:233-241)."""
    from kubernetes_autoscaler_amd.objects import Container, pod_requests
    C = lambda cpu, mem=0, always=False: Container({"cpu": cpu, "memory": mem}, always)
    assert pod_requests([C(100, 10), C(200, 20)]) == {"cpu": 300, "memory": 30}
    # a plain init container only sets a floor
    assert pod_requests([C(100, 10), C(200, 20)], [C(500, 5)]) == {"cpu": 500, "memory": 30}
    # sidecars add to the total; a plain init container runs next to the sidecars started before it
    got = pod_requests([C(100)], [C(50, 0, True), C(400), C(30, 0, True)])
    assert got == {"cpu": 450, "memory": 0}            # running: 100 + 50 + 30 = 180; init step 2: 400 + 50 = 450
    got = pod_requests([C(100)], [C(50, 0, True), C(10), C(30, 0, True)], overhead={"cpu": 7, "memory": 3})
    assert got == {"cpu": 187, "memory": 3}            # running 180 beats every init step (50, 60, 80); + overhead
    # pod-level requests replace cpu / memory only, overhead still on top; other resources keep the container sums
    got = pod_requests([Container({"cpu": 100, "memory": 10, "example.com/gpu": 1})], pod_level={"cpu": 1000, "example.com/gpu": 9}, overhead={"cpu": 1})
    assert got == {"cpu": 1001, "memory": 10, "example.com/gpu": 1}
    assert pod_requests([]) == {}
    # resources only some containers name
    assert pod_requests([Container({"cpu": 1}), Container({"ephemeral-storage": 5})], [Container({"memory": 9})]) == {"cpu": 1, "ephemeral-storage": 5, "memory": 9}


def test_similar_pods_scheduling_like_the_reference_tests():
    """simulator/scheduling/similar_pods_test.go:32-150 on the mirror's memo (scheduling.SimilarPodsScheduling)."""
    import json, os
    from kubernetes_autoscaler_amd.objects import Pod
    from kubernetes_autoscaler_amd.scheduling import SimilarPodsScheduling
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))["similar_pods_scheduling"]

    def build(p, labels=None):
        vol = p.get("volume", "")
        return Pod(name=p["name"], requests={"cpu": p["cpu"], "memory": p["mem"]}, controller_uid=p["controller"], daemonset=bool(p.get("daemonset")),
                   labels=dict(labels or {}), spec_extra="" if vol in ("", "projected") else vol)
    for case in G["cases"]:
        memo = SimilarPodsScheduling()
        if "generate" in case:
            g = case["generate"]
            pods = [build({"name": f"p{i}", "cpu": g["cpu"], "mem": g["mem"], "controller": g["controller"]}, {g["unique_label"]: f"l{i}"}) for i in range(g["count"])]
            assert not any(memo.is_similar_unschedulable(p) for p in pods)
            for p in pods:
                memo.set_unschedulable(p)
            assert [memo.is_similar_unschedulable(p) for p in pods] == [True] * (g["count"] - 1) + [False], case["name"]
        else:
            pods = {k: build(v) for k, v in case["pods"].items()}
            for op in case["ops"]:
                if op[0] == "set":
                    memo.set_unschedulable(pods[op[1]])
                else:
                    assert memo.is_similar_unschedulable(pods[op[1]]) == op[2], (case["name"], op)
        assert memo.overflowing_controller_count() == case["overflowing"], case["name"]
