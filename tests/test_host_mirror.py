"""CPU tests of the host-side mirror of the reference interfaces (limiter, thresholds, expander chain
configuration, object builders) against the reference's own unit-test vectors."""
import json
import os

import pytest

from kubernetes_autoscaler_amd import estimator as est
from kubernetes_autoscaler_amd import expander, objects

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
