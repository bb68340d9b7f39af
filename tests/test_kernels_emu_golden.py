"""The product encoder + product kernels (compiled for the host, run by the wave emulator) against the
reference's known-answer vectors and the oracle.  CPU only; the same scenarios run on the MI355X in
test_gpu_parity.py."""
import json
import os

import pytest

from harness import GroupSpec, Scenario, assert_matches_oracle, encode, run_emu, run_oracle
from kubernetes_autoscaler_amd.objects import (NodeInfo, build_test_pod, make_node, make_pod_equivalence_group, with_host_port,
                                               with_labels, with_max_skew, with_namespace)

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["binpacking_estimate"]


def golden_scenario(case, fastpath=False, template_pods=None):
    setup = GOLD["setup"]
    ex, t = setup["existing_node"], setup["template"]
    pegs = []
    for g in case["pegs"]:
        opts = [with_namespace(setup["namespace"]), with_labels(setup["labels"])]
        if g.get("host_port"):
            opts.append(with_host_port(g["host_port"]))
        if g.get("max_skew"):
            opts.append(with_max_skew(*g["max_skew"]))
        pegs.append(make_pod_equivalence_group(build_test_pod("estimatee", g["cpu"], g["mem"], *opts), g["count"]))
    tmpl = NodeInfo(make_node(case["millicores"], case["memory_mib"], template_pods or t["pods"], t["name"], t["zone"]))
    return Scenario(pegs=pegs, groups=[GroupSpec(tmpl, max_nodes=case["max_nodes"])],
                    existing=[NodeInfo(make_node(ex["cpu"], ex["mem_mib"], ex["pods"], ex["name"], ex["zone"]))], fastpath=fastpath)


@pytest.mark.parametrize("fastpath", [False, True])
@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: c["name"])
def test_golden_rows(case, fastpath):
    if fastpath and not case["check_fastpath"]:
        pytest.skip("reference only asserts fastpath parity for single-group rows")
    sc = golden_scenario(case, fastpath)
    res, _ = run_emu(encode(sc), fastpath=fastpath)
    assert (int(res.node_count[0]), int(res.pods_scheduled[0])) == (case["expect_nodes"], case["expect_pods"])
    assert_matches_oracle(res, run_oracle(sc), case["name"])


@pytest.mark.parametrize("lds_budget", [0, 1024])
def test_benchmark_vector(lds_budget):
    """BenchmarkBinpackingEstimate: 51000 pods -> 2595 nodes.  lds_budget=1024 forces the HBM-scratch variants."""
    b = GOLD["benchmark"]
    sc = golden_scenario(b, template_pods=b["template_pods"])
    res, _ = run_emu(encode(sc), lds_budget=lds_budget)
    assert (int(res.node_count[0]), int(res.pods_scheduled[0])) == (b["expect_nodes"], b["expect_pods"])
    assert_matches_oracle(res, run_oracle(sc), "benchmark")


@pytest.mark.parametrize("case", GOLD["topology_spread_cases"], ids=lambda c: c["name"])
def test_topology_spread_rows_are_delegated(case):
    """Rows 6-8 need PodTopologySpread (and pods landing on nodes already in the cluster): the engine fails closed —
    the group comes back CASIM_NG_UNSUPPORTED with nothing placed — and the shim runs the Go estimator; the oracle
    itself reproduces the reference's numbers (test_oracle_golden.py)."""
    sc = golden_scenario(case)
    res, _ = run_emu(encode(sc))
    assert int(res.status[0]) == 1 and int(res.node_count[0]) == 0 and int(res.pods_scheduled[0]) == 0
    est, _ = run_oracle(sc)[0]
    assert (est.node_count, est.pods_scheduled) == (case["expect_nodes"], case["expect_pods"])
