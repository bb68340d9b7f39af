"""SURVEY §8c determinism caveat: the reference's node list order is Go map iteration, so a workload's
(NodeCount, Pods) is only a well-defined reference answer if it does not depend on that order.  The bench /
BASELINE workloads and the reference's golden rows must pass this probe: the oracle with a freshly shuffled node
list for EVERY scheduling attempt gives the same node count, pod count and per-PEG placed counts as the canonical
insertion order."""
import pytest

from harness import GroupSpec, Scenario, run_oracle
from kubernetes_autoscaler_amd import workloads
from test_kernels_emu_golden import GOLD, golden_scenario


def scenario_of(w):
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing,
                    lanes=w.lanes)


def summary(res):
    return [(e.node_count, e.pods_scheduled, e.nodes_added, e.limiter_nodes, list(e.order), list(e.placed)) for e, _ in res]


CASES = {
    "C0": lambda: workloads.config_c0(),
    "C1 (40 PEGs x 50 pods, cap 64)": lambda: workloads.config_c1(n_pegs=40, pods_per_peg=50, cap=64),
    "C1 full size": lambda: workloads.config_c1(),
    "C2 (6 groups, 60 PEGs)": lambda: workloads.config_c2(n_groups=6, n_pegs=60, pods_per_peg=10, cap=20),
    "C4 (6 groups, 60 PEGs)": lambda: workloads.config_c4(n_groups=6, n_pegs=60, pods_per_peg=10, cap=20),
}


# Workloads whose partially consumed PEGs leave order-dependent residuals ("not order-independent in
# general"): the reference itself has no single answer for pods_scheduled there — a Go run and the canonical order can
# legitimately differ by a few pods — while the node count (what the expander consumes) is the same under every order.
ORDER_DEPENDENT_PODS = {"C1 full size", "C4 (6 groups, 60 PEGs)"}


@pytest.mark.parametrize("name", list(CASES))
def test_bench_workloads_and_the_node_order(name):
    sc = scenario_of(CASES[name]())
    want = summary(run_oracle(sc))
    for seed in (11, 22, 33):
        got = summary(run_oracle(sc, list_shuffle_seed=seed))
        if name not in ORDER_DEPENDENT_PODS:
            assert got == want, f"{name}: node order {seed} changes the result"
            continue
        for g, w in zip(got, want):
            assert g[0] == w[0] and g[2] == w[2] and g[3] == w[3] and g[4] == w[4], f"{name}: node count / PEG order depend on the node order"
            assert abs(g[1] - w[1]) <= max(2, w[1] // 20), f"{name}: pods scheduled {g[1]} vs {w[1]}"


@pytest.mark.parametrize("case", GOLD["cases"] + GOLD["topology_spread_cases"], ids=lambda c: c["name"])
def test_golden_rows_do_not_depend_on_the_node_order(case):
    sc = golden_scenario(case)
    want = [(e.node_count, e.pods_scheduled) for e, _ in run_oracle(sc)]
    assert want == [(case["expect_nodes"], case["expect_pods"])]
    for seed in (5, 6, 7):
        assert [(e.node_count, e.pods_scheduled) for e, _ in run_oracle(sc, list_shuffle_seed=seed)] == want
