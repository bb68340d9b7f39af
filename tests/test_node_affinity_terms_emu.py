"""Required node affinity with several nodeSelectorTerms (ORed) and matchFields on metadata.name
(V/component-helpers/scheduling/corev1/nodeaffinity/nodeaffinity.go:60-107,187-198,260-291): the encoder folds the whole
term list into ONE per-node bit, the kernels are unchanged.  Product encoder + kernels under the wave emulator against the
object-level oracle, bit for bit, through every consumer: TrySchedulePods (f1), the removal loop (f4), the estimator in
template mode and on the whole snapshot (K_est, f3).  The two WithNodeNamesAffinity rows of TestRunFiltersOnNode
(plugin_runner_test.go:121-136) are the reference's own known answers for matchFields."""
import pytest

import test_cluster_estimate_emu as ce
import test_kernels_emu_fuzz as kf
import test_removal_emu as rm
import test_sched_emu as se
from harness import SchedCase, assert_matches_oracle, encode, run_emu, run_oracle, sched_emu
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.objects import (NodeInfo, NodeSelectorTerm, Requirement, build_test_node, build_test_pod,
                                               with_node_names_affinity)
from test_oracle_golden import GOLD


@pytest.mark.parametrize("case", [c for c in GOLD["run_filters_on_node"]["cases"] if "affinity" in c["name"]], ids=lambda c: c["name"])
def test_reference_node_names_affinity_rows_on_the_device_path(case):
    G = GOLD["run_filters_on_node"]
    nd = G["node"]
    cpu, mem = G["pods"][case["test"]]
    pod = build_test_pod(case["test"], cpu, mem, with_node_names_affinity(*G["node_names_affinity"][case["test"]]))
    sc = SchedCase(nodes=[NodeInfo(build_test_node(nd["name"], nd["cpu"], nd["mem"]))], pods=[pod])
    _, node_out, _, n_sched, _ = sched_emu(sc)
    assert (node_out[0] == 0) == case["ok"] and n_sched == int(case["ok"])
    se.check(sc, case["name"])


def test_terms_are_ored_and_fields_anded():
    nodes = [NodeInfo(build_test_node(f"n{i}", 1000, 1 << 30)) for i in range(4)]
    nodes[1].node.labels["disk"] = "ssd"
    nodes[3].node.labels["disk"] = "ssd"
    def pod(name, terms):
        p = build_test_pod(name, 100, 0)
        p.node_affinity_terms = terms
        return p
    f = lambda op, *v: Requirement("metadata.name", op, list(v))
    pods = [
        pod("a", [NodeSelectorTerm(match_fields=[f("In", "n2")]), NodeSelectorTerm([Requirement("disk", "In", ["ssd"])], [f("NotIn", "n1")])]),  # n2 or n3
        pod("b", [NodeSelectorTerm(match_fields=[f("In", "n0"), f("In", "n1")])]),           # conflicting names: nowhere
        pod("c", [NodeSelectorTerm(), NodeSelectorTerm([Requirement("disk", "Exists", [])])]),   # empty term dropped: n1 or n3 (n3: the search resumes after n2)
        pod("d", []),                                                                        # a selector without terms: nowhere
        pod("e", [NodeSelectorTerm([Requirement("disk", "Exists", ["x"])]), NodeSelectorTerm(match_fields=[f("In", "n0", "n1")])]),  # both fail to parse
        pod("g", [NodeSelectorTerm([Requirement("disk", "DoesNotExist", [])], [f("NotIn", "n0")])]),   # n2 only
    ]
    sc = SchedCase(nodes=nodes, pods=pods, last_index=0)
    node_out, _, n = se.check(sc, "ored terms")
    assert list(node_out) == [2, -1, 3, -1, -1, 2] and n == 3   # c starts after lastIndex = 2: n3 before n1


@pytest.mark.parametrize("seed", range(300))
def test_fuzz_try_schedule_pods(seed):
    w = workloads.fuzz_pending(seed, max_nodes=24, max_pods=60)
    if not workloads.add_random_node_affinity_terms(seed, w.pods, w.nodes):
        pytest.skip("no pod drew a term list")
    se.check(se.case_of(w), w.name)


@pytest.mark.parametrize("seed", range(150))
def test_fuzz_try_schedule_pods_with_domain_rules(seed):
    """spread constraints count a node only when it matches the pod's required node affinity (nodeAffinityPolicy: Honor)"""
    w = workloads.fuzz_pending_domains(seed, max_nodes=24, max_pods=50)
    if not workloads.add_random_node_affinity_terms(seed, w.pods, w.nodes):
        pytest.skip("no pod drew a term list")
    se.check(se.case_of(w), w.name)


@pytest.mark.parametrize("seed", range(200))
def test_fuzz_removals(seed):
    w = workloads.fuzz_removals(seed)
    if not workloads.add_random_node_affinity_terms(seed, [p for info in w.nodes for p in info.pods], w.nodes):
        pytest.skip("no pod drew a term list")
    rm.check(rm.case_of(w), w.name)


@pytest.mark.parametrize("seed", range(150))
def test_fuzz_estimate_template_mode(seed):
    """an Estimate: terms over template labels only; nothing may be delegated"""
    w = workloads.fuzz(7000 + seed)
    if not workloads.add_random_node_affinity_terms(seed, [p for pg in w.pegs for p in pg.pods], [g.template for g in w.groups], allow_per_node=False):
        pytest.skip("no pod drew a term list")
    sc = kf.scenario_of(w)
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


def test_template_mode_delegates_terms_that_read_the_node_name():
    w = workloads.fuzz(7001)
    w.pegs[0].pods[0].node_affinity_terms = [NodeSelectorTerm(match_fields=[Requirement("metadata.name", "NotIn", ["x"])])]
    for p in w.pegs[0].pods[1:]:
        p.node_affinity_terms = w.pegs[0].pods[0].node_affinity_terms
    enc = encode(kf.scenario_of(w))
    assert enc.pegs.flags[0] & _abi.PEG_UNSUPPORTED


@pytest.mark.parametrize("seed", range(100))
def test_fuzz_estimate_on_cluster(seed):
    w = workloads.fuzz_estimate_domains(seed)
    if not workloads.add_random_node_affinity_terms(seed, [p for pg in w.pegs for p in pg.pods], [g.template for g in w.groups] + list(w.existing), allow_per_node=False):
        pytest.skip("no pod drew a term list")
    sc = ce.scenario_of(w)
    if ce.cluster_estimate_emu(sc)[0] == 1:
        pytest.skip("delegated (hostname anti-affinity with an unnamed node)")
    ce.check(sc, w.name)
