"""BinpackingNodeEstimator.Estimate on the whole snapshot (K_est, casim_estimate.h) under the wave
emulator against the object-level oracle — all eight TestBinpackingEstimate rows (the three PodTopologySpread rows
included: the hostname retry places pods on the node that is already in the cluster) and scenarios the template-mode
packer handles too (both paths must agree with the oracle)."""
import pytest

from harness import GroupSpec, Scenario, assert_cluster_estimate_matches, cluster_estimate_emu, run_oracle
from kubernetes_autoscaler_amd import workloads
from test_kernels_emu_golden import GOLD, golden_scenario


def check(sc, what="", lds=(0, 64)):
    want = run_oracle(sc)
    for gi in range(len(sc.groups)):
        est, ids = want[gi]
        for b in lds:
            assert_cluster_estimate_matches(cluster_estimate_emu(sc, gi, lds_budget=b), est, ids, f"{what} group {gi} lds={b}")
    return want


@pytest.mark.parametrize("case", GOLD["cases"] + GOLD["topology_spread_cases"], ids=lambda c: c["name"])
def test_golden_rows(case):
    sc = golden_scenario(case)
    want = check(sc, case["name"])
    assert (want[0][0].node_count, want[0][0].pods_scheduled) == (case["expect_nodes"], case["expect_pods"])


def scenario_of(w):
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing,
                    lanes=w.lanes)


@pytest.mark.parametrize("seed", range(120))
def test_fuzz_same_as_template_mode(seed):
    """the feature-mix fuzz of the packer (taints, selectors, ports, hostname and zone anti-affinity, preloaded pods,
    every limiter sign): K_est must give the oracle's answer too"""
    w = workloads.fuzz(4000 + seed, max_groups=2, max_pegs=8)
    for pg in w.pegs:   # per-pod walks: keep the groups small
        del pg.pods[12:]
    check(scenario_of(w), w.name, lds=(0,))


@pytest.mark.parametrize("seed", range(400))
def test_fuzz_domain_rules(seed):
    w = workloads.fuzz_estimate_domains(seed)
    sc = scenario_of(w)
    if cluster_estimate_emu(sc)[0] == 1:
        # the only legitimate delegation here: hostname anti-affinity next to a node without a hostname label
        assert any("kubernetes.io/hostname" not in info.node.labels for info in w.existing)
        assert any(t.topology_key == "kubernetes.io/hostname" for pg in w.pegs for t in pg.pods[0].anti_affinity)
        pytest.skip("delegated: hostname anti-affinity with an unnamed node")
    check(sc, w.name)


@pytest.mark.parametrize("seed", range(150))
def test_fuzz_node_taints_policy_honor(seed):
    """K_est with tainted nodes (existing and, sometimes, the template) and nodeTaintsPolicy: Honor on some constraints."""
    import dataclasses
    import random
    from kubernetes_autoscaler_amd.objects import Taint, Toleration
    w = workloads.fuzz_estimate_domains(5000 + seed)
    rng = random.Random(400 + seed)
    for info in w.existing:
        if rng.random() < 0.3:
            info.node.taints.append(Taint("dedicated", "x", rng.choice(["NoSchedule", "NoExecute", "PreferNoSchedule"])))
    if rng.random() < 0.25:
        w.groups[0].template.node.taints.append(Taint("dedicated", "x", "NoSchedule"))
    for pg in w.pegs:
        pod = pg.pods[0]
        pod.tolerations = rng.choice([[], [Toleration("dedicated", "Equal", "x", "")], [Toleration("", "Exists", "", "")]])
        pod.spread_constraints = [dataclasses.replace(c, node_taints_policy=("Honor" if rng.random() < 0.6 else "Ignore"),
                                                      node_affinity_policy=("Ignore" if rng.random() < 0.4 else "Honor")) for c in pod.spread_constraints]
    sc = scenario_of(w)
    if cluster_estimate_emu(sc)[0] == 1:
        pytest.skip("delegated (hostname anti-affinity next to an unnamed node)")
    check(sc, w.name)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_domain_rules_with_extended_resources(seed):
    """more than two resource lanes: the `<.., 8>` instantiation of estimate_kernel (4 lanes on even seeds, 8 on odd ones); the template,
    the existing nodes and most PEGs get amounts on the extra lanes, sometimes the binding ones"""
    import copy
    from kubernetes_autoscaler_amd.workloads import SplitMix64
    lanes = ("cpu", "memory", "ephemeral-storage", "example.com/gpu") + (("example.com/fpga", "hugepages-2Mi", "example.com/nic", "example.com/license") if seed % 2 else ())
    w = copy.deepcopy(workloads.fuzz_estimate_domains(2000 + seed))
    rng = SplitMix64(0xE57A + seed)
    for info in [g.template for g in w.groups] + list(w.existing):
        for lane in lanes[2:]:
            v = rng.pick([0, 1, 2, 4, 8])
            info.node.allocatable[lane] = v
            info.node.capacity[lane] = v
    for pg in w.pegs:
        if pg.pods:
            extra = {lane: rng.pick([1, 1, 2]) for lane in lanes[2:] if rng.chance(1, 3)}
            for p in {id(p): p for p in pg.pods}.values():
                p.requests.update(extra)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=lanes)
    if cluster_estimate_emu(sc)[0] == 1:
        pytest.skip("delegated: hostname anti-affinity with an unnamed node")
    check(sc, f"{w.name} {len(lanes)} lanes")
