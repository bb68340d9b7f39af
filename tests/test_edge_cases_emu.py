"""Edge cases of the boundary (empty / ragged inputs, maximum widths, limiter signs, large counts) through
the encoder + emulated kernels vs the oracle.  The same cases run on the GPU in test_gpu_parity.py."""
import numpy as np
import pytest

from harness import GroupSpec, Scenario, assert_matches_oracle, encode, run_emu, run_emu_feasibility, run_oracle
from kubernetes_autoscaler_amd.objects import (GiB, MiB, Node, NodeInfo, Pod, PodAffinityTerm, PodEquivalenceGroup, Taint, Toleration,
                                               build_test_pod, make_node, make_pod_equivalence_group)


def tmpl(cpu=4000, mem=8 * GiB, pods=110, name="t", labels=None, taints=None):
    cap = {"cpu": cpu, "memory": mem, "pods": pods}
    return NodeInfo(Node(name=name, labels=dict(labels or {}), taints=list(taints or []), allocatable=dict(cap), capacity=dict(cap)))


def peg(cpu, mem, n, **kw):
    return PodEquivalenceGroup(pods=[Pod(name="p", requests={"cpu": cpu, "memory": mem}, **kw)] * n)


def edge_scenarios():
    yield "no pegs", Scenario(pegs=[], groups=[GroupSpec(tmpl(), 5)])
    yield "group without schedulable pegs", Scenario(pegs=[peg(100, MiB, 3)], groups=[GroupSpec(tmpl(), 5, 0, []), GroupSpec(tmpl(), 5)])
    yield "empty peg (Exemplar() == nil) between real ones", Scenario(
        pegs=[peg(500, GiB, 4), PodEquivalenceGroup(pods=[]), peg(100, MiB, 2)], groups=[GroupSpec(tmpl(), 0)])
    yield "limiter forbids (-1)", Scenario(pegs=[peg(100, MiB, 10)], groups=[GroupSpec(tmpl(), -1)])
    yield "limiter of one node", Scenario(pegs=[peg(3000, GiB, 5), peg(100, MiB, 50)], groups=[GroupSpec(tmpl(), 1)])
    yield "zero requests: only pod slots bind", Scenario(pegs=[peg(0, 0, 250)], groups=[GroupSpec(tmpl(pods=100), 0)])
    yield "pod bigger than the node", Scenario(pegs=[peg(9000, MiB, 3), peg(100, MiB, 3)], groups=[GroupSpec(tmpl(), 4)])
    yield "template already over-committed by preloaded pods", Scenario(
        pegs=[peg(100, MiB, 3)],
        groups=[GroupSpec(NodeInfo(tmpl().node, [Pod(name="ds", requests={"cpu": 5000, "memory": MiB})]), 4)])
    yield "no pod slots left on the template", Scenario(
        pegs=[peg(100, MiB, 3)], groups=[GroupSpec(NodeInfo(tmpl(pods=1).node, [Pod(name="ds", requests={"cpu": 1, "memory": 1})]), 4)])
    yield "one huge peg, unlimited nodes (HBM/LDS generic store)", Scenario(pegs=[peg(1000, GiB, 5000)], groups=[GroupSpec(tmpl(), 0)])
    yield "1500 nodes (beyond the register packer)", Scenario(pegs=[peg(1000, GiB, 6000), peg(250, 64 * MiB, 999)], groups=[GroupSpec(tmpl(), 1500)])
    yield "2^20 tiny pods", Scenario(pegs=[peg(1, 1, 1 << 20)], groups=[GroupSpec(tmpl(cpu=64000, mem=64 * GiB, pods=60000), 30)])
    yield "values above 2^31 with gcd 1 (int64 store)", Scenario(
        pegs=[peg(100, 3 * GiB + 1, 7), peg(50, 5 * GiB + 3, 5)], groups=[GroupSpec(tmpl(mem=64 * GiB + 7), 0)])
    # the register packer's loop behind a dry limiter: 200 PEGs over four record chunks, three nodes, PEGs that still fit
    # the leftovers scattered among PEGs that fit nowhere (nothing may be recorded for those)
    yield "limiter dry after three nodes, 200 PEGs", Scenario(
        pegs=[peg(100 + (i * 37) % 1900, (1 + (i * 11) % 24) * 128 * MiB, 1 + i % 4) for i in range(200)], groups=[GroupSpec(tmpl(), 3)])
    yield "limiter dry from the first PEG on (one node), self-excluding PEGs in between", Scenario(
        pegs=[peg(150 + 10 * (i % 7), 64 * MiB, 2, labels={"app": f"a{i}"},
                  **({"anti_affinity": [PodAffinityTerm("kubernetes.io/hostname", match_labels={"app": f"a{i}"})]} if i % 5 == 0 else {}))
              for i in range(150)],
        groups=[GroupSpec(tmpl(), 1)])
    # the PEG record carries the pods that fit an empty node in 22 bits: one slot more and the batch takes the generic packer
    yield "2^22 - 1 pod slots on the template (largest record field)", Scenario(
        pegs=[peg(1, 1, 3000), peg(2, 1, 10)], groups=[GroupSpec(tmpl(cpu=64000, mem=64 * GiB, pods=(1 << 22) - 1), 2)])
    yield "2^22 pod slots on the template (generic packer)", Scenario(
        pegs=[peg(1, 1, 3000), peg(2, 1, 10)], groups=[GroupSpec(tmpl(cpu=64000, mem=64 * GiB, pods=1 << 22), 2)])
    yield "lastIndex far beyond the list", Scenario(pegs=[peg(1000, GiB, 9), peg(300, MiB, 20)], groups=[GroupSpec(tmpl(), 0, last_index=12345)])
    # 70 distinct taints / 70 label requirements: two mask words of each kind
    taints = [Taint(f"k{i}", "v", "NoSchedule") for i in range(70)]
    labels = {f"l{i}": "x" for i in range(70)}
    tol_all = [Toleration(key=f"k{i}", operator="Exists") for i in range(70)]
    tol_most = tol_all[:69]
    yield "two mask words", Scenario(
        pegs=[peg(100, MiB, 5, tolerations=tol_all, node_selector={f"l{i}": "x" for i in range(0, 70, 3)}),
              peg(100, MiB, 5, tolerations=tol_most), peg(200, MiB, 5, tolerations=tol_all, node_selector={"l69": "y"})],
        groups=[GroupSpec(tmpl(labels=labels, taints=taints), 3)], device_csr=True)
    # eight resource lanes
    lanes = ("cpu", "memory", "ephemeral-storage", "r3", "r4", "r5", "r6", "r7")
    alloc = {"cpu": 8000, "memory": 16 * GiB, "ephemeral-storage": 100 * GiB, "r3": 4, "r4": 8, "r5": 0, "r6": 1 << 40, "r7": 3, "pods": 50}
    node = Node(name="r8", labels={}, allocatable=dict(alloc), capacity=dict(alloc))
    yield "eight resource lanes", Scenario(
        pegs=[PodEquivalenceGroup(pods=[Pod(name="a", requests={"cpu": 500, "memory": GiB, "r3": 1, "r6": 1 << 38})] * 9),
              PodEquivalenceGroup(pods=[Pod(name="b", requests={"cpu": 100, "r5": 1})] * 4),
              PodEquivalenceGroup(pods=[Pod(name="c", requests={"cpu": 100, "r7": 2, "ephemeral-storage": 30 * GiB})] * 7)],
        groups=[GroupSpec(NodeInfo(node), 0)], lanes=lanes)


CASES = list(edge_scenarios())


@pytest.mark.parametrize("name,sc", CASES, ids=[c[0] for c in CASES])
def test_edge_case(name, sc):
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), name)


def test_no_groups():
    sc = Scenario(pegs=[peg(100, MiB, 3)], groups=[])
    res, best = run_emu(encode(sc), kinds=[0])
    assert len(res.node_count) == 0 and best[0] == -1 and best[1] == 0


def test_unsupported_peg_delegates_its_groups_only():
    ok = peg(100, MiB, 3)
    bad = PodEquivalenceGroup(pods=[Pod(name="x", requests={"cpu": 100, "memory": MiB}, topology_spread=True)] * 2)
    sc = Scenario(pegs=[ok, bad], groups=[GroupSpec(tmpl(), 5, 0, [0]), GroupSpec(tmpl(), 5, 0, [0, 1])])
    res, _ = run_emu(encode(sc))
    assert list(res.status) == [0, 1]                      # CASIM_NG_UNSUPPORTED only where the PEG is used
    assert int(res.node_count[0]) == 1 and int(res.pods_scheduled[0]) == 3
    assert int(res.node_count[1]) == 0 and int(res.pods_scheduled[1]) == 0


def test_feasibility_bits_match_check_predicates():
    from kubernetes_autoscaler_amd import workloads
    from oracle_driver import OracleScenario
    for seed in (1003, 1010, 1021, 1033):
        w = workloads.fuzz(seed)
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index) for g in w.groups], existing=w.existing, device_csr=True)
        bits = run_emu_feasibility(encode(sc))
        s = OracleScenario()
        for info in w.existing:
            s.add_existing(info)
        for gi, g in enumerate(w.groups):
            t = s.node(g.template)
            for pi, pg in enumerate(w.pegs):
                want = s.check_predicates(t, pg.exemplar())[0]
                got = bool((int(bits[gi, pi // 64]) >> (pi % 64)) & 1)
                assert got == want, (seed, gi, pi)


def test_unsupported_peg_never_drops_out_of_a_device_built_list():
    """peg_offsets == NULL (the engine derives SchedulablePodGroups): a PEG that needs a predicate outside the encoded
    subset stays on the list of every group whose encoded Filters it passes — those groups come back
    CASIM_NG_UNSUPPORTED — and is dropped only where the encoded part already rejects it."""
    from kubernetes_autoscaler_amd import workloads
    from kubernetes_autoscaler_amd.objects import TopologySpreadConstraint
    w = workloads.config_c2(n_groups=3, n_pegs=6, pods_per_peg=3, cap=10)
    pod = w.pegs[1].pods[0]   # schedulable on two of the three groups
    pod.spread_constraints = [TopologySpreadConstraint(2, "kubernetes.io/hostname", 0, dict(pod.labels))]
    pod.topology_spread = True
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)
    res, _ = run_emu(encode(sc))
    from oracle_driver import OracleScenario
    s = OracleScenario()
    passes = [s.check_predicates(s.node(g.template), pod)[0] for g in w.groups]
    s.close()
    assert any(passes) and not all(passes)
    for i, ok in enumerate(passes):
        listed = 1 in [int(x) for x in res.group(i)[0]]
        assert listed == ok and (int(res.status[i]) == 1) == ok
