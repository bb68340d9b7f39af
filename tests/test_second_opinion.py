"""The product encoder + kernels and the oracle against an INDEPENDENT evaluator of the Filter verdicts (tests/second_opinion.py: written from
the doc comments of the Kubernetes API types, sharing no code with the encoder or the oracle) — VERDICT r4 next #9: where the reference
holds no known answers for a14 / a15 / a17 / a18 inside an Estimate, encoder and oracle must not be each other's only witness.

Checked cell by cell: the SchedulablePodGroups matrix (does PEG g fit a fresh node of template t) from
  (1) the evaluator, (2) the oracle's CheckPredicates, (3) the product's feasibility kernel under the wave emulator (casim_feasibility)."""
import numpy as np
import pytest

from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.objects import (ContainerPort, LABEL_HOSTNAME, LABEL_ZONE, Node, NodeInfo, NodeSelectorTerm, Pod, PodAffinityTerm, PodEquivalenceGroup,
                                               Requirement, Taint, Toleration)
from harness import GroupSpec, Scenario, encode, run_emu_feasibility
from oracle_driver import OracleScenario
import second_opinion as so


def _matrices(sc):
    want = np.array([[so.fits_fresh_template(pg.exemplar(), g.template) for pg in sc.pegs] for g in sc.groups], bool)
    s = OracleScenario(lanes=sc.lanes)
    for info in sc.existing:
        s.add_existing(info)
    oracle = np.array([[s.check_predicates(s.node(g.template), pg.exemplar())[0] for pg in sc.pegs] for g in sc.groups], bool)
    s.close()
    enc = encode(sc)
    bits = run_emu_feasibility(enc)
    unsupported = [bool(int(enc.pegs.flags[i]) & 0x20) for i in range(len(sc.pegs))]
    enc.close()
    dev = np.array([[bool((int(bits[gi][k >> 6]) >> (k & 63)) & 1) for k in range(len(sc.pegs))] for gi in range(len(sc.groups))], bool)
    return want, oracle, dev, unsupported


def _check(sc, what):
    want, oracle, dev, unsupported = _matrices(sc)
    for gi in range(len(sc.groups)):
        for k in range(len(sc.pegs)):
            assert bool(oracle[gi][k]) == bool(want[gi][k]), f"{what}: ORACLE says {oracle[gi][k]}, the independent evaluator {want[gi][k]} for PEG {k} ({sc.pegs[k].exemplar()}) on {sc.groups[gi].template.node}"
            if not unsupported[k]:
                assert bool(dev[gi][k]) == bool(want[gi][k]), f"{what}: ENCODER + KERNEL say {dev[gi][k]}, the independent evaluator {want[gi][k]} for PEG {k} ({sc.pegs[k].exemplar()}) on {sc.groups[gi].template.node}"
    return int(want.sum()), int(want.size)


@pytest.mark.parametrize("block", range(10))
def test_fuzz_families_against_the_independent_evaluator(block):
    """the rich fuzz family of the parity suites (taints of three effects, tolerations with empty operators / keys / effects, node selectors,
    host ports with and without IP and protocol, hostname and zone anti-affinity in both directions, DaemonSet pods on the template,
    unschedulable templates, pods without requests): 60 scenarios per block"""
    yes = cells = 0
    for seed in range(60):
        w = workloads.fuzz(60000 + 100 * block + seed, max_groups=5, max_pegs=14)
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=True)
        a, b = _check(sc, f"fuzz {60000 + 100 * block + seed}")
        yes += a; cells += b
    assert 0 < yes < cells      # (both verdicts occur)


def _rand_requirement(rng, keys, values):
    op = rng.choice(["In", "NotIn", "Exists", "DoesNotExist", "Gt", "Lt"])
    key = rng.choice(keys)
    if op in ("In", "NotIn"):
        vals = tuple(rng.choice(values, size=int(rng.integers(1, 3))))
    elif op in ("Gt", "Lt"):
        vals = (str(rng.choice(["3", "7", "-2", "+5", "x", "007", "9223372036854775808", ""])),) if rng.integers(0, 6) else ("1", "2")
    else:
        vals = ()
    return Requirement(str(key), str(op), tuple(str(v) for v in vals))


@pytest.mark.parametrize("block", range(6))
def test_label_requirements_of_every_operator(block):
    """labels.Requirement.Matches through nodeSelector, the single required term and ORed nodeSelectorTerms (matchExpressions and
    matchFields): In / NotIn / Exists / DoesNotExist / Gt / Lt with values that parse, values that do not, signs, leading zeros, overflow"""
    rng = np.random.default_rng(4200 + block)
    keys = ["pool", "tier", "gen", "rack"]
    values = ["a", "b", "5", "10", "-3", "+4", "0x10", "9223372036854775807", ""]
    for case in range(40):
        groups = []
        for gi in range(int(rng.integers(2, 6))):
            labels = {LABEL_HOSTNAME: f"n{gi}"}
            for k in keys:
                if rng.integers(0, 3):
                    labels[k] = str(rng.choice(values))
            cap = {"cpu": 4000, "memory": 8 << 30, "pods": 110}
            groups.append(GroupSpec(NodeInfo(Node(name=f"n{gi}", labels=labels, allocatable=dict(cap), capacity=dict(cap)), []), 0, 0, None))
        pegs = []
        for i in range(int(rng.integers(3, 10))):
            pod = Pod(name=f"p{i}", requests={"cpu": 100, "memory": 1 << 20})
            kind = int(rng.integers(0, 4))
            if kind == 0:
                pod.node_selector = {str(rng.choice(keys)): str(rng.choice(values[:4]))}
            elif kind == 1:
                pod.node_affinity = [_rand_requirement(rng, keys, values) for _ in range(int(rng.integers(1, 4)))]
            elif kind == 2:
                terms = []
                for _ in range(int(rng.integers(0, 4))):
                    t = NodeSelectorTerm(match_expressions=[_rand_requirement(rng, keys, values) for _ in range(int(rng.integers(0, 3)))])
                    terms.append(t)
                pod.node_affinity_terms = terms
            pegs.append(PodEquivalenceGroup(pods=[pod] * 2))
        sc = Scenario(pegs=pegs, groups=groups, existing=[], device_csr=True)
        _check(sc, f"requirements {block}/{case}")


@pytest.mark.parametrize("block", range(4))
def test_tolerations_of_every_shape(block):
    """ToleratesTaint: every combination of (key | empty) x (Exists | Equal | "" | an unknown operator) x (value matches | not) x
    (effect empty | same | other) against taints of the three effects"""
    rng = np.random.default_rng(5300 + block)
    keys, vals, effects = ["dedicated", "gpu", "spot"], ["yes", "no", ""], ["NoSchedule", "NoExecute", "PreferNoSchedule"]
    for case in range(50):
        groups = []
        for gi in range(int(rng.integers(2, 6))):
            taints = [Taint(str(rng.choice(keys)), str(rng.choice(vals)), str(rng.choice(effects))) for _ in range(int(rng.integers(0, 4)))]
            cap = {"cpu": 4000, "memory": 8 << 30, "pods": 110}
            node = Node(name=f"n{gi}", labels={LABEL_HOSTNAME: f"n{gi}"}, taints=taints, allocatable=dict(cap), capacity=dict(cap))
            node.unschedulable = bool(rng.integers(0, 8) == 0)
            groups.append(GroupSpec(NodeInfo(node, []), 0, 0, None))
        pegs = []
        for i in range(int(rng.integers(3, 10))):
            tols = []
            for _ in range(int(rng.integers(0, 4))):
                op = str(rng.choice(["Exists", "Equal", "", "Exists"]))
                key = str(rng.choice(keys + [""])) if op == "Exists" else str(rng.choice(keys))
                tols.append(Toleration(key=key, operator=op, value=str(rng.choice(vals)), effect=str(rng.choice(["", "NoSchedule", "NoExecute", "PreferNoSchedule"]))))
            if rng.integers(0, 5) == 0:
                tols.append(Toleration(key="node.kubernetes.io/unschedulable", operator="Exists", effect=str(rng.choice(["", "NoSchedule", "NoExecute"]))))
            pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": 100, "memory": 1 << 20}, tolerations=tols)] * 2))
        sc = Scenario(pegs=pegs, groups=groups, existing=[], device_csr=True)
        _check(sc, f"tolerations {block}/{case}")


@pytest.mark.parametrize("block", range(4))
def test_host_ports_and_anti_affinity_against_the_template_pods(block):
    """NodePorts triples (IP / protocol defaults, 0.0.0.0 wildcards) and hostname / zone anti-affinity in both directions against the
    DaemonSet pods of the template, namespaces listed and defaulted"""
    rng = np.random.default_rng(6400 + block)
    for case in range(50):
        groups = []
        for gi in range(int(rng.integers(2, 5))):
            labels = {LABEL_HOSTNAME: f"n{gi}"}
            if rng.integers(0, 3):
                labels[LABEL_ZONE] = f"z{int(rng.integers(0, 2))}"
            cap = {"cpu": 4000, "memory": 8 << 30, "pods": int(rng.choice([2, 3, 110]))}
            pre = []
            for d in range(int(rng.integers(0, 3))):
                ds = Pod(name=f"ds{gi}-{d}", namespace=str(rng.choice(["kube-system", "default"])), labels={"app": str(rng.choice(["ds", "web", "db"]))},
                         requests={"cpu": int(rng.choice([100, 1500])), "memory": 64 << 20})
                if rng.integers(0, 2):
                    ds.host_ports = [ContainerPort(int(rng.choice([80, 8080])), host_ip=str(rng.choice(["", "10.0.0.1", "0.0.0.0"])), protocol=str(rng.choice(["", "TCP", "UDP"])))]
                if rng.integers(0, 3) == 0:
                    ds.anti_affinity = [PodAffinityTerm(str(rng.choice([LABEL_HOSTNAME, LABEL_ZONE])), match_labels={"app": str(rng.choice(["web", "db"]))},
                                                        namespaces=tuple(["default"] if rng.integers(0, 2) else []))]
                pre.append(ds)
            groups.append(GroupSpec(NodeInfo(Node(name=f"n{gi}", labels=labels, allocatable=dict(cap), capacity=dict(cap)), pre), 0, 0, None))
        pegs = []
        for i in range(int(rng.integers(3, 9))):
            pod = Pod(name=f"p{i}", namespace=str(rng.choice(["default", "kube-system"])), labels={"app": str(rng.choice(["web", "db", "cache"]))},
                      requests={"cpu": int(rng.choice([0, 100, 3000])), "memory": int(rng.choice([0, 1 << 20]))})
            if rng.integers(0, 2):
                pod.host_ports = [ContainerPort(int(rng.choice([80, 8080, 0])), host_ip=str(rng.choice(["", "10.0.0.1", "10.0.0.2", "0.0.0.0"])), protocol=str(rng.choice(["", "TCP", "UDP"])))]
            if rng.integers(0, 2):
                pod.anti_affinity = [PodAffinityTerm(str(rng.choice([LABEL_HOSTNAME, LABEL_ZONE])), match_labels={"app": str(rng.choice(["ds", "web", "db"]))},
                                                    namespaces=tuple([[], ["kube-system"], ["default", "kube-system"]][int(rng.integers(0, 3))]))]
            pegs.append(PodEquivalenceGroup(pods=[pod] * 2))
        sc = Scenario(pegs=pegs, groups=groups, existing=[], device_csr=True)
        _check(sc, f"ports / anti-affinity {block}/{case}")


@pytest.mark.parametrize("block", range(4))
def test_snapshot_nodes_with_and_without_the_hostname_label(block):
    """Per-node mode (RunFiltersOnNode on the nodes of a snapshot): the independent evaluator against the oracle AND against the product's
    feasibility matrix (encoder + feas kernel under the emulator), cell by cell — on fuzz_pending clusters as generated (every node named), with
    kubernetes.io/hostname taken off every node (hostname anti-affinity inert: DESIGN 17e-3, nothing delegated) and taken off every other node
    (classes with hostname terms are delegated by the encoder: those are compared evaluator vs oracle only)."""
    import copy
    from harness import SchedCase, run_emu_feasibility, sched_encode
    from kubernetes_autoscaler_amd import _abi
    from kubernetes_autoscaler_amd.objects import LABEL_HOSTNAME, NodeInfo
    from oracle_driver import OracleScenario
    from kubernetes_autoscaler_amd import workloads as W
    cells = device_cells = delegated = 0
    for seed in range(40 * block, 40 * block + 40):
        for mode in ("named", "stripped", "mixed"):
            w = W.fuzz_pending(seed)
            nodes = [NodeInfo(copy.deepcopy(n.node), list(n.pods)) for n in w.nodes]
            for i, n in enumerate(nodes):
                if mode == "stripped" or (mode == "mixed" and i % 2 == 0):
                    n.node.labels.pop(LABEL_HOSTNAME, None)
            canon = {}
            for p in w.pods:
                canon.setdefault(p.spec_key(), p)
            classes = list(canon.values())
            orc = OracleScenario()
            for info in nodes:
                orc.add_existing(info)
            enc, pc = sched_encode(SchedCase(nodes=nodes, pods=classes))
            bits = run_emu_feasibility(enc)
            for ci, p in enumerate(classes):
                c = int(pc[ci])
                unsupported = bool(enc.pegs.flags[c] & _abi.PEG_UNSUPPORTED)
                if mode != "mixed":
                    assert not unsupported, (seed, mode, p.name)
                delegated += int(unsupported)
                for ni in range(len(nodes)):
                    want = so.fits_existing_node(p, ni, nodes)
                    ok, plug, _ = orc.run_filters_on_node(ni, p)
                    assert ok == want, (seed, mode, p.name, nodes[ni].node.name, "oracle", plug)
                    cells += 1
                    if not unsupported:
                        got = bool((int(bits[ni][c >> 6]) >> (c & 63)) & 1)
                        assert got == want, (seed, mode, p.name, nodes[ni].node.name, "device")
                        device_cells += 1
            enc.close()
            orc.close()
    assert cells >= 5000 and device_cells >= 4000 and delegated >= 1, (cells, device_cells, delegated)
