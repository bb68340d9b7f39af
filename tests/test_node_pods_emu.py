"""casim_results.node_pods: the pods on every node an estimate added — what estimationAnalyserFunc receives as
newNodesWithPods (binpacking_estimator.go:157-159, :58-61) — from both packers under the wave emulator vs the oracle."""
import numpy as np
import pytest

from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.tables import TableSet
from harness import GroupSpec, Scenario, encode, run_emu_tables
from oracle_driver import OracleScenario


def _oracle_node_pods(sc):
    s = OracleScenario(lanes=sc.lanes)
    for info in sc.existing:
        s.add_existing(info)
    out = []
    for g in sc.groups:
        ids = list(range(len(sc.pegs))) if g.pegs is None else list(g.pegs)
        est = s.estimate(s.node(g.template), [sc.pegs[i] for i in ids], max_nodes=g.max_nodes, last_index=g.last_index, node_pods_cap=4096)
        out.append(est)
    s.close()
    return out


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("seed", range(60))
def test_pods_per_added_node_match_the_oracle(seed, generic):
    w = workloads.fuzz(1000 + seed)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing)
    enc = encode(sc)
    ts = TableSet.from_encoder(enc)
    res, _ = run_emu_tables(ts, node_pods_capacity=8192, generic=generic)
    for i, est in enumerate(_oracle_node_pods(sc)):
        if int(res.status[i]) != 0:
            continue
        a, b = int(res.node_pods_offsets[i]), int(res.node_pods_offsets[i + 1])
        assert b - a == est.nodes_added == int(res.nodes_added[i])
        assert list(res.node_pods[a:b]) == list(est.node_pods), (seed, i)
        assert int((res.node_pods[a:b] > 0).sum()) == est.node_count
        assert res.nodes_with_pods(i, w.groups[i].template.node.name) == [f"{w.groups[i].template.node.name}-e-{j}" for j, n in enumerate(est.node_pods) if n > 0]
    enc.close()


def test_capacity_cuts_the_lists_off_without_failing():
    w = workloads.config_c0()
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups])
    enc = encode(sc)
    ts = TableSet.from_encoder(enc)
    full, _ = run_emu_tables(ts, node_pods_capacity=64)
    cut, _ = run_emu_tables(ts, node_pods_capacity=3)
    assert int(full.node_pods_offsets[1]) == int(full.nodes_added[0]) > 3
    assert list(cut.node_pods) == list(full.node_pods[:3]) and int(cut.node_pods_offsets[1]) == 3
    enc.close()
