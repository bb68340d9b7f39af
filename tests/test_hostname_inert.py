"""Hostname anti-affinity when NO node carries kubernetes.io/hostname — the shape of the reference's BuildTestNode clusters (utils/test/test_utils.go:367-400
sets no labels), BenchmarkRunFiltersUntilPassingNode among them (plugin_runner_test.go:524-583).  InterPodAffinity counts and blocks through the topology
pairs of the nodes' labels (V/kubernetes/pkg/scheduler/framework/plugins/interpodaffinity/filtering.go), so such terms are inert as a whole; the per-node
encoder used to delegate every PEG with a hostname term as soon as one node lacked the label.  Now: no node has it -> no node bits, nothing delegated,
oracle-exact; some have it, some do not -> delegated as before; an update session that brings the first labelled node asks for a full finalize."""
import copy
import ctypes as C
import json
import os

import pytest

from harness import (EmuContext, RemovalCase, SchedCase, assert_removal_matches, assert_sched_matches, removal_device, removal_oracle, sched_emu,
                     sched_encode, sched_oracle)
from kubernetes_autoscaler_amd import _abi, workloads as W
from kubernetes_autoscaler_amd.objects import (LABEL_HOSTNAME, NodeInfo, build_test_node, build_test_pod, with_labels,
                                               with_pod_hostname_anti_affinity)

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


def benchmark_cluster(b=None):
    b = b or GOLD["benchmark_run_filters_until_passing_node"]
    pod = build_test_pod("p", b["pod_cpu"], b["pod_mem"], with_pod_hostname_anti_affinity(b["label"]), with_labels(b["label"]))
    nodes = []
    for i in range(b["nodes"]):
        info = NodeInfo(build_test_node(f"n-{i}", b["node_cpu"], b["node_mem"]))
        for j in range(b["pods_per_node"]):
            info.pods.append(build_test_pod(f"p-{i}-{j}", b["running_pod_cpu"], b["running_pod_mem"], with_labels(b["label"])))
        nodes.append(info)
    nodes.append(NodeInfo(build_test_node(f"n-{b['nodes']}", b["last_node_cpu"], b["last_node_mem"])))
    return SchedCase(nodes=nodes, pods=[pod], last_index=b["last_index"]), b


def test_benchmark_run_filters_until_passing_node():
    """the oracle finds the benchmark's one passing node; K_sched under the emulator runs the call itself (status 0: not delegated) and agrees"""
    case, b = benchmark_cluster()
    assert all(LABEL_HOSTNAME not in n.node.labels for n in case.nodes)
    want = sched_oracle(case)
    assert list(want[0]) == [b["expect_node_index"]] and want[2] == 1
    got = sched_emu(case)
    assert_sched_matches(got, want, "BenchmarkRunFiltersUntilPassingNode")
    enc, _ = sched_encode(case)
    assert enc.pegs.w_excl == 0 and not (enc.pegs.flags[0] & _abi.PEG_UNSUPPORTED)   # no node bits: the term is inert
    enc.close()


def _strip(nodes):
    out = [NodeInfo(copy.deepcopy(n.node), list(n.pods)) for n in nodes]
    for n in out:
        n.node.labels.pop(LABEL_HOSTNAME, None)
    return out


def _hostname_terms(pods):
    return sum(1 for p in pods for t in p.anti_affinity if t.topology_key == LABEL_HOSTNAME)


@pytest.mark.parametrize("base", range(0, 160, 40))
def test_fuzz_clusters_without_the_hostname_label(base):
    """fuzz_pending / fuzz_removals with the label taken off EVERY node: nothing is delegated, every field equals the oracle's"""
    terms = 0
    for seed in range(base, base + 40):
        w = W.fuzz_pending(seed)
        nodes = _strip(w.nodes)
        terms += _hostname_terms(w.pods) + _hostname_terms([p for n in nodes for p in n.pods])
        case = SchedCase(nodes=nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)
        assert_sched_matches(sched_emu(case), sched_oracle(case), w.name)
        r = W.fuzz_removals(seed)
        nodes = _strip(r.nodes)
        terms += _hostname_terms([p for n in nodes for p in n.pods])
        rc = RemovalCase(nodes=nodes, candidates=r.candidates, destination=r.destination, hints=r.hints, persist=r.persist,
                         max_removable=r.max_removable, last_index=r.last_index)
        assert_removal_matches(removal_device(rc, EmuContext(0)), removal_oracle(rc), r.name)
    assert terms >= 20, terms      # (the corpus does carry such terms)


def test_some_nodes_with_the_label_some_without_is_still_delegated():
    case, _ = benchmark_cluster(dict(GOLD["benchmark_run_filters_until_passing_node"], nodes=6))
    case.nodes[2].node.labels[LABEL_HOSTNAME] = case.nodes[2].node.name
    enc, _ = sched_encode(case)
    assert enc.pegs.flags[0] & _abi.PEG_UNSUPPORTED
    enc.close()
    for n in case.nodes:       # every node named: the term counts (every small node runs a pod it matches; the big one is free)
        n.node.labels[LABEL_HOSTNAME] = n.node.name
    enc, _ = sched_encode(case)
    assert not (enc.pegs.flags[0] & _abi.PEG_UNSUPPORTED) and enc.pegs.w_excl >= 1
    enc.close()
    assert_sched_matches(sched_emu(case), sched_oracle(case), "all named")


def test_the_first_labelled_node_of_an_update_session_asks_for_a_full_finalize():
    from kubernetes_autoscaler_amd._ffi import lib
    case, _ = benchmark_cluster(dict(GOLD["benchmark_run_filters_until_passing_node"], nodes=6))
    enc, _ = sched_encode(case)
    enc.begin_update()
    relabelled = NodeInfo(copy.deepcopy(case.nodes[1].node), list(case.nodes[1].pods))
    relabelled.node.labels[LABEL_HOSTNAME] = relabelled.node.name
    enc.reset_group(1, relabelled)
    n = C.c_int32(0)
    assert lib.casim_enc_refinalize(enc._h, None, 0, C.byref(n)) == _abi.ENC_NEEDS_FULL
    enc.close()
    # without hostname terms anywhere the same update stays incremental
    plain = SchedCase(nodes=case.nodes, pods=[build_test_pod("q", 100, 1000)], last_index=0)
    enc, _ = sched_encode(plain)
    enc.begin_update()
    enc.reset_group(1, relabelled)
    ok, changed = enc.refinalize()
    assert ok and list(changed) == [1]
    enc.close()
