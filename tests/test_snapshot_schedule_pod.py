"""The SchedulePod / SchedulePodOnAnyNodeMatching rows of the reference's snapshot operation table
(simulator/clustersnapshot/predicate/predicate_snapshot_test.go:400-511, transcribed to tests/golden/reference_vectors.json
`snapshot_schedule_pod`) — row a9 of SURVEY section 8: success and failure of scheduling ONE pod onto a named node /
onto any acceptable node, the state unchanged on failure.  Oracle: RunFiltersOnNode (orc_run_filters_on_snapshot_node) for SchedulePod,
the cyclic search (orc_try_schedule_pods with the IsNodeAcceptable mask) for SchedulePodOnAnyNodeMatching.  Product: K_sched
(casim_try_schedule_pods) — SchedulePod(pod, node) is the search restricted to that one node — under the wave emulator here, on the
MI355X in tests/test_gpu_round5.py."""
import json
import os

import pytest

from kubernetes_autoscaler_amd.objects import NodeInfo, build_test_node, build_test_pod
from harness import SchedCase, assert_sched_matches, sched_emu, sched_oracle
from oracle_driver import OracleScenario

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))["snapshot_schedule_pod"]
CASES = G["cases"]


def build(case):
    nodes = [NodeInfo(build_test_node(n, *G["fixtures"]["nodes"][n]), []) for n in case["nodes"]]
    pod = build_test_pod(case["pod"], *G["fixtures"]["pods"][case["pod"]])
    if case["op"] == "SchedulePod":
        acceptable = [1 if n == case["on"] else 0 for n in case["nodes"]]
    else:
        acceptable = [1] * len(nodes) if case["acceptable"] == "all" else [1 if n in case["acceptable"] else 0 for n in case["nodes"]]
    want = case["nodes"].index(case["want_node"]) if case["want_node"] is not None else -1
    return nodes, pod, acceptable, want


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_the_snapshot_row(case):
    nodes, pod, acceptable, want = build(case)
    if case["op"] == "SchedulePod":      # RunFiltersOnNode on the named node: passes or FailingPredicateError (plugin + reason)
        s = OracleScenario()
        for info in nodes:
            s.add_existing(info)
        ok, plugin, _ = s.run_filters_on_node(case["nodes"].index(case["on"]), pod)
        s.close()
        assert ok == (want >= 0)
        if not ok:
            assert plugin == "NodeResourcesFit"      # "the pod is too big for the node"
    node_out, _, n = sched_oracle(SchedCase(nodes=nodes, pods=[pod], acceptable=acceptable))
    assert (int(node_out[0]), n) == (want, 1 if want >= 0 else 0)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_product_kernels_reproduce_the_snapshot_row_under_the_emulator(case):
    nodes, pod, acceptable, want = build(case)
    sc = SchedCase(nodes=nodes, pods=[pod], acceptable=acceptable)
    got = sched_emu(sc)
    assert_sched_matches(got, sched_oracle(sc), case["name"])
    assert int(got[1][0]) == want
    # "the state shouldn't change on error": a second pod that fits finds the cluster as it was
    if want < 0 and case["pod"] == "largePod":
        small = build_test_pod("specialPod", *G["fixtures"]["pods"]["specialPod"])
        sc2 = SchedCase(nodes=nodes, pods=[pod, small], acceptable=[1] * len(nodes))
        got2 = sched_emu(sc2)
        assert_sched_matches(got2, sched_oracle(sc2), case["name"] + " + a pod that fits")
        assert int(got2[1][0]) == -1 and int(got2[1][1]) >= 0      # (which node: the cyclic search from lastIndex + 1, as the oracle says)
