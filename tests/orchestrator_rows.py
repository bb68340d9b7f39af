"""Builds the reference's orchestrator test rows (tests/golden/reference_vectors.json: orchestrator_scale_up, transcribed from
core/scaleup/orchestrator/orchestrator_test.go) as objects of the host mirror and checks a ScaleUpDecision against a row's expectation.
Shared by the oracle test (CPU), the emulator test (CPU) and the MI355X test."""
import json
import os
from typing import Dict, List

from kubernetes_autoscaler_amd.estimator import NodeGroup
from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, PodEquivalenceGroup, Taint, Toleration
from kubernetes_autoscaler_amd.scaleup import ScaleUpDecision, decide_scale_up, estimates_from_results
from harness import GroupSpec, Scenario

GPU = "nvidia.com/gpu"
LANES = ("cpu", "memory", GPU)
ROWS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))["orchestrator_scale_up"]["cases"]


def _node(n) -> Node:
    """buildTestNode (orchestrator_test.go:1276-1283) over BuildTestNode / AddGpusToNode (utils/test/test_utils.go:367-401,457-468)"""
    cap = {"cpu": n["cpu"], "memory": n["mem"], "pods": 100}
    node = Node(name=n["name"], labels={}, allocatable=dict(cap), capacity=dict(cap))
    if n.get("gpu", 0) > 0:
        node.taints.append(Taint(GPU, "present", "NoSchedule"))
        node.allocatable[GPU] = node.capacity[GPU] = n["gpu"]
        node.labels["cloud.google.com/gke-accelerator"] = "nvidia-tesla-k80"
    if n.get("group"):
        node.labels["ng"] = n["group"]
    return node


def _pod(p) -> Pod:
    """buildTestPod (:1285-1300) over BuildTestPod / RequestGpuForPod / TolerateGpuForPod (test_utils.go:38-72,332-347)"""
    pod = Pod(name=p["name"], requests={"cpu": p["cpu"], "memory": p["mem"]})
    if p.get("gpu", 0) > 0:
        pod.requests[GPU] = p["gpu"]
    if p.get("tolerates_gpu"):
        pod.tolerations.append(Toleration(key=GPU, operator="Exists"))
    return pod


class Row:
    def __init__(self, row):
        self.row = row
        nodes = {n["name"]: _node(n) for n in row["nodes"]}
        running: Dict[str, List[Pod]] = {name: [] for name in nodes}
        for p in row["pods"]:
            running[p["node"]].append(_pod(p))
        self.existing = [NodeInfo(nodes[n["name"]], running[n["name"]]) for n in row["nodes"]]
        self.pegs = [PodEquivalenceGroup(pods=[_pod(p)]) for p in row["extra_pods"]]    # no controller: one group per pod (equivalence/groups.go:69-73)
        cfg = row.get("groups", {})
        self.node_groups: List[NodeGroup] = []
        self.templates: Dict[str, NodeInfo] = {}
        for n in row["nodes"]:                                   # groups found through their nodes: template = the node without its pods
            g = n.get("group")
            if g and g not in self.templates:
                size = sum(1 for m in row["nodes"] if m.get("group") == g)
                self.node_groups.append(NodeGroup(g, cfg.get(g, {}).get("max", 10), size))
                self.templates[g] = NodeInfo(nodes[n["name"]], [])
        for g, t in row.get("templates", {}).items():           # NodeTemplateConfigs: groups without nodes
            self.node_groups.append(NodeGroup(g, cfg.get(g, {}).get("max", 10), 0))
            cap = {"cpu": t["cpu"], "memory": t["mem"], "pods": 100}
            self.templates[g] = NodeInfo(Node(name=f"template-{g}", labels={}, allocatable=dict(cap), capacity=dict(cap)), [])

    def scenario(self) -> Scenario:
        """the estimator of these tests has NO thresholds (newEstimatorBuilder, orchestrator_test.go:2452-2462): max_nodes 0 = unlimited"""
        return Scenario(pegs=self.pegs, groups=[GroupSpec(self.templates[ng.id()], 0, 0, None) for ng in self.node_groups], existing=self.existing,
                        lanes=LANES, device_csr=True)

    def decide(self, per_group) -> ScaleUpDecision:
        """per_group[i] = (order, placed, node_count) of node group i"""
        row, exp = self.row, self.row["expect"]
        want = exp.get("option_chosen")

        def choose(options):
            if want is None:
                return options[0]
            hits = [o for o in options if [o.node_group.id(), o.node_count] == want]
            assert hits, f"{row['name']}: the option the reference's test picks, {want}, is not among {[(o.node_group.id(), o.node_count) for o in options]}"
            return hits[0]
        stop = (lambda options: len(options) >= 1) if row.get("stop_binpacking_after_first_option") else None
        return decide_scale_up(self.pegs, estimates_from_results(self.pegs, self.node_groups, per_group), len(self.existing),
                               all_or_nothing=row.get("all_or_nothing", False), zero_or_max_node_scaling=row.get("zero_or_max_node_scaling", False),
                               max_nodes_total=row.get("max_nodes_total", 0), stop_binpacking=stop, choose=choose)

    def check(self, d: ScaleUpDecision, what=""):
        row, exp = self.row, self.row["expect"]
        tag = f"{what} {row['name']} ({row['ref']})"
        assert d.scale_up == exp["scale_up"], f"{tag}: scale-up {d.scale_up}, reason {d.reason!r}"
        names = lambda pods: sorted(p.name for p in pods)
        if "expansion_options" in exp:
            assert sorted([o.node_group.id(), o.node_count] for o in d.options) == sorted(exp["expansion_options"]), f"{tag}: expander input"
        if "n_options" in exp:
            assert len(d.options) == exp["n_options"], f"{tag}: options {[(o.node_group.id(), o.node_count) for o in d.options]}"
        if exp.get("final") is not None:
            assert [d.final_group, d.final_size_change] == exp["final"], f"{tag}: final {d.final_group} +{d.final_size_change}"
        if "reason" in exp:
            assert d.reason == exp["reason"], f"{tag}: reason {d.reason!r}"
        for key, got in (("triggered", d.pods_triggered_scale_up), ("remaining", d.pods_remain_unschedulable), ("awaiting", d.pods_await_evaluation)):
            if key in exp:
                assert names(got) == sorted(exp[key]), f"{tag}: {key} {names(got)}"


def per_group_of_batch(res):
    return [(res.group(i)[0], res.group(i)[1], int(res.node_count[i])) for i in range(len(res.node_count))]


def per_group_of_oracle(out):
    return [([ids[k] for k in est.order], list(est.placed), est.node_count) for est, ids in out]
