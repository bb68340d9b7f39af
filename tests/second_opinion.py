"""An INDEPENDENT second derivation of the string-level Filter verdicts.

The product's host encoder (csrc/casim_encoder.cpp) and the oracle (oracle/casim_oracle.c) were written by the same hand from the same Go
lines; a test that only compares the two cannot see a misreading they share.  The reference holds no known answers for TaintToleration /
NodeAffinity / NodePorts / InterPodAffinity inside an Estimate (vendored plugins ship no tests).  This module is a third opinion: a
brute-force, object-level evaluator written from the DOCUMENTED contracts of the Kubernetes API types — the doc comments quoted below —
NOT from the oracle and not from the encoder.  It shares no code with either (plain Python over the objects of
kubernetes_autoscaler_amd.objects; no bit masks, no dictionaries of taints, no table).  tests/test_second_opinion.py fuzzes
encoder -> kernels and the oracle against it.

What it answers: does pod P pass the scheduler's Filter plugins on a FRESH node made from template T (T's DaemonSet pods already on it)
— the CheckPredicates question of ScaleUpOrchestrator.SchedulablePodGroups.

Contracts restated (V = vendor/k8s.io of the reference):
  * V/api/core/v1/toleration.go:40-51 (ToleratesTaint doc comment):
      "1. Empty toleration.effect means to match all taint effects, otherwise taint effect must equal to toleration.effect.
       2. If toleration.operator is 'Exists', it means to match all taint values.
       3. Empty toleration.key means to match all taint keys.  If toleration.key is empty, toleration.operator must be 'Exists';
          this combination means to match all taint values and all taint keys."
    + types.go (Toleration.Operator): "Valid operators are Exists and Equal. Defaults to Equal."
  * TaintToleration plugin doc: a pod is rejected by a node's taints with effect NoSchedule or NoExecute that it does not tolerate
    (PreferNoSchedule only scores).
  * V/api/core/v1/types.go NodeSelectorRequirement / labels.Requirement.Matches doc (apimachinery/pkg/labels/selector.go:235-246):
      "(1) The operator is Exists and Labels has the Requirement's key.  (2) The operator is In, Labels has the Requirement's key and
       Labels' value for that key is in Requirement's value set.  (3) The operator is NotIn, Labels has the Requirement's key and
       Labels' value for that key is not in Requirement's value set.  (4) The operator is DoesNotExist or NotIn and Labels does not have
       the Requirement's key.  (5) The operator is GreaterThanOperator or LessThanOperator, and Labels has the Requirement's key and the
       corresponding value satisfies mathematical inequality."
  * PodSpec.nodeSelector doc: "Selector which must match a node's labels for the pod to be scheduled on that node" (every key = value);
    NodeSelector doc: "A node selector represents the union of the results of one or more label queries over a set of nodes; that is, it
    represents the OR of the selectors represented by the node selector terms"; NodeSelectorTerm doc: "A null or empty node selector term
    matches no objects. The requirements of them are ANDed."
  * NodeSpec.unschedulable doc + NodeUnschedulable plugin: an unschedulable node only takes pods that tolerate the taint
    node.kubernetes.io/unschedulable:NoSchedule.
  * ContainerPort.hostPort / HostPortInfo doc (V/kube-scheduler/framework/types.go:596-640): a (hostIP, protocol, hostPort) triple
    conflicts with one in use when protocol and port are equal and the IPs are equal or either is 0.0.0.0; empty hostIP means 0.0.0.0,
    empty protocol means TCP; hostPort 0 = none.
  * PodAntiAffinity.requiredDuringSchedulingIgnoredDuringExecution doc: "If the anti-affinity requirements specified by this field are not
    met at scheduling time, the pod will not be scheduled onto the node"; PodAffinityTerm doc: the term selects pods by labelSelector
    in the listed namespaces ("null or empty namespaces list and null namespaceSelector means 'this pod's namespace'"), co-located = on
    nodes "whose value of the label with key topologyKey matches that of any node on which a pod of the set of pods is running".
    The scheduler applies it in both directions: the incoming pod's terms against pods on the node, and the terms of pods already there
    against the incoming pod.
  * NodeResourcesFit doc: the pod's requests fit allocatable minus what the pods on the node request, per resource the pod asks for, and
    the node's pod count stays within allocatable "pods"."""
from typing import Dict, Iterable

from kubernetes_autoscaler_amd.objects import NodeInfo, Pod, Requirement

UNSCHEDULABLE_TAINT = ("node.kubernetes.io/unschedulable", "NoSchedule")


def tolerates(tol, key: str, value: str, effect: str) -> bool:
    if tol.effect != "" and tol.effect != effect:
        return False
    if tol.key != "" and tol.key != key:
        return False
    op = tol.operator or "Equal"
    if op == "Exists":
        return True
    if op == "Equal":
        return tol.value == value
    return False          # (Lt / Gt: behind a feature gate that is off)


def taints_allow(pod: Pod, taints) -> bool:
    for t in taints:
        if t.effect not in ("NoSchedule", "NoExecute"):
            continue
        if not any(tolerates(tol, t.key, t.value, t.effect) for tol in pod.tolerations):
            return False
    return True


def _as_int(s: str):
    """strconv.ParseInt(s, 10, 64): optional sign, decimal digits, 64-bit range"""
    body = s[1:] if s[:1] in "+-" else s
    if not body or not body.isdigit() or not body.isascii():
        return None
    v = int(s)
    return v if -(1 << 63) <= v < (1 << 63) else None


def requirement_matches(r: Requirement, labels: Dict[str, str]) -> bool:
    has = r.key in labels
    if r.operator == "In":
        return has and labels[r.key] in r.values
    if r.operator == "NotIn":
        return not has or labels[r.key] not in r.values
    if r.operator == "Exists":
        return has
    if r.operator == "DoesNotExist":
        return not has
    if r.operator in ("Gt", "Lt"):
        if not has or len(r.values) != 1:
            return False
        a, b = _as_int(labels[r.key]), _as_int(r.values[0])
        if a is None or b is None:
            return False
        return a > b if r.operator == "Gt" else a < b
    return False


def node_affinity_allows(pod: Pod, node) -> bool:
    for k, v in pod.node_selector.items():
        if node.labels.get(k) != v:
            return False
    if pod.node_affinity:          # ONE required term (ANDed requirements)
        if not all(requirement_matches(r, node.labels) for r in pod.node_affinity):
            return False
    if pod.node_affinity_terms is not None:
        ok = False
        for term in pod.node_affinity_terms:
            if not term.match_expressions and not term.match_fields:
                continue                     # "a null or empty node selector term matches no objects"
            good = all(requirement_matches(r, node.labels) for r in term.match_expressions)
            for f in term.match_fields:      # metadata.name, In / NotIn with one value
                if f.key != "metadata.name" or len(f.values) != 1 or f.operator not in ("In", "NotIn"):
                    good = False
                else:
                    good = good and ((node.name == f.values[0]) == (f.operator == "In"))
            ok = ok or good
        if not ok:
            return False
    return True


def _triple(p):
    return (p.host_ip or "0.0.0.0", p.protocol or "TCP", int(p.host_port))


def ports_allow(pod: Pod, pods_on_node: Iterable[Pod]) -> bool:
    used = [_triple(p) for q in pods_on_node for p in q.host_ports if p.host_port > 0]
    for p in pod.host_ports:
        if p.host_port <= 0:
            continue
        ip, proto, port = _triple(p)
        for uip, uproto, uport in used:
            if uproto == proto and uport == port and (uip == ip or uip == "0.0.0.0" or ip == "0.0.0.0"):
                return False
    return True


def _term_selects(term, owner: Pod, other: Pod) -> bool:
    """does `term` (owned by `owner`) select the pod `other`?  namespaces: the listed ones, or the owner's own when none are listed and
    there is no namespace selector (selectors over namespaces are left to the dedicated tests: not generated by the fuzz families here)"""
    if term.namespace_selector is not None:
        raise NotImplementedError("namespaceSelector: covered by tests/test_namespace_selector_emu.py")
    spaces = list(term.namespaces) or [owner.namespace]
    if other.namespace not in spaces:
        return False
    return all(requirement_matches(r, other.labels) for r in term.requirements())


def anti_affinity_allows(pod: Pod, node, pods_on_node: Iterable[Pod]) -> bool:
    """fresh node of a template: the only pods in its topology domains that the estimate's snapshot holds are the ones on the node itself
    (hostname), and — for a zone-wide key — none elsewhere in these scenarios (existing nodes are empty)"""
    for other in pods_on_node:
        for term in pod.anti_affinity:
            if term.topology_key in node.labels and _term_selects(term, pod, other):
                return False
        for term in other.anti_affinity:
            if term.topology_key in node.labels and _term_selects(term, other, pod):
                return False
    return True


def resources_allow(pod: Pod, info: NodeInfo) -> bool:
    node = info.node
    if len(info.pods) + 1 > node.allowed_pods():
        return False
    for name, want in pod.requests.items():
        if want <= 0:
            continue
        used = sum(int(q.requests.get(name, 0)) for q in info.pods)
        if want > int(node.allocatable.get(name, 0)) - used:
            return False
    return True


def fits_fresh_template(pod: Pod, template: NodeInfo) -> bool:
    node = template.node
    if node.unschedulable and not any(tolerates(t, UNSCHEDULABLE_TAINT[0], "", UNSCHEDULABLE_TAINT[1]) for t in pod.tolerations):
        return False
    return (taints_allow(pod, node.taints) and node_affinity_allows(pod, node) and ports_allow(pod, template.pods) and
            anti_affinity_allows(pod, node, template.pods) and resources_allow(pod, template))


# ---- per-node mode: a pod against the nodes of a SNAPSHOT (RunFiltersOnNode; filter-out-schedulable, the removal loop) ------------------------------
# PodAffinityTerm doc (V/api/core/v1/types.go): "This pod should be co-located (affinity) or not co-located (anti-affinity) with the pods matching the
# labelSelector in the specified namespaces, where co-located is defined as running on a node whose value of the label with key topologyKey matches
# that of any node on which any of the selected pods is running.  Empty topologyKey is not allowed."  A node WITHOUT the key has no value to match:
# it is co-located with nothing under that key — BuildTestNode clusters carry no kubernetes.io/hostname at all.
def co_located(topology_key: str, a, b) -> bool:
    return topology_key in a.labels and topology_key in b.labels and a.labels[topology_key] == b.labels[topology_key]


def fits_existing_node(pod: Pod, index: int, cluster) -> bool:
    """does `pod` pass the Filter plugins on node `index` of the snapshot `cluster` (a list of NodeInfo, every node with the pods running on it)?
    Anti-affinity looks at every pod of the snapshot: the incoming pod's terms against pods co-located with the node, the running pods' terms
    against the incoming pod (symmetry: "the pod will not be scheduled onto the node" holds for existing pods' required anti-affinity too)."""
    info = cluster[index]
    node = info.node
    if node.unschedulable and not any(tolerates(t, UNSCHEDULABLE_TAINT[0], "", UNSCHEDULABLE_TAINT[1]) for t in pod.tolerations):
        return False
    if not (taints_allow(pod, node.taints) and node_affinity_allows(pod, node) and ports_allow(pod, info.pods) and resources_allow(pod, info)):
        return False
    for other_info in cluster:
        for other in other_info.pods:
            for term in pod.anti_affinity:
                if co_located(term.topology_key, node, other_info.node) and _term_selects(term, pod, other):
                    return False
            for term in other.anti_affinity:
                if co_located(term.topology_key, node, other_info.node) and _term_selects(term, other, pod):
                    return False
    return True
