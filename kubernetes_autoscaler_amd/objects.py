"""Host-side object model mirroring the slice of the Kubernetes API the scale-up simulation reads.

Names follow the reference so that tests read like the reference's tests:
  Pod / Node / Taint / Toleration ...            k8s.io/api/core/v1
  NodeInfo                                       CA/simulator/framework/infos.go:57
  PodEquivalenceGroup                            CA/estimator/estimator.go:37-48
  build_test_pod / build_test_node / ...         CA/utils/test/test_utils.go:38,367
  make_node                                      CA/estimator/binpacking_estimator_test.go:44-64
(`CA/` = /root/reference/cluster-autoscaler/.)

Quantities are plain integers: cpu in millicores (Quantity.MilliValue), everything else in base
units (Quantity.Value)."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

MiB = 1024 * 1024
GiB = 1024 * MiB

LABEL_HOSTNAME = "kubernetes.io/hostname"
LABEL_ZONE = "topology.kubernetes.io/zone"

RES_CPU, RES_MEMORY, RES_EPHEMERAL = "cpu", "memory", "ephemeral-storage"


@dataclass
class Toleration:
    key: str = ""
    operator: str = ""      # "" == Equal (toleration.go:62)
    value: str = ""
    effect: str = ""


@dataclass
class Taint:
    key: str
    value: str = ""
    effect: str = "NoSchedule"


@dataclass
class Requirement:
    """labels.Requirement / v1.NodeSelectorRequirement / metav1.LabelSelectorRequirement."""
    key: str
    operator: str            # In, NotIn, Exists, DoesNotExist, Gt, Lt
    values: Sequence[str] = ()


@dataclass
class PodAffinityTerm:
    """v1.PodAffinityTerm.  namespace_selector: None = not set; [] = the empty selector (every namespace); it is resolved
    against NAMESPACE_LISTER (AffinityTerm.Matches, V/kube-scheduler/framework/types.go:390-395)."""
    topology_key: str
    match_labels: Dict[str, str] = field(default_factory=dict)
    match_expressions: List[Requirement] = field(default_factory=list)
    namespaces: Sequence[str] = ()
    namespace_selector: Optional[List[Requirement]] = None

    def requirements(self) -> List[Requirement]:
        reqs = [Requirement(k, "In", (v,)) for k, v in sorted(self.match_labels.items())]
        return reqs + list(self.match_expressions)


# The namespace lister of the scheduler framework handle (an informer in the reference, CA/simulator/framework/handle.go:
# cluster state, not an argument of the path): namespace name -> labels.  Encoder() snapshots it when it is created.
NAMESPACE_LISTER: Dict[str, Dict[str, str]] = {}


class namespaces:
    """with namespaces({"team-a": {"tier": "prod"}, "default": {}}): ...  temporarily replaces NAMESPACE_LISTER."""

    def __init__(self, table: Dict[str, Dict[str, str]]):
        self.table = {k: dict(v) for k, v in table.items()}

    def __enter__(self):
        self.saved = dict(NAMESPACE_LISTER)
        NAMESPACE_LISTER.clear()
        NAMESPACE_LISTER.update(self.table)
        return self

    def __exit__(self, *exc):
        NAMESPACE_LISTER.clear()
        NAMESPACE_LISTER.update(self.saved)
        return False


@dataclass
class TopologySpreadConstraint:
    """DoNotSchedule topologySpreadConstraint (labelSelector in its matchLabels form; `{}` selects nothing to count,
    common.go:145-148).  Evaluated on the device in per-node mode (TrySchedulePods, removal loop, estimator on the
    cluster); the template-mode batch delegates PEGs that carry one."""
    max_skew: int
    topology_key: str
    min_domains: int = 0                   # 0 = nil (treated as 1)
    match_labels: Dict[str, str] = field(default_factory=dict)
    node_taints_policy: str = "Ignore"     # "Honor": tainted nodes the pod does not tolerate are no domain members
    node_affinity_policy: str = "Honor"    # "Ignore": the pod's required node affinity does not gate domain membership
    match_label_keys: Tuple[str, ...] = ()  # matchLabelKeys: the pod's own values of these labels join the selector

    def effective_match_labels(self, pod_labels: Dict[str, str]) -> Dict[str, str]:
        """filterTopologySpreadConstraints (common.go:96-107): selector AND key == <the incoming pod's value> for every
        matchLabelKeys entry the pod carries (mergeLabelSetWithSelector :130-143)."""
        out = dict(self.match_labels)
        for k in self.match_label_keys:
            if k in pod_labels:
                if k in out and out[k] != pod_labels[k]:
                    out[k] = "\x01unsatisfiable"   # key == a AND key == b: nothing matches (no label value can hold a control character)
                else:
                    out[k] = pod_labels[k]
        return out


@dataclass
class ContainerPort:
    host_port: int
    host_ip: str = ""
    protocol: str = ""


@dataclass
class NodeSelectorTerm:
    """v1.NodeSelectorTerm: matchExpressions (node labels) AND matchFields (metadata.name, In / NotIn with one value)."""
    match_expressions: List[Requirement] = field(default_factory=list)
    match_fields: List[Requirement] = field(default_factory=list)


@dataclass
class Pod:
    name: str
    namespace: str = "default"
    labels: Dict[str, str] = field(default_factory=dict)
    # total pod requests per resource name (resource.PodRequests); cpu in millicores
    requests: Dict[str, int] = field(default_factory=dict)
    tolerations: List[Toleration] = field(default_factory=list)
    node_selector: Dict[str, str] = field(default_factory=dict)
    node_affinity: List[Requirement] = field(default_factory=list)   # ONE required term, ANDed
    # requiredDuringSchedulingIgnoredDuringExecution.nodeSelectorTerms, ORed (None = not set); exclusive with node_affinity
    node_affinity_terms: Optional[List[NodeSelectorTerm]] = None
    host_ports: List[ContainerPort] = field(default_factory=list)
    anti_affinity: List[PodAffinityTerm] = field(default_factory=list)
    # PodAffinity.RequiredDuringSchedulingIgnoredDuringExecution (namespaces and / or a namespace_selector, resolved through the lister)
    affinity: List[PodAffinityTerm] = field(default_factory=list)
    # first container's requests as AsApproximateFloat64 for the fastpath chooser; None = derive
    fastpath_cpu: Optional[float] = None
    fastpath_mem: Optional[float] = None
    has_containers: bool = True
    topology_spread: bool = False          # outside the encoded subset -> fallback
    spread_constraints: List[TopologySpreadConstraint] = field(default_factory=list)
    unsupported_reason: str = ""           # anything else outside the encoded subset
    # identity / ownership: only the filter-out-schedulable pass reads these (hints.go, similar_pods.go)
    uid: str = ""
    controller_uid: str = ""               # drain.ControllerRef(pod).UID, "" = no controller
    daemonset: bool = False                # pod_utils.IsDaemonSetPod
    priority: int = 0                      # corev1helpers.PodPriority
    # digest of the spec fields the model does not carry (volumes after sanitization, ...): only equality matters
    spec_extra: str = ""

    def spec_key(self):
        """Hashable scheduling-relevant spec + labels: two pods with equal keys are interchangeable for every
        encoded Filter (what SimilarPodsSchedulingInfo.Match compares, similar_pods.go:48-50)."""
        return (self.namespace, tuple(sorted(self.labels.items())), tuple(sorted(self.requests.items())),
                tuple((t.key, t.operator, t.value, t.effect) for t in self.tolerations),
                tuple(sorted(self.node_selector.items())),
                tuple((r.key, r.operator, tuple(r.values)) for r in self.node_affinity),
                None if self.node_affinity_terms is None else tuple(
                    (tuple((r.key, r.operator, tuple(r.values)) for r in t.match_expressions),
                     tuple((r.key, r.operator, tuple(r.values)) for r in t.match_fields)) for t in self.node_affinity_terms),
                tuple((h.host_port, h.host_ip, h.protocol) for h in self.host_ports),
                tuple((t.topology_key, tuple(sorted(t.match_labels.items())),
                       tuple((r.key, r.operator, tuple(r.values)) for r in t.match_expressions), tuple(t.namespaces),
                       None if t.namespace_selector is None else tuple((r.key, r.operator, tuple(r.values)) for r in t.namespace_selector))
                      for t in self.anti_affinity),
                tuple((t.topology_key, tuple(sorted(t.match_labels.items())),
                       tuple((r.key, r.operator, tuple(r.values)) for r in t.match_expressions), tuple(t.namespaces),
                       None if t.namespace_selector is None else tuple((r.key, r.operator, tuple(r.values)) for r in t.namespace_selector))
                      for t in self.affinity),
                self.topology_spread, tuple((c.max_skew, c.topology_key, c.min_domains, tuple(sorted(c.match_labels.items())), c.node_taints_policy, c.node_affinity_policy, tuple(c.match_label_keys))
                                            for c in self.spread_constraints),
                self.unsupported_reason, self.has_containers, self.spec_extra)

    def fastpath_requests(self):
        """Containers[0].Resources.Requests.{Cpu,Memory}().AsApproximateFloat64()
        (binpacking_estimator.go:451-458).  A NewMilliQuantity(v) is float64(v) * 10**-3,
        a NewQuantity(v) is float64(v)  (quantity.go:468-483)."""
        if not self.has_containers:
            return 0.0, 0.0
        cpu = self.fastpath_cpu if self.fastpath_cpu is not None else float(self.requests.get(RES_CPU, 0)) * 1e-3
        mem = self.fastpath_mem if self.fastpath_mem is not None else float(self.requests.get(RES_MEMORY, 0))
        return cpu, mem


@dataclass
class Node:
    name: str
    labels: Dict[str, str] = field(default_factory=dict)
    taints: List[Taint] = field(default_factory=list)
    allocatable: Dict[str, int] = field(default_factory=dict)   # incl. "pods"
    capacity: Dict[str, int] = field(default_factory=dict)
    unschedulable: bool = False

    def allowed_pods(self) -> int:
        return int(self.allocatable.get("pods", 0))


@dataclass
class NodeInfo:
    """framework.NodeInfo: a node plus the pods already on it (DaemonSet pods for a template)."""
    node: Node
    pods: List[Pod] = field(default_factory=list)


@dataclass
class PodEquivalenceGroup:
    pods: List[Pod] = field(default_factory=list)

    def exemplar(self) -> Optional[Pod]:
        return self.pods[0] if self.pods else None


# ---------------------------------------------------------------------------------------------
# builders mirroring CA/utils/test/test_utils.go and the estimator tests
# ---------------------------------------------------------------------------------------------
@dataclass
class Container:
    """What resource.PodRequests reads of a v1.Container: its requests and, for init containers, restartPolicy: Always
    (a sidecar)."""
    requests: Dict[str, int] = field(default_factory=dict)
    restart_always: bool = False


def pod_requests(containers: Sequence[Container], init_containers: Sequence[Container] = (), overhead: Optional[Dict[str, int]] = None,
                 pod_level: Optional[Dict[str, int]] = None) -> Dict[str, int]:
    """resource.PodRequests (V/component-helpers/resource/helpers.go:151-191 over AggregateContainerRequests :198-281), which
    podutils.PodRequests (CA/utils/pod/pod.go:88-96) and NodeResourcesFit.PreFilter (fit.go:321-347) call: the value that goes
    into Pod.requests / the request lanes of the C ABI.

        sum of the containers and of the sidecars (restartable init containers)                        :216-258
        max with every init step: a plain init container + the sidecars started before it, a sidecar    :243-267
            step = the sidecars so far
        pod-level requests, when set, replace cpu / memory / hugepages-* (PodLevelResources)            :157-179
        + overhead                                                                                      :182-185

    Status-based variants (UseStatusResources: in-place resize of RUNNING containers, :205-214) do not apply to the pending
    pods this path schedules: they have no container statuses."""
    reqs: Dict[str, int] = {}

    def add(dst, src):
        for k, v in src.items():
            dst[k] = dst.get(k, 0) + int(v)

    def vmax(dst, src):
        for k, v in src.items():
            if k not in dst or int(v) > dst[k]:
                dst[k] = int(v)

    for c in containers:
        add(reqs, c.requests)
    sidecars: Dict[str, int] = {}
    init_max: Dict[str, int] = {}
    for c in init_containers:
        if c.restart_always:
            add(reqs, c.requests)
            add(sidecars, c.requests)
            step = sidecars
        else:
            step = {}
            add(step, c.requests)
            add(step, sidecars)
        vmax(init_max, step)
    vmax(reqs, init_max)
    if pod_level:
        for k, v in pod_level.items():
            if k in (RES_CPU, RES_MEMORY) or k.startswith("hugepages-"):
                reqs[k] = int(v)
    if overhead:
        add(reqs, overhead)
    return reqs


def build_test_pod(name: str, cpu: int, mem: int, *options) -> Pod:
    """BuildTestPod(name, cpuMilli, memBytes, opts...)  test_utils.go:38-70."""
    pod = Pod(name=name, namespace="default")
    if cpu >= 0:
        pod.requests[RES_CPU] = cpu
    if mem >= 0:
        pod.requests[RES_MEMORY] = mem
    for opt in options:
        opt(pod)
    return pod


def with_namespace(ns):
    def f(pod): pod.namespace = ns
    return f


def with_labels(labels):
    def f(pod): pod.labels = dict(labels)
    return f


def with_host_port(port):
    """WithHostPort  test_utils.go:152-163."""
    def f(pod):
        if port > 0:
            pod.host_ports = [ContainerPort(host_port=port)]
    return f


def with_max_skew(max_skew, key, min_domains):
    """WithMaxSkew  test_utils.go:165-184 — topology spread is outside the encoded subset."""
    def f(pod):
        if max_skew > 0:
            pod.topology_spread = True
            pod.spread_constraints = [TopologySpreadConstraint(max_skew, key, min_domains, {"app": "estimatee"})]
    return f


def with_node_names_affinity(*node_names):
    """WithNodeNamesAffinity (CA/utils/test/test_utils.go:200-220): ONE term, ONE matchFields requirement
    metadata.name In node_names."""
    def f(pod): pod.node_affinity_terms = [NodeSelectorTerm(match_fields=[Requirement("metadata.name", "In", list(node_names))])]
    return f


def with_pod_hostname_anti_affinity(labels):
    """WithPodHostnameAntiAffinity  test_utils.go:223-240."""
    def f(pod): pod.anti_affinity = [PodAffinityTerm(topology_key=LABEL_HOSTNAME, match_labels=dict(labels))]
    return f


def with_node_selector(sel):
    def f(pod): pod.node_selector = dict(sel)
    return f


def with_tolerations(tols):
    def f(pod): pod.tolerations = list(tols)
    return f


def build_test_node(name: str, cpu_milli: int, mem: int, pods: int = 100) -> Node:
    """BuildTestNode  test_utils.go:367-400: pods capacity 100, allocatable == capacity."""
    cap = {"pods": pods}
    if cpu_milli >= 0:
        cap[RES_CPU] = cpu_milli
    if mem >= 0:
        cap[RES_MEMORY] = mem
    return Node(name=name, labels={}, capacity=dict(cap), allocatable=dict(cap))


def make_node(cpu: int, mem_mib: int, pod_count: int, name: str, zone: str) -> Node:
    """makeNode  binpacking_estimator_test.go:44-64 (memory given in MiB)."""
    cap = {RES_CPU: cpu, RES_MEMORY: mem_mib * MiB, "pods": pod_count}
    return Node(name=name, labels={LABEL_HOSTNAME: name, LABEL_ZONE: zone}, capacity=dict(cap), allocatable=dict(cap))


def make_pod_equivalence_group(pod: Pod, count: int) -> PodEquivalenceGroup:
    """makePodEquivalenceGroup  binpacking_estimator_test.go:34-42 (one pointer repeated)."""
    return PodEquivalenceGroup(pods=[pod] * count)
