"""Host-side mirror of pod equivalence grouping:

    BuildPodGroups / groupPodsBySchedulingProperties   CA/core/scaleup/equivalence/groups.go:39-104
    SchedulablePodGroups (the PEG x node-group matrix)   CA/core/scaleup/orchestrator/orchestrator.go:535-570

Grouping is string hashing and stays on the host: `casim_enc_group_pods` in libcasim's encoder (group_pods_native); the
Python restatement below is the checker for it.  What moves to the device is the matrix: every group's exemplar against every node-group template in ONE call
(`Context.feasibility` -> feas_kernel), instead of G x NG CheckPredicates runs."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from .encoder import Encoder
from .engine import Context
from .objects import NodeInfo, Pod, PodEquivalenceGroup

MAX_EQUIVALENCE_GROUPS_BY_CONTROLLER = 10  # groups.go:58


@dataclass
class PodGroup:
    """equivalence.PodGroup (groups.go:31-36)."""
    pods: List[Pod]
    scheduling_errors: Dict[str, str] = field(default_factory=dict)
    schedulable_groups: List[str] = field(default_factory=list)
    schedulable: bool = False


def group_pods_by_scheduling_properties(pods: Sequence[Pod]) -> List[List[Pod]]:
    """groupPodsBySchedulingProperties (groups.go:62-93).  The reference returns a Go map (group ids are
    meaningless); here the groups come in creation order.  Pods without a controller and DaemonSet pods are
    singletons; a controller gets at most 10 reusable groups, later distinct specs are singletons too."""
    groups: List[List[Pod]] = []
    by_controller: Dict[str, List[tuple]] = {}   # uid -> [(group index, spec key of the representant)]
    for pod in pods:
        if not pod.controller_uid or pod.daemonset:
            groups.append([pod])
            continue
        egs = by_controller.setdefault(pod.controller_uid, [])
        key = pod.spec_key()
        hit = next((gid for gid, k in egs if k == key), None)   # match(): labels DeepEqual + PodSpecSemanticallyEqual
        if hit is not None:
            groups[hit].append(pod)
            continue
        if len(egs) < MAX_EQUIVALENCE_GROUPS_BY_CONTROLLER:
            egs.append((len(groups), key))
        groups.append([pod])
    return groups


def group_pods_native(pods: Sequence[Pod], share_specs: bool = False) -> List[List[Pod]]:
    """The same grouping through libcasim (casim_enc_group_pods): what a shim calls instead of BuildPodGroups."""
    enc = Encoder()
    try:
        gid, n = enc.group_pods(pods, share_specs=share_specs)
    finally:
        enc.close()
    groups: List[List[Pod]] = [[] for _ in range(n)]
    for p, g in zip(pods, gid):
        groups[int(g)].append(p)
    return groups


def build_pod_groups(pods: Sequence[Pod], native: bool = True) -> List[PodGroup]:
    """BuildPodGroups (groups.go:39-49); native = through the C ABI (the product path), else the Python restatement above (test scaffolding)."""
    return [PodGroup(pods=g) for g in (group_pods_native(pods) if native else group_pods_by_scheduling_properties(pods))]


@dataclass
class SchedulingError:
    """clustersnapshot.SchedulingError of kind FailingPredicateError (CA/simulator/clustersnapshot/scheduling_error.go:29-52):
    the Filter plugin that rejected the pod on the node and its reasons; what the orchestrator keeps per node group in
    eg.SchedulingErrors (orchestrator.go:553-567) and NoScaleUp events print."""
    failing_predicate_name: str
    failing_predicate_reasons: List[str]
    unknown: bool = False     # the pod needs a predicate outside the encoded subset: run the Go CheckPredicates for the real answer

    def verbose_error(self) -> str:
        return f"predicate {self.failing_predicate_name!r} failed: {', '.join(self.failing_predicate_reasons)}"


def decode_scheduling_error(code: int, lanes: Sequence[str]) -> Optional[SchedulingError]:
    """One uint16 of casim_feasibility_reasons -> SchedulingError (None = the pod fits)."""
    from . import _abi
    code = int(code)
    plugin = code & _abi.PLUGIN_MASK
    if plugin == 0:
        return None
    if plugin == 15:
        return SchedulingError(_abi.PLUGIN_NAMES[15], [], unknown=True)
    if plugin == 6:
        reasons = ["Too many pods"] if code & _abi.REASON_TOO_MANY_PODS else []
        for r, name in enumerate(lanes):   # fitsRequest order: pods, then cpu, memory, ephemeral-storage, scalar resources
            if code & _abi.reason_insufficient(r):
                reasons.append(f"Insufficient {name}")
        return SchedulingError(_abi.PLUGIN_NAMES[6], reasons)
    return SchedulingError(_abi.PLUGIN_NAMES[plugin], [_abi.PLUGIN_REASONS[plugin]])


def schedulable_pod_groups_with_errors(ctx: Context, pod_groups: Sequence["PodGroup"], templates: Dict[str, NodeInfo], lanes=None):
    """SchedulablePodGroups + the SchedulingErrors the orchestrator records for the cells that fail: returns
    (ok [node group][pod group], errors {node group name: {pod group index: SchedulingError}})."""
    enc = Encoder() if lanes is None else Encoder(lanes=lanes)
    for g in pod_groups:
        enc.add_peg(PodEquivalenceGroup(pods=g.pods))
    names = list(templates)
    for n in names:
        enc.add_group(templates[n], pegs=None)
    enc.finalize()
    codes = ctx.feasibility_reasons(enc.pegs, enc.groups, enc.port_block)
    lane_names = enc.lanes
    enc.close()
    ok = codes == 0
    errors: Dict[str, Dict[int, SchedulingError]] = {}
    for i, n in enumerate(names):
        errors[n] = {j: decode_scheduling_error(codes[i, j], lane_names) for j in range(len(pod_groups)) if codes[i, j] != 0}
    for j, g in enumerate(pod_groups):
        g.schedulable_groups = [names[i] for i in range(len(names)) if ok[i, j]]
        g.schedulable = bool(g.schedulable_groups)
        g.scheduling_errors = {n: errors[n][j] for n in names if j in errors[n]}
    return ok, errors


def schedulable_pod_groups(ctx: Context, pod_groups: Sequence[PodGroup], templates: Dict[str, NodeInfo], lanes=None) -> np.ndarray:
    """SchedulablePodGroups for every node group at once: bool matrix [node group][pod group] (the exemplar of the
    group passes CheckPredicates on the template).  Also fills PodGroup.schedulable / schedulable_groups the way
    orchestrator.go:1049-1051 + processors consume them."""
    enc = Encoder() if lanes is None else Encoder(lanes=lanes)
    for g in pod_groups:
        enc.add_peg(PodEquivalenceGroup(pods=g.pods))
    names = list(templates)
    for n in names:
        enc.add_group(templates[n], pegs=None)
    enc.finalize()
    bits = ctx.feasibility(enc.pegs, enc.groups)
    enc.close()
    ok = np.zeros((len(names), len(pod_groups)), bool)
    for i in range(len(names)):
        for j in range(len(pod_groups)):
            ok[i, j] = bool((int(bits[i, j >> 6]) >> (j & 63)) & 1)
    for j, g in enumerate(pod_groups):
        g.schedulable_groups = [names[i] for i in range(len(names)) if ok[i, j]]
        g.schedulable = bool(g.schedulable_groups)
    return ok
