"""Host-side mirror of pod equivalence grouping:

    BuildPodGroups / groupPodsBySchedulingProperties   CA/core/scaleup/equivalence/groups.go:39-104
    SchedulablePodGroups (the PEG x node-group matrix)   CA/core/scaleup/orchestrator/orchestrator.go:535-570

Grouping is string hashing and stays on the host (the Go shim keeps calling BuildPodGroups); what moves to the
device is the matrix: every group's exemplar against every node-group template in ONE call
(`Context.feasibility` -> feas_kernel), instead of G x NG CheckPredicates runs."""
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

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


def build_pod_groups(pods: Sequence[Pod]) -> List[PodGroup]:
    """BuildPodGroups (groups.go:39-49)."""
    return [PodGroup(pods=g) for g in group_pods_by_scheduling_properties(pods)]


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
