"""Host-side mirror of the filter-out-schedulable pass:

    filterOutSchedulablePodListProcessor.Process / filterOutSchedulableByPacking
        CA/core/podlistprocessor/filter_out_schedulable.go:48-133
    HintingSimulator.TrySchedulePods          CA/simulator/scheduling/hinting_simulator.go:53-135
    Hints                                     CA/simulator/scheduling/hints.go
    SimilarPodsScheduling (bookkeeping only)  CA/simulator/scheduling/similar_pods.go:38-107

Same names, argument meaning and results as the reference; the per-pod RunFilters* loop is ONE device call
(casim_try_schedule_pods -> K_sched_static + K_sched in csrc/casim_sched.h).  Nothing here evaluates a
Filter on the host: without an MI355X the Context cannot be created."""
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from .encoder import Encoder
from .engine import Context
from .objects import NodeInfo, Pod, PodEquivalenceGroup

MAX_PODS_PER_OWNER_REF = 10  # similar_pods.go:52


def hint_key_from_pod(pod: Pod) -> str:
    """HintKeyFromPod (hints.go:28-33)."""
    return pod.uid if pod.uid else f"{pod.namespace}/{pod.name}"


class Hints:
    """Two-generation hint map (hints.go:36-75): DropOld keeps only the hints set since the last DropOld."""

    def __init__(self):
        self.current: Dict[str, str] = {}
        self.old: Dict[str, str] = {}

    def get(self, key: str) -> Optional[str]:
        if key in self.current:
            return self.current[key]
        return self.old.get(key)

    def set(self, key: str, node_name: str):
        self.current[key] = node_name

    def drop_old(self):
        self.old = self.current
        self.current = {}


class SimilarPodsScheduling:
    """similar_pods.go:38-107: per controller UID the specs already found unschedulable (at most 10: beyond that the controller
    "overflows" and nothing more is cached for it).  Match = DeepEqual(labels) AND PodSpecSemanticallyEqual (utils/utils.go:64-75),
    which ignores projected volumes and the per-pod fields the sanitizer drops — exactly what Pod.spec_key() compares: labels,
    namespace and every scheduling-relevant spec field of the model, `spec_extra` standing for the spec fields the model does not
    carry (a non-projected volume changes it, a projected one does not).  DaemonSet pods are never cached (:84-86), pods without a
    controller neither cached nor found (:66-69, :84)."""

    def __init__(self):
        self.items: Dict[str, List[tuple]] = {}
        self.overflowing_controllers = set()

    def is_similar_unschedulable(self, pod: Pod) -> bool:
        if not pod.controller_uid:
            return False
        key = pod.spec_key()
        return any(k == key for k in self.items.get(pod.controller_uid, ()))

    def set_unschedulable(self, pod: Pod):
        if not pod.controller_uid or pod.daemonset:
            return
        pm = self.items.setdefault(pod.controller_uid, [])
        if len(pm) >= MAX_PODS_PER_OWNER_REF:
            self.overflowing_controllers.add(pod.controller_uid)
            return
        pm.append(pod.spec_key())

    def overflowing_controller_count(self) -> int:
        return len(self.overflowing_controllers)


@dataclass
class Status:
    """scheduling.Status: a pod and the node it can be moved to."""
    pod: Pod
    node_name: str


class UnsupportedPredicate(Exception):
    """A pending pod needs a Filter outside the encoded subset (required pod affinity, volumes, ...): the shim runs the
    reference HintingSimulator for this loop iteration."""


def encode_pending_pods(nodes: Sequence[NodeInfo], pods: Sequence[Pod], lanes=None):
    """Classes = distinct pending-pod specs (first-seen order); node records = the snapshot's nodes with their pods.
    Returns (encoder, pod_class[P])."""
    enc = Encoder(explicit_self_exclusion=True) if lanes is None else Encoder(lanes=lanes, explicit_self_exclusion=True)
    class_of: Dict[tuple, int] = {}
    pod_class = np.zeros(len(pods), np.int32)
    for i, p in enumerate(pods):
        k = p.spec_key()
        c = class_of.get(k)
        if c is None:
            c = enc.add_peg(PodEquivalenceGroup(pods=[p]))
            class_of[k] = c
        pod_class[i] = c
    for info in nodes:
        enc.add_group(info, pegs=[])
    enc.finalize()
    return enc, pod_class


class HintingSimulator:
    """scheduling.NewHintingSimulator(): keeps the hints between loop iterations."""

    def __init__(self, ctx: Context, lanes=None):
        self.ctx = ctx
        self.lanes = lanes
        self.hints = Hints()
        self.last_index = 0  # the snapshot's lastIndexOrderMapping survives between calls in the reference store

    def try_schedule_pods(self, nodes: Sequence[NodeInfo], pods: Sequence[Pod], break_on_failure: bool = False,
                          is_node_acceptable: Optional[Callable[[NodeInfo], bool]] = None) -> Tuple[List[Status], int]:
        """TrySchedulePods (:53-83): statuses of the pods that found a node (processing order) and the number of
        controllers with more than 10 distinct failing specs.  `nodes` is the snapshot in list order; the caller
        forks / commits it (the scheduled pods are NOT appended to `nodes` here)."""
        if not pods:
            return [], 0
        enc, pod_class = encode_pending_pods(nodes, pods, self.lanes)
        name_to_index = {info.node.name: i for i, info in enumerate(nodes)}
        hint = np.full(len(pods), -1, np.int32)
        for i, p in enumerate(pods):
            h = self.hints.get(hint_key_from_pod(p))
            if h is not None:
                hint[i] = name_to_index.get(h, -1)  # hinted node left the cluster: look elsewhere (:94-97)
        ctrl: Dict[str, int] = {}
        similar = [(-1 if not p.controller_uid or p.daemonset else ctrl.setdefault(p.controller_uid, len(ctrl))) for p in pods]
        acceptable = None
        if is_node_acceptable is not None:
            acceptable = np.array([1 if is_node_acceptable(info) else 0 for info in nodes], np.uint8)
        status, node_out, last_index, _ = self.ctx.try_schedule_pods(
            enc.pegs, enc.groups, pod_class, hint_node=hint, node_acceptable=acceptable, break_on_failure=break_on_failure,
            last_index=self.last_index, rules=enc.rules, similar_key=similar)
        enc.close()
        if status == _abi.NG_UNSUPPORTED:
            raise UnsupportedPredicate("pending pods need a predicate outside the encoded subset")
        self.last_index = last_index
        statuses: List[Status] = []
        similar_pods = SimilarPodsScheduling()   # (the device decided every pod; this replays the reference's bookkeeping for the metric)
        for i, p in enumerate(pods):
            m = int(node_out[i])
            if m >= 0:
                name = nodes[m].node.name
                self.hints.set(hint_key_from_pod(p), name)
                statuses.append(Status(p, name))
                continue
            # trySchedule (hinting_simulator.go:112-135): a pod whose like already failed is not tried again, the others are recorded
            if not similar_pods.is_similar_unschedulable(p):
                similar_pods.set_unschedulable(p)
            if break_on_failure:
                break
        return statuses, similar_pods.overflowing_controller_count()

    def drop_old_hints(self):
        self.hints.drop_old()


class FilterOutSchedulablePodListProcessor:
    """NewFilterOutSchedulablePodListProcessor(nodeFilter) (filter_out_schedulable.go:40-45)."""

    def __init__(self, ctx: Context, node_filter: Optional[Callable[[NodeInfo], bool]] = None, lanes=None):
        self.scheduling_simulator = HintingSimulator(ctx, lanes)
        self.node_filter = node_filter

    def process(self, nodes: Sequence[NodeInfo], unschedulable_pods: List[Pod]) -> List[Pod]:
        """Process (:48-89) = filterOutSchedulableByPacking (:97-133): highest priority first, returns the pods that
        still need help.  sort.Slice is unstable; the canonical order here keeps ties in input order."""
        candidates = sorted(unschedulable_pods, key=lambda p: -p.priority)
        statuses, _ = self.scheduling_simulator.try_schedule_pods(nodes, candidates, False, self.node_filter)
        scheduled = {id(s.pod) for s in statuses}
        left = [p for p in candidates if id(p) not in scheduled]
        self.scheduling_simulator.drop_old_hints()
        return left
