"""Host-side mirror of the scale-down removal simulation:

    Planner.categorizeNodes loop                 CA/core/scaledown/planner/planner.go:286-336
    RemovalSimulator.SimulateNodeRemoval         CA/simulator/cluster.go:131-172
    findPlaceFor / replaceWithTaintedGhostNode   CA/simulator/cluster.go:190-265
    GetPodsToMove (host policy, stays host)      CA/simulator/drain.go:49-86

The per-candidate Fork -> unschedule -> TrySchedulePods(breakOnFailure) -> Commit/Revert chain runs on the device as
ONE call over all candidates (casim_simulate_node_removals -> K_sched with transactions).  What stays here is what
the reference also keeps outside the scheduler simulation: which pods of a node have to move (drainability rules,
PDBs), the hint map, and the snapshot objects.  When an earlier removal moved pods onto a later candidate, that
candidate lists them again after its own pods (the device keeps the log of committed moves); only when such a pod
is "sticky" — the host has to re-run its drainability / PDB rules for it — does the device stop in front of that
candidate, and this loop re-submits the rest from the updated snapshot: the point where the reference calls
GetPodsToMove again."""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import _abi
from .encoder import Encoder
from .engine import Context
from .objects import Node, NodeInfo, Pod, PodEquivalenceGroup
from .scheduling import Hints, UnsupportedPredicate, hint_key_from_pod

# simulator.UnremovableReason (cluster.go:78-103), the two this path produces
NO_PLACE_TO_MOVE_PODS = "NoPlaceToMovePods"
BLOCKED_BY_POD = "BlockedByPod"
NO_NODE_INFO = "NoNodeInfo"


@dataclass
class NodeToBeRemoved:
    """simulator.NodeToBeRemoved (cluster.go:42-54)."""
    node: object
    pods_to_reschedule: List[Pod]
    daemon_set_pods: List[Pod] = field(default_factory=list)


@dataclass
class UnremovableNode:
    """simulator.UnremovableNode (cluster.go:56-63)."""
    node: object
    reason: str


def default_pods_to_move(info: NodeInfo) -> Optional[List[Pod]]:
    """GetPodsToMove with rules that let everything drain: every non-DaemonSet pod moves (drain.go:64-84).
    Return None for a node blocked by a pod (BlockDrain)."""
    return [p for p in info.pods if not p.daemonset]


class RemovalSimulator:
    """simulator.NewRemovalSimulator(listers, clusterSnapshot, deleteOptions, drainabilityRules, persist)."""

    def __init__(self, ctx: Context, snapshot: List[NodeInfo], persist_successful_simulations: bool = True,
                 pods_to_move: Callable[[NodeInfo], Optional[List[Pod]]] = default_pods_to_move,
                 is_sticky: Callable[[Pod], bool] = lambda p: False, ext_capacity: Optional[int] = None, lanes=None):
        self.ctx = ctx
        self.snapshot = snapshot            # list order == the order lastIndex refers to; mutated when persisting
        self.can_persist = persist_successful_simulations
        self.pods_to_move = pods_to_move
        self.is_sticky = is_sticky          # pod -> True when GetPodsToMove must see it again before it moves twice (PDBs)
        self.ext_capacity = ext_capacity    # None = engine default; 0 = re-submit at every candidate with arrivals
        self.lanes = lanes
        self.hints = Hints()
        self.last_index = 0
        self.device_calls = 0

    def drop_old_hints(self):
        self.hints.drop_old()

    # ---- one device call over `names` (prefix semantics) ----------------------------------------------------
    def _submit(self, names: Sequence[str], lists: List[List[Pod]], destinations: Dict[str, bool], max_removable: int,
                atomic: Optional[Sequence[int]] = None):
        enc = Encoder(explicit_self_exclusion=True) if self.lanes is None else Encoder(lanes=self.lanes, explicit_self_exclusion=True)
        class_of: Dict[tuple, int] = {}
        pod_class: List[int] = []
        hint: List[int] = []
        sticky: List[int] = []
        off = [0]
        pos = {info.node.name: i for i, info in enumerate(self.snapshot)}
        for lst in lists:
            for p in lst:
                k = p.spec_key()
                c = class_of.get(k)
                if c is None:
                    c = enc.add_peg(PodEquivalenceGroup(pods=[p]))
                    class_of[k] = c
                pod_class.append(c)
                h = self.hints.get(hint_key_from_pod(p))
                hint.append(pos.get(h, -1) if h is not None else -1)
                sticky.append(1 if self.is_sticky(p) else 0)
            off.append(len(pod_class))
        for info in self.snapshot:
            enc.add_group(info, pegs=[])
        enc.finalize()
        dest = np.array([1 if destinations.get(info.node.name, False) else 0 for info in self.snapshot], np.uint8)
        self.device_calls += 1
        res = self.ctx.simulate_node_removals(enc.pegs, enc.groups, [pos[n] for n in names], off, pod_class, hint, dest,
                                              persist=self.can_persist, max_removable=max_removable, last_index=self.last_index,
                                              pod_sticky=sticky if any(sticky) else None, ext_capacity=self.ext_capacity,
                                              rules=enc.rules, cand_atomic=atomic if atomic is not None and any(atomic) else None)
        enc.close()
        if res.status == _abi.NG_UNSUPPORTED:
            raise UnsupportedPredicate("pods to move need a predicate outside the encoded subset")
        return res, off

    def simulate_node_removals(self, candidates: Sequence[str], destinations: Dict[str, bool],
                               max_removable: Optional[int] = None, is_atomic: Optional[Callable[[str], bool]] = None):
        """The categorizeNodes loop (planner.go:300-330): SimulateNodeRemoval per candidate in order, successful
        simulations persisted when `can_persist`, the removed node dropped from `destinations` (:318).
        `max_removable` = unneededNodesLimit(), None for no limit; 0 is a real limit and stops the loop before the
        first candidate (planner.go:303 compares `>=`).  Nodes for which `is_atomic(name)` holds (their group scales to zero or
        max, atomicScaleDownNode :339-357) do not count toward it.
        Returns (removable: List[NodeToBeRemoved], unremovable: List[UnremovableNode], skipped: names not evaluated)."""
        removable: List[NodeToBeRemoved] = []
        unremovable: List[UnremovableNode] = []
        todo = list(candidates)
        counted = 0   # len(removableList) - atomicScaleDownNodesCount
        while todo:
            if max_removable is not None and counted >= max_removable:
                break
            by_name = {info.node.name: info for info in self.snapshot}
            # GetPodsToMove on the CURRENT snapshot; a node blocked by a pod never reaches the device
            names, lists = [], []
            for n in todo:
                info = by_name.get(n)
                names.append(n)
                lists.append(self.pods_to_move(info) if info is not None else None)
            cut = next((i for i, lst in enumerate(lists) if lst is None), len(names))
            if cut == 0:
                n = todo.pop(0)
                if n in by_name:
                    unremovable.append(UnremovableNode(by_name[n].node, BLOCKED_BY_POD))
                else:   # cluster.go:138-146: not in the snapshot, decided before any simulation
                    unremovable.append(UnremovableNode(Node(name=n), NO_NODE_INFO))
                continue
            # the ABI field counts what is LEFT and keeps 0 for "no limit": a limited call always has left >= 1 here
            left = (max_removable - counted) if max_removable is not None else 0
            infos = list(self.snapshot)   # node indices of this call refer to this list
            atomic = [1 if is_atomic(n) else 0 for n in names[:cut]] if is_atomic is not None else None
            res, off = self._submit(names[:cut], lists[:cut], destinations, left, atomic)
            self.last_index = res.last_index
            flat = [p for lst in lists[:cut] for p in lst]
            again: Dict[int, list] = {}   # candidate -> [(pod, destination)] for the pods it listed again
            for k, e, m in zip(res.ext_candidate.tolist(), res.ext_pod.tolist(), res.ext_node.tolist()):
                again.setdefault(k, []).append((flat[e], m))
            n_done = res.n_processed
            for k in range(res.n_processed):
                if int(res.removable[k]) == 2:   # not evaluated: the removable limit was reached inside the call
                    n_done = k
                    break
                info = by_name[names[k]]
                moved = list(zip(lists[k], (int(res.node_out[i]) for i in range(off[k], off[k + 1])))) + again.get(k, [])
                for p, m in moved:
                    if m >= 0:  # hints.Set on every placement, reverted simulation or not (hinting_simulator.go:108,133)
                        self.hints.set(hint_key_from_pod(p), infos[m].node.name)
                if int(res.removable[k]) == 1:
                    removable.append(NodeToBeRemoved(info.node, [p for p, _ in moved], [p for p in info.pods if p.daemonset]))
                    if atomic is None or not atomic[k]:
                        counted += 1
                    if self.can_persist:
                        # Commit: the pods now run on their destinations (arrival order), the node leaves the list
                        for p, m in moved:
                            infos[m].pods.append(p)
                        self.snapshot.remove(info)
                        destinations.pop(names[k], None)   # planner.go:318 (the planner always persists)
                else:
                    unremovable.append(UnremovableNode(info.node, NO_PLACE_TO_MOVE_PODS))
            if n_done == 0:
                break
            todo = todo[n_done:]
        return removable, unremovable, todo


class Planner:
    """The simulating part of planner.Planner.UpdateClusterState (planner.go:118-141): recently evicted pods are put
    back into the snapshot (injectPods :256-270, one TrySchedulePods call on the device), then categorizeNodes
    (:286-336) walks the eligible candidates with persisted removal simulations.  Eligibility, PDB accounting and
    the unneeded-time bookkeeping around it stay with the caller (host policy, SURVEY §8 out of scope)."""

    def __init__(self, ctx: Context, snapshot: List[NodeInfo],
                 pods_to_move: Callable[[NodeInfo], Optional[List[Pod]]] = default_pods_to_move,
                 is_sticky: Callable[[Pod], bool] = lambda p: False, lanes=None):
        from .scheduling import HintingSimulator
        self.snapshot = snapshot
        self.rs = RemovalSimulator(ctx, snapshot, True, pods_to_move, is_sticky, lanes=lanes)
        self.actuation_injector = HintingSimulator(ctx, lanes)

    def inject_pods(self, pods: Sequence[Pod]) -> bool:
        """injectPods: breakOnFailure, any node acceptable; what was placed stays in the snapshot.  False when not
        every pod found a node (the reference logs a warning and goes on)."""
        if not pods:
            return True
        # one lastIndex for both simulators: it belongs to the snapshot's plugin runner (plugin_runner.go:51)
        self.actuation_injector.last_index = self.rs.last_index
        statuses, _ = self.actuation_injector.try_schedule_pods(self.snapshot, list(pods), break_on_failure=True)
        self.rs.last_index = self.actuation_injector.last_index
        by_name = {info.node.name: info for info in self.snapshot}
        for s in statuses:
            by_name[s.node_name].pods.append(s.pod)
        return len(statuses) == len(pods)

    @staticmethod
    def unneeded_nodes_limit(previously_unneeded: int, max_scale_down_parallelism: int, scale_down_unneeded_time: float,
                             min_update_interval: float) -> int:
        """unneededNodesLimit (planner.go:385-400): previously unneeded + 2 N, capped by N * (U / I) + N loops' worth."""
        n = max_scale_down_parallelism
        limit = previously_unneeded + 2 * n
        interval = max(1, round(min_update_interval * 1e9))                 # time.Duration arithmetic (integer ns)
        u = max(round(scale_down_unneeded_time * 1e9), interval)
        upper = n * (u // interval) + n
        return min(upper, limit)

    def update_cluster_state(self, pod_destinations: Sequence[str], eligible_candidates: Sequence[str],
                             recent_evictions: Sequence[Pod] = (), unneeded_nodes_limit: Optional[int] = None,
                             is_atomic: Optional[Callable[[str], bool]] = None):
        """Returns (removable, unremovable, skipped) of the categorizeNodes loop."""
        self.inject_pods(recent_evictions)
        out = self.rs.simulate_node_removals(list(eligible_candidates), {n: True for n in pod_destinations}, unneeded_nodes_limit,
                                             is_atomic)
        self.rs.drop_old_hints()
        self.actuation_injector.drop_old_hints()
        return out
