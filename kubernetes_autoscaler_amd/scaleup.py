"""The scale-up simulation loop of ScaleUpOrchestrator.prepareScaleUp
(CA/core/scaleup/orchestrator/orchestrator.go:1043-1085) as ONE batched device call:

    for ng in validNodeGroups: SchedulablePodGroups(...)       :1049-1051  -> feas_kernel + CSR kernels
    for ng in validNodeGroups: ComputeExpansionOption(...)     :1053-1068  -> order_kernel + pack_kernel
    ExpanderStrategy.BestOption(options)                       :1079       -> option_kernel (+ RCCL when sharded)

The reference runs the two loops sequentially on one goroutine; every node group is an independent
Estimate (Fork/Revert), so here all of them are simulated by one launch, one wavefront per group."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from .encoder import Encoder
from .engine import BatchResult, Context, Problem
from .estimator import (ClusterSnapshotView, EstimationContext, NodeGroup, ThresholdBasedEstimationLimiter, _fastpath_eligible,
                        encode_cluster_estimate)
from .expander import ChainStrategy, Option
from .objects import NodeInfo, Pod, PodEquivalenceGroup


@dataclass
class ScaleUpPlan:
    """What prepareScaleUp derives: one Option per node group that can help + the chosen one."""
    options: List[Option]
    best: Optional[Option]
    n_equally_good: int
    schedulable_pod_groups: Dict[str, List[int]]      # node group id -> PEG indices (SchedulablePodGroups)
    result: BatchResult
    delegated: List[str] = field(default_factory=list)  # groups with CASIM_NG_UNSUPPORTED (Go estimator's job)


class ScaleUpSimulator:
    def __init__(self, engine_ctx: Context, limiter: ThresholdBasedEstimationLimiter, expander: ChainStrategy = None,
                 max_nodes_total: int = 0, fastpath: bool = False, lanes: Sequence[str] = ("cpu", "memory")):
        self.ctx = engine_ctx
        self.limiter = limiter
        self.expander = expander or ChainStrategy()
        self.max_nodes_total = max_nodes_total
        self.fastpath = fastpath
        self.lanes = lanes

    def prepare_scale_up(self, pegs: List[PodEquivalenceGroup], node_groups: List[NodeGroup], node_infos: Dict[str, NodeInfo],
                         snapshot: ClusterSnapshotView, all_or_nothing: bool = False,
                         similar_node_groups: Optional[Dict[str, List[NodeGroup]]] = None) -> ScaleUpPlan:
        """`similar_node_groups`: node group id -> its similar groups (option.SimilarNodeGroups, what the balancing processor
        found); they feed SngCapacityThreshold through the EstimationContext exactly as orchestrator.go:409-412 does."""
        similar_node_groups = similar_node_groups or {}
        enc = Encoder(lanes=self.lanes)
        for pg in pegs:
            enc.add_peg(pg)
        for info in snapshot.existing:
            for p in info.pods:
                enc.add_existing_pod(p, info.node.labels)
        for ng in node_groups:
            # estimatorBuilder(..., NewEstimationContext(MaxNodesTotal, similarNodeGroups, currentNodeCount))  :409-412
            context = EstimationContext(self.max_nodes_total, list(similar_node_groups.get(ng.id(), [])), len(snapshot.existing))
            self.limiter.start_estimation(pegs, ng, context)
            enc.add_group(node_infos[ng.id()], max_nodes=self.limiter.device_max_nodes(), existing_nodes=len(snapshot.existing),
                          last_index=snapshot.last_index, pegs=None)   # None: SchedulablePodGroups runs on the device
            self.limiter.end_estimation()
        enc.finalize()
        rerun: Dict[int, dict] = {}   # groups the batch delegated and K_est estimated: index -> result
        with Problem(self.ctx, enc.pegs, enc.groups, self.fastpath) as prob:
            prob.run()
            res = prob.fetch()
            # Groups whose PEGs carry domain rules (PodTopologySpread, zone anti-affinity) come back UNSUPPORTED from the
            # template-mode batch: Estimate them on the whole snapshot and let them join the expander reduce.
            can_rerun = not self.fastpath or not any(pg.pods and _fastpath_eligible(pg.pods[0]) for pg in pegs)
            for i, ng in enumerate(node_groups):
                if int(res.status[i]) == 0 or not can_rerun:
                    continue
                order, _ = res.group(i)
                ids = sorted(int(x) for x in order)   # SchedulablePodGroups of this group (device feasibility)
                context = EstimationContext(self.max_nodes_total, list(similar_node_groups.get(ng.id(), [])), len(snapshot.existing))
                self.limiter.start_estimation(pegs, ng, context)
                maxn = self.limiter.device_max_nodes()
                self.limiter.end_estimation()
                enc2 = encode_cluster_estimate(self.lanes, [pegs[j] for j in ids], snapshot.existing, node_infos[ng.id()], maxn)
                try:
                    rc, out = self.ctx.estimate_on_cluster(enc2.pegs, enc2.groups, len(snapshot.existing), maxn, snapshot.last_index,
                                                           enc2.rules, enc2.port_block)
                finally:
                    enc2.close()
                if rc != 0:
                    continue
                out["peg_ids"] = ids
                prob.set_group_result(i, out)
                rerun[i] = out
            # orchestrator.go:1057-1063: empty options and, for all-or-nothing, partial ones are dropped BEFORE
            # ExpanderStrategy.BestOption (:1079) sees the list: the mask goes to the device reduce with the chain
            total_pods = sum(len(pg.pods) for pg in pegs)
            options, schedulable, delegated = [], {}, []
            by_group: Dict[int, Option] = {}
            valid = np.zeros(len(node_groups), np.uint8)
            for i, ng in enumerate(node_groups):
                order, placed = res.group(i)
                schedulable[ng.id()] = sorted(int(x) for x in order)
                node_count = int(res.node_count[i])
                if i in rerun:
                    r = rerun[i]
                    order, placed, node_count = [r["peg_ids"][int(k)] for k in r["order"]], r["placed"], r["node_count"]
                elif int(res.status[i]) != 0:
                    delegated.append(ng.id())
                    continue
                pods: List[Pod] = []
                for pg_id, n in zip(order, placed):
                    pods.extend(pegs[int(pg_id)].pods[:int(n)])
                opt = Option(node_group=ng, node_count=node_count, pods=pods)
                if not pods or opt.node_count == 0:
                    continue
                if all_or_nothing and len(pods) < total_pods:
                    continue
                options.append(opt)
                by_group[i] = opt
                valid[i] = 1
            best_idx, n_best, best_set = self.expander.best_option_index(prob, valid=valid)
        best = by_group.get(best_idx) if best_idx >= 0 else None
        return ScaleUpPlan(options, best, n_best, schedulable, res, delegated)


# ---------------------------------------------------------------------------------------------------------------------
# The host rules of ScaleUpOrchestrator.ScaleUp around the simulation (orchestrator.go:86-196, 380-437, 1043-1180): what turns the
# per-group answers of SchedulablePodGroups + Estimate into the expander's option list, the final size change and the three pod
# sets of status.ScaleUpStatus.  Pure host logic over results — `estimates` come from the device (tests also feed it the CPU checker's answers).
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class GroupEstimate:
    """One node group after the two loops of prepareScaleUp (:1049-1068)."""
    node_group: NodeGroup
    schedulable: List[int]        # PEG indices whose sample pod passed CheckPredicates on the template (SchedulablePodGroups :535-570)
    node_count: int               # Estimate's first result
    pods: List[Pod]               # ... and its second: the pods that fit, in placement order


@dataclass
class ScaleUpDecision:
    scale_up: bool
    options: List[Option]                  # what ExpanderStrategy.BestOption receives (:1079)
    best: Optional[Option]
    final_group: Optional[str]
    final_size_change: int
    pods_triggered_scale_up: List[Pod]
    pods_remain_unschedulable: List[Pod]
    pods_await_evaluation: List[Pod]
    reason: str = ""


def decide_scale_up(pegs: List[PodEquivalenceGroup], estimates: List[GroupEstimate], n_existing_nodes: int, *, all_or_nothing: bool = False,
                    zero_or_max_node_scaling: bool = False, max_nodes_total: int = 0, stop_binpacking=None, choose=None) -> ScaleUpDecision:
    """`choose(options) -> Option | None`: the expander (default: the first option, what a test's mock strategy picks when it is given
    nothing).  `stop_binpacking(options) -> bool`: processors.BinpackingLimiter.StopBinpacking, asked after every group (:1065)."""
    n_pods = sum(len(pg.pods) for pg in pegs)
    schedulable_somewhere = [False] * len(pegs)
    failing_on: Dict[str, set] = {}                    # group id -> PEGs whose sample pod failed there (eg.SchedulingErrors)
    for ge in estimates:                               # first loop (:1049-1051): every group, whatever the second loop does
        ok = set(ge.schedulable)
        for i in ok:
            schedulable_somewhere[i] = True
        failing_on[ge.node_group.id()] = {i for i in range(len(pegs)) if i not in ok}
    options: List[Option] = []
    for ge in estimates:                               # second loop (:1053-1068)
        node_count, pods = (ge.node_count, list(ge.pods)) if ge.schedulable else (0, [])
        if zero_or_max_node_scaling:                   # ComputeExpansionOption :421-434
            if all_or_nothing and node_count > ge.node_group.max_size():
                node_count, pods = 0, []
            if node_count > 0:
                node_count = ge.node_group.max_size()
        if pods and node_count > 0 and not (all_or_nothing and len(pods) < n_pods):
            options.append(Option(node_group=ge.node_group, node_count=node_count, pods=pods))
        if stop_binpacking is not None and stop_binpacking(options):
            break

    def remaining(all_unschedulable=False):            # GetRemainingPods :803-819 / markAllGroupsAsUnschedulable
        return [p for i, pg in enumerate(pegs) if all_unschedulable or not schedulable_somewhere[i] for p in pg.pods]

    def none(reason, all_unschedulable=False):
        return ScaleUpDecision(False, options, None, None, 0, [], remaining(all_unschedulable), [], reason)
    if not options:
        return none("no expansion options")
    best = choose(options) if choose is not None else options[0]
    if best is None or best.node_count <= 0:
        return none("expander filtered out all options", True)
    new_nodes = best.node_count
    if max_nodes_total > 0 and n_existing_nodes + new_nodes > max_nodes_total:   # GetCappedNewNodeCount
        new_nodes = max_nodes_total - n_existing_nodes
        if new_nodes < 1:
            return none("max node total count reached", True)
    if new_nodes < best.node_count and all_or_nothing:
        return none("all-or-nothing", True)            # abortAllOrNothing :895-903, :1120
    capacity = best.node_group.max_size() - best.node_group.target_size()      # the one-group form of balanceScaleUps' capacity check (:1160-1169)
    if capacity < new_nodes:
        if all_or_nothing:
            return none("all-or-nothing", True)
        new_nodes = capacity
    if new_nodes <= 0:
        return none("node group at its max size", True)
    bad = failing_on.get(best.node_group.id(), set())
    awaiting = [p for i, pg in enumerate(pegs) if schedulable_somewhere[i] and i in bad for p in pg.pods]   # GetPodsAwaitingEvaluation :998-1009
    return ScaleUpDecision(True, options, best, best.node_group.id(), new_nodes, list(best.pods), remaining(), awaiting)


def estimates_from_results(pegs: List[PodEquivalenceGroup], node_groups: List[NodeGroup], per_group) -> List[GroupEstimate]:
    """`per_group[i]` = (PEG ids in processing order, pods placed per entry, node count) of group i (a BatchResult row, or the same three things from any other estimator)."""
    out = []
    for ng, (order, placed, node_count) in zip(node_groups, per_group):
        pods: List[Pod] = []
        for pg_id, n in zip(order, placed):
            pods.extend(pegs[int(pg_id)].pods[:int(n)])
        out.append(GroupEstimate(ng, sorted(int(x) for x in order), int(node_count), pods))
    return out
