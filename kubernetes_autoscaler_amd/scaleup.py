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
