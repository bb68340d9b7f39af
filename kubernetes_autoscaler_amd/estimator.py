"""Host-side mirror of the reference's estimator package (CA/estimator/*.go) around the HIP engine.

Same names, argument meaning and behaviour as the reference so that tests read like the reference's:
  Threshold / NewStaticThreshold / NewSngCapacityThreshold / NewClusterCapacityThreshold
  EstimationLimiter / NewThresholdBasedEstimationLimiter      CA/estimator/threshold_based_limiter.go
  EstimationContext                                           CA/estimator/estimation_context.go:24
  EstimationPodOrderer / NewDecreasingPodOrderer              CA/estimator/decreasing_pod_orderer.go
  Estimator / NewBinpackingNodeEstimator / EstimatorBuilder   CA/estimator/{estimator,binpacking_estimator}.go
The limiter arithmetic is host logic (it only produces `max_nodes`); ordering, bin-packing and every
predicate check run on the device — `BinpackingNodeEstimator.estimate` has no host implementation."""
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

from . import _abi
from .encoder import Encoder
from .engine import Context, Problem
from .objects import NodeInfo, Pod, PodEquivalenceGroup

BINPACKING_ESTIMATOR_NAME = "binpacking"
GPU_BINPACKING_ESTIMATOR_NAME = "gpu-binpacking"
AVAILABLE_ESTIMATORS = [BINPACKING_ESTIMATOR_NAME, GPU_BINPACKING_ESTIMATOR_NAME]


# ---- cloudprovider.NodeGroup: the three methods the thresholds read ---------------------------
@dataclass
class NodeGroup:
    id_: str
    max_size_: int = 0
    target_size_: int = 0

    def id(self) -> str:
        return self.id_

    def max_size(self) -> int:
        return self.max_size_

    def target_size(self) -> int:
        return self.target_size_


@dataclass
class EstimationContext:
    """estimation_context.go:24-60."""
    cluster_max_node_limit_: int = 0
    similar_node_groups_: List[NodeGroup] = field(default_factory=list)
    current_node_count_: int = 0

    def similar_node_groups(self):
        return self.similar_node_groups_

    def cluster_max_node_limit(self):
        return self.cluster_max_node_limit_

    def current_node_count(self):
        return self.current_node_count_


# ---- thresholds ---------------------------------------------------------------------------------
class StaticThreshold:
    """static_threshold.go."""

    def __init__(self, max_nodes: int, max_duration: float = 0.0):
        self.max_nodes, self.max_duration = max_nodes, max_duration

    def node_limit(self, node_group, context) -> int:
        return self.max_nodes

    def duration_limit(self, node_group, context) -> float:
        return self.max_duration


class SngCapacityThreshold:
    """sng_capacity_threshold.go:34-59."""

    def node_limit(self, node_group, context) -> int:
        if context is None:
            return 0
        total = self._capacity(node_group) + sum(self._capacity(g) for g in context.similar_node_groups())
        return -1 if total <= 0 else total

    @staticmethod
    def _capacity(ng) -> int:
        cap = ng.max_size() - ng.target_size()
        return cap if cap > 0 else 0

    def duration_limit(self, node_group, context) -> float:
        return 0.0


class ClusterCapacityThreshold:
    """cluster_capacity_threshold.go:33-41."""

    def node_limit(self, node_group, context) -> int:
        if context is None or context.cluster_max_node_limit() == 0:
            return 0
        if context.cluster_max_node_limit() < 0 or context.cluster_max_node_limit() <= context.current_node_count():
            return -1
        return context.cluster_max_node_limit() - context.current_node_count()

    def duration_limit(self, node_group, context) -> float:
        return 0.0


def get_min_limit(base, target):
    """getMinLimit  threshold_based_limiter.go:45-53."""
    if base < 0 or target < 0:
        return -1
    if (base == 0 or base > target) and target > 0:
        return target
    return base


class ThresholdBasedEstimationLimiter:
    """thresholdBasedEstimationLimiter  threshold_based_limiter.go:26-77.  On the device path only
    StartEstimation's fold matters (max_nodes travels to the packer, which counts the grants itself);
    PermissionToAddNode is kept for interface parity and for the duration cut-off between batches."""

    def __init__(self, thresholds: Sequence):
        self.thresholds = list(thresholds)
        self.max_nodes = 0
        self.max_duration = 0.0
        self.nodes = 0
        self.start = 0.0

    def start_estimation(self, pegs, node_group, context):
        self.start = time.monotonic()
        self.nodes = 0
        self.max_nodes = 0
        self.max_duration = 0.0
        for t in self.thresholds:
            self.max_nodes = get_min_limit(self.max_nodes, t.node_limit(node_group, context))
            self.max_duration = get_min_limit(self.max_duration, t.duration_limit(node_group, context))

    def end_estimation(self):
        pass

    def permission_to_add_node(self) -> bool:
        if self.max_nodes < 0 or (self.max_nodes > 0 and self.nodes >= self.max_nodes):
            return False
        if self.max_duration < 0 or (self.max_duration > 0 and self.start and time.monotonic() > self.start + self.max_duration):
            return False
        self.nodes += 1
        return True

    def device_max_nodes(self) -> int:
        """What the packer receives: a negative duration limit forbids every node, like :62-66."""
        return -1 if self.max_duration < 0 else self.max_nodes


class DecreasingPodOrderer:
    """decreasing_pod_orderer.go: the order itself is computed by order_kernel on the device (score =
    cpuReq/cpuAlloc + memReq/memAlloc in float64, descending).  This object only selects it."""
    name = "decreasing"


@dataclass
class ClusterSnapshotView:
    """What Estimate needs from clustersnapshot.ClusterSnapshot on the device path: how many nodes
    the snapshot already lists (they occupy list positions), the runner's lastIndex
    (plugin_runner.go:33-36) and the pods that can interact through non-hostname anti-affinity."""
    existing: List[NodeInfo] = field(default_factory=list)
    last_index: int = 0


def encode_cluster_estimate(lanes, pegs: List[PodEquivalenceGroup], existing: List[NodeInfo], template: NodeInfo, max_nodes: int):
    """Per-node tables for casim_estimate_on_cluster: the snapshot's nodes with their pods, then clones of the template
    named like the estimator names new nodes (<template>-e-<i>, hostname label = name; addNewNodeToSnapshot
    binpacking_estimator.go:326-342, SanitizedNodeInfo node_info_utils.go:93-137).  One clone per node the limiter may
    grant, at most one per pod plus the one that may stay empty."""
    import copy
    from .objects import LABEL_HOSTNAME
    enc = Encoder(lanes=lanes, explicit_self_exclusion=True)
    for pg in pegs:
        enc.add_peg(pg)
    for info in existing:
        enc.add_group(info, pegs=[])
    pods_total = sum(max(len(pg.pods), 1) for pg in pegs)
    n_clones = max_nodes if max_nodes > 0 else pods_total + 1
    n_clones = max(1, min(n_clones, pods_total + 1))
    for i in range(n_clones):
        node = copy.copy(template.node)
        node.name = f"{template.node.name}-e-{i}"
        node.labels = dict(template.node.labels)
        node.labels[LABEL_HOSTNAME] = node.name
        enc.add_group(NodeInfo(node, list(template.pods)), pegs=[])
    enc.finalize()
    return enc


def _fastpath_eligible(pod: Pod) -> bool:
    """shouldUseFastPath (binpacking_estimator.go:411-425): no topology spread, no anti-affinity on a non-hostname key."""
    from .objects import LABEL_HOSTNAME
    return not pod.topology_spread and not pod.spread_constraints and all(t.topology_key == LABEL_HOSTNAME for t in pod.anti_affinity)


class BinpackingNodeEstimator:
    """NewBinpackingNodeEstimator(clusterSnapshot, limiter, podOrderer, context, analyser, fastpath)
    binpacking_estimator.go:66-82 — backed by libcasim (HIP)."""

    def __init__(self, engine_ctx: Context, cluster_snapshot: ClusterSnapshotView, limiter: ThresholdBasedEstimationLimiter,
                 pod_orderer: Optional[DecreasingPodOrderer] = None, context: Optional[EstimationContext] = None,
                 estimation_analyser_func: Optional[Callable] = None, fastpath_binpacking_enabled: bool = False,
                 lanes: Sequence[str] = ("cpu", "memory"), fallback: Optional[Callable] = None, prefetch: Optional["PrefetchShared"] = None):
        self.prefetch = prefetch   # filled by PrefetchNodeGroupListProcessor.process before the orchestrator's loops
        self.engine_ctx = engine_ctx
        self.snapshot = cluster_snapshot
        self.limiter = limiter
        self.pod_orderer = pod_orderer or DecreasingPodOrderer()
        self.context = context
        self.analyser = estimation_analyser_func
        self.fastpath = fastpath_binpacking_enabled
        self.lanes = lanes
        self.fallback = fallback   # the Go estimator in the real shim; None here => unsupported raises

    def estimate(self, pegs: List[PodEquivalenceGroup], node_template: NodeInfo, node_group) -> Tuple[int, List[Pod]]:
        """Estimate  binpacking_estimator.go:102-161: (node count, pods that fit, in placement order)."""
        self.limiter.start_estimation(pegs, node_group, self.context)
        try:
            if self.prefetch is not None and self.analyser is None:   # (the analyser wants the pods per node: per-call path)
                hit = self.prefetch.lookup(pegs, node_template, node_group, self.limiter.device_max_nodes(), len(self.snapshot.existing),
                                           runner_last_index=self.snapshot.last_index)
                if hit is None and self.prefetch.last_miss == _abi.PREFETCH_MISS_LAST_INDEX and \
                        self.prefetch.rechain(node_group, node_template, self.snapshot.last_index):
                    # the chain was left (an earlier group ran elsewhere or was skipped): the rest of the loop as one chained batch from the
                    # runner's lastIndex of now (gpubinpacking/prefetch.go: rechain) — this lookup and the following ones hit again
                    hit = self.prefetch.lookup(pegs, node_template, node_group, self.limiter.device_max_nodes(), len(self.snapshot.existing),
                                               runner_last_index=self.snapshot.last_index)
                if hit is not None and hit["status"] == 0:
                    pods = []
                    for k, n in zip(hit["order"], hit["placed"]):
                        pods.extend(pegs[int(k)].pods[:int(n)])
                    self.limiter.nodes = hit["limiter_nodes"]
                    if self.prefetch.chain:   # the runner moves on exactly as if this Estimate had run here (plugin_runner.go:138)
                        self.snapshot.last_index = hit["last_index_out"]
                    return hit["node_count"], pods
                # miss (another PEG subset, another limiter answer, unknown group) or a delegated group: the per-call path below
            enc = Encoder(lanes=self.lanes)
            ids = [enc.add_peg(pg) for pg in pegs]
            for info in self.snapshot.existing:
                for p in info.pods:
                    enc.add_existing_pod(p, info.node.labels)
            enc.add_group(node_template, max_nodes=self.limiter.device_max_nodes(), existing_nodes=len(self.snapshot.existing),
                          last_index=self.snapshot.last_index, pegs=ids)
            enc.finalize()
            with Problem(self.engine_ctx, enc.pegs, enc.groups, self.fastpath, node_pods=self.analyser is not None) as prob:
                prob.run()
                res = prob.fetch()
            if int(res.status[0]) != 0:
                # PEGs with domain rules (PodTopologySpread, zone anti-affinity): Estimate on the whole snapshot (K_est),
                # unless the fastpath would pick up one of the other PEGs (not modelled there)
                if not self.fastpath or not any(pg.pods and _fastpath_eligible(pg.pods[0]) for pg in pegs):
                    out = self._estimate_on_cluster(pegs, node_template)
                    if out is not None:
                        return out
                if self.fallback is not None:
                    return self.fallback(pegs, node_template, node_group)
                raise NotImplementedError("a PEG needs a predicate outside the encoded subset (delegate to the Go estimator)")
            order, placed = res.group(0)
            pods: List[Pod] = []
            for pg_id, n in zip(order, placed):
                pods.extend(pegs[int(pg_id)].pods[:int(n)])
            self.snapshot.last_index = int(res.last_index_out[0])   # the runner's lastIndex persists (plugin_runner.go:138)
            self.limiter.nodes = int(res.limiter_nodes[0])
            if self.analyser is not None:
                # estimationAnalyserFunc(clusterSnapshot, nodeGroup, newNodesWithPods)  binpacking_estimator.go:157-159: the names
                # of the added nodes that received a pod, from the per-node pod counts the device kept
                names = res.nodes_with_pods(0, node_template.node.name)
                # tryFastPath books the extrapolated nodes as "<lastNodeName>-fake-<j>", j = 1 .. (binpacking_estimator.go:311-321): the
                # device returns them as a count only (node_count - listed nodes), the names are rebuilt here
                fakes = int(res.node_count[0]) - len(names)
                if self.fastpath and fakes > 0 and int(res.nodes_added[0]) > 0:
                    last = f"{node_template.node.name}-e-{int(res.nodes_added[0]) - 1}"
                    names = names + [f"{last}-fake-{j}" for j in range(1, fakes + 1)]
                self.analyser(self.snapshot, node_group, {n: True for n in names})
            return int(res.node_count[0]), pods
        finally:
            self.limiter.end_estimation()

    def _estimate_on_cluster(self, pegs, node_template):
        enc2 = encode_cluster_estimate(self.lanes, pegs, self.snapshot.existing, node_template, self.limiter.device_max_nodes())
        try:
            rc, out = self.engine_ctx.estimate_on_cluster(enc2.pegs, enc2.groups, len(self.snapshot.existing), self.limiter.device_max_nodes(),
                                                          self.snapshot.last_index, enc2.rules, enc2.port_block)
        finally:
            enc2.close()
        if rc != 0:
            return None
        pods: List[Pod] = []
        for pg_id, n in zip(out["order"], out["placed"]):
            pods.extend(pegs[int(pg_id)].pods[:int(n)])
        self.snapshot.last_index = out["last_index_out"]
        self.limiter.nodes = out["limiter_nodes"]
        return out["node_count"], pods


class PrefetchShared:
    """What the shim's NodeGroupListProcessor wrapper and its estimators share (gpubinpacking.shared in integration/go): the
    prefetch cache and the identities the keys are made of.  Keys: a PEG is its exemplar pod object (the orchestrator passes the
    same *PodEquivalenceGroup values to every Estimate of a loop), a node group is (Id(), identity of its template NodeInfo).
    Every group of a batch is estimated from the lastIndex the snapshot's runner has when the batch is filled — the estimators
    of a prefetched loop do not thread lastIndex from one group into the next (the per-call path does)."""

    def __init__(self, engine_ctx: Context, limiter: "ThresholdBasedEstimationLimiter", max_nodes_total: int = 0, fastpath: bool = False,
                 lanes: Sequence[str] = ("cpu", "memory"), chain_last_index: bool = True):
        """chain_last_index (default since round 5, casim_options.chain_last_index): the batch estimates the groups in the order the processor
        handed them over, each from the lastIndex its predecessor left — what the orchestrator's loop does on one snapshot
        (plugin_runner.go:138).  A lookup then comes with the runner's CURRENT lastIndex: it hits only while the Estimate() calls arrive in
        the batch's order, and a hit moves the runner on (BinpackingNodeEstimator.estimate).  False: every group from the loop's lastIndex
        (rounds 3-4; hits never moved the runner)."""
        from .engine import PrefetchCache
        self.chain = bool(chain_last_index)
        self.ctx, self.limiter, self.max_nodes_total, self.fastpath, self.lanes = engine_ctx, limiter, max_nodes_total, fastpath, lanes
        self.cache = PrefetchCache(engine_ctx)
        self.loop_last_index = 0
        self.enc = None          # the loop's tables stay until the next fill (the Go shim keeps its session: Shared.sess): rechain reads them
        self.batch_keys, self.peg_keys, self.rechains, self.last_miss = [], [], 0, 0

    MAX_RECHAINS = 4             # batches one loop may spend on re-chaining (gpubinpacking/prefetch.go: maxRechains)

    def _drop_loop_tables(self):
        if self.enc is not None:
            self.enc.close()
            self.enc = None
        self.batch_keys, self.peg_keys, self.rechains = [], [], 0

    def close(self):
        self._drop_loop_tables()
        self.cache.close()

    @staticmethod
    def peg_key(pg: PodEquivalenceGroup) -> int:
        return id(pg.pods[0]) & 0xFFFFFFFFFFFFFFFF

    @staticmethod
    def group_key(node_group, template: NodeInfo) -> int:
        return hash((node_group.id(), id(template))) & 0xFFFFFFFFFFFFFFFF

    def fill(self, pegs: List[PodEquivalenceGroup], node_groups, node_infos, snapshot: ClusterSnapshotView, similar_node_groups=None):
        """ONE casim_estimate_batch over every PEG and every candidate group; SchedulablePodGroups on the device."""
        similar_node_groups = similar_node_groups or {}
        self._drop_loop_tables()
        self.loop_last_index = snapshot.last_index
        enc = Encoder(lanes=self.lanes)
        for pg in pegs:
            enc.add_peg(pg)
        for info in snapshot.existing:
            for p in info.pods:
                enc.add_existing_pod(p, info.node.labels)
        gkeys = []
        for ng in node_groups:
            context = EstimationContext(self.max_nodes_total, list(similar_node_groups.get(ng.id(), [])), len(snapshot.existing))
            self.limiter.start_estimation(pegs, ng, context)   # the reference's own limiter decides max_nodes (threshold_based_limiter.go:34-43)
            enc.add_group(node_infos[ng.id()], max_nodes=self.limiter.device_max_nodes(), existing_nodes=len(snapshot.existing),
                          last_index=snapshot.last_index, pegs=None)
            self.limiter.end_estimation()
            gkeys.append(self.group_key(ng, node_infos[ng.id()]))
        enc.finalize()
        pkeys = [self.peg_key(pg) for pg in pegs]
        try:
            self.cache.fill(enc.pegs, enc.groups, gkeys, pkeys, self.fastpath, chain_last_index=self.chain)
        except Exception:
            enc.close()
            raise
        self.enc, self.batch_keys, self.peg_keys = enc, gkeys, pkeys

    def rechain(self, node_group, node_template, last_index: int) -> bool:
        """gpubinpacking/prefetch.go rechain: the REST of the loop's batch, from this group on, estimated again as one chained batch that starts
        from `last_index` (rows of the loop's group table through casim_enc_group_rows, the first group's last_index overridden).  False:
        nothing to re-chain (unchained mode, unknown group, the batch's last group, this loop's budget spent)."""
        import numpy as np
        key = self.group_key(node_group, node_template)
        if not self.chain or self.enc is None or self.rechains >= self.MAX_RECHAINS or key not in self.batch_keys:
            return False
        pos = self.batch_keys.index(key)
        n = len(self.batch_keys) - pos
        if n < 2:
            return False
        self.rechains += 1
        rows = self.enc.group_rows(range(pos, pos + n))
        li = np.full(n, int(last_index), np.int32)   # (the chain reads the FIRST group's entry)
        rows.last_index = li.ctypes.data_as(_abi.i32p)
        self.cache.fill(self.enc.pegs, rows, self.batch_keys[pos:], self.peg_keys, self.fastpath, chain_last_index=True)
        return True

    def lookup(self, pegs, node_template, node_group, max_nodes: int, existing_nodes: int, runner_last_index: Optional[int] = None):
        """runner_last_index: the snapshot runner's lastIndex at the time of THIS Estimate() (chain mode); None = the loop's (unchained mode)"""
        li = self.loop_last_index if (runner_last_index is None or not self.chain) else runner_last_index
        hit, out = self.cache.lookup(self.group_key(node_group, node_template), [self.peg_key(pg) for pg in pegs], max_nodes, existing_nodes, li)
        self.last_miss = 0 if hit else int(out["miss_reason"])
        return out if hit else None


class PrefetchNodeGroupListProcessor:
    """gpubinpacking.WrapNodeGroupListProcessor: processors.NodeGroupListProcessor.Process sees every candidate node group, every
    template NodeInfo and the pending pods before ScaleUp's two loops run (CA/core/scaleup/orchestrator/orchestrator.go:121-123);
    the wrapper lets the inner processor answer, then fills the shared cache with one batch."""

    def __init__(self, inner, shared: PrefetchShared, build_pod_groups: Callable):
        self.inner, self.shared, self.build_pod_groups = inner, shared, build_pod_groups
        self.pegs: List[PodEquivalenceGroup] = []

    def process(self, snapshot: ClusterSnapshotView, node_groups, node_infos, unschedulable_pods):
        if self.inner is not None:
            node_groups, node_infos = self.inner.process(snapshot, node_groups, node_infos, unschedulable_pods)
        self.pegs = self.build_pod_groups(unschedulable_pods)   # equivalence.BuildPodGroups: the groups the orchestrator will pass on
        self.shared.fill(self.pegs, node_groups, node_infos, snapshot)
        return node_groups, node_infos


def new_estimator_builder(name: str, limiter, orderer=None, analyser=None, fastpath: bool = False, engine_ctx: Optional[Context] = None,
                          prefetch: Optional[PrefetchShared] = None):
    """NewEstimatorBuilder  estimator.go:62-77: returns func(clusterSnapshot, context) Estimator."""
    if name != GPU_BINPACKING_ESTIMATOR_NAME:
        raise ValueError(f"unknown estimator: {name} (this package provides only {GPU_BINPACKING_ESTIMATOR_NAME})")
    if engine_ctx is None:
        engine_ctx = Context(0)

    def builder(cluster_snapshot: ClusterSnapshotView, context: EstimationContext):
        return BinpackingNodeEstimator(engine_ctx, cluster_snapshot, limiter, orderer, context, analyser, fastpath, prefetch=prefetch)
    return builder
