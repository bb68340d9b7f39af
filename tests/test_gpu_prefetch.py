"""-m gpu: the estimator shim's two modes through the C ABI on the MI355X.
Prefetch mode: the NodeGroupListProcessor wrapper fills ONE batch (casim_prefetch_fill), every Estimate() of the orchestrator's
loop is a lookup; per-call mode: one casim_estimate_batch per Estimate().  Both against the oracle, group by group, and the
cache-key protocol's miss paths (another PEG subset, another limiter answer, a group the batch did not hold)."""
import numpy as np
import pytest

import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd import estimator as est
from kubernetes_autoscaler_amd.objects import PodEquivalenceGroup
from oracle_driver import OracleScenario

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = kaa.Context(0)
    yield c
    c.close()


def _oracle_per_group(w, last_index=0):
    """SchedulablePodGroups + Estimate, group by group, every group from the same entry lastIndex."""
    out = []
    for g in w.groups:
        s = OracleScenario(lanes=w.lanes)
        t = s.node(g.template)
        ids = [i for i, pg in enumerate(w.pegs) if s.check_predicates(t, pg.pods[0])[0]]
        e = s.estimate(t, [w.pegs[i] for i in ids], max_nodes=g.max_nodes, last_index=last_index)
        out.append((ids, e))
        s.close()
    return out


def _setup(w):
    ngs = [est.NodeGroup(f"ng{i}", max_size_=g.max_nodes, target_size_=0) for i, g in enumerate(w.groups)]
    infos = {ng.id(): g.template for ng, g in zip(ngs, w.groups)}
    limiter = est.ThresholdBasedEstimationLimiter([est.SngCapacityThreshold(), est.ClusterCapacityThreshold()])
    return ngs, infos, limiter


def test_prefetch_mode_answers_every_estimate_of_the_loop_from_one_batch(ctx):
    w = workloads.config_c2(n_groups=12, n_pegs=80, pods_per_peg=10, cap=20)
    ngs, infos, limiter = _setup(w)
    shared = est.PrefetchShared(ctx, limiter)
    pods = [p for pg in w.pegs for p in pg.pods]
    proc = est.PrefetchNodeGroupListProcessor(None, shared, lambda _pods: w.pegs)   # (BuildPodGroups has its own tests: test_groups_*)
    snapshot = est.ClusterSnapshotView()
    proc.process(snapshot, ngs, infos, pods)
    builder = est.new_estimator_builder(est.GPU_BINPACKING_ESTIMATOR_NAME, limiter, engine_ctx=ctx, prefetch=shared)
    want = _oracle_per_group(w)
    for ng, g, (ids, e) in zip(ngs, w.groups, want):
        # the orchestrator's loop: SchedulablePodGroups filtered the PEGs, a fresh estimator per group, ONE Estimate
        estimator = builder(snapshot, est.EstimationContext(0, [], 0))
        n, got_pods = estimator.estimate([w.pegs[i] for i in ids], infos[ng.id()], ng)
        assert (n, len(got_pods)) == (e.node_count, e.pods_scheduled), ng.id()
    st = shared.cache.stats()
    assert st["fills"] == 1 and st["groups_cached"] == len(ngs) and st["hits"] == len(ngs) and st["miss_pegs"] == st["miss_limits"] == st["miss_group"] == 0
    shared.close()


def test_cache_misses_take_the_per_call_path_and_stay_exact(ctx):
    w = workloads.config_c2(n_groups=6, n_pegs=60, pods_per_peg=8, cap=15)
    ngs, infos, limiter = _setup(w)
    shared = est.PrefetchShared(ctx, limiter)
    snapshot = est.ClusterSnapshotView()
    shared.fill(w.pegs, ngs[:5], infos, snapshot)            # group 5 is not part of the batch
    builder = est.new_estimator_builder(est.GPU_BINPACKING_ESTIMATOR_NAME, limiter, engine_ctx=ctx, prefetch=shared)
    want = _oracle_per_group(w)
    # (a) a PEG subset the batch did not answer: the orchestrator dropped one schedulable PEG
    ids, _ = want[0]
    sub = ids[:-1]
    s = OracleScenario(lanes=w.lanes); t = s.node(w.groups[0].template)
    e = s.estimate(t, [w.pegs[i] for i in sub], max_nodes=w.groups[0].max_nodes, last_index=0); s.close()
    n, got = builder(snapshot, est.EstimationContext(0, [], 0)).estimate([w.pegs[i] for i in sub], infos["ng0"], ngs[0])
    assert (n, len(got)) == (e.node_count, e.pods_scheduled)
    # (b) the limiter answers differently (the node group's max size changed since the batch)
    ids, _ = want[1]
    smaller = est.NodeGroup("ng1", max_size_=max(1, w.groups[1].max_nodes // 2), target_size_=0)
    s = OracleScenario(lanes=w.lanes); t = s.node(w.groups[1].template)
    e = s.estimate(t, [w.pegs[i] for i in ids], max_nodes=smaller.max_size_, last_index=0); s.close()
    n, got = builder(snapshot, est.EstimationContext(0, [], 0)).estimate([w.pegs[i] for i in ids], infos["ng1"], smaller)
    assert (n, len(got)) == (e.node_count, e.pods_scheduled)
    # (c) a group the batch did not hold
    ids, e = want[5]
    n, got = builder(snapshot, est.EstimationContext(0, [], 0)).estimate([w.pegs[i] for i in ids], infos["ng5"], ngs[5])
    assert (n, len(got)) == (e.node_count, e.pods_scheduled)
    # (d) and a plain hit in between
    ids, e = want[2]
    n, got = builder(snapshot, est.EstimationContext(0, [], 0)).estimate([w.pegs[i] for i in ids], infos["ng2"], ngs[2])
    assert (n, len(got)) == (e.node_count, e.pods_scheduled)
    st = shared.cache.stats()
    assert (st["miss_pegs"], st["miss_limits"], st["miss_group"], st["hits"]) == (1, 1, 1, 1), st
    shared.close()


def test_lookup_returns_positions_in_the_callers_list(ctx):
    """order_out indexes the list Estimate() received, whatever ids the PEGs had in the batch"""
    w = workloads.config_c2(n_groups=4, n_pegs=40, pods_per_peg=5, cap=10)
    ngs, infos, limiter = _setup(w)
    shared = est.PrefetchShared(ctx, limiter)
    snapshot = est.ClusterSnapshotView()
    shared.fill(w.pegs, ngs, infos, snapshot)
    for ng, g, (ids, e) in zip(ngs, w.groups, _oracle_per_group(w)):
        hit = shared.lookup([w.pegs[i] for i in ids], infos[ng.id()], ng, g.max_nodes, 0)
        assert hit is not None and hit["n_pegs"] == len(ids)
        assert sorted(int(k) for k in hit["order"]) == list(range(len(ids)))
        assert [int(k) for k in hit["order"]] == [int(k) for k in e.order] and [int(x) for x in hit["placed"]] == [int(x) for x in e.placed]
    shared.close()
