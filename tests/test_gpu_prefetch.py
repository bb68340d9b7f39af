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


def _oracle_per_group(w, last_index=0, chain=False):
    """SchedulablePodGroups + Estimate, group by group, every group from the same entry lastIndex — or (chain) from the lastIndex its
    predecessor left, as the orchestrator's loop runs them on one snapshot (plugin_runner.go:138)."""
    out = []
    for g in w.groups:
        s = OracleScenario(lanes=w.lanes)
        t = s.node(g.template)
        ids = [i for i, pg in enumerate(w.pegs) if s.check_predicates(t, pg.pods[0])[0]]
        e = s.estimate(t, [w.pegs[i] for i in ids], max_nodes=g.max_nodes, last_index=last_index)
        if chain:
            last_index = e.last_index_out
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
    want = _oracle_per_group(w, chain=True)   # (the batch chains lastIndex through the groups since round 5: casim_options.chain_last_index)
    for ng, g, (ids, e) in zip(ngs, w.groups, want):
        # the orchestrator's loop: SchedulablePodGroups filtered the PEGs, a fresh estimator per group, ONE Estimate
        estimator = builder(snapshot, est.EstimationContext(0, [], 0))
        n, got_pods = estimator.estimate([w.pegs[i] for i in ids], infos[ng.id()], ng)
        assert (n, len(got_pods)) == (e.node_count, e.pods_scheduled), ng.id()
        assert snapshot.last_index == e.last_index_out                      # a hit moves the runner on, like the Estimate it stands for
    st = shared.cache.stats()
    assert st["fills"] == 1 and st["groups_cached"] == len(ngs) and st["hits"] == len(ngs) and st["miss_pegs"] == st["miss_limits"] == st["miss_group"] == 0
    shared.close()


def test_cache_misses_take_the_per_call_path_and_stay_exact(ctx):
    w = workloads.config_c2(n_groups=6, n_pegs=60, pods_per_peg=8, cap=15)
    ngs, infos, limiter = _setup(w)
    shared = est.PrefetchShared(ctx, limiter, chain_last_index=False)   # (the key protocol of the unchained batch: every group from the loop's lastIndex)
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
    shared = est.PrefetchShared(ctx, limiter, chain_last_index=False)
    snapshot = est.ClusterSnapshotView()
    shared.fill(w.pegs, ngs, infos, snapshot)
    for ng, g, (ids, e) in zip(ngs, w.groups, _oracle_per_group(w)):
        hit = shared.lookup([w.pegs[i] for i in ids], infos[ng.id()], ng, g.max_nodes, 0)
        assert hit is not None and hit["n_pegs"] == len(ids)
        assert sorted(int(k) for k in hit["order"]) == list(range(len(ids)))
        assert [int(k) for k in hit["order"]] == [int(k) for k in e.order] and [int(x) for x in hit["placed"]] == [int(x) for x in e.placed]
    shared.close()


def test_a_chained_batch_hits_only_while_the_calls_arrive_in_its_order(ctx):
    """casim_options.chain_last_index in the prefetch cache: group i was estimated from the lastIndex group i - 1 left, so its entry answers
    an Estimate() that comes with exactly that lastIndex.  In order: every call hits and equals the sequential oracle loop.  Out of order
    (the orchestrator skipped a group): the call comes with another lastIndex — a miss on the limits unless the two happen to agree — and
    whichever path answers, the result is what the reference computes for that call with the runner's real lastIndex."""
    w = workloads.config_c2(n_groups=8, n_pegs=70, pods_per_peg=9, cap=18)
    ngs, infos, limiter = _setup(w)
    want = _oracle_per_group(w, chain=True)
    assert len({e.last_index_out for _, e in want}) > 1            # (the chain hands different values on)
    shared = est.PrefetchShared(ctx, limiter)
    snapshot = est.ClusterSnapshotView()
    shared.fill(w.pegs, ngs, infos, snapshot)
    builder = est.new_estimator_builder(est.GPU_BINPACKING_ESTIMATOR_NAME, limiter, engine_ctx=ctx, prefetch=shared)
    for ng, (ids, e) in zip(ngs, want):
        n, got = builder(snapshot, est.EstimationContext(0, [], 0)).estimate([w.pegs[i] for i in ids], infos[ng.id()], ng)
        assert (n, len(got), snapshot.last_index) == (e.node_count, e.pods_scheduled, e.last_index_out), ng.id()
    st = shared.cache.stats()
    assert st["hits"] == len(ngs) and st["miss_limits"] == 0, st
    # a lookup that comes with a lastIndex the batch did not use for the group misses on the limits
    ids = want[3][0]
    assert shared.lookup([w.pegs[i] for i in ids], infos[ngs[3].id()], ngs[3], w.groups[3].max_nodes, 0, runner_last_index=want[2][1].last_index_out + 1) is None
    assert shared.cache.stats()["miss_limits"] == 1
    # the same loop with groups 1 and 4 skipped: every call still equals the oracle's Estimate from the runner's real lastIndex
    snapshot = est.ClusterSnapshotView()
    shared.fill(w.pegs, ngs, infos, snapshot)
    li = 0
    for k in (0, 2, 3, 5, 6, 7):
        ids = want[k][0]
        s = OracleScenario(lanes=w.lanes); t = s.node(w.groups[k].template)
        e = s.estimate(t, [w.pegs[i] for i in ids], max_nodes=w.groups[k].max_nodes, last_index=li); s.close()
        n, got = builder(snapshot, est.EstimationContext(0, [], 0)).estimate([w.pegs[i] for i in ids], infos[ngs[k].id()], ngs[k])
        assert (n, len(got), snapshot.last_index) == (e.node_count, e.pods_scheduled, e.last_index_out), k
        li = e.last_index_out
    shared.close()
