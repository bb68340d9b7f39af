"""-m gpu: the entry points added in round 2 on a real MI355X, through the C ABI, against the oracle / the single-device
path: batches of simulations (PEG ranges, per-simulation expander, validity mask), SchedulingError codes, the multi-device
context (one process, several contexts, RCCL or host reduce), the timed call, the all-or-nothing rule."""
import numpy as np
import pytest

import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import estimate_batch_timed
from kubernetes_autoscaler_amd.tables import TableSet
from harness import (GroupSpec, Scenario, assert_matches_oracle, encode, encode_batch, run_emu_tables, run_gpu_tables, run_oracle)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = kaa.Context(0)
    yield c
    c.close()


def _scenario(seed, groups=5):
    w = workloads.fuzz(seed, max_groups=groups, max_pegs=14)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)


def _oracle_of(scs, bases):
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc)])
    return want


def test_batched_simulations_match_the_oracle(ctx):
    for seed in range(40):
        scs = [_scenario(1000 * seed + k) for k in range(1 + seed % 6)]
        enc, ts, bases = encode_batch(scs)
        res, exp = run_gpu_tables(ts, ctx, kinds=[_abi.EXPANDER_LEAST_NODES, _abi.EXPANDER_LEAST_WASTE])
        assert_matches_oracle(res, _oracle_of(scs, bases), f"batch {seed}")
        _, emu = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES, _abi.EXPANDER_LEAST_WASTE])
        assert list(exp["best"]) == list(emu["best"]) and list(exp["n_best"]) == list(emu["n_best"])
        assert exp["keys"].tolist() == emu["keys"].tolist() and list(exp["packed"]) == list(emu["packed"])
        enc.close()


def test_large_tiled_batch_on_the_device(ctx):
    """4096 simulations in one launch (the bench's shape, small simulations): multi-block scan, wave-per-group orderer."""
    scs = [_scenario(31 + k, groups=4) for k in range(8)]
    enc, ts, bases = encode_batch(scs)
    big = ts.tile(512)
    res, exp = run_gpu_tables(big, ctx, kinds=[_abi.EXPANDER_LEAST_NODES])
    base, bexp = run_gpu_tables(ts, ctx, kinds=[_abi.EXPANDER_LEAST_NODES])
    assert_matches_oracle(base, _oracle_of(scs, bases), "tile base")
    ng, nnz = ts.n_groups, int(base.offsets[-1])
    for k in (0, 1, 255, 511):
        assert list(res.node_count[k * ng:(k + 1) * ng]) == list(base.node_count)
        assert list(res.pods_scheduled[k * ng:(k + 1) * ng]) == list(base.pods_scheduled)
        assert list(res.placed[k * nnz:(k + 1) * nnz]) == list(base.placed)
        assert list(exp["packed"][k * ts.n_sims:(k + 1) * ts.n_sims]) == list(bexp["packed"])
    enc.close()


def test_batch_with_lists_beyond_the_one_wave_networks(ctx):
    """A batch (>= 2048 groups: one wave per group in the orderer, LDS sized for lists of <= 256 PEGs) whose simulations
    mix short lists with lists of 300 and 700 PEGs: the long ones sort in the HBM slab of the same launch; the packer
    walks them through several record chunks, most of it behind a dry limiter."""
    from harness import mixed_list_simulations
    scs = mixed_list_simulations()
    enc, ts, bases = encode_batch(scs)
    want = _oracle_of(scs, bases)
    base, _ = run_gpu_tables(ts, ctx)
    assert_matches_oracle(base, want, "mixed lists, small launch")
    big = ts.tile(256)   # 3072 groups
    res, _ = run_gpu_tables(big, ctx)
    ng, nnz = ts.n_groups, int(base.offsets[-1])
    for k in (0, 1, 128, 255):
        assert list(res.node_count[k * ng:(k + 1) * ng]) == list(base.node_count)
        assert list(res.pods_scheduled[k * ng:(k + 1) * ng]) == list(base.pods_scheduled)
        assert list(res.last_index_out[k * ng:(k + 1) * ng]) == list(base.last_index_out)
        assert list(res.placed[k * nnz:(k + 1) * nnz]) == list(base.placed)
        assert [o - k * ts.n_pegs for o in res.order[k * nnz:(k + 1) * nnz]] == list(base.order)
    enc.close()


def test_streamed_batch_equals_one_problem():
    """casim_options.n_streams: the simulations of a batch spread over 1 / 3 / 4 / 16 internal streams of ONE context give the results
    of ONE unstreamed problem — group arrays, CSR offsets, PEG ids in the whole batch's numbering, per-simulation expander winners and
    the packed keys written into slices of one device tensor (tests/tools/streamed_batch_check.py, its own process: torch
    first, so that the tensor and libcasim share one HIP runtime)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "streamed_batch_check.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["streams_checked"] == [1, 3, 4, 16] and out["simulations"] == 11


def test_reason_codes_on_the_device(ctx):
    from test_reasons_emu import emu_reasons, oracle_codes
    for seed in range(60):
        w = workloads.fuzz(9000 + seed, max_groups=5, max_pegs=14)
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], existing=w.existing,
                      device_csr=True)
        enc = encode(sc)
        got = ctx.feasibility_reasons(enc.pegs, enc.groups, enc.port_block)
        assert got.tolist() == emu_reasons(enc).tolist(), seed
        want = oracle_codes(sc)
        for j in range(len(w.pegs)):
            if enc.pegs.flags[j] & _abi.PEG_UNSUPPORTED:
                continue
            assert list(got[:, j]) == list(want[:, j]), (seed, j)
        enc.close()


def test_schedulable_pod_groups_with_errors(ctx):
    from kubernetes_autoscaler_amd.equivalence import build_pod_groups, schedulable_pod_groups_with_errors
    from kubernetes_autoscaler_amd.objects import NodeInfo, Taint, build_test_node, build_test_pod
    small, large = build_test_pod("small", 100, 0), build_test_pod("large", 1500, 0)
    tainted = build_test_node("tainted", 4000, 2000000)
    tainted.taints = [Taint("dedicated", "x", "NoSchedule")]
    ok, errors = schedulable_pod_groups_with_errors(ctx, build_pod_groups([small, large]), {"n1000": NodeInfo(build_test_node("n1000", 1000, 2000000)),
                                                                                             "tainted": NodeInfo(tainted)})
    assert ok.tolist() == [[True, False], [False, False]]
    assert errors["n1000"][1].failing_predicate_name == "NodeResourcesFit" and errors["n1000"][1].failing_predicate_reasons == ["Insufficient cpu"]
    assert errors["tainted"][0].failing_predicate_name == "TaintToleration"


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_device_context(ctx, devices):
    """One process, several contexts (a 1-GPU box: the same device named several times), keys reduced on the host."""
    with kaa.MultiContext(devices, use_rccl=False) as m:
        for seed in range(12):
            scs = [_scenario(3100 + 10 * seed + k, groups=7) for k in range(1 + seed % 4)]
            enc, ts, bases = encode_batch(scs)
            pegs, groups = ts.structs()
            for kinds in ([_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_LEAST_WASTE, _abi.EXPANDER_MOST_PODS]):
                got, exp = m.estimate_batch(pegs, groups, kinds=kinds)
                assert_matches_oracle(got, _oracle_of(scs, bases), f"multi {devices} seed {seed}")
                _, one = run_gpu_tables(ts, ctx, kinds=kinds)
                assert list(exp["best"]) == list(one["best"]) and list(exp["packed"]) == list(one["packed"])
            info = m.info()
            assert info["devices"] == len(devices) and sum(info["groups_per_device"]) == ts.n_groups and not info["rccl"]
            enc.close()


def test_multi_device_context_reduces_through_rccl():
    """ncclCommInitAll + ncclAllReduce(min, int64) inside libcasim, on the hardware, over every visible device (tests/tools/
    mctx_rccl_check.py, its own process: HIP runtime and RCCL must come from one place, see there)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "mctx_rccl_check.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])    # (RCCL prints its banner on stdout too)
    assert out["rccl"] and out["last_reduce_by_rccl"] and out["batches"] == 24


def test_timed_call_reports_phases_and_the_same_results(ctx):
    w = workloads.config_c2(n_groups=8, n_pegs=60, pods_per_peg=6, cap=10)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)
    enc, ts, bases = encode_batch([sc])
    pegs, groups = ts.structs()
    arrs, ph, exp = estimate_batch_timed(ctx, pegs, groups, kinds=[_abi.EXPANDER_LEAST_NODES])
    res, one = run_gpu_tables(ts, ctx, kinds=[_abi.EXPANDER_LEAST_NODES])
    assert list(arrs["node_count"][:ts.n_groups]) == list(res.node_count) and list(arrs["pods_scheduled"][:ts.n_groups]) == list(res.pods_scheduled)
    assert int(exp["best"][0]) == int(one["best"][0])
    assert all(ph[k] >= 0 for k in ph) and ph["wall_ms"] >= ph["pack_ms"] > 0
    assert_matches_oracle(res, _oracle_of([sc], bases), "timed")
    enc.close()


def test_all_or_nothing_drops_partial_options_before_the_expander(ctx):
    """orchestrator.go:1057-1063 / :1079: with all-or-nothing a partial option never reaches ExpanderStrategy.BestOption."""
    from kubernetes_autoscaler_amd import estimator as est
    from kubernetes_autoscaler_amd import expander
    from kubernetes_autoscaler_amd.objects import GiB, NodeInfo, Pod, PodEquivalenceGroup, build_test_node
    from kubernetes_autoscaler_amd.scaleup import ScaleUpSimulator
    # small nodes, capped at 2: least-waste likes it but it cannot take all pods; big nodes take everything
    small = est.NodeGroup("small", 2, 0)
    big = est.NodeGroup("big", 10, 0)
    infos = {"small": NodeInfo(build_test_node("small-t", 1000, 1 * GiB)), "big": NodeInfo(build_test_node("big-t", 16000, 64 * GiB))}
    pods = [Pod(name=f"p{i}", requests={"cpu": 500, "memory": 256 * 1024 * 1024}) for i in range(8)]
    peg = PodEquivalenceGroup(pods=pods)
    limiter = est.ThresholdBasedEstimationLimiter([est.SngCapacityThreshold()])
    for chain in ([expander.LEAST_WASTE], [expander.LEAST_NODES], [expander.MOST_PODS]):
        sim = ScaleUpSimulator(ctx, limiter, expander.ChainStrategy(chain))
        free = sim.prepare_scale_up([peg], [small, big], infos, est.ClusterSnapshotView())
        aon = sim.prepare_scale_up([peg], [small, big], infos, est.ClusterSnapshotView(), all_or_nothing=True)
        assert {o.node_group.id() for o in free.options} == {"small", "big"}
        assert [o.node_group.id() for o in aon.options] == ["big"] and aon.best.node_group.id() == "big" and aon.n_equally_good == 1
    # similar node groups widen the SngCapacityThreshold (orchestrator.go:409-412)
    sim = ScaleUpSimulator(ctx, limiter, expander.ChainStrategy([expander.MOST_PODS]))
    alone = sim.prepare_scale_up([peg], [small], infos, est.ClusterSnapshotView())
    twin = est.NodeGroup("small-b", 3, 1)
    wider = sim.prepare_scale_up([peg], [small], infos, est.ClusterSnapshotView(), similar_node_groups={"small": [twin]})
    assert alone.options[0].node_count == 2 and wider.options[0].node_count == 4


def test_node_pods_and_the_analyser_hook(ctx):
    """casim_results.node_pods on the device == the oracle's per-node counts; BinpackingNodeEstimator hands the analyser the
    names of the added nodes that hold a pod (binpacking_estimator.go:157-159)."""
    from kubernetes_autoscaler_amd import estimator as est
    from kubernetes_autoscaler_amd.engine import Problem
    from test_node_pods_emu import _oracle_node_pods
    for seed in range(40):
        w = workloads.fuzz(1000 + seed)
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing)
        enc = encode(sc)
        for generic in (False, True):
            with Problem(ctx, enc.pegs, enc.groups, False, generic, node_pods=True) as p:
                p.run()
                res = p.fetch()
            for i, e in enumerate(_oracle_node_pods(sc)):
                if int(res.status[i]) != 0:
                    continue
                a, b = int(res.node_pods_offsets[i]), int(res.node_pods_offsets[i + 1])
                assert list(res.node_pods[a:b]) == list(e.node_pods), (seed, i, generic)
        enc.close()
    seen = {}
    w = workloads.config_c0()
    limiter = est.ThresholdBasedEstimationLimiter([est.StaticThreshold(10)])
    e = est.BinpackingNodeEstimator(ctx, est.ClusterSnapshotView(), limiter, estimation_analyser_func=lambda snap, ng, nodes: seen.update(nodes))
    n, pods = e.estimate(w.pegs, w.groups[0].template, est.NodeGroup("c0", 10, 0))
    assert n == len(seen) > 0 and all(k.startswith("c0-template-e-") for k in seen)
    # fastpath: the extrapolated nodes reach the analyser as "<lastNodeName>-fake-<j>" (binpacking_estimator.go:311-321), so that
    # len(newNodesWithPods) == the returned node count as in the reference.  One PEG of many identical pods: tryFastPath
    # simulates ONE node and books the rest by arithmetic.
    import copy
    from kubernetes_autoscaler_amd.objects import PodEquivalenceGroup
    seen.clear()
    pod = w.pegs[0].pods[0]
    one = [PodEquivalenceGroup(pods=[copy.copy(pod) for _ in range(60)])]
    limiter = est.ThresholdBasedEstimationLimiter([est.StaticThreshold(100)])
    e = est.BinpackingNodeEstimator(ctx, est.ClusterSnapshotView(), limiter, estimation_analyser_func=lambda snap, ng, nodes: seen.update(nodes),
                                    fastpath_binpacking_enabled=True)
    n, pods = e.estimate(one, w.groups[0].template, est.NodeGroup("c0", 100, 0))
    assert n == len(seen) > 1 and len(pods) == 60, (n, seen)
    fake = sorted(k for k in seen if "-fake-" in k)
    assert len(fake) == n - 1 and all(k.startswith("c0-template-e-0-fake-") for k in fake), seen


def test_resident_cluster_iteration(ctx):
    """casim_cluster_*: filter-out-schedulable committed into the resident node table, a reverted pass, the planner's removal loop
    on the committed image — vs the oracle threading one snapshot through the same sequence."""
    from harness import resident_iteration
    total = 0
    for seed in range(30):
        w = workloads.fuzz_pending(300 + seed)
        w.hints = None
        out = resident_iteration(lambda classes, nodes: kaa.ResidentCluster(ctx, classes, nodes), w)
        assert out["stats"]["full_uploads"] == 1
        total += out["scheduled"]
    assert total > 0
    w = workloads.pending_scale(2000, 20000, 32, 4)      # LDS-resident state, long runs
    out = resident_iteration(lambda classes, nodes: kaa.ResidentCluster(ctx, classes, nodes), w, n_candidates=50)
    assert out["scheduled"] > 0


def test_required_pod_affinity_on_the_device(ctx):
    """a18 complete: required pod affinity as domain rules of kind 2 — TrySchedulePods, the removal loop and Estimate on the
    snapshot on the MI355X vs the oracle (same families as tests/test_pod_affinity_emu.py)."""
    from kubernetes_autoscaler_amd.workloads import add_random_pod_affinity
    from harness import (RemovalCase, SchedCase, assert_cluster_estimate_matches, assert_removal_matches, assert_sched_matches,
                         cluster_estimate_gpu, removal_device, removal_oracle, sched_gpu, sched_oracle)
    placed = 0
    for seed in range(120):
        w = workloads.fuzz_pending_domains(7000 + seed)
        add_random_pod_affinity(seed, w.pods + [p for info in w.nodes for p in info.pods], frac=0.6)
        case = SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)
        got = sched_gpu(case, ctx)
        assert_sched_matches(got, sched_oracle(case), w.name)
        placed += int(got[3])
    assert placed > 0
    for seed in range(80):
        w = workloads.fuzz_removals_domains(7000 + seed)
        add_random_pod_affinity(seed, [p for info in w.nodes for p in info.pods], frac=0.5, apps=("app0", "app1", "app2"))
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=w.max_removable, last_index=w.last_index)
        assert_removal_matches(removal_device(case, ctx), removal_oracle(case), w.name)
    for seed in range(120):
        w = workloads.fuzz_estimate_domains(7000 + seed)
        add_random_pod_affinity(seed, [pg.pods[0] for pg in w.pegs] + [p for info in w.existing for p in info.pods] + list(w.groups[0].template.pods),
                                frac=0.6, apps=("app0", "app1", "app2"))
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes)
        got = cluster_estimate_gpu(sc, ctx)
        if got[0] == 1:
            continue
        est, ids = run_oracle(sc)[0]
        assert_cluster_estimate_matches(got, est, ids, w.name)


def test_estimator_mirror_runs_affinity_groups_on_the_snapshot(ctx):
    """BinpackingNodeEstimator: a PEG with required pod affinity leaves the template-mode batch (CASIM_NG_UNSUPPORTED) and is
    estimated by casim_estimate_on_cluster — same answer as the oracle's Estimate."""
    from kubernetes_autoscaler_amd import estimator as est
    from kubernetes_autoscaler_amd.objects import LABEL_ZONE, GiB, MiB, NodeInfo, Pod, PodAffinityTerm, PodEquivalenceGroup
    from kubernetes_autoscaler_amd.workloads import _node
    tmpl = NodeInfo(_node("t", 2000, 8 * GiB, 110, {LABEL_ZONE: "z0"}))
    web = Pod(name="web", labels={"app": "web"}, requests={"cpu": 500, "memory": 256 * MiB}, affinity=[PodAffinityTerm(LABEL_ZONE, match_labels={"app": "web"})])
    far = Pod(name="far", labels={"app": "far"}, requests={"cpu": 100, "memory": 64 * MiB}, affinity=[PodAffinityTerm(LABEL_ZONE, match_labels={"app": "absent"})])
    pegs = [PodEquivalenceGroup([web] * 9), PodEquivalenceGroup([far] * 3)]
    e = est.BinpackingNodeEstimator(ctx, est.ClusterSnapshotView(), est.ThresholdBasedEstimationLimiter([est.StaticThreshold(10)]))
    n, pods = e.estimate(pegs, tmpl, est.NodeGroup("ng", 10, 0))
    sc = Scenario(pegs=pegs, groups=[GroupSpec(tmpl, 10, 0, None)])
    want, _ = run_oracle(sc)[0]
    assert (n, len(pods)) == (want.node_count, want.pods_scheduled) == (3, 9)     # the zone-affine series fills 3 nodes, "far" finds no partner
