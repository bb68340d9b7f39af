"""-m gpu: the HIP path on a real MI355X, through the C ABI, against the oracle — bit-exact.
Same scenarios as the emulated CPU tests plus the BASELINE.json configs at full size."""
import json
import os

import numpy as np
import pytest

import kubernetes_autoscaler_amd as kaa
from harness import GroupSpec, Scenario, assert_matches_oracle, encode, run_gpu, run_oracle
from kubernetes_autoscaler_amd import workloads
from test_kernels_emu_golden import GOLD, golden_scenario

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = kaa.Context(0)
    yield c
    c.close()


def scenario_of(w, fastpath=False, device_csr=False):
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups],
                    existing=w.existing, lanes=w.lanes, fastpath=fastpath, device_csr=device_csr)


def test_native_library_is_loaded(ctx):
    maps = open("/proc/self/maps").read()
    assert "libcasim.so" in maps


@pytest.mark.parametrize("fastpath", [False, True])
@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: c["name"])
def test_golden_rows(ctx, case, fastpath):
    if fastpath and not case["check_fastpath"]:
        pytest.skip("reference only asserts fastpath parity for single-group rows")
    sc = golden_scenario(case, fastpath)
    res, _ = run_gpu(encode(sc), ctx, fastpath=fastpath)
    assert (int(res.node_count[0]), int(res.pods_scheduled[0])) == (case["expect_nodes"], case["expect_pods"])
    assert_matches_oracle(res, run_oracle(sc), case["name"])


def test_benchmark_vector(ctx):
    b = GOLD["benchmark"]
    sc = golden_scenario(b, template_pods=b["template_pods"])
    res, _ = run_gpu(encode(sc), ctx)
    assert (int(res.node_count[0]), int(res.pods_scheduled[0])) == (b["expect_nodes"], b["expect_pods"])
    assert_matches_oracle(res, run_oracle(sc), "benchmark")


def test_fuzz_batches(ctx):
    """Every fuzz family of the CPU suite, on the GPU (one problem per seed)."""
    for base, n, kw, fast, dcsr in ((0, 60, dict(rich=False), False, False), (1000, 150, {}, False, False),
                                    (2000, 40, {}, True, False), (3000, 40, {}, False, True),
                                    (5000, 60, dict(max_groups=3, max_pegs=48), False, False),
                                    (7000, 40, dict(max_groups=3, max_pegs=260, rich=False), False, False),   # long lists: the dry-limiter loop
                                    (7000, 40, dict(max_groups=3, max_pegs=260), False, False)):              # ... with exclusion state
        for seed in range(n):
            sc = scenario_of(workloads.fuzz(base + seed, **kw), fastpath=fast, device_csr=dcsr)
            res, _ = run_gpu(encode(sc), ctx, fastpath=fast)
            assert_matches_oracle(res, run_oracle(sc), f"fuzz {base + seed}")


@pytest.mark.parametrize("name", ["C0", "C1", "C2", "C3", "C4", "R1", "R2"])
def test_baseline_configs_full_size(ctx, name):
    """BASELINE configs C0-C4 and the reference's own benchmark regimes: R1 = BenchmarkRunOnceScaleUp (10 000 singleton PEGs ->
    200 nodes, core/bench/benchmark_runonce_test.go:395-418,493-503), R2 = BenchmarkBinpackingEstimate (2595 nodes / 51 000 pods,
    estimator/binpacking_estimator_test.go:256-303)."""
    w = workloads.CONFIGS[name]()
    sc = scenario_of(w, device_csr=name in ("C2", "C3", "C4", "R1"))
    res, _ = run_gpu(encode(sc), ctx)
    oracle = run_oracle(sc)
    assert_matches_oracle(res, oracle, name)
    if name == "R1":
        assert (int(res.node_count[0]), int(res.pods_scheduled[0])) == (200, 10000)
    if name == "R2":
        assert (int(res.node_count[0]), int(res.pods_scheduled[0])) == (2595, 51000)


@pytest.mark.parametrize("n_pegs,cap,seed,generic", [(2000, 64, 0, False), (2049, 200, 1, False), (5000, 256, 2, False), (5000, 1000, 3, True),
                                                     (8191, 300, 4, True), (12000, 700, 5, False), (20000, 1024, 6, False),
                                                     (20000, 5000, 7, False), (16385, 40, 8, False), (3000, 3000, 9, False)])
def test_thousands_of_pegs_in_one_group(ctx, n_pegs, cap, seed, generic):
    """2 000 - 20 000 PEGs in ONE group (the HBM-slab sort of order_kernel, list bound > 1024; SURVEY N7: singleton PEGs are the
    reference's own benchmark regime), every node store."""
    sc = scenario_of(workloads.config_many_pegs(seed, n_pegs, cap))
    res, _ = run_gpu(encode(sc), ctx, generic=generic)
    assert_matches_oracle(res, run_oracle(sc), f"many {n_pegs} cap {cap}")


def test_r1_with_the_int64_packer_and_smaller_scale_ups(ctx):
    for nodes, generic in ((200, True), (1, False), (7, False), (60, False), (60, True)):
        sc = scenario_of(workloads.config_r1(nodes, max_ng_size=10000 if nodes == 200 else 1000))
        res, _ = run_gpu(encode(sc), ctx, generic=generic)
        assert int(res.node_count[0]) == nodes
        assert_matches_oracle(res, run_oracle(sc), f"R1 {nodes}")


def test_batched_simulations_are_independent(ctx):
    """B independent C1-shaped simulations in one launch == the same simulations one by one."""
    w = workloads.batch_of(workloads.config_c1, 24, n_pegs=40, pods_per_peg=20, cap=64)
    sc = scenario_of(w)
    res, _ = run_gpu(encode(sc), ctx)
    assert_matches_oracle(res, run_oracle(sc), "C1 batch")


def test_size_independent_properties(ctx):
    """Properties that hold at any size (checked at C3 scale): every group's placed[] is a prefix count,
    pods = sum(placed), nodes <= limiter cap, sum of requests == sum over PEGs, idempotent re-run."""
    w = workloads.config_c3()
    sc = scenario_of(w, device_csr=True)
    enc = encode(sc)
    with kaa.Problem(ctx, enc.pegs, enc.groups) as p:
        p.run(); a = p.fetch()
        p.run(); b = p.fetch()
    for f in ("node_count", "pods_scheduled", "order", "placed", "last_index_out", "req_cpu_sum"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    counts = np.array([len(pg.pods) for pg in w.pegs])
    cpu = np.array([pg.pods[0].requests["cpu"] for pg in w.pegs])
    for i, g in enumerate(w.groups):
        order, placed = a.group(i)
        assert len(set(order.tolist())) == len(order)
        assert np.all(placed >= 0) and np.all(placed <= counts[order])
        assert int(a.pods_scheduled[i]) == int(placed.sum())
        assert int(a.req_cpu_sum[i]) == int((placed.astype(np.int64) * cpu[order]).sum())
        assert 0 <= int(a.node_count[i]) <= int(a.nodes_added[i]) <= g.max_nodes


# ---- the reference-interface mirror on the GPU ------------------------------------------------------
def _est(ctx, max_nodes, fastpath=False):
    from kubernetes_autoscaler_amd import estimator as est
    from kubernetes_autoscaler_amd.objects import NodeInfo, make_node
    snap = est.ClusterSnapshotView(existing=[NodeInfo(make_node(100, 100, 10, "oldnode", "zone-jupiter"))])
    limiter = est.ThresholdBasedEstimationLimiter([est.StaticThreshold(max_nodes, 0.0)])
    return est.BinpackingNodeEstimator(ctx, snap, limiter, est.DecreasingPodOrderer(), None, None, fastpath)


@pytest.mark.parametrize("case", GOLD["cases"] + GOLD["topology_spread_cases"], ids=lambda c: c["name"])
def test_binpacking_estimate_like_the_reference_test(ctx, case):
    """TestBinpackingEstimate (binpacking_estimator_test.go:66-254), all eight rows, written against the mirror API: the
    PodTopologySpread rows go through K_est (Estimate on the whole snapshot), with and without the fastpath flag."""
    from kubernetes_autoscaler_amd.objects import (NodeInfo, build_test_pod, make_node, make_pod_equivalence_group,
                                                   with_host_port, with_labels, with_max_skew, with_namespace)
    pegs = []
    for g in case["pegs"]:
        opts = [with_namespace("universe"), with_labels({"app": "estimatee"})]
        if g.get("host_port"):
            opts.append(with_host_port(g["host_port"]))
        if g.get("max_skew"):
            opts.append(with_max_skew(*g["max_skew"]))
        pegs.append(make_pod_equivalence_group(build_test_pod("estimatee", g["cpu"], g["mem"], *opts), g["count"]))
    node_info = NodeInfo(make_node(case["millicores"], case["memory_mib"], 10, "template", "zone-mars"))
    nodes, pods = _est(ctx, case["max_nodes"]).estimate(pegs, node_info, None)
    assert (nodes, len(pods)) == (case["expect_nodes"], case["expect_pods"])
    if "expect_placed_by_input_peg" in case:
        assert pods == pegs[1].pods   # expectProcessedPods == highResourcePodGroup.Pods
    if case["check_fastpath"]:
        fn, fp = _est(ctx, case["max_nodes"], fastpath=True).estimate(pegs, node_info, None)
        assert (fn, fp) == (nodes, pods)


@pytest.mark.parametrize("kinds", [[0], [2], [1], [1, 0]])
def test_expander_chain_on_device(ctx, kinds):
    from test_expander_emu import oracle_chain
    for seed in range(12):
        w = workloads.fuzz(7000 + seed, max_groups=9, max_pegs=10, rich=False)
        sc = scenario_of(w)
        res, best = run_gpu(encode(sc), ctx, kinds=kinds, group_id_base=100)
        assert_matches_oracle(res, run_oracle(sc), f"fuzz {7000 + seed}")
        want = oracle_chain(res, kinds, [g.template.node.capacity["cpu"] for g in w.groups],
                            [g.template.node.capacity["memory"] for g in w.groups])
        bi, nb, bset, key = best
        assert [i for i in range(len(bset)) if bset[i]] == want and bi == (want[0] if want else -1)
        assert key[9] == (100 + want[0] if want else 0x7FFFFFFFFFFFFFFF)


def test_prepare_scale_up_batched(ctx):
    """ScaleUpSimulator.prepare_scale_up == SchedulablePodGroups + ComputeExpansionOption + expander, composed from the oracle."""
    from kubernetes_autoscaler_amd import estimator as est
    from kubernetes_autoscaler_amd.expander import ChainStrategy
    from kubernetes_autoscaler_amd.scaleup import ScaleUpSimulator
    w = workloads.config_c2(n_groups=12, n_pegs=80, pods_per_peg=10, cap=20)
    ngs = [est.NodeGroup(f"ng{i}", max_size_=g.max_nodes, target_size_=0) for i, g in enumerate(w.groups)]
    infos = {ng.id(): g.template for ng, g in zip(ngs, w.groups)}
    limiter = est.ThresholdBasedEstimationLimiter([est.SngCapacityThreshold(), est.ClusterCapacityThreshold()])
    sim = ScaleUpSimulator(ctx, limiter, ChainStrategy(["least-waste", "least-nodes"]), max_nodes_total=0)
    plan = sim.prepare_scale_up(w.pegs, ngs, infos, est.ClusterSnapshotView())
    sc = scenario_of(w, device_csr=True)
    oracle = run_oracle(sc)
    assert_matches_oracle(plan.result, oracle, "prepare_scale_up")
    for ng, (o, ids) in zip(ngs, oracle):
        assert plan.schedulable_pod_groups[ng.id()] == sorted(ids)
    assert plan.best is not None and plan.best in plan.options
    assert all(o.node_count > 0 and o.pods for o in plan.options)


def test_prepare_scale_up_with_spread_constraints(ctx):
    """A PEG with a hostname spread constraint makes the batch delegate every group it is schedulable on; the simulator
    re-estimates those groups on the whole snapshot (K_est) and they take part in the expander reduce."""
    from kubernetes_autoscaler_amd import estimator as est
    from kubernetes_autoscaler_amd.expander import ChainStrategy
    from kubernetes_autoscaler_amd.objects import NodeInfo, TopologySpreadConstraint, make_node
    from kubernetes_autoscaler_amd.scaleup import ScaleUpSimulator
    from oracle_driver import OracleScenario
    w = workloads.config_c2(n_groups=6, n_pegs=30, pods_per_peg=6, cap=30)
    spread_pod = w.pegs[0].pods[0]
    spread_pod.spread_constraints = [TopologySpreadConstraint(2, "kubernetes.io/hostname", 0, dict(spread_pod.labels))]
    spread_pod.topology_spread = True
    existing = [NodeInfo(make_node(4000, 8000, 20, f"old{i}", "zone-x")) for i in range(3)]
    ngs = [est.NodeGroup(f"ng{i}", max_size_=g.max_nodes, target_size_=0) for i, g in enumerate(w.groups)]
    infos = {ng.id(): g.template for ng, g in zip(ngs, w.groups)}
    limiter = est.ThresholdBasedEstimationLimiter([est.SngCapacityThreshold(), est.ClusterCapacityThreshold()])
    sim = ScaleUpSimulator(ctx, limiter, ChainStrategy(["least-nodes"]), max_nodes_total=0)
    plan = sim.prepare_scale_up(w.pegs, ngs, infos, est.ClusterSnapshotView(existing=existing))
    assert not plan.delegated
    # the oracle, group by group, on the PEGs that pass CheckPredicates on the template
    want = {}
    for ng, g in zip(ngs, w.groups):
        s = OracleScenario()
        for info in existing:
            s.add_existing(info)
        t = s.node(g.template)
        ids = [i for i, pg in enumerate(w.pegs) if s.check_predicates(t, pg.pods[0])[0]]
        e = s.estimate(t, [w.pegs[i] for i in ids], max_nodes=g.max_nodes, last_index=0)
        want[ng.id()] = (e.node_count, e.pods_scheduled)
        s.close()
    got = {o.node_group.id(): (o.node_count, len(o.pods)) for o in plan.options}
    assert got == {k: v for k, v in want.items() if v[0] > 0 and v[1] > 0}
    assert plan.best is not None and plan.best.node_count == min(v[0] for v in got.values())


# ---- edge cases of the boundary on the GPU ------------------------------------------------------------
from test_edge_cases_emu import CASES as EDGE_CASES  # noqa: E402


@pytest.mark.parametrize("name,sc", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_case(ctx, name, sc):
    res, _ = run_gpu(encode(sc), ctx)
    assert_matches_oracle(res, run_oracle(sc), name)


def test_generic_packer_equals_register_packer(ctx):
    """force_generic_packer: the int64 LDS packer and the int32 register packer agree bit for bit."""
    for name in ("C0", "C1", "C2"):
        w = workloads.CONFIGS[name]()
        sc = scenario_of(w)
        enc = encode(sc)
        a, _ = run_gpu(enc, ctx)
        b, _ = run_gpu(enc, ctx, generic=True)
        for f in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "req_cpu_sum", "req_mem_sum", "order", "placed"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (name, f)
        assert_matches_oracle(b, run_oracle(sc), name)


# ---- Estimate on the whole snapshot (K_est) --------------------------------------------------------
@pytest.mark.parametrize("case", GOLD["cases"] + GOLD["topology_spread_cases"], ids=lambda c: c["name"])
def test_cluster_estimate_golden_rows(ctx, case):
    from harness import assert_cluster_estimate_matches, cluster_estimate_gpu
    sc = golden_scenario(case)
    est, ids = run_oracle(sc)[0]
    assert (est.node_count, est.pods_scheduled) == (case["expect_nodes"], case["expect_pods"])
    assert_cluster_estimate_matches(cluster_estimate_gpu(sc, ctx), est, ids, case["name"])


def test_cluster_estimate_fuzz(ctx):
    from harness import assert_cluster_estimate_matches, cluster_estimate_gpu
    delegated = 0
    for seed in range(300):
        w = workloads.fuzz_estimate_domains(seed)
        sc = scenario_of(w)
        got = cluster_estimate_gpu(sc, ctx)
        if got[0] == 1:
            delegated += 1
            continue
        est, ids = run_oracle(sc)[0]
        assert_cluster_estimate_matches(got, est, ids, w.name)
    assert delegated < 40


def test_fuzz_node_affinity_terms_in_an_estimate(ctx):
    """Several nodeSelectorTerms (ORed) over template labels: one more label-requirement bit per distinct term list."""
    ran = 0
    for seed in range(150):
        w = workloads.fuzz(7000 + seed)
        if workloads.add_random_node_affinity_terms(seed, [p for pg in w.pegs for p in pg.pods], [g.template for g in w.groups], allow_per_node=False):
            sc = scenario_of(w)
            res, _ = run_gpu(encode(sc), ctx)
            assert_matches_oracle(res, run_oracle(sc), f"node terms {seed}")
            ran += 1
    assert ran > 100
