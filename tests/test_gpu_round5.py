"""-m gpu: round 5 on a real MI355X through the C ABI.
  * casim_options.chain_last_index: the groups of a simulation as successive Estimate() calls (lastIndex handed on, plugin_runner.go:138) —
    the packer's fixed-point passes against the oracle's sequential loop, both packers, batches, streamed parts;
  * resources by name (ABI 9): the reference's GPU-pool orchestrator rows through the call sequence of the Go binding;
  * feas_stream_kernel: every instantiation against the oracle and against the LDS-staged kernel it replaces;
  * the a3 quotient with a divisor of one next to a very large PEG."""
import os

import numpy as np
import pytest

import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, PodEquivalenceGroup
from harness import (GroupSpec, Scenario, assert_matches_oracle, encode, encode_batch, run_gpu, run_gpu_tables, run_oracle)
from orchestrator_rows import Row, per_group_of_batch

pytestmark = pytest.mark.gpu
GiB = 1 << 30


@pytest.fixture(scope="module")
def ctx():
    c = kaa.Context(0)
    yield c
    c.close()


def _scenario(seed, device_csr, max_groups=7, max_pegs=14, existing=True):
    w = workloads.fuzz(seed, max_groups=max_groups, max_pegs=max_pegs)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None if device_csr else g.pegs) for g in w.groups],
                    existing=w.existing if existing else [], lanes=w.lanes, device_csr=device_csr)


# ---- chain_last_index ---------------------------------------------------------------------------------------------------------------
def test_chained_groups_match_the_sequential_loop_on_the_device(ctx):
    changed = 0
    for seed in range(120):
        sc = _scenario(31000 + seed, device_csr=seed % 2 == 0)
        want = run_oracle(sc, chain=True)
        changed += any(a.last_index_out != b.last_index_out for (a, _), (b, _) in zip(want, run_oracle(sc)))
        for generic in (False, True):
            enc = encode(sc)
            res, _ = run_gpu(enc, ctx, chain=True, generic=generic)
            enc.close()
            assert_matches_oracle(res, want, f"seed {seed} generic={generic}")
    assert changed > 10     # (the corpus exercises the chain)


def test_chained_batches_and_streamed_parts_on_the_device(ctx):
    for seed in range(12):
        scs = [_scenario(32000 + 100 * seed + k, device_csr=True, max_groups=6, existing=False) for k in range(2 + seed % 5)]
        enc, ts, bases = encode_batch(scs)
        want = []
        for sc, (pb, _) in zip(scs, bases):
            want.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc, chain=True)])
        for kw in ({}, {"generic": True}, {"n_streams": 3}):
            res, _ = run_gpu_tables(ts, ctx, chain=True, **kw)
            assert_matches_oracle(res, want, f"batch {seed} {kw}")
        enc.close()


def test_c2_and_c4_as_one_chained_simulation_on_the_device(ctx):
    for name in ("C2", "C4", "C3"):
        w = workloads.CONFIGS[name]()
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes,
                      device_csr=True)
        enc = encode(sc)
        res, _ = run_gpu(enc, ctx, chain=True)
        enc.close()
        assert_matches_oracle(res, run_oracle(sc, chain=True), f"{name} chained")


# ---- resources by name ------------------------------------------------------------------------------------------------------------------
def test_gpu_pool_rows_through_the_named_resource_calls_on_the_device(ctx):
    from test_named_lanes import GPU_ROWS
    assert len(GPU_ROWS) == 3
    for row in GPU_ROWS:
        for generic in (False, True):
            r = Row(row)
            sc = r.scenario()
            enc = encode(sc, named_lanes=True)
            assert enc.lanes == ("cpu", "memory", "ephemeral-storage", "nvidia.com/gpu")
            res, _ = run_gpu(enc, ctx, generic=generic)
            enc.close()
            assert_matches_oracle(res, run_oracle(sc), row["name"])
            r.check(r.decide(per_group_of_batch(res)), "MI355X, named lanes:")


def test_hugepages_and_a_name_without_a_lane_on_the_device(ctx):
    from test_named_lanes import _tmpl
    huge = "hugepages-2Mi"
    pods = [Pod(name="hp", requests={"cpu": 100, "memory": 64 << 20, huge: 1 * GiB})] * 3
    sc = Scenario(pegs=[PodEquivalenceGroup(pods=pods)], groups=[GroupSpec(_tmpl("with-hugepages", {huge: 2 * GiB}), 0, 0, None), GroupSpec(_tmpl("without"), 0, 0, None)],
                  existing=[], lanes=("cpu", "memory", huge), device_csr=True)
    enc = encode(sc, named_lanes=True)
    res, _ = run_gpu(enc, ctx)
    enc.close()
    assert_matches_oracle(res, run_oracle(sc), "hugepages")
    assert (int(res.node_count[0]), int(res.pods_scheduled[0]), int(res.pods_scheduled[1])) == (2, 3, 0)
    # six names: the pod that asks for the sixth is delegated, never estimated without its request
    rq = {"cpu": 100, "memory": 1 << 28}
    names = [f"example.com/dev{i}" for i in range(6)]
    pegs = [PodEquivalenceGroup(pods=[Pod(name="five", requests={**rq, **{n: 1 for n in names[:5]}})] * 2),
            PodEquivalenceGroup(pods=[Pod(name="sixth", requests={**rq, names[5]: 1})] * 2)]
    sc = Scenario(pegs=pegs, groups=[GroupSpec(_tmpl("t", {n: 4 for n in names}), 0, 0, None)], existing=[], lanes=("cpu", "memory"), device_csr=True)
    enc = encode(sc, named_lanes=True)
    assert int(enc.pegs.flags[1]) & _abi.PEG_UNSUPPORTED and not int(enc.pegs.flags[0]) & _abi.PEG_UNSUPPORTED
    res, _ = run_gpu(enc, ctx)
    enc.close()
    assert int(res.status[0]) == _abi.NG_UNSUPPORTED


# ---- feas_stream_kernel ------------------------------------------------------------------------------------------------------------------
def _both(ts, ctx, **kw):
    os.environ.pop("CASIM_NO_FEAS_STREAM", None)
    new, _ = run_gpu_tables(ts, ctx, **kw)
    os.environ["CASIM_NO_FEAS_STREAM"] = "1"
    try:
        old, _ = run_gpu_tables(ts, ctx, **kw)
    finally:
        os.environ.pop("CASIM_NO_FEAS_STREAM", None)
    for f in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "offsets", "order", "placed"):
        assert np.array_equal(getattr(new, f), getattr(old, f)), f
    return new


def test_streaming_feasibility_kernel_on_the_device(ctx):
    """the emulator corpus of tests/test_feas_stream_emu.py on the MI355X (resident problems: the mask31 instantiations; the one-shot calls of
    the other GPU tests run the general ones)"""
    from test_feas_stream_emu import _sc, _want
    for seed in range(24):
        scs = [_sc(52000 + 97 * seed + k, max_groups=6, max_pegs=90 if seed % 3 == 0 else 14, rich=seed % 2 == 0) for k in range(2 + seed % 5)]
        enc, ts, bases = encode_batch(scs)
        res = _both(ts, ctx)
        assert_matches_oracle(res, _want(scs, bases), f"seed {seed}")
        enc.close()


def test_the_headline_batch_takes_the_streaming_kernel_and_stays_exact(ctx):
    """C2 x 256 simulations as the bench builds them: the problem reports feas_stream_kernel<lean, mask31>, its feasibility launch can be
    timed alone, and every group equals the oracle (bench.verify_headline) — with streams and without"""
    import bench
    from kubernetes_autoscaler_amd.tables import TableSet
    seeds = 4
    ts = bench.simulation_tables(workloads.config_c2, range(seeds), kaa.Encoder, TableSet).tile(64)
    pegs, groups = ts.structs()
    for n_streams in (0, 4):
        with kaa.Problem(ctx, pegs, groups, n_streams=n_streams) as prob:
            ms, info = prob.time_feasibility(10)
            assert info["stream"] and info["lean"] and ms > 0, info    # (C2's 32 label pairs reach bit 31: the mask64 instantiation)
            prob.run()
            res = prob.fetch()
        chk = bench.verify_headline(workloads, workloads.config_c2, seeds, ts, res)
        assert chk["headline_bit_exact"] and chk["groups_compared"] == ts.n_groups, chk
    row = bench.feasibility_roofline(kaa, ctx, workloads, TableSet, "C2", 512, 4, iters=10)
    assert row["bit_exact"] and row["kernel"].startswith("feas_stream_kernel<lean, ") and 0 < row["frac"] < 1.0, row
    print("roofline_feasibility (C2 x 512, a small launch):", {k: row[k] for k in ("kernel_ms", "achieved", "frac", "algorithmic_bytes_per_launch")})


def test_gates_dictionaries_and_long_simulations_on_the_device(ctx):
    import test_feas_stream_emu as m
    from kubernetes_autoscaler_amd.objects import Taint, Toleration
    # more than 64 groups per simulation, rows that end inside a word
    rng = np.random.default_rng(7)
    pegs = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": int(rng.choice([100, 500, 2000, 9000])), "memory": int(rng.choice([1, 4, 40])) << 28})] * int(rng.integers(1, 4)))
            for i in range(130)]
    groups = [GroupSpec(m._tmpl(f"g{i}", cpu=int(rng.choice([1000, 4000, 16000])), mem=int(rng.choice([2, 16, 64])) * GiB), 3, 0, None) for i in range(70)]
    scs = [Scenario(pegs=pegs, groups=groups, device_csr=True), Scenario(pegs=pegs[:130], groups=groups[:3], device_csr=True)]
    enc, ts, bases = encode_batch(scs)
    assert_matches_oracle(_both(ts, ctx), m._want(scs, bases), "70 groups")
    enc.close()
    # dictionaries beyond bit 31
    gs = [GroupSpec(m._tmpl(f"t{i}", taints=[Taint(f"k{i}", "v", "NoSchedule")], labels={f"l{i}": "x"}), 0, 0, None) for i in range(40)]
    ps = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": 100, "memory": 1 << 28}, node_selector={f"l{i}": "x"},
                                        tolerations=[Toleration(key=f"k{i}", operator="Exists")])] * 2) for i in range(40)]
    scs = [Scenario(pegs=ps, groups=gs, device_csr=True), Scenario(pegs=ps[5:], groups=gs[3:], device_csr=True)]
    enc, ts, bases = encode_batch(scs)
    want = m._want(scs, bases)
    assert_matches_oracle(_both(ts, ctx), want, "40 keys")
    assert [ids for _, ids in want[:40]] == [[i] for i in range(40)]
    enc.close()


# ---- ADVICE r4 ---------------------------------------------------------------------------------------------------------------------------
def test_a3_quotient_with_a_divisor_of_one_and_a_very_large_peg(ctx):
    """need = ceil(rem / cn) with cn = 1 (one pod per node) and rem up to 2^25: the raw v_rcp_f64 estimate (~2^-24 relative) is off by more
    than the +-1 the fix-up repairs from quotients of 2^24 on; with the Newton step it is exact.  The limiter caps what is created, so the
    oracle finishes quickly; nodes_added / limiter grants / pods must agree."""
    for count, cap in ((1 << 25, 900), ((1 << 24) + 12345, 64), ((1 << 25) - 1, 1)):
        cap_d = {"cpu": 1000, "memory": 4 * GiB, "pods": 110}
        tmpl = NodeInfo(Node(name="one-pod-per-node", labels={}, allocatable=dict(cap_d), capacity=dict(cap_d)), [])
        big = PodEquivalenceGroup(pods=[Pod(name="big", requests={"cpu": 600, "memory": 1 * GiB})])   # (one object; the count travels as a number below)
        sc = Scenario(pegs=[big], groups=[GroupSpec(tmpl, cap, 0, None)], existing=[], device_csr=True)
        enc = encode(sc)
        # the PEG's pod count, written where the encoder put it (2^25 pod objects are not needed to say "2^25")
        np.ctypeslib.as_array(enc.pegs.count, shape=(1,))[0] = count
        res, _ = run_gpu(enc, ctx)
        enc.close()
        assert (int(res.status[0]), int(res.node_count[0]), int(res.pods_scheduled[0]), int(res.nodes_added[0])) == (0, cap, cap, cap), (count, cap, res.node_count, res.pods_scheduled)


# ---- the snapshot's SchedulePod rows (predicate_snapshot_test.go:400-511) -------------------------------------------------------------------
def test_snapshot_schedule_pod_rows_on_the_device(ctx):
    from harness import SchedCase, assert_sched_matches, sched_gpu, sched_oracle
    from test_snapshot_schedule_pod import CASES, build
    for case in CASES:
        nodes, pod, acceptable, want = build(case)
        sc = SchedCase(nodes=nodes, pods=[pod], acceptable=acceptable)
        got = sched_gpu(sc, ctx)
        assert_sched_matches(got, sched_oracle(sc), case["name"])
        assert int(got[1][0]) == want, case["name"]


# ---- the removal loop as one wave over per-class fit masks (removals_lean_kernel) ---------------------------------------------------------
def test_lean_removal_kernel_and_k_sched_agree_with_the_oracle_on_the_device(ctx, monkeypatch):
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import fuzz_removals, fuzz_removals_plain, removal_scale
    lean = 0
    cases = [fuzz_removals_plain(s) for s in range(90)] + [fuzz_removals(s) for s in range(30)] + [removal_scale(700, pods_per_node=12, frac_candidates=0.3, seed=2)]
    for w in cases:
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=w.max_removable, last_index=w.last_index)
        want = removal_oracle(case)
        monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
        assert_removal_matches(removal_device(case, ctx), want, f"{w.name} lean")
        lean += int(kaa.Context.last_removals_info()["lean"])
        monkeypatch.setenv("CASIM_NO_LEAN_REMOVALS", "1")
        assert_removal_matches(removal_device(case, ctx), want, f"{w.name} K_sched")
        assert not kaa.Context.last_removals_info()["lean"]
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    assert lean >= 100, lean


def test_runs_of_replicas_a_word_of_nodes_at_a_time_on_the_device(ctx, monkeypatch):
    """runs of identical pods (workloads.fuzz_removals_runs: replicas, tight and loose clusters, runs that come round the list or fail half way,
    pods listed again, clusters of > 64 mask words): the one-wave kernel placing a word of nodes at a time (schedule_run), the same kernel pod by
    pod (CASIM_LEAN_BULK_MIN=0) and K_sched — each against the oracle in every field"""
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import fuzz_removals_runs
    lean = 0
    for seed in range(160):
        w = fuzz_removals_runs(seed)
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=w.max_removable, last_index=w.last_index, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64)
        want = removal_oracle(case)
        for name, env in (("a word at a time", {}), ("pod by pod", {"CASIM_LEAN_BULK_MIN": "0"}), ("K_sched", {"CASIM_NO_LEAN_REMOVALS": "1"})):
            for k in ("CASIM_LEAN_BULK_MIN", "CASIM_NO_LEAN_REMOVALS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            assert_removal_matches(removal_device(case, ctx), want, f"{w.name} {name}")
            if not env:
                lean += int(kaa.Context.last_removals_info()["lean"])
    for k in ("CASIM_LEAN_BULK_MIN", "CASIM_NO_LEAN_REMOVALS"):
        monkeypatch.delenv(k, raising=False)
    assert lean >= 120, lean


def test_a_removal_log_smaller_than_the_worst_case_on_the_device(ctx, monkeypatch):
    """the one-wave kernel with a 256-entry log (CASIM_LEAN_LOG_CAP): it finishes (squeezing dead entries out when the log fills up) or gives up
    at the commit that does not fit and K_sched answers — the oracle's results either way (tests/test_removal_lean_emu.py: the same under the emulator)"""
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import fuzz_removals_plain, fuzz_removals_runs
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    monkeypatch.setenv("CASIM_LEAN_HBM_LOG", "0")    # (the fall-back to K_sched; with the HBM log allowed the one-wave kernel would answer again)
    finished = squeezed = gave_up = 0
    from kubernetes_autoscaler_amd.workloads import runonce_scale_down
    for w in ([fuzz_removals_plain(s) for s in range(0, 400, 9)] + [fuzz_removals_runs(s) for s in range(80)] +
              [runonce_scale_down(n, ppn) for n, ppn in ((20, 6), (30, 5), (40, 4), (25, 8), (60, 3))]):
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=w.max_removable, last_index=w.last_index, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64)
        want = removal_oracle(case)
        monkeypatch.delenv("CASIM_LEAN_LOG_CAP", raising=False)
        assert_removal_matches(removal_device(case, ctx), want, f"{w.name}")
        eligible = bool(kaa.Context.last_removals_info()["lean"])
        monkeypatch.setenv("CASIM_LEAN_LOG_CAP", "256")
        assert_removal_matches(removal_device(case, ctx), want, f"{w.name} small log")
        ran_lean = bool(kaa.Context.last_removals_info()["lean"])
        assert not (ran_lean and not eligible)
        moves = sum(len(lst) for lst, r in zip(case.pod_lists(), want["removable"]) if r == 1) + sum(1 for c, _, _ in want["ext"] if want["removable"][c] == 1)
        finished += int(ran_lean)
        squeezed += int(ran_lean and case.persist and moves > 256)
        gave_up += int(eligible and not ran_lean)
    monkeypatch.delenv("CASIM_LEAN_LOG_CAP", raising=False)
    monkeypatch.delenv("CASIM_LEAN_HBM_LOG", raising=False)
    assert finished >= 20 and squeezed >= 2 and gave_up >= 5, (finished, squeezed, gave_up)
    # default policy: an LDS log that gives up hands over to the log in HBM — still the one-wave kernel, still the oracle's results
    monkeypatch.setenv("CASIM_LEAN_LOG_CAP", "256")
    for w in [runonce_scale_down(n, ppn) for n, ppn in ((20, 40), (40, 20), (30, 30))]:
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, ext_capacity=4000)
        assert_removal_matches(removal_device(case, ctx), removal_oracle(case), f"{w.name} LDS log, then HBM log")
        assert kaa.Context.last_removals_info()["lean"]
    monkeypatch.delenv("CASIM_LEAN_LOG_CAP", raising=False)


def test_chain_strategy_rows_on_the_device(ctx):
    """TestChainStrategy_BestOption (expander/factory/chain_test.go:57-134) through packer + option_kernel on the MI355X (tests/test_chain_strategy_rows.py)"""
    from harness import run_gpu
    from test_chain_strategy_rows import GOLD, check_row
    for case in GOLD["cases"]:
        check_row(case, lambda enc, kinds: run_gpu(enc, ctx, kinds=kinds))


def test_clusters_without_the_hostname_label_on_the_device(ctx):
    """BenchmarkRunFiltersUntilPassingNode (plugin_runner_test.go:524-583: 5 001 nodes built by BuildTestNode — no labels —, a pod with a hostname
    anti-affinity term, one node with room) and fuzz clusters with the label taken off every node: the terms are inert (no node carries the
    topology key), nothing is delegated, every field equals the oracle's (tests/test_hostname_inert.py: the same under the emulator)"""
    from harness import RemovalCase, SchedCase, assert_removal_matches, assert_sched_matches, removal_device, removal_oracle, sched_gpu, sched_oracle
    from test_hostname_inert import _strip, benchmark_cluster
    from kubernetes_autoscaler_amd import workloads as W
    case, b = benchmark_cluster()
    got = sched_gpu(case, ctx)
    assert_sched_matches(got, sched_oracle(case), "BenchmarkRunFiltersUntilPassingNode")
    assert list(got[1]) == [b["expect_node_index"]]
    for seed in range(200, 260):
        w = W.fuzz_pending(seed)
        sc = SchedCase(nodes=_strip(w.nodes), pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)
        assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name)
        r = W.fuzz_removals(seed)
        rc = RemovalCase(nodes=_strip(r.nodes), candidates=r.candidates, destination=r.destination, hints=r.hints, persist=r.persist,
                         max_removable=r.max_removable, last_index=r.last_index)
        assert_removal_matches(removal_device(rc, ctx), removal_oracle(rc), r.name)


def test_the_removal_log_in_hbm_on_the_device(ctx, monkeypatch):
    """removals_lean_kernel<., true, true> (the log of committed moves in HBM, 32-bit pod indices): forced on the run fuzz and the plain fuzz
    (CASIM_LEAN_HBM_LOG=1), picked by itself for BenchmarkRunOnceScaleDown's cluster at 1 650 nodes (66 000 pods) — the oracle's results in every field"""
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import fuzz_removals_plain, fuzz_removals_runs, runonce_scale_down
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    monkeypatch.setenv("CASIM_LEAN_HBM_LOG", "1")
    ran = 0
    for w in [fuzz_removals_runs(s) for s in range(120)] + [fuzz_removals_plain(s) for s in range(0, 400, 5)]:
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=w.max_removable, last_index=w.last_index, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64)
        assert_removal_matches(removal_device(case, ctx), removal_oracle(case), f"{w.name} log in HBM")
        ran += int(kaa.Context.last_removals_info()["lean"])
    assert ran >= 150, ran
    monkeypatch.delenv("CASIM_LEAN_HBM_LOG", raising=False)
    w = runonce_scale_down(1650)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates, ext_capacity=70000)
    want = removal_oracle(case)
    got = removal_device(case, ctx)
    assert kaa.Context.last_removals_info()["lean"]
    assert_removal_matches(got, want, w.name)
    assert int((got.removable == 1).sum()) == 990


def test_the_reference_scale_down_benchmark_on_the_device(ctx, monkeypatch):
    """BenchmarkRunOnceScaleDown at full size (core/bench/benchmark_runonce_test.go:505-521: 400 nodes at 40 %, verifyToBeDeleted(240)): both
    removal kernels == the oracle in every field, and the reference's own number comes out."""
    import json, os
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import runonce_scale_down
    b = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["benchmark_runonce_scale_down"]
    w = runonce_scale_down(b["nodes"], b["pods_per_node"])
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates, ext_capacity=40 * b["nodes"] * b["pods_per_node"])
    want = removal_oracle(case)
    assert sum(1 for r in want["removable"] if r == 1) == b["expect_to_be_deleted"]
    for force_k_sched in (False, True):
        if force_k_sched:
            monkeypatch.setenv("CASIM_NO_LEAN_REMOVALS", "1")
        else:
            monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
        got = removal_device(case, ctx)
        assert_removal_matches(got, want, f"{w.name} {'K_sched' if force_k_sched else 'default kernel'}")
        assert int((got.removable == 1).sum()) == b["expect_to_be_deleted"]
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)


def test_lean_removal_kernel_long_transactions_and_wide_clusters_on_the_device(ctx, monkeypatch):
    from harness import RemovalCase, assert_removal_matches, removal_device, removal_oracle
    from kubernetes_autoscaler_amd.workloads import fuzz_removals_plain
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    seen_big = seen_long = 0
    for seed in range(400):
        w = fuzz_removals_plain(seed)
        big, long_ = len(w.nodes) > 4096, any(len(w.nodes[c].pods) > 256 for c in w.candidates)
        if not (big or long_):
            continue
        case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                           max_removable=w.max_removable, last_index=w.last_index)
        assert_removal_matches(removal_device(case, ctx), removal_oracle(case), w.name)
        assert kaa.Context.last_removals_info()["lean"]
        seen_big += big; seen_long += long_
    assert seen_big >= 5 and seen_long >= 10


# ---- casim_pegs.req32 / req_unit (ABI 10) and simulations that share their PEG rows ------------------------------------------------------------
def test_narrowed_requests_and_shared_peg_rows_on_the_device(ctx):
    """the batch with int64 requests == the batch with req32 + req_unit (req NULL) == the batch whose simulations point into ONE copy of the PEG
    tables (TableSet.tile_groups), resident and cut into streamed parts; sizes beyond the device gcd pass's threshold included"""
    fields = ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum", "placed")
    for seed, tiles in ((0, 1), (1, 3), (2, 40)):
        scs = [_scenario(52000 + 10 * seed + k, device_csr=True, existing=False) for k in range(4)]
        if len({sc.lanes for sc in scs}) > 1:
            scs = [scs[0]] * 4
        enc, ts, _ = encode_batch(scs)
        big, shared = ts.tile(tiles), ts.tile_groups(tiles)
        kinds = [_abi.EXPANDER_LEAST_WASTE]
        for streams in (0, 3):
            wide, ew = run_gpu_tables(big, ctx, kinds=kinds, n_streams=streams)
            narrow, en = run_gpu_tables(big, ctx, kinds=kinds, n_streams=streams, narrow_requests=True)
            sh, es = run_gpu_tables(shared, ctx, kinds=kinds, n_streams=streams, narrow_requests=True)
            for f in fields:
                assert np.array_equal(getattr(wide, f), getattr(narrow, f)), (seed, streams, f)
                assert np.array_equal(getattr(wide, f), getattr(sh, f)), (seed, streams, f, "shared")
            assert np.array_equal(wide.order, narrow.order) and np.array_equal(np.asarray(wide.order) % ts.n_pegs, sh.order)
            assert np.array_equal(ew["packed"], en["packed"]) and np.array_equal(ew["packed"], es["packed"])
        enc.close()


def test_narrowed_requests_on_the_headline_shape(ctx):
    """C2 simulations (the bench's batch, a small one): req32 input through the device's own narrowing-free path against the int64 input that takes the
    device gcd pass (>= 2^17 request values)"""
    from kubernetes_autoscaler_amd.tables import TableSet
    sets = []
    for s in range(4):
        w = workloads.config_c2(seed_offset=s)
        enc = kaa.Encoder(lanes=w.lanes)
        for pg in w.pegs:
            enc.add_peg(pg)
        for g in w.groups:
            enc.add_group(g.template, max_nodes=g.max_nodes, existing_nodes=0, last_index=g.last_index, pegs=None)
        enc.finalize()
        sets.append(TableSet.from_encoder(enc).as_one_simulation())
        enc.close()
    ts = TableSet.concat(sets).tile(48)      # 192 simulations x 400 PEGs x 2 lanes = 153 600 values
    kinds = [_abi.EXPANDER_LEAST_NODES]
    wide, ew = run_gpu_tables(ts, ctx, kinds=kinds, n_streams=4)
    narrow, en = run_gpu_tables(ts, ctx, kinds=kinds, n_streams=4, narrow_requests=True)
    for f in ("node_count", "pods_scheduled", "last_index_out", "placed", "order", "req_cpu_sum", "req_mem_sum"):
        assert np.array_equal(getattr(wide, f), getattr(narrow, f)), f
    assert np.array_equal(ew["packed"], en["packed"])


# ---- the orderer's ranks once per (simulation, allocatable pair) ---------------------------------------------------------------------------------
def test_ranked_orderer_equals_the_per_group_sort_on_the_device(ctx, monkeypatch):
    from kubernetes_autoscaler_amd.engine import Problem
    from kubernetes_autoscaler_amd.tables import TableSet
    fields = ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum", "order", "placed")
    # (a) fuzz batches, the path forced on
    for seed in range(12):
        scs = [_scenario(56000 + 10 * seed + k, device_csr=True, existing=False) for k in range(2 + seed % 3)]
        if len({sc.lanes for sc in scs}) > 1:
            scs = [scs[0], scs[0]]
        enc, ts, bases = encode_batch(scs)
        monkeypatch.setenv("CASIM_RANK_ONCE", "0")
        a, ea = run_gpu_tables(ts.tile(3), ctx, kinds=[_abi.EXPANDER_LEAST_WASTE], n_streams=seed % 3)
        monkeypatch.setenv("CASIM_RANK_ONCE", "1")
        b, eb = run_gpu_tables(ts.tile(3), ctx, kinds=[_abi.EXPANDER_LEAST_WASTE], n_streams=seed % 3)
        for f in fields:
            assert np.array_equal(getattr(a, f), getattr(b, f)), (seed, f)
        assert np.array_equal(ea["packed"], eb["packed"])
        enc.close()
    # (b) C3-shaped: the automatic rule takes it (info says so), results equal the forced-off run and the oracle's for one simulation
    monkeypatch.delenv("CASIM_RANK_ONCE", raising=False)
    sets, ws = [], []
    for s in range(3):
        w = workloads.config_c3(seed_offset=s)
        enc = kaa.Encoder(lanes=w.lanes)
        for pg in w.pegs:
            enc.add_peg(pg)
        for g in w.groups:
            enc.add_group(g.template, max_nodes=g.max_nodes, existing_nodes=0, last_index=g.last_index, pegs=None)
        enc.finalize()
        sets.append(TableSet.from_encoder(enc).as_one_simulation())
        enc.close(); ws.append(w)
    ts = TableSet.concat(sets).tile(4)
    pegs, groups = ts.structs()
    with Problem(ctx, pegs, groups, False, False, n_streams=0) as p:
        assert p.info()["ranked_orderer"]
        p.run(); auto = p.fetch()
    monkeypatch.setenv("CASIM_RANK_ONCE", "0")
    with Problem(ctx, pegs, groups, False, False, n_streams=0) as p:
        assert not p.info()["ranked_orderer"]
        p.run(); off = p.fetch()
    monkeypatch.delenv("CASIM_RANK_ONCE")
    for f in fields:
        assert np.array_equal(getattr(auto, f), getattr(off, f)), f
    sc = Scenario(pegs=ws[0].pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in ws[0].groups], device_csr=True, lanes=ws[0].lanes)
    want = run_oracle(sc)
    ng = len(ws[0].groups)
    assert [int(x) for x in auto.node_count[:ng]] == [e.node_count for e, _ in want] and [int(x) for x in auto.pods_scheduled[:ng]] == [e.pods_scheduled for e, _ in want]
