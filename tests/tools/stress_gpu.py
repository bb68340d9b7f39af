#!/usr/bin/env python3
"""One-off randomized stress with seeds the test suites do not use: every device path against the oracle — on the MI355X,
or (CASIM_STRESS_EMU=1) the same product kernels under the wave emulator on the CPU.
Usage: stress_gpu.py [first_seed] [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import workloads as W  # noqa: E402
from harness import (EmuContext, GroupSpec, RemovalCase, Scenario, SchedCase, assert_cluster_estimate_matches, assert_matches_oracle,  # noqa: E402
                     assert_removal_matches, assert_sched_matches, cluster_estimate_emu, cluster_estimate_gpu, encode, removal_device,
                     removal_oracle, run_emu, run_gpu, run_oracle, sched_gpu, sched_oracle)

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
EMU = os.environ.get("CASIM_STRESS_EMU") == "1"
ctx = EmuContext(0) if EMU else kaa.Context(0)
if EMU:   # same call shapes as the GPU helpers
    cluster_estimate_gpu = lambda sc, _ctx: cluster_estimate_emu(sc)          # noqa: E731
    run_gpu = lambda enc, _ctx, fastpath=False, generic=False: run_emu(enc, fastpath=fastpath, generic=generic)  # noqa: E731
t0 = time.time()
stats = {}


def bump(k):
    stats[k] = stats.get(k, 0) + 1


def scen(w, **kw):
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing,
                    lanes=w.lanes, **kw)


for seed in range(first, first + count):
    if (seed - first) % 1000 == 0 and seed > first:
        print(f"... {seed - first} seeds OK, {time.time() - t0:.0f} s", flush=True)
    for gen in (W.fuzz_pending, W.fuzz_pending_domains):
        w = gen(seed)
        sc = SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)
        assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name); bump(gen.__name__)
    # spread constraints with non-default node-inclusion policies next to tainted nodes (eligibility rows of the encoder)
    from test_sched_emu import honor_taints_variant
    w = honor_taints_variant(W.fuzz_pending_domains(seed + 7_000_000), seed)
    sc = SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)
    assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name); bump("fuzz_pending_policies")
    for gen in (W.fuzz_removals, W.fuzz_removals_domains):
        w = gen(seed)
        rc = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                         max_removable=w.max_removable, last_index=w.last_index)
        assert_removal_matches(removal_device(rc, ctx), removal_oracle(rc), w.name); bump(gen.__name__)
        if rc.max_removable > 0:   # candidates of atomically scaled groups do not count toward the limit
            import random
            rng = random.Random(seed)
            rc.atomic = [1 if rng.random() < 0.4 else 0 for _ in rc.candidates]
            assert_removal_matches(removal_device(rc, ctx), removal_oracle(rc), w.name + " atomic"); bump(gen.__name__ + "_atomic")
    w = W.fuzz_estimate_domains(seed)
    sc = scen(w)
    got = cluster_estimate_gpu(sc, ctx)
    if got[0] == 0:
        est, ids = run_oracle(sc)[0]
        assert_cluster_estimate_matches(got, est, ids, w.name); bump("fuzz_estimate_domains")
    else:
        bump("fuzz_estimate_domains_delegated")
    for fast in (False, True):
        w = W.fuzz(seed)
        sc = scen(w, fastpath=fast)
        res, _ = run_gpu(encode(sc), ctx, fastpath=fast)
        assert_matches_oracle(res, run_oracle(sc), w.name); bump(f"fuzz_packer_fast{int(fast)}")
    # lists derived on the device: one launch for feasibility + offsets + lists + order (front_kernel), long rows on odd seeds
    w = W.fuzz(seed + 3_000_000, max_groups=6, max_pegs=150 if seed % 2 else 24)
    sc = scen(w, device_csr=True)
    res, _ = run_gpu(encode(sc), ctx)
    assert_matches_oracle(res, run_oracle(sc), w.name); bump("fuzz_packer_device_lists")
    # round 4: the register store on int64 lanes (forced on the same scenario; wide lanes that select it by themselves) and the LDS store
    enc = encode(sc)
    want = run_oracle(sc)
    for generic in (2, 1):
        if EMU:
            res, _ = run_emu(enc, generic=generic)
        else:
            from harness import run_gpu as _run_gpu
            res, _ = _run_gpu(enc, ctx, generic=generic)
        assert_matches_oracle(res, want, w.name + f" generic={generic}"); bump(f"fuzz_packer_generic{generic}")
    enc.close()
    from test_pack_i64_emu import shape_scenario
    sc = shape_scenario(seed, 1 << 20, 1024) if seed % 5 else shape_scenario(seed, 1 << 40, 1 << 9)
    res, _ = run_gpu(encode(sc), ctx)
    assert_matches_oracle(res, run_oracle(sc), f"wide lanes {seed}"); bump("wide_lanes_int64_store")
    # round 4: hostname-level required pod affinity inside the template packers (node bits of NEED polarity, the series walked twice)
    from test_pod_affinity_emu import _batch_affinity_workload
    from kubernetes_autoscaler_amd.objects import LABEL_HOSTNAME, LABEL_ZONE
    w, _n = _batch_affinity_workload(1_000_000 + seed, keys=(LABEL_HOSTNAME, LABEL_HOSTNAME, LABEL_HOSTNAME, LABEL_ZONE))
    fast = seed % 4 == 3
    sc = scen(w, device_csr=seed % 2 == 0, fastpath=fast)
    res, _ = run_gpu(encode(sc), ctx, fastpath=fast)
    if not any(int(x) != 0 for x in res.status):
        assert_matches_oracle(res, run_oracle(sc), f"hostname affinity {seed}"); bump("hostname_affinity")
    else:
        bump("hostname_affinity_other_predicate_delegated")
    # more than two resource lanes in K_sched (every fourth seed)
    if seed % 4 == 0:
        from test_sched_lanes_emu import LANES4, LANES8, with_extra_resources
        lanes = LANES8 if seed % 8 == 0 else LANES4
        w = W.fuzz_pending(seed + 5_000_000)
        nodes, pods = with_extra_resources(w.nodes, w.pods, lanes, seed)
        sc = SchedCase(nodes=nodes, pods=pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index, lanes=lanes)
        assert_sched_matches(sched_gpu(sc, ctx), sched_oracle(sc), w.name + " lanes"); bump("fuzz_pending_lanes")
    # round 5: the removal loop as one wave over per-class fit masks AND through K_sched (plain clusters: the shape the lean kernel takes)
    w = W.fuzz_removals_plain(seed)
    rc = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                     max_removable=w.max_removable, last_index=w.last_index)
    want = removal_oracle(rc)
    os.environ.pop("CASIM_NO_LEAN_REMOVALS", None)
    assert_removal_matches(removal_device(rc, ctx), want, w.name + " lean"); bump("fuzz_removals_plain_lean")
    os.environ["CASIM_NO_LEAN_REMOVALS"] = "1"
    assert_removal_matches(removal_device(rc, ctx), want, w.name + " K_sched"); bump("fuzz_removals_plain_k_sched")
    os.environ.pop("CASIM_NO_LEAN_REMOVALS", None)
    # round 5, second session: runs of replicas — the one-wave kernel a word of nodes at a time, pod by pod (CASIM_LEAN_BULK_MIN=0), K_sched
    w = W.fuzz_removals_runs(seed)
    rc = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                     max_removable=w.max_removable, last_index=w.last_index, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64)
    want = removal_oracle(rc)
    for tag, env in (("word_at_a_time", {}), ("pod_by_pod", {"CASIM_LEAN_BULK_MIN": "0"}), ("k_sched", {"CASIM_NO_LEAN_REMOVALS": "1"}), ("log_in_hbm", {"CASIM_LEAN_HBM_LOG": "1"})):
        for k in ("CASIM_LEAN_BULK_MIN", "CASIM_NO_LEAN_REMOVALS", "CASIM_LEAN_HBM_LOG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        assert_removal_matches(removal_device(rc, ctx), want, w.name + " " + tag); bump("fuzz_removals_runs_" + tag)
    for k in ("CASIM_LEAN_BULK_MIN", "CASIM_NO_LEAN_REMOVALS", "CASIM_LEAN_HBM_LOG"):
        os.environ.pop(k, None)
    # round 5: batches — chained groups, the ranked orderer forced on, requests narrowed by the caller, PEG rows shared between the tiles
    if seed % 2 == 0:
        from harness import encode_batch, run_emu_tables, run_gpu_tables
        import numpy as np
        scs = [Scenario(pegs=x.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in x.groups], device_csr=True, lanes=x.lanes)
               for x in (W.fuzz(seed * 7 + k, max_groups=6, max_pegs=14) for k in range(2 + seed % 3))]
        if len({sc.lanes for sc in scs}) > 1:
            scs = [scs[0], scs[0]]
        enc, ts, bases = encode_batch(scs)
        chain = seed % 4 == 0
        want = []
        for sc, (pb, _) in zip(scs, bases):
            want.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc, chain=chain)])
        run_t = (lambda t, **kw: run_emu_tables(t, kinds=None, chain=chain, **kw)[0]) if EMU else (lambda t, **kw: run_gpu_tables(t, ctx, kinds=None, chain=chain, **kw)[0])
        os.environ["CASIM_RANK_ONCE"] = "1"
        os.environ["CASIM_RANK_SHARE"] = str(seed % 3 == 0 and 1 or 0)
        res = run_t(ts, narrow_requests=True)
        os.environ.pop("CASIM_RANK_ONCE", None); os.environ.pop("CASIM_RANK_SHARE", None)
        assert_matches_oracle(res, want, f"batch ranked + req32 {seed}"); bump("batch_ranked_req32")
        sh = run_t(ts.tile_groups(2))
        ng = ts.n_groups
        assert list(sh.node_count[:ng]) == list(res.node_count) and list(sh.node_count[ng:]) == list(res.node_count) and \
            list(sh.order[:len(res.order)]) == list(res.order), f"shared PEG rows {seed}"
        bump("batch_shared_peg_rows")
        enc.close()
    # round 6, second session: batches of small estimates on the lean register packer (existing nodes, unschedulable templates, every limiter sign)
    # and C4-shaped batches behind a dry limiter (the anti-affinity packer's record words and its idle steps)
    if seed % 2 == 1:
        from harness import encode_batch, run_emu_tables, run_gpu_tables
        from test_lean_batches_emu import _want, lean_batch
        fast = seed % 4 == 3
        scs = lean_batch(3_000_000 + seed, fast)
        enc, ts, bases = encode_batch(scs)
        res = run_emu_tables(ts, fastpath=fast)[0] if EMU else run_gpu_tables(ts, ctx, fastpath=fast)[0]
        assert_matches_oracle(res, _want(scs, bases), f"lean batch {seed}"); bump("lean_batches")
        enc.close()
        scs = [Scenario(pegs=x.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in x.groups], device_csr=True, lanes=x.lanes)
               for x in (W.config_c4(4_000_000 + seed * 3 + k, n_groups=3, n_pegs=24 + seed % 40, pods_per_peg=1 + seed % 5, cap=2 + seed % 7) for k in range(2))]
        enc, ts, bases = encode_batch(scs)
        res = run_emu_tables(ts)[0] if EMU else run_gpu_tables(ts, ctx)[0]
        assert_matches_oracle(res, _want(scs, bases), f"C4-shaped batch {seed}"); bump("c4_shaped_batches")
        enc.close()
print("stress OK", "(emulator)" if EMU else "(MI355X)", stats, f"{time.time() - t0:.0f} s")
if not EMU:
    ctx.close()
