"""-m gpu: round 3 on a real MI355X through the C ABI.
  * the headline three ways are ONE answer: resident step through the context's internal streams (casim_options.n_streams),
    enter -> return (casim_estimate_batch_query, streamed and not) and the int64 packer (force_generic_packer) — bit-equal;
  * streams inside one casim_ctx == the unstreamed problem, also after many resident steps and with a validity mask."""
import os

import numpy as np
import pytest

import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import BatchCall
from kubernetes_autoscaler_amd.tables import TableSet
from harness import GroupSpec, Scenario, assert_matches_oracle, encode, encode_batch, run_gpu_tables, run_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = [_abi.EXPANDER_LEAST_NODES]
FIELDS = ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum")


@pytest.fixture(scope="module")
def ctx():
    c = kaa.Context(0)
    yield c
    c.close()


def _same(a, b, what):
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), (what, f)
    nnz = int(a.offsets[-1])
    assert np.array_equal(a.order[:nnz], b.order[:nnz]) and np.array_equal(a.placed[:nnz], b.placed[:nnz]), what


def _c2_batch(n_seeds, tile):
    from bench import simulation_tables
    ts = simulation_tables(workloads.config_c2, range(n_seeds), kaa.Encoder, TableSet)
    return ts.tile(tile)


def test_headline_rows_are_one_answer(ctx):
    """(a) resident, streams inside libcasim  (b) enter -> return, streamed and unstreamed  (c) int64 packer: bit-equal on a C2
    batch (the headline workload: 6 seeds tiled to 96 simulations = 1920 node groups), and seed 0 equals the oracle."""
    ts = _c2_batch(6, 16)
    pegs, groups = ts.structs()
    base, bexp = run_gpu_tables(ts, ctx, kinds=KINDS)                       # one part, one stream
    for k in (2, 4, 5):
        res, exp = run_gpu_tables(ts, ctx, kinds=KINDS, n_streams=k)        # (a)
        _same(res, base, f"resident k={k}")
        assert list(exp["best"]) == list(bexp["best"]) and list(exp["packed"]) == list(bexp["packed"])
    for k in (0, 4):
        call = BatchCall(ctx, pegs, groups, kinds=KINDS, n_streams=k)       # (b)
        for _ in range(3):
            res, exp = call.call()
        _same(res, base, f"enter-return k={k}")
        assert list(exp["best"]) == list(bexp["best"]) and list(exp["packed"]) == list(bexp["packed"])
    res, exp = run_gpu_tables(ts, ctx, kinds=KINDS, n_streams=4, generic=True)   # (c)
    _same(res, base, "int64")
    assert list(exp["packed"]) == list(bexp["packed"])
    w = workloads.config_c2(0)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)
    want = run_oracle(sc)
    ng = len(w.groups)
    for i, (est, ids) in enumerate(want):
        order, placed = base.group(i)
        assert list(order) == [ids[k] for k in est.order] and list(placed) == list(est.placed)
        assert (int(base.node_count[i]), int(base.pods_scheduled[i])) == (est.node_count, est.pods_scheduled)


def test_streamed_problem_many_resident_steps_and_validity_mask(ctx):
    scs = []
    for k in range(37):
        w = workloads.fuzz(5200 + k, max_groups=5, max_pegs=14)
        scs.append(Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True))
    enc, ts, bases = encode_batch(scs)
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc)])
    rng = np.random.default_rng(3)
    valid = (rng.random(ts.n_groups) < 0.6).astype(np.uint8)
    pegs, groups = ts.structs()
    with kaa.Problem(ctx, pegs, groups) as p:
        p.run(); bexp = p.best_option_sims([_abi.EXPANDER_LEAST_WASTE], valid=valid, n_sims=ts.n_sims)
    with kaa.Problem(ctx, pegs, groups, n_streams=7) as p:
        assert 2 <= p.info()["parts"] <= 7      # (7 asked; the context takes as many lanes as the runtime has concurrent hardware queues for)
        for _ in range(25):
            p.run()
            p.best_option_sims([_abi.EXPANDER_LEAST_WASTE], fetch=False, n_sims=ts.n_sims)
        res = p.fetch()
        exp = p.best_option_sims([_abi.EXPANDER_LEAST_WASTE], valid=valid, n_sims=ts.n_sims)
        with pytest.raises(kaa.CasimError):
            p.best_option([_abi.EXPANDER_LEAST_NODES])        # one reduce over every group: not on a streamed batch
    assert_matches_oracle(res, want, "streamed fuzz batch")
    assert list(exp["best"]) == list(bexp["best"]) and exp["keys"].tolist() == bexp["keys"].tolist() and list(exp["best_set"]) == list(bexp["best_set"])
    enc.close()


def test_register_packer_beyond_its_node_slots_with_generic_retry(ctx):
    """node bounds > 1024: register packer first, generic retry launch for the groups that really overflow (same launch)"""
    w = workloads.config_retry_mix()
    for device_csr in (False, True):
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], device_csr=device_csr)
        enc = encode(sc)
        with kaa.Problem(ctx, enc.pegs, enc.groups) as p:
            assert p.info()["fast_packer_slots_per_lane"] == 16
            p.run(); res = p.fetch()
            p.run(); again = p.fetch()            # resident re-run: the retry marks are rewritten every pass
        oracle = run_oracle(sc)
        assert oracle[0][0].nodes_added < 1024 < oracle[1][0].nodes_added
        assert_matches_oracle(res, oracle, f"retry csr={device_csr}")
        assert_matches_oracle(again, oracle, f"retry again csr={device_csr}")


def test_resident_cluster_iteration_with_domain_rules(ctx):
    """commits on the resident cluster feed the PodTopologySpread / zone anti-affinity / pod-affinity counters of the
    later calls of the iteration (reverted pass, removal loop) — vs the oracle threading one snapshot through the sequence."""
    from harness import resident_iteration
    total = 0
    for seed in range(30):
        w = workloads.fuzz_pending_domains(8300 + seed)
        w.hints = None
        out = resident_iteration(lambda classes, nodes: kaa.ResidentCluster(ctx, classes, nodes), w, with_rules=True)
        total += out["scheduled"]
    assert total > 0


def test_packer_self_check_ran_and_found_the_two_builds_identical(ctx):
    """libcasim carries the register packer twice (with / without the experimental structurizer option) and
    compares the two on a built-in corpus before the first context of the process is handed out."""
    info = ctx.pack_build_info()
    if info["forced_by_env"]:
        pytest.skip("CASIM_PACK_BUILD forces a build")
    # lazy since round 4: every instantiation this process has launched so far went through its 18 (16 without exclusion words) case families
    ts = _c2_batch(1, 2)                 # (kept alive: the structs point into its arrays)
    pegs, groups = ts.structs()
    with kaa.Problem(ctx, pegs, groups) as p:
        p.run(); p.fetch()
    info = ctx.pack_build_info()
    assert info["batches_compared"] >= 15 and info["batches_differing"] == 0 and info["build"] == "option", info


def test_packer_self_check_corpus_runs_everything_it_generates_in_time():
    """every batch of the grown corpus (fastpath, singleton runs, zone words with NEED polarity, caller lists, PEGs
    outside the simple shape; alone and combined) must be RUNNABLE by the plain build — a batch it refuses is not compared — and the whole
    check has to stay a start-up cost nobody notices (own process: the verdict is per process)."""
    import re, subprocess, sys
    code = "import sys; sys.path.insert(0, %r)\nimport kubernetes_autoscaler_amd as kaa\nctx = kaa.Context(0)\nprint(ctx.pack_build_info())\n" % ROOT
    env = dict(os.environ); env["CASIM_PACK_SELFCHECK_VERBOSE"] = "1"; env["CASIM_PACK_SELFCHECK"] = "eager"; env.pop("CASIM_PACK_BUILD", None)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    ms_ = re.findall(r"packer self-check: (\d+) batches compared, (\d+) differing, (\d+) not runnable, ([0-9.]+) ms", p.stderr)
    assert len(ms_) == 18, p.stderr[-2000:]     # (one line per instantiation — 2 / 4 int32 lanes, 2 int64 lanes — cumulative figures)
    compared, differing, skipped, ms = int(ms_[-1][0]), int(ms_[-1][1]), int(ms_[-1][2]), float(ms_[-1][3])
    print(f"self-check, eager: {compared} batches, {ms:.1f} ms")
    assert compared == 306 and differing == 0 and skipped == 0
    assert ms < 900.0   # (includes the first launches of 36 kernel instantiations: code-object loading, not compute)


def test_packer_self_check_is_lazy_and_cheap_at_start_up():
    """the default: a context costs nothing; the first problem that needs an instantiation checks THAT one (16-18 batches, both builds) — a
    process that runs C2 batches pays for one instantiation, not for twelve (< 50 ms at start-up)"""
    import re, subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport kubernetes_autoscaler_amd as kaa\nfrom kubernetes_autoscaler_amd import workloads\n"
            "from harness import GroupSpec, Scenario, encode\nctx = kaa.Context(0)\nprint('before', ctx.pack_build_info())\n"
            "w = workloads.CONFIGS['C2']()\nsc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=True)\n"
            "enc = encode(sc)\nfor _ in range(2):\n    with kaa.Problem(ctx, enc.pegs, enc.groups) as p:\n        p.run(); p.fetch()\nprint('after', ctx.pack_build_info())\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ); env["CASIM_PACK_SELFCHECK_VERBOSE"] = "1"; env.pop("CASIM_PACK_BUILD", None); env.pop("CASIM_PACK_SELFCHECK", None)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "'batches_compared': 0" in p.stdout.split("after")[0]          # nothing ran when the context came up
    ms_ = re.findall(r"packer self-check: (\d+) batches compared, (\d+) differing, (\d+) not runnable, ([0-9.]+) ms", p.stderr)
    assert len(ms_) == 1, p.stderr[-2000:]                                  # ONE instantiation, checked once (the second problem found it done)
    compared, differing, skipped, ms = int(ms_[0][0]), int(ms_[0][1]), int(ms_[0][2]), float(ms_[0][3])
    print(f"self-check, lazy: {compared} batches, {ms:.1f} ms")
    assert compared == 16 and differing == 0 and skipped == 0 and ms < 50.0


def test_both_packer_builds_agree_with_the_oracle(ctx):
    """casim_options.pack_build: the plain and the option build of every instantiation the fuzz families reach, both against the oracle."""
    for seed in range(60):
        w = workloads.fuzz(7000 + seed, max_groups=3, max_pegs=260, rich=seed % 2 == 0)
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes,
                      device_csr=seed % 3 == 0)
        enc = encode(sc)
        want = run_oracle(sc)
        for build in (_abi.PACK_BUILD_PLAIN, _abi.PACK_BUILD_OPTION):
            with kaa.Problem(ctx, enc.pegs, enc.groups, pack_build=build) as p:
                p.run(); res = p.fetch()
            assert_matches_oracle(res, want, f"seed {seed} build {build}")
        enc.close()
    w = workloads.CONFIGS["C2"]()
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=True)
    enc = encode(sc); want = run_oracle(sc)
    for build in (_abi.PACK_BUILD_PLAIN, _abi.PACK_BUILD_OPTION):
        with kaa.Problem(ctx, enc.pegs, enc.groups, pack_build=build) as p:
            assert p.info()["fast_packer_slots_per_lane"] > 0
            p.run(); res = p.fetch()
        assert_matches_oracle(res, want, f"C2 build {build}")
    enc.close()


def test_a_failing_self_check_retires_the_option_build():
    """CASIM_PACK_SELFCHECK_FAULT=1 flips one word of the option build's results inside the comparison: the process must come up on the
    plain build, say so, and still match the oracle (own process: the verdict is per process and device)."""
    import json, subprocess, sys
    code = (
        "import json, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import kubernetes_autoscaler_amd as kaa\n"
        "from kubernetes_autoscaler_amd import workloads\n"
        "from harness import GroupSpec, Scenario, encode, run_oracle, assert_matches_oracle\n"
        "ctx = kaa.Context(0)\n"
        "w = workloads.CONFIGS['C1']()\n"
        "sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes)\n"
        "enc = encode(sc)\n"
        "with kaa.Problem(ctx, enc.pegs, enc.groups) as p:\n"
        "    p.run(); res = p.fetch(); fast = p.info()['fast_packer_slots_per_lane']\n"
        "assert_matches_oracle(res, run_oracle(sc), 'C1 on the plain build')\n"
        "print(json.dumps({'info': ctx.pack_build_info(), 'fast': fast}))\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ); env["CASIM_PACK_SELFCHECK_FAULT"] = "1"; env.pop("CASIM_PACK_BUILD", None)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["info"]["build"] == "plain" and out["info"]["batches_differing"] == 1 and out["fast"] > 0, out
    assert "retired for this process" in p.stderr


def test_runs_of_identical_singleton_pegs_on_the_device(ctx):
    """SingletonRuns (csrc/casim_pipeline.h): adjacent identical controller-less pods estimated as one row — equal to the oracle, which
    estimates every singleton PEG on its own, and to the unmerged run; BenchmarkRunOnceScaleUp's 10 000 singletons at full size."""
    from kubernetes_autoscaler_amd.engine import Problem
    for seed in range(80):
        w = workloads.fuzz_singleton_runs(seed)
        for dcsr in (False, True):
            sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=dcsr)
            enc = encode(sc); want = run_oracle(sc)
            for generic in (False, True):
                with Problem(ctx, enc.pegs, enc.groups, force_generic_packer=generic) as p:
                    p.run(); res = p.fetch()
                assert_matches_oracle(res, want, f"singleton runs {seed} csr={dcsr} generic={generic}")
            enc.close()
    w = workloads.CONFIGS["R1"]()
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=True)
    enc = encode(sc); want = run_oracle(sc)
    with Problem(ctx, enc.pegs, enc.groups) as p:
        p.run(); res = p.fetch()
        tot, kms = p.time(iters=5)
    assert_matches_oracle(res, want, "R1, 10 000 singleton PEGs as one row")
    assert int(res.nodes_added[0]) == 200 and tot < 2.0, (tot, kms)     # (14.5 ms as 10 000 dependent steps)
    enc.close()


def test_affinity_terms_with_namespace_selectors_on_the_device(ctx):
    """VERDICT r2 missing #4 (second half): required pod-affinity terms with a namespaceSelector inside TrySchedulePods on the MI355X"""
    from kubernetes_autoscaler_amd.objects import namespaces
    from kubernetes_autoscaler_amd.workloads import add_random_pod_affinity
    from harness import SchedCase, assert_sched_matches, sched_gpu, sched_oracle
    seen = 0
    for seed in range(120):
        w = workloads.fuzz_pending_domains(7600 + seed)
        everybody = w.pods + [p for info in w.nodes for p in info.pods]
        add_random_pod_affinity(seed, everybody, frac=0.6)
        table = workloads.add_random_namespace_selectors(seed, everybody)
        seen += sum(1 for p in everybody for t in p.affinity if t.namespace_selector is not None)
        with namespaces(table):
            case = SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)
            assert_sched_matches(sched_gpu(case, ctx), sched_oracle(case), w.name)
    assert seen > 100


def test_static_pod_affinity_in_template_mode_on_the_device(ctx):
    """required pod affinity whose verdict is fixed by the existing cluster (non-hostname keys, no partner inside the batch) stays in
    casim_estimate_batch: both packers vs the oracle"""
    from test_pod_affinity_emu import _static_affinity_workload
    from kubernetes_autoscaler_amd.engine import Problem
    checked = 0
    for seed in range(100):
        w, n_aff = _static_affinity_workload(seed)
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=seed % 2 == 0)
        enc = encode(sc)
        for generic in (False, True):
            with Problem(ctx, enc.pegs, enc.groups, force_generic_packer=generic) as p:
                p.run(); res = p.fetch()
            if any(int(s) != 0 for s in res.status):
                continue
            assert_matches_oracle(res, run_oracle(sc), f"static affinity {seed} generic={generic}")
            checked += 1
        enc.close()
    assert checked > 120


def test_pod_affinity_towards_partners_of_the_batch_on_the_device(ctx):
    """zone-level required pod affinity whose partner is another PEG of the batch (or the PEG itself): group bits of NEED polarity
    (casim_pegs.zone_polarity) in K_feas and in both packers, the caller's lists and device-derived lists, vs the oracle"""
    from test_pod_affinity_emu import _batch_affinity_workload
    from kubernetes_autoscaler_amd.engine import Problem
    from kubernetes_autoscaler_amd.objects import LABEL_ZONE
    checked = with_need = 0
    for seed in range(160):
        w, n_aff = _batch_affinity_workload(seed, keys=(LABEL_ZONE, LABEL_ZONE, "pool-0", "pool"))
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=seed % 2 == 0)
        enc = encode(sc)
        need = bool(enc.pegs.zone_polarity) and enc.pegs.w_zone > 0 and any(int(enc.pegs.zone_polarity[k]) for k in range(enc.pegs.w_zone))
        for generic in (False, True):
            with Problem(ctx, enc.pegs, enc.groups, force_generic_packer=generic) as p:
                p.run(); res = p.fetch()
            if any(int(s) != 0 for s in res.status):
                continue
            assert_matches_oracle(res, run_oracle(sc), f"batch affinity {seed} generic={generic}")
            checked += 1; with_need += 1 if need else 0
        enc.close()
    assert checked > 200 and with_need > 40


def test_front_kernel_equals_the_separate_launches_and_the_oracle(ctx):
    """A call of <= 1024 groups runs feasibility, list offsets, lists and PEG order as ONE launch (front_kernel: the blocks hand each
    other their counts through device-scope release / acquire words — across XCDs on the real chip).  Same bits as the four
    separate launches (casim_options.no_front_kernel) and as the oracle: feature-mix fuzz, long rows, 200 groups with empty rows,
    three runs of one resident problem (the epoch of the tickets), the baseline configs C0-C2."""
    from kubernetes_autoscaler_amd.engine import Problem
    from kubernetes_autoscaler_amd.objects import NodeInfo, Pod, PodEquivalenceGroup
    from kubernetes_autoscaler_amd.workloads import _node, SplitMix64

    def both(sc, fastpath=False, generic=False, runs=1):
        enc = encode(sc)
        out = []
        for no_front in (False, True):
            with Problem(ctx, enc.pegs, enc.groups, fastpath, generic, no_front_kernel=no_front) as p:
                assert p.info()["front_kernel"] == (not no_front)
                for _ in range(runs):
                    p.run()
                out.append(p.fetch())
        _same(out[0], out[1], "front vs separate")
        enc.close()
        return out[0]

    def scenario(w, fastpath=False):
        return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups],
                        existing=w.existing, lanes=w.lanes, fastpath=fastpath, device_csr=True)

    for seed in range(60):
        fast = seed % 3 == 1
        sc = scenario(workloads.fuzz(3000 + seed), fastpath=fast)
        assert_matches_oracle(both(sc, fastpath=fast, generic=seed % 3 == 2, runs=1 + 2 * (seed % 5 == 0)), run_oracle(sc), f"fuzz {seed}")
    for seed in range(12):
        sc = scenario(workloads.fuzz(7000 + seed, max_groups=3, max_pegs=260, rich=seed % 2 == 0))
        assert_matches_oracle(both(sc), run_oracle(sc), f"long rows {seed}")
    rng = SplitMix64(0xF207)
    pegs = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": 100 * (1 + rng.below(8)), "memory": (128 << 20) * (1 + rng.below(6))},
                                          node_selector=({"pool": f"p{rng.below(5)}"} if rng.chance(2, 3) else {}))] * (1 + rng.below(5))) for i in range(150)]
    groups = [GroupSpec(NodeInfo(_node(f"g{k}", 50 if k % 9 == 4 else 1000 * (1 + rng.below(8)), (1 + rng.below(16)) << 30, 30,
                                       {"pool": f"p{rng.below(7)}"} if rng.chance(3, 4) else {})), max_nodes=rng.pick([0, 2, 5, 20]), last_index=0, pegs=None)
              for k in range(200)]
    sc = Scenario(pegs=pegs, groups=groups, device_csr=True)
    assert_matches_oracle(both(sc, runs=3), run_oracle(sc), "200 groups")
    for cfg in (workloads.config_c0, workloads.config_c1, workloads.config_c2):
        w = cfg()
        sc = scenario(w)
        assert_matches_oracle(both(sc), run_oracle(sc), w.name)
    # blocks that do not wait: with a poll limit of 0 every block counts the rows in front of it itself (what a block does on a
    # saturated chip when a predecessor has not published in time)
    os.environ["CASIM_FRONT_SPIN"] = "0"
    try:
        assert_matches_oracle(both(Scenario(pegs=pegs, groups=groups[:90], device_csr=True), runs=2), run_oracle(Scenario(pegs=pegs, groups=groups[:90], device_csr=True)),
                              "90 groups, recount")
        for seed in range(30):
            sc = scenario(workloads.fuzz(3000 + seed))
            assert_matches_oracle(both(sc), run_oracle(sc), f"recount fuzz {seed}")
        w = workloads.config_c2()
        assert_matches_oracle(both(scenario(w)), run_oracle(scenario(w)), "C2 recount")
    finally:
        del os.environ["CASIM_FRONT_SPIN"]


def test_one_call_with_expander_waits_for_the_device_once(ctx):
    """casim_estimate_batch_query on a single simulation: the upload is not waited for on its own, the expander's answer travels inside
    the results slab and the offsets come with the fetch — ONE wait per call.  Winner, survivor set, key block, packed key, results and
    offsets equal the resident problem's (casim_problem_* with a wait per step) and the oracle's, call after call on one context."""
    import ctypes as C
    from kubernetes_autoscaler_amd.engine import Problem, alloc_results, finish_results, _ptr
    from kubernetes_autoscaler_amd._ffi import lib

    def scenario(w):
        return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups],
                        existing=w.existing, lanes=w.lanes, device_csr=True)

    chains = ([_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_MOST_PODS, _abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_LEAST_NODES, _abi.EXPANDER_MOST_PODS])
    for seed in range(45):
        w = workloads.fuzz(3000 + seed) if seed % 3 else workloads.fuzz(5000 + seed, max_groups=4, max_pegs=48)
        sc = scenario(w)
        enc = encode(sc)
        kinds = chains[seed % 3]
        with Problem(ctx, enc.pegs, enc.groups) as p:
            p.run()
            want = p.fetch()
            wb = p.best_option_sims(kinds, per_sim=False)
        ng = enc.groups.n_groups
        st, arrs = alloc_results(ng, ng * enc.pegs.n_pegs)
        off = np.zeros(ng + 1, np.int32)
        ks = (C.c_int32 * len(kinds))(*kinds)
        got = dict(best=np.full(1, -1, np.int32), n_best=np.zeros(1, np.int32), best_set=np.full(max(ng, 1), 7, np.uint8), keys=np.zeros((1, 10), np.int64),
                   packed=np.zeros(1, np.int64))
        q = _abi.OptionQuery(kinds=ks, n_kinds=len(kinds), best_out=_ptr(got["best"], C.c_int32), n_best_out=_ptr(got["n_best"], C.c_int32),
                             best_set_out=_ptr(got["best_set"], C.c_uint8), key_out=_ptr(got["keys"], C.c_int64), packed_out=_ptr(got["packed"], C.c_int64))
        opts = _abi.Options()
        for rep in range(2):   # (the second call reuses the pooled blocks and staging of the first)
            assert lib.casim_estimate_batch_query(ctx._h, C.byref(enc.pegs), C.byref(enc.groups), C.byref(opts), C.byref(st), _ptr(off, C.c_int32), C.byref(q)) == 0
            res = finish_results(arrs, ng, int(off[ng]), off.copy())
            _same(res, want, f"seed {seed} call {rep}")
            assert int(got["best"][0]) == int(wb["best"][0]) and int(got["n_best"][0]) == int(wb["n_best"][0]), seed
            assert np.array_equal(got["best_set"][:ng], wb["best_set"][:ng]) and np.array_equal(got["keys"], wb["keys"]) and np.array_equal(got["packed"], wb["packed"]), seed
        assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")
        enc.close()
