"""-m gpu: round 3 on a real MI355X through the C ABI.
  * the headline three ways are ONE answer: resident step through the context's internal streams (casim_options.n_streams),
    enter -> return (casim_estimate_batch_query, streamed and not) and the int64 packer (force_generic_packer) — bit-equal;
  * streams inside one casim_ctx == the unstreamed problem, also after many resident steps and with a validity mask."""
import numpy as np
import pytest

import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import BatchCall
from kubernetes_autoscaler_amd.tables import TableSet
from harness import GroupSpec, Scenario, assert_matches_oracle, encode, encode_batch, run_gpu_tables, run_oracle

pytestmark = pytest.mark.gpu
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
        assert p.info()["parts"] == 7
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
