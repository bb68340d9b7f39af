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
                                    (5000, 60, dict(max_groups=3, max_pegs=48), False, False)):
        for seed in range(n):
            sc = scenario_of(workloads.fuzz(base + seed, **kw), fastpath=fast, device_csr=dcsr)
            res, _ = run_gpu(encode(sc), ctx, fastpath=fast)
            assert_matches_oracle(res, run_oracle(sc), f"fuzz {base + seed}")


@pytest.mark.parametrize("name", ["C0", "C1", "C2", "C3", "C4"])
def test_baseline_configs_full_size(ctx, name):
    w = workloads.CONFIGS[name]()
    sc = scenario_of(w, device_csr=name in ("C2", "C3", "C4"))
    res, _ = run_gpu(encode(sc), ctx)
    assert_matches_oracle(res, run_oracle(sc), name)


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
