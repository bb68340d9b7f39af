"""Feature-mix fuzz: product encoder + product kernels (wave emulator) vs the oracle, bit for bit.
Covers taints/tolerations, nodeSelector, host ports (shared / wildcard), hostname and zone
anti-affinity (self and cross), preloaded DaemonSet pods, every limiter sign, lastIndex / existing
nodes, zero requests, ties in the orderer score, fastpath, device-side feasibility + CSR."""
import numpy as np
import pytest

from harness import GroupSpec, Scenario, assert_matches_oracle, encode, run_emu, run_oracle
from kubernetes_autoscaler_amd import workloads


def scenario_of(w, fastpath=False, device_csr=False):
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups],
                    existing=w.existing, lanes=w.lanes, fastpath=fastpath, device_csr=device_csr)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_plain(seed):
    sc = scenario_of(workloads.fuzz(seed, rich=False))
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(150))
def test_fuzz_rich(seed):
    sc = scenario_of(workloads.fuzz(1000 + seed))
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("rich", [False, True], ids=["plain", "rich"])
@pytest.mark.parametrize("seed", range(40))
def test_fuzz_long_lists(seed, rich):
    """Up to 260 PEGs per group (several 64-record chunks), limits of 1 / 3 / 7 nodes common: the register packer's loop behind
    a dry limiter across chunk boundaries, with (rich) and without exclusion state."""
    sc = scenario_of(workloads.fuzz(7000 + seed, max_groups=3, max_pegs=260, rich=rich))
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_fastpath(seed):
    sc = scenario_of(workloads.fuzz(2000 + seed), fastpath=True)
    res, _ = run_emu(encode(sc), fastpath=True)
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_device_csr(seed):
    """peg_offsets == NULL: the feasibility kernel + CSR compaction derive SchedulablePodGroups."""
    sc = scenario_of(workloads.fuzz(3000 + seed), device_csr=True)
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(20))
def test_fuzz_hbm_scratch(seed):
    """Force the HBM-scratch variants of the order and pack kernels (tiny LDS budget)."""
    sc = scenario_of(workloads.fuzz(4000 + seed))
    res, _ = run_emu(encode(sc), lds_budget=256)
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_many_pegs(seed):
    """More PEGs per group: several 64-node slots per sweep, long a2 rounds, binary search on T."""
    sc = scenario_of(workloads.fuzz(5000 + seed, max_groups=3, max_pegs=48))
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_plain_generic_packer(seed):
    """Same resource-only scenarios through the MemStore (int64, LDS) packer instead of the register one."""
    sc = scenario_of(workloads.fuzz(seed, rich=False))
    enc = encode(sc)
    res_fast, _ = run_emu(enc)
    res_gen, _ = run_emu(enc, generic=True)
    assert_matches_oracle(res_gen, run_oracle(sc), f"seed {seed}")
    for f in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "req_cpu_sum", "req_mem_sum", "order", "placed"):
        assert (getattr(res_fast, f) == getattr(res_gen, f)).all(), f


@pytest.mark.parametrize("seed", range(30))
def test_fuzz_fast_packer_shapes(seed):
    """Resource-only scenarios sized to hit every register-packer instantiation (64 / 256 / 1024 nodes,
    2 and 4 resource lanes) and odd gcds."""
    from kubernetes_autoscaler_amd.objects import NodeInfo, Node, Pod, PodEquivalenceGroup
    rng = workloads.SplitMix64(9000 + seed)
    lanes = ("cpu", "memory") if seed % 2 == 0 else ("cpu", "memory", "ephemeral-storage", "example.com/gpu")
    unit = rng.pick([1, 7, 1000, 1 << 20, 3 * (1 << 26)])
    cap_nodes = rng.pick([5, 64, 65, 200, 256, 257, 700, 1024])
    alloc = {"cpu": 1000 * rng.pick([4, 16, 64]), "memory": unit * rng.pick([64, 1000, 4096]), "pods": rng.pick([8, 30, 110])}
    if len(lanes) == 4:
        alloc["ephemeral-storage"] = 100 * unit
        alloc["example.com/gpu"] = rng.pick([0, 4, 8])
    node = Node(name=f"shape{seed}", labels={}, allocatable=dict(alloc), capacity=dict(alloc))
    pegs = []
    for i in range(1 + rng.below(40)):
        req = {"cpu": 50 * rng.below(60), "memory": unit * rng.below(40)}
        if len(lanes) == 4:
            req["ephemeral-storage"] = unit * rng.below(5)
            req["example.com/gpu"] = rng.pick([0, 0, 0, 1, 2])
        pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests=req)] * rng.pick([1, 3, 50, 400])))
    sc = Scenario(pegs=pegs, groups=[GroupSpec(NodeInfo(node), max_nodes=cap_nodes, last_index=rng.below(5))],
                  existing=[], lanes=lanes)
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(60))
def test_host_loops_cut_over_threads(seed, monkeypatch):
    """The gcd pass / int32 tables / staging copies of ProblemT::init run on up to four host threads for million-PEG batches;
    CASIM_HOST_GRAIN makes them do so on fuzz-sized tables: same results (lanes with every kind of gcd: the shapes family)."""
    monkeypatch.setenv("CASIM_HOST_GRAIN", "3")
    if seed < 30:
        test_fuzz_fast_packer_shapes(seed)
    else:
        sc = scenario_of(workloads.fuzz(1000 + seed))
        res, _ = run_emu(encode(sc))
        assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(60))
def test_gcd_and_int32_tables_on_the_device(seed, monkeypatch):
    """Round 4: for big request tables the gcd fold and the int32 quotients run on the DEVICE (gcd_reduce_kernel / scale_requests_kernel,
    casim_kernels.h: the host pass was the longest stage of an enter -> return call); CASIM_DEV_GCD_MIN=1 sends fuzz-sized tables down
    that path: same results as the oracle — lanes with every kind of gcd (the shapes family), lanes that do not narrow at all (generic
    packer), 3- and 4-lane tables."""
    monkeypatch.setenv("CASIM_DEV_GCD_MIN", "1")
    if seed < 30:
        test_fuzz_fast_packer_shapes(seed)
    else:
        sc = scenario_of(workloads.fuzz(1000 + seed))
        res, _ = run_emu(encode(sc))
        assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


def test_device_gcd_gives_the_tables_of_the_host_pass(monkeypatch):
    """the same batch of simulations through both forms of the pass: every result array identical, and the register packer is chosen
    either way (the verdict on the gcd-scaled ranges is the same)"""
    import kubernetes_autoscaler_amd as kaa
    from kubernetes_autoscaler_amd.tables import TableSet
    from harness import run_emu_tables
    import bench
    ts = bench.simulation_tables(workloads.config_c2, range(2), kaa.Encoder, TableSet)
    a, _ = run_emu_tables(ts)
    monkeypatch.setenv("CASIM_DEV_GCD_MIN", "1")
    b, _ = run_emu_tables(ts)
    for f in ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum", "order", "placed"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
