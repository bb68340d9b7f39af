"""Feature-mix fuzz: product encoder + product kernels (wave emulator) vs the oracle, bit for bit.
Covers taints/tolerations, nodeSelector, host ports (shared / wildcard), hostname and zone
anti-affinity (self and cross), preloaded DaemonSet pods, every limiter sign, lastIndex / existing
nodes, zero requests, ties in the orderer score, fastpath, device-side feasibility + CSR."""
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
