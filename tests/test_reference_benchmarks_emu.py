"""The reference's OWN benchmark regimes for the estimator, as parity-tested workloads:

  R1  BenchmarkRunOnceScaleUp      CA/core/bench/benchmark_runonce_test.go:395-418,493-503 — 10 000 controller-less pods =
      10 000 singleton PodEquivalenceGroups -> one node group, verifyTargetSize(200)
  R2  BenchmarkBinpackingEstimate  CA/estimator/binpacking_estimator_test.go:256-303 — 2595 nodes / 51 000 pods

plus fuzz with 2 000 - 20 000 PEGs per group: order_kernel's HBM-slab sort (list bound > 1024 entries) and the packer's
per-PEG loop at the length the reference benchmarks.  Product kernels under the wave emulator vs the oracle, bit for bit."""
import pytest

from harness import GroupSpec, Scenario, assert_matches_oracle, encode, run_emu, run_oracle
from kubernetes_autoscaler_amd import workloads


def scenario_of(w, device_csr=False):
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups],
                    existing=w.existing, lanes=w.lanes, device_csr=device_csr)


@pytest.mark.parametrize("device_csr", [False, True], ids=["given-lists", "device-csr"])
def test_r1_benchmark_run_once_scale_up(device_csr):
    w = workloads.config_r1()
    assert len(w.pegs) == 10000 and all(len(pg.pods) == 1 for pg in w.pegs)
    sc = scenario_of(w, device_csr)
    oracle = run_oracle(sc)
    assert (oracle[0][0].node_count, oracle[0][0].pods_scheduled) == (200, 10000)   # verifyTargetSize(200)
    res, _ = run_emu(encode(sc))
    assert (int(res.node_count[0]), int(res.pods_scheduled[0])) == (200, 10000)
    assert_matches_oracle(res, oracle, "R1")


@pytest.mark.parametrize("nodes", [1, 7, 60])
def test_r1_smaller_scale_ups(nodes):
    """the same scenario for other target sizes (setupScaleUp(nodes)), generic and register packer"""
    sc = scenario_of(workloads.config_r1(nodes, max_ng_size=1000))
    oracle = run_oracle(sc)
    assert oracle[0][0].node_count == nodes
    enc = encode(sc)
    for generic in (False, True):
        res, _ = run_emu(enc, generic=generic)
        assert_matches_oracle(res, oracle, f"R1 {nodes} nodes generic={generic}")


@pytest.mark.parametrize("generic", [False, True], ids=["default", "generic-packer"])
def test_r2_benchmark_binpacking_estimate(generic):
    sc = scenario_of(workloads.config_r2())
    oracle = run_oracle(sc)
    assert (oracle[0][0].node_count, oracle[0][0].pods_scheduled) == (2595, 51000)
    res, _ = run_emu(encode(sc), generic=generic)
    assert (int(res.node_count[0]), int(res.pods_scheduled[0])) == (2595, 51000)
    assert_matches_oracle(res, oracle, "R2")


@pytest.mark.parametrize("n_pegs,cap,seed,kw", [
    (2000, 64, 0, {}), (2049, 200, 1, {}), (5000, 256, 2, {}), (5000, 1000, 3, dict(lds_budget=4096)),
    (8191, 300, 4, dict(generic=True)), (12000, 700, 5, {}), (20000, 1024, 6, {}), (9000, 5000, 7, {}),
], ids=lambda v: str(v) if not isinstance(v, dict) else ("+".join(v) or "default"))
def test_fuzz_thousands_of_pegs_in_one_group(n_pegs, cap, seed, kw):
    """2 000 - 20 000 PEGs in ONE group, scores tying freely: the sort runs in the HBM slab (npad > 1024), the packer walks
    thousands of dependent steps through every store (registers <= 1024 nodes, LDS, HBM slab beyond)."""
    sc = scenario_of(workloads.config_many_pegs(seed, n_pegs, cap))
    res, _ = run_emu(encode(sc), **kw)
    assert_matches_oracle(res, run_oracle(sc), f"many {n_pegs} cap {cap}")


def test_fuzz_thousands_of_pegs_device_csr_two_groups():
    """two groups sharing 3000 PEGs, lists derived on the device (feasibility rows of 47 words -> the unfolded CSR count)"""
    w = workloads.config_many_pegs(11, 3000, 128)
    from kubernetes_autoscaler_amd.objects import NodeInfo
    from kubernetes_autoscaler_amd.workloads import GiB, GroupPlan, _node
    w.groups.append(GroupPlan(NodeInfo(_node("many-small", 2000, 8 * GiB, 30)), max_nodes=90, last_index=3))
    sc = scenario_of(w, device_csr=True)
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), "many device csr")


def test_distinct_scores_fails_loudly_when_the_draw_space_is_exhausted():
    with pytest.raises(ValueError, match="distinct scores"):
        workloads.config_c1(n_pegs=30000)


def test_register_packer_with_node_bounds_beyond_its_slots_and_the_generic_retry():
    """Node BOUNDS above the register packer's 1024 slots no longer send a batch to the int64 packer: the groups start in the
    register packer, and only those that really create a 1025th node are packed again by the generic packer's retry launch.
    One launch with both kinds: a roomy template (tens of nodes) and a tiny one (> 1024 nodes), unlimited and limited."""
    w = workloads.config_retry_mix()
    for device_csr in (False, True):
        sc = scenario_of(w, device_csr)
        oracle = run_oracle(sc)
        assert oracle[0][0].nodes_added < 1024 < oracle[1][0].nodes_added and oracle[2][0].nodes_added == 1500
        res, _ = run_emu(encode(sc))
        assert_matches_oracle(res, oracle, f"retry csr={device_csr}")
