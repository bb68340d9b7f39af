"""K_sched with more than two resource lanes (extended resources: ephemeral-storage, device plugins, ...): the `<.., 8>` instantiations of
sched_kernel — TrySchedulePods, TrySchedulePods with domain rules, and the removal loop — against the oracle, bit for bit.  The fuzz
workloads of the two-lane suites, with extra resources added to every node and to most pod specs (some of them the binding one).
CPU only (wave emulator); the same cases run on the MI355X in test_gpu_sched.py."""
import copy

import pytest

from harness import EmuContext, RemovalCase, SchedCase, assert_removal_matches, assert_sched_matches, removal_device, removal_oracle, sched_emu, sched_oracle
from kubernetes_autoscaler_amd.workloads import SplitMix64, fuzz_pending, fuzz_pending_domains, fuzz_removals

LANES4 = ("cpu", "memory", "ephemeral-storage", "example.com/gpu")
LANES8 = LANES4 + ("example.com/fpga", "hugepages-2Mi", "example.com/nic", "example.com/license")
GiB = 1 << 30


def with_extra_resources(nodes, pods, lanes, seed):
    """deep copies of the workload's objects with allocatable amounts / requests on the lanes beyond cpu and memory; pods that were one
    object (one class) stay one object"""
    rng = SplitMix64(0xA17E5 + seed)
    nodes = copy.deepcopy(nodes)
    memo = {}
    pods = copy.deepcopy(pods, memo)
    amounts = {"ephemeral-storage": [0, 10 * GiB, 100 * GiB], "hugepages-2Mi": [0, 1 * GiB, 4 * GiB]}
    for info in nodes:
        for lane in lanes[2:]:
            v = rng.pick(amounts.get(lane, [0, 1, 2, 4, 8]))
            info.node.allocatable[lane] = v
            info.node.capacity[lane] = v
    seen = set()
    for p in list(pods) + [q for info in nodes for q in info.pods]:
        if id(p) in seen:
            continue
        seen.add(id(p))
        for lane in lanes[2:]:
            if rng.chance(1, 3):
                p.requests[lane] = rng.pick([1 * GiB, 20 * GiB] if lane in amounts else [1, 1, 2, 3])
    return nodes, pods


def remap(w_nodes, new_nodes, seq):
    """lists of NodeInfo objects of the workload (acceptable / destination / candidates are given by index or by object)"""
    return seq


@pytest.mark.parametrize("lanes", [LANES4, LANES8], ids=["4-lanes", "8-lanes"])
@pytest.mark.parametrize("seed", range(60))
def test_try_schedule_pods_with_extended_resources(seed, lanes):
    w = fuzz_pending(seed) if seed % 2 == 0 else fuzz_pending_domains(seed)
    nodes, pods = with_extra_resources(w.nodes, w.pods, lanes, seed)
    case = SchedCase(nodes=nodes, pods=pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index,
                     lanes=lanes)
    want = sched_oracle(case)
    for lds in (0, 64):
        assert_sched_matches(sched_emu(case, lds_budget=lds), want, f"{w.name} {len(lanes)} lanes lds={lds}")


@pytest.mark.parametrize("lanes", [LANES4, LANES8], ids=["4-lanes", "8-lanes"])
@pytest.mark.parametrize("seed", range(40))
def test_removal_loop_with_extended_resources(seed, lanes):
    w = fuzz_removals(seed)
    nodes, _ = with_extra_resources(w.nodes, [], lanes, seed)
    case = RemovalCase(nodes=nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist, max_removable=w.max_removable,
                       last_index=w.last_index, lanes=lanes)
    want = removal_oracle(case)
    for lds in (0, 64):
        assert_removal_matches(removal_device(case, EmuContext(lds)), want, f"{w.name} {len(lanes)} lanes lds={lds}")


@pytest.mark.parametrize("seed", range(30))
def test_uploads_that_do_not_fit_the_packed_slab(seed, monkeypatch):
    """CASIM_TEST_SMALL_UPLOAD_BOUND: the bound of the packed upload is 512 bytes, so most columns of a call take the fallback copy of
    their own (and the call waits for them before its locals die): same results — TrySchedulePods, the removal loop, K_est, Estimate."""
    from harness import GroupSpec, Scenario, assert_cluster_estimate_matches, assert_matches_oracle, cluster_estimate_emu, encode, run_emu, run_oracle
    from kubernetes_autoscaler_amd import workloads
    monkeypatch.setenv("CASIM_TEST_SMALL_UPLOAD_BOUND", "1")
    w = fuzz_pending_domains(40 + seed)
    case = SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)
    assert_sched_matches(sched_emu(case), sched_oracle(case), w.name)
    w = fuzz_removals(40 + seed)
    rc = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist, max_removable=w.max_removable,
                     last_index=w.last_index)
    assert_removal_matches(removal_device(rc, EmuContext(0)), removal_oracle(rc), w.name)
    w = workloads.fuzz_estimate_domains(40 + seed)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes)
    got = cluster_estimate_emu(sc)
    if got[0] == 0:
        est, ids = run_oracle(sc)[0]
        assert_cluster_estimate_matches(got, est, ids, w.name)
    w = workloads.fuzz(3000 + seed)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes,
                  device_csr=True)
    res, _ = run_emu(encode(sc))
    assert_matches_oracle(res, run_oracle(sc), w.name)
