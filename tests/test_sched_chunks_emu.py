"""K_sched with one, two and four waves (CASIM_SCHED_THREADS) on clusters of several hundred nodes, so that runs cross chunk
boundaries of the node range, wrap around the list and reach rounds 2.. in every workgroup shape — f1, f4 and the resident
iteration vs the object-level CPU oracle, bit for bit, node store in LDS and in the HBM slab.  CPU only (wave emulator)."""
import pytest

from harness import EmuCluster, EmuContext, RemovalCase, SchedCase, assert_removal_matches, assert_sched_matches, removal_device, removal_oracle, \
    resident_iteration, sched_emu, sched_oracle
from kubernetes_autoscaler_amd import workloads


def sched_case(w):
    return SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)


def removal_case(w):
    return RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                       max_removable=w.max_removable, last_index=w.last_index)


@pytest.fixture(params=[64, 128, 256])
def small_workgroup(request, monkeypatch):
    monkeypatch.setenv("CASIM_SCHED_THREADS", str(request.param))
    return request.param


@pytest.mark.parametrize("seed", range(60))
def test_pending_pods_across_chunks(seed, small_workgroup):
    w = workloads.fuzz_pending(7000 + seed, max_nodes=420, max_pods=300)
    case = sched_case(w)
    want = sched_oracle(case)
    for lds in (0, 64):
        assert_sched_matches(sched_emu(case, lds_budget=lds), want, f"{w.name} T={small_workgroup} lds={lds}")


@pytest.mark.parametrize("seed", range(60))
def test_pending_pods_with_domain_rules_across_chunks(seed, small_workgroup):
    w = workloads.fuzz_pending_domains(7000 + seed, max_nodes=300, max_pods=160)
    case = sched_case(w)
    want = sched_oracle(case)
    for lds in (0, 64):
        assert_sched_matches(sched_emu(case, lds_budget=lds), want, f"{w.name} T={small_workgroup} lds={lds}")


@pytest.mark.parametrize("seed", range(60))
def test_removal_loop_across_chunks(seed, small_workgroup):
    w = workloads.fuzz_removals(7000 + seed, max_nodes=300)
    case = removal_case(w)
    want = removal_oracle(case)
    for lds in (0, 64):
        assert_removal_matches(removal_device(case, EmuContext(lds)), want, f"{w.name} T={small_workgroup} lds={lds}")


@pytest.mark.parametrize("seed", range(40))
def test_removal_loop_with_domain_rules_across_chunks(seed, small_workgroup):
    w = workloads.fuzz_removals_domains(7000 + seed, max_nodes=260)
    case = removal_case(w)
    want = removal_oracle(case)
    for lds in (0, 64):
        assert_removal_matches(removal_device(case, EmuContext(lds)), want, f"{w.name} T={small_workgroup} lds={lds}")


def test_scale_shapes_with_a_small_workgroup(small_workgroup):
    w = workloads.pending_scale(700, 5000, 24, 11)
    case = sched_case(w)
    assert_sched_matches(sched_emu(case, lds_budget=64), sched_oracle(case), w.name)
    w = workloads.removal_scale(900, pods_per_node=4, frac_candidates=0.1, seed=12)
    case = removal_case(w)
    assert_removal_matches(removal_device(case, EmuContext(64)), removal_oracle(case), w.name)


@pytest.mark.parametrize("seed", range(12))
def test_resident_iteration_across_chunks(seed, small_workgroup):
    w = workloads.fuzz_pending(7100 + seed, max_nodes=300, max_pods=200)
    for lds in (0, 64):
        resident_iteration(lambda c, n, b=lds: EmuCluster(c, n, lds_budget=b), w)
