"""The removal loop as ONE wave over per-class fit masks (csrc/casim_sched.h removals_lean_kernel, round 5) under the wave emulator:
every case through the lean kernel AND through K_sched's general transaction loop (CASIM_NO_LEAN_REMOVALS=1), both against the oracle —
removable flag per candidate, destination of every pod, the pods listed again (ext), lastIndex, candidates decided.
The shapes the lean kernel takes: no domain rules, no node-local exclusion words, <= 64 classes, <= 4 lanes, state within the LDS budget;
everything else keeps K_sched (checked here too: the info word says which kernel ran)."""
import ctypes as C

import pytest

from harness import EmuContext, RemovalCase, assert_removal_matches, emu_lib, removal_device, removal_oracle
from kubernetes_autoscaler_amd.objects import NodeInfo, Pod, build_test_pod
from kubernetes_autoscaler_amd.workloads import GiB, MiB, _node, fuzz_removals, fuzz_removals_plain, fuzz_removals_runs, removal_scale, runonce_scale_down


def last_kernel():
    info = (C.c_int32 * 4)()
    emu_lib().emu_last_removals_info(info)
    return list(info)


def both(case, what="", monkeypatch=None, expect_lean=True):
    want = removal_oracle(case)
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    got = removal_device(case, EmuContext(0))
    lean = last_kernel()[0]
    assert_removal_matches(got, want, f"{what} lean kernel")
    monkeypatch.setenv("CASIM_NO_LEAN_REMOVALS", "1")
    got = removal_device(case, EmuContext(0))
    assert last_kernel()[0] == 0
    assert_removal_matches(got, want, f"{what} K_sched")
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    if expect_lean is not None and want["n_processed"] + len(want["removable"]) > 0:
        assert lean == (1 if expect_lean else 0), f"{what}: kernel {lean}"
    return want, lean


def case_of(w, **kw) -> RemovalCase:
    return RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                       max_removable=w.max_removable, last_index=w.last_index, **kw)


@pytest.mark.parametrize("seed", range(400))
def test_fuzz_plain_clusters(seed, monkeypatch):
    w = fuzz_removals_plain(seed)
    both(case_of(w), w.name, monkeypatch)


@pytest.mark.parametrize("seed", range(120))
def test_fuzz_plain_with_sticky_atomic_and_small_ext_tables(seed, monkeypatch):
    w = fuzz_removals_plain(5000 + seed, max_nodes=25)
    import random
    rnd = random.Random(seed)
    pods = [p for c in w.candidates for p in w.nodes[c].pods if not p.daemonset]
    sticky = {id(p) for p in pods if rnd.random() < 0.1} if seed % 3 == 0 else None
    atomic = [1 if rnd.random() < 0.3 else 0 for _ in w.candidates] if seed % 2 == 0 else None
    ext = rnd.choice([None, None, 0, 1, 3, 10])
    both(case_of(w, sticky=sticky, atomic=atomic, ext_capacity=ext), w.name, monkeypatch)


@pytest.mark.parametrize("seed", range(150))
def test_general_fuzz_takes_the_kernel_its_shape_allows(seed, monkeypatch):
    """fuzz_removals: a third of the seeds carry host ports / hostname anti-affinity (exclusion words): K_sched; the rest: the lean kernel"""
    w = fuzz_removals(seed)
    both(case_of(w), w.name, monkeypatch, expect_lean=None)


def test_share_of_the_general_fuzz_on_the_lean_kernel(monkeypatch):
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    n = 0
    for seed in range(60):
        removal_device(case_of(fuzz_removals(seed)), EmuContext(0))
        n += last_kernel()[0]
    assert 20 <= n < 60, n


def test_three_and_four_resource_lanes(monkeypatch):
    lanes = ("cpu", "memory", "ephemeral-storage", "example.com/gpu")
    for seed in range(12):
        nodes = []
        for i in range(12):
            node = _node(f"n{i}", 4000, 8 * GiB, 20)
            node.allocatable["ephemeral-storage"] = node.capacity["ephemeral-storage"] = 100 * GiB
            node.allocatable["example.com/gpu"] = node.capacity["example.com/gpu"] = (i + seed) % 3
            info = NodeInfo(node)
            for j in range((i * 7 + seed) % 4):
                rq = {"cpu": 500, "memory": 1 * GiB, "ephemeral-storage": (10 + 20 * ((i + j) % 3)) * GiB}
                if seed % 2 and node.allocatable["example.com/gpu"] > j and (i + j + seed) % 2:   # (only when the lane is there)
                    rq["example.com/gpu"] = 1
                info.pods.append(Pod(name=f"p{i}-{j}", labels={"app": "x"}, requests=rq, controller_uid=f"rs{j}"))
            nodes.append(info)
        order = [(3 * k + seed) % 12 for k in range(8)]
        order = list(dict.fromkeys(order))
        both(RemovalCase(nodes=nodes, candidates=order, lanes=lanes[:3 + seed % 2], last_index=seed % 13), f"lanes seed {seed}", monkeypatch)


def test_more_than_sixty_four_classes_keep_k_sched(monkeypatch):
    nodes = [NodeInfo(_node(f"n{i}", 64000, 256 * GiB, 200)) for i in range(6)]
    for i in range(3):
        for j in range(30):
            nodes[i].pods.append(build_test_pod(f"p{i}-{j}", 10 + i * 30 + j, 1))   # 90 distinct requests
    both(RemovalCase(nodes=nodes, candidates=[0, 1, 2]), "90 classes", monkeypatch, expect_lean=False)


def test_a_transaction_longer_than_the_lds_ring(monkeypatch):
    """one candidate with 300 + pods: placements beyond the ring come back from node_out; committed, then listed again by the next candidate;
    and the same loop without persistence (every transaction reverted pod by pod)"""
    nodes = [NodeInfo(_node(f"n{i}", 64000, 256 * GiB, 800)) for i in range(4)]
    for j in range(310):
        nodes[0].pods.append(build_test_pod(f"a{j}", 10 + (j % 3), 1))
    for j in range(5):
        nodes[1].pods.append(build_test_pod(f"b{j}", 100, 1))
    for persist in (True, False):
        w, _ = both(RemovalCase(nodes=nodes, candidates=[0, 1, 2], persist=persist), f"long transaction persist={persist}", monkeypatch)
        assert list(w["removable"]) == [1, 1, 1]
    # a failing long transaction: the last pod finds no node, 300 + placements are taken back
    small = [NodeInfo(_node(f"n{i}", 3200, 256 * GiB, 800)) for i in range(3)]
    for j in range(310):
        small[0].pods.append(build_test_pod(f"a{j}", 10, 1))
    small[0].pods.append(build_test_pod("big", 3000, 1))
    small[1].pods.append(build_test_pod("c", 1000, 1))
    w, _ = both(RemovalCase(nodes=small, candidates=[0, 1]), "long transaction reverted", monkeypatch)
    assert list(w["removable"]) == [0, 1]


def test_bench_shape_small(monkeypatch):
    w = removal_scale(300, pods_per_node=12, frac_candidates=0.3, seed=4)
    want, lean = both(case_of(w), w.name, monkeypatch)
    assert lean == 1 and want["n_processed"] == len(w.candidates)


def test_the_reference_scale_down_benchmark_shape_small(monkeypatch):
    """BenchmarkRunOnceScaleDown's cluster at 50 nodes (benchmark_runonce_test.go:424-452): 60 % of the nodes go, as at its 400 (golden vector
    benchmark_runonce_scale_down; tests/test_gpu_round5.py runs the full size on the device) — through both removal kernels.  Every candidate
    after the first lists pods that arrived from earlier removals: 40-pod transactions plus the ext path, all night."""
    for n in (50, 110):   # (at 110 nodes the log passes 2 048 entries with most of them dead: squeezed before it is full)
        w = runonce_scale_down(n)
        want, lean = both(case_of(w, ext_capacity=n * 40 * 40), w.name, monkeypatch, expect_lean=None)
        assert lean == 1 and want["n_processed"] == n and sum(1 for r in want["removable"] if r == 1) == n * 6 // 10 and len(want["ext"]) > 1000


def test_a_log_smaller_than_the_worst_case_gives_up_and_k_sched_answers(monkeypatch):
    """The one-wave kernel's LDS log is sized to what fits when `pods + ext_capacity` does not (casim_sched.h: lean_optimistic_): when it fills up,
    moves onto nodes that were removed since are squeezed out; a commit that still does not fit ends the kernel and the host runs the call again
    through K_sched — the oracle's results either way.  CASIM_LEAN_LOG_CAP makes the log 256 entries, so that all three outcomes (fits, fits after
    squeezing, gives up) occur on small cases."""
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    monkeypatch.setenv("CASIM_LEAN_HBM_LOG", "0")    # (the fall-back to K_sched is what this test is after; test_an_lds_log_that_gives_up_... has the other one)
    finished = squeezed = gave_up = 0
    cases = [case_of(fuzz_removals_plain(s)) for s in range(0, 400, 7)] + [case_of(runonce_scale_down(n), ext_capacity=4000) for n in (5, 10, 20, 40)]
    cases += [case_of(w, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64) for w in (fuzz_removals_runs(s) for s in range(0, 240, 5)) if len(w.nodes) < 1000]
    # (every node a candidate, few pods each: many more moves than live entries — pods travel from node to node as the cluster empties)
    cases += [case_of(runonce_scale_down(n, ppn), ext_capacity=4000) for n, ppn in ((20, 6), (30, 5), (40, 4), (25, 8), (60, 3))]
    for case in cases:
        want = removal_oracle(case)
        monkeypatch.delenv("CASIM_LEAN_LOG_CAP", raising=False)
        removal_device(case, EmuContext(0))
        eligible = last_kernel()[0] == 1
        monkeypatch.setenv("CASIM_LEAN_LOG_CAP", "256")
        got = removal_device(case, EmuContext(0))
        assert_removal_matches(got, want, "small log")
        lean = last_kernel()[0] == 1
        assert not (lean and not eligible)
        moves = sum(len(lst) for lst, r in zip(case.pod_lists(), want["removable"]) if r == 1) + sum(1 for c, _, _ in want["ext"] if want["removable"][c] == 1)
        finished += int(lean)
        squeezed += int(lean and case.persist and moves > 256)      # more committed moves than entries: it finished because dead ones went
        gave_up += int(eligible and not lean)
    monkeypatch.delenv("CASIM_LEAN_LOG_CAP", raising=False)
    assert finished >= 20 and squeezed >= 2 and gave_up >= 3, (finished, squeezed, gave_up)


@pytest.mark.parametrize("seed", range(240))
def test_fuzz_runs_of_replicas(seed, monkeypatch):
    """runs of identical pods: the one-wave kernel a word of nodes at a time (schedule_run), the same kernel pod by pod (CASIM_LEAN_BULK_MIN=0)
    and K_sched, all three against the oracle in every field"""
    w = fuzz_removals_runs(seed)
    case = case_of(w, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64)
    monkeypatch.delenv("CASIM_LEAN_BULK_MIN", raising=False)
    want, lean = both(case, w.name, monkeypatch, expect_lean=None)
    monkeypatch.setenv("CASIM_LEAN_BULK_MIN", "0")
    got = removal_device(case, EmuContext(0))
    assert_removal_matches(got, want, f"{w.name} lean kernel, pod by pod")
    assert last_kernel()[0] == lean


def test_the_run_fuzz_reaches_what_it_is_for():
    """long runs, runs that come round the list, failed (reverted) transactions, pods listed again — and most cases on the one-wave kernel"""
    longest = failed = again = 0
    for seed in range(240):
        w = fuzz_removals_runs(seed)
        case = case_of(w, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64)
        want = removal_oracle(case)
        failed += sum(1 for r in want["removable"] if r == 0)
        again += len(want["ext"])
        for lst in case.pod_lists():
            run = 1
            for a, b in zip(lst, lst[1:]):
                run = run + 1 if a.spec_key() == b.spec_key() else 1
                longest = max(longest, run)
    assert longest >= 40 and failed >= 300 and again >= 3000, (longest, failed, again)


@pytest.mark.parametrize("base", range(0, 240, 40))
def test_the_log_in_hbm(base, monkeypatch):
    """removals_lean_kernel<., true, true>: the log of committed moves in HBM (calls whose live moves outgrow LDS or that list more than 65 536 pods).
    CASIM_LEAN_HBM_LOG=1 sends every eligible call through it: the run fuzz and the plain fuzz against the oracle in every field."""
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    monkeypatch.setenv("CASIM_LEAN_HBM_LOG", "1")
    ran = 0
    for seed in range(base, base + 40):
        for w in (fuzz_removals_runs(seed), fuzz_removals_plain(seed)):
            case = case_of(w, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64)
            assert_removal_matches(removal_device(case, EmuContext(0)), removal_oracle(case), f"{w.name} log in HBM")
            ran += last_kernel()[0]
    assert ran >= 60, ran


def test_more_pods_than_sixteen_bits(monkeypatch):
    """BenchmarkRunOnceScaleDown's cluster at 1 650 nodes: 66 000 pods to move — beyond the 16-bit pod indices of the LDS log; the HBM log (32-bit)
    is picked by itself and the one-wave kernel answers: 60 % of the nodes go"""
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    monkeypatch.delenv("CASIM_LEAN_HBM_LOG", raising=False)
    w = runonce_scale_down(1650)
    case = case_of(w, ext_capacity=70000)
    want = removal_oracle(case)
    got = removal_device(case, EmuContext(0))
    assert last_kernel()[0] == 1
    assert_removal_matches(got, want, w.name)
    assert sum(1 for r in want["removable"] if r == 1) == 990


def test_an_lds_log_that_gives_up_hands_over_to_the_log_in_hbm(monkeypatch):
    """default policy: the LDS log first; when it gives up, the same one-wave kernel runs again with its log in HBM (K_sched only where that is not
    possible) — the oracle's results, and the kernel that answered is the one-wave kernel"""
    monkeypatch.delenv("CASIM_NO_LEAN_REMOVALS", raising=False)
    monkeypatch.delenv("CASIM_LEAN_HBM_LOG", raising=False)
    handed_over = 0
    cases = [case_of(runonce_scale_down(n, ppn), ext_capacity=4000) for n, ppn in ((20, 40), (40, 20), (30, 30))]
    cases += [case_of(w, ext_capacity=4 * sum(len(n.pods) for n in w.nodes) + 64) for w in (fuzz_removals_runs(s) for s in range(0, 240, 6)) if len(w.nodes) < 1000]
    for case in cases:
        want = removal_oracle(case)
        monkeypatch.setenv("CASIM_LEAN_LOG_CAP", "256")
        monkeypatch.setenv("CASIM_LEAN_HBM_LOG", "0")
        removal_device(case, EmuContext(0))
        gives_up = last_kernel()[0] == 0
        monkeypatch.delenv("CASIM_LEAN_HBM_LOG", raising=False)
        got = removal_device(case, EmuContext(0))
        assert_removal_matches(got, want, "LDS log, then HBM log")
        monkeypatch.delenv("CASIM_LEAN_LOG_CAP", raising=False)
        removal_device(case, EmuContext(0))
        eligible = last_kernel()[0] == 1
        if eligible and gives_up:
            monkeypatch.setenv("CASIM_LEAN_LOG_CAP", "256")
            removal_device(case, EmuContext(0))
            assert last_kernel()[0] == 1      # (with the small LDS log AND the HBM log allowed: still the one-wave kernel)
            monkeypatch.delenv("CASIM_LEAN_LOG_CAP", raising=False)
            handed_over += 1
    assert handed_over >= 3, handed_over
