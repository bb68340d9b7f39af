"""Batches of SMALL estimates on the lean register packer (pack_fast_kernel<2, 1, 0>: cpu + memory on narrowed lanes, no exclusion words, <= 64
simulated nodes) — the shape of the headline batch — with what BASELINE's C2 generator never varies: nodes already in the cluster (the
rotation origin of tryToScheduleOnExistingNodes: scheduling_opts.go:54-59), entry lastIndex values beyond the list, unschedulable
templates next to pods that tolerate everything (next-fit on the newest node by name, binpacking_estimator.go:198-209), DaemonSet pods on
the template, pod limits of 1 / 3 / 10 / 110, limiter values of every sign, zero requests, PEGs of 1 .. 255 pods, tryFastPath.
workloads.fuzz_lean was written for an experiment of round 6 (an estimate per LANE: profiles/r15a_wide_packer_ab.txt — bit-exact, slower,
not kept); the workloads stay.  CPU: product kernels under the wave emulator against the oracle, simulation by simulation."""
import os

import pytest

from kubernetes_autoscaler_amd import workloads
from harness import GroupSpec, Scenario, assert_matches_oracle, emu_lib, encode_batch, run_emu_tables, run_oracle


def lean_scenario(seed, fastpath=False):
    w = workloads.fuzz_lean(seed)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], existing=w.existing,
                    device_csr=True, fastpath=fastpath)


def lean_batch(seed, fastpath=False):
    return [lean_scenario(1000 * seed + k, fastpath) for k in range(2 + seed % 5)]


def _want(scs, bases, chain=False):
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc, chain=chain)])
    return want


@pytest.mark.parametrize("fastpath", [False, True])
@pytest.mark.parametrize("seed", range(24))
def test_batches_of_small_estimates_match_the_oracle(seed, fastpath):
    scs = lean_batch(seed + (500 if fastpath else 0), fastpath)
    enc, ts, bases = encode_batch(scs)
    res, _ = run_emu_tables(ts, fastpath=fastpath)
    assert emu_lib().emu_last_packer() == 201, "the lean register packer was not taken"
    assert_matches_oracle(res, _want(scs, bases), f"lean batch {seed} fastpath {fastpath}")
    enc.close()


@pytest.mark.parametrize("seed", range(8))
def test_the_same_in_batch_geometry_and_chained(seed, monkeypatch):
    """CASIM_TEST_BATCH_GROUPS: one-wave orderer blocks and the simulation-major kernels with a few groups; chain_last_index: lastIndex
    handed from group to group inside every simulation (plugin_runner.go:138)."""
    monkeypatch.setenv("CASIM_TEST_BATCH_GROUPS", "2")
    scs = lean_batch(900 + seed)
    enc, ts, bases = encode_batch(scs)
    res, _ = run_emu_tables(ts, chain=True)
    assert_matches_oracle(res, _want(scs, bases, chain=True), f"lean batch {seed}, chained")
    enc.close()


def test_the_workloads_reach_what_they_are_for():
    """existing nodes, unschedulable templates with a tolerating PEG, every limiter sign, groups that hit their limiter and groups that do not."""
    seen = dict(existing=0, unschedulable=0, tolerates_all=0, unlimited=0, refuses=0, big_peg=0)
    for seed in range(120):
        w = workloads.fuzz_lean(seed)
        seen["existing"] += bool(w.existing)
        seen["unschedulable"] += any(g.template.node.unschedulable for g in w.groups)
        seen["tolerates_all"] += any(any(t.key == "" and t.operator == "Exists" for t in (pg.pods[0].tolerations or [])) for pg in w.pegs)
        seen["unlimited"] += any(g.max_nodes == 0 for g in w.groups)
        seen["refuses"] += any(g.max_nodes < 0 for g in w.groups)
        seen["big_peg"] += any(len(pg.pods) == 255 for pg in w.pegs)
    assert all(v >= 5 for v in seen.values()), seen


@pytest.mark.parametrize("seed", range(24))
def test_c4_shaped_batches_behind_a_dry_limiter(seed):
    """The anti-affinity register packer (pack_fast_kernel<2, 1, 2>): its records carry the PEG's exclusion words (DevResults::rec_xw) and,
    behind a dry limiter, a PEG WITHOUT words is decided by the lean store's three compares before any word logic (casim_pack.h: `idle`).
    BASELINE config C4 in small: half of the PEGs self-anti-affine on the hostname, a tenth anti-affine to another PEG's label, limits of 2 .. 8
    nodes — most steps run behind the dry limiter, PEGs with and without words alternate."""
    scs = []
    for k in range(2):
        w = workloads.config_c4(4_100_000 + seed * 3 + k, n_groups=3, n_pegs=24 + (seed * 7) % 40, pods_per_peg=1 + seed % 5, cap=2 + seed % 7)
        scs.append(Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True))
    enc, ts, bases = encode_batch(scs)
    assert ts.dims["w_excl"] in (1, 2), ts.dims
    res, _ = run_emu_tables(ts)
    assert emu_lib().emu_last_packer() == 201
    assert_matches_oracle(res, _want(scs, bases), f"C4-shaped batch {seed}")
    enc.close()
