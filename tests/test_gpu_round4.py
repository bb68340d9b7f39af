"""-m gpu: round 4 on a real MI355X through the C ABI.
  * the reference's orchestrator rows (tests/golden/reference_vectors.json: orchestrator_scale_up) through the product kernels: the chain
    SchedulablePodGroups -> Estimate per node group -> expander input -> status sets, both packers;
  * casim_cluster_forget_commits on the device;
  * (further down) what round 4 adds to the batch path."""
import os

import numpy as np
import pytest

import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from harness import assert_matches_oracle, encode, run_gpu, run_oracle
from orchestrator_rows import ROWS, Row, per_group_of_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = kaa.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("generic", [False, True], ids=["register-packer", "int64-packer"])
def test_reference_orchestrator_rows_on_the_device(ctx, generic):
    for row in ROWS:
        r = Row(row)
        sc = r.scenario()
        enc = encode(sc)
        res, _ = run_gpu(enc, ctx, generic=generic)
        enc.close()
        assert_matches_oracle(res, run_oracle(sc), row["name"])
        r.check(r.decide(per_group_of_batch(res)), "MI355X:")


def test_rules_from_a_fresh_snapshot_need_forget_commits_on_the_device(ctx):
    """the emulator test of tests/test_resident_cluster_emu.py on the MI355X: second pass with rules re-encoded from the committed snapshot"""
    from test_resident_cluster_emu import _second_pass_with_fresh_rules
    import test_resident_cluster_emu as mod
    saved = mod.EmuCluster
    mod.EmuCluster = lambda classes, nodes, lds_budget=0: kaa.ResidentCluster(ctx, classes, nodes)
    try:
        compared = 0
        for seed in range(30):
            w = workloads.fuzz_pending_domains(8100 + seed)
            if len(w.pods) < 2:
                continue
            got, want, committed = _second_pass_with_fresh_rules(w, forget=True)
            if got is None or committed == 0:
                continue
            compared += 1
            assert got == want, f"seed {seed}"
        assert compared >= 10
    finally:
        mod.EmuCluster = saved


def test_winners_only_on_the_device(ctx):
    """casim_options.winners_only through casim_estimate_batch_query on the MI355X: one part and streamed parts against the full answer
    (the emulator form is tests/test_winners_only_emu.py); a C2 batch like the bench's enter_return row."""
    from bench import _same_winners, simulation_tables
    from kubernetes_autoscaler_amd.engine import BatchCall
    from kubernetes_autoscaler_amd.tables import TableSet
    from test_winners_only_emu import _batch, _check
    KINDS = [_abi.EXPANDER_LEAST_NODES]
    enc, ts, _ = _batch(37)
    pegs, groups = ts.structs()
    full, fexp = BatchCall(ctx, pegs, groups, kinds=KINDS).call()
    for k in (0, 3):
        call = BatchCall(ctx, pegs, groups, kinds=KINDS, n_streams=k, winners_only=True)
        for _ in range(2):
            got, gexp = call.call()
        assert _check(full, fexp, got, gexp, f"n_streams {k}") > 0
    enc.close()
    c2 = simulation_tables(workloads.config_c2, range(4), kaa.Encoder, TableSet).tile(24)     # 96 simulations
    pegs, groups = c2.structs()
    full = BatchCall(ctx, pegs, groups, kinds=KINDS, n_streams=4).call()
    win = BatchCall(ctx, pegs, groups, kinds=KINDS, n_streams=4, winners_only=True).call()
    assert _same_winners(win, full)
    assert int(win[0].winner_offsets[-1]) * 8 < int(full[0].offsets[-1])      # (an eighth of the lists at most: 20 groups per simulation)


def test_int64_register_packer_on_the_device(ctx):
    """pack_fast64_kernel through the C ABI (the emulator form is tests/test_pack_i64_emu.py): forced on fuzz scenarios against the oracle,
    selected by itself for byte-granular co-prime amounts (problem_info [1] == 8), amounts beyond 2^53, and the headline batch bit-equal to
    the int32 store, resident and streamed."""
    from harness import GroupSpec, Scenario, run_gpu_tables
    from test_gpu_round3 import KINDS, _c2_batch, _same
    from test_pack_i64_emu import shape_scenario
    import test_kernels_emu_fuzz as F
    from kubernetes_autoscaler_amd.engine import Problem
    on64 = 0
    for seed in range(40):
        sc = F.scenario_of(workloads.fuzz(seed, rich=seed % 3 == 0))
        enc = encode(sc)
        with Problem(ctx, enc.pegs, enc.groups, False, 2) as p:
            on64 += p.info()["fast_packer_lanes"] == 8
            p.run(); res = p.fetch()
        enc.close()
        assert_matches_oracle(res, run_oracle(sc), f"forced int64 store, seed {seed}")
    assert on64 >= 20
    for seed in range(30):
        sc = shape_scenario(seed, 1 << 20, 1024) if seed % 3 else shape_scenario(seed, 1 << 40, 1 << 9)
        enc = encode(sc)
        with Problem(ctx, enc.pegs, enc.groups) as p:
            assert p.info()["fast_packer_lanes"] == 8, seed
            p.run(); res = p.fetch()
        enc.close()
        assert_matches_oracle(res, run_oracle(sc), f"wide lanes, seed {seed}")
    ts = _c2_batch(6, 16)
    base, bexp = run_gpu_tables(ts, ctx, kinds=KINDS)
    for k in (0, 4):
        res, exp = run_gpu_tables(ts, ctx, kinds=KINDS, n_streams=k, generic=2)
        _same(res, base, f"int64 register store, n_streams {k}")
        assert list(exp["best"]) == list(bexp["best"]) and list(exp["packed"]) == list(bexp["packed"])
