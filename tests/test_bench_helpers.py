"""bench.py's own checker: verify_headline compares EVERY group of the timed batch (all tiles of the distinct seeds, sharded or not) with
the oracle's scale-up simulation of its seed (VERDICT r3 next #1a).  Here on a small C2 batch whose results come from the product kernels
under the wave emulator — the bench itself needs the MI355X."""
import numpy as np

import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.tables import TableSet
from harness import run_emu_tables

import bench


def test_verify_headline_sees_every_tile_and_every_shard_and_a_single_wrong_entry():
    seeds = 2
    full = bench.simulation_tables(workloads.config_c2, range(seeds), kaa.Encoder, TableSet).tile(2)
    res, _ = run_emu_tables(full)
    out = bench.verify_headline(workloads, workloads.config_c2, seeds, full, res)
    assert out["headline_bit_exact"] and out["groups_compared"] == full.n_groups == 80 and out["simulations_compared"] == 4, out
    for rank in range(2):   # N > 1: every rank holds some groups of every simulation
        mine = full.shard(rank, 2)
        r2, _ = run_emu_tables(mine)
        o2 = bench.verify_headline(workloads, workloads.config_c2, seeds, mine, r2)
        assert o2["headline_bit_exact"] and o2["groups_compared"] == mine.n_groups and o2["simulations_compared"] == 4, o2
    # one pod count off by one in one group of the LAST tile: caught, and located
    a = int(res.offsets[full.n_groups - 3])
    res.placed[a] += 1
    bad = bench.verify_headline(workloads, workloads.config_c2, seeds, full, res)
    assert not bad["headline_bit_exact"] and bad["groups_differing"] == 1 and bad["first_differing_group"] == full.n_groups - 3
    res.placed[a] -= 1
    res.last_index_out[7] += 1
    bad = bench.verify_headline(workloads, workloads.config_c2, seeds, full, res)
    assert bad["groups_differing"] == 1 and bad["first_differing_group"] == 7


def test_sched_issue_roofline_reads_the_committed_counters_or_says_why_not():
    r = bench.sched_issue_roofline("try_schedule_pods", 2.2)
    assert "error" in r or (r["bound"] in ("valu_issue", "salu_issue") and 0 < r["frac"] < 1.5 and r["simds_in_use"] <= 4)
    assert "error" in bench.sched_issue_roofline("no_such_row", 1.0)
