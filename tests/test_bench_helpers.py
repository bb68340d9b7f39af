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


def _canned_full_line():
    """the full result dict of a real run (round 4's committed bench output: 22 KB, the size the driver's parser lost)"""
    import json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r09i_bench.json")
    for ln in open(path):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError("no JSON line in " + path)


def test_the_contract_line_is_compact_complete_and_the_last_line_of_stdout(tmp_path, capfd):
    """< 4 KB, carries roofline + cpu_baseline + the wall / int64 regimes, and nothing follows it on stdout — not even
    what C code printed into a stdio buffer before it (RCCL's banner)."""
    import ctypes, json
    out = _canned_full_line()
    out["roofline_feasibility"] = {"bound": "hbm", "kernel": "feas_sim_kernel<true>", "kernel_ms": 0.0236, "algorithmic_bytes_per_launch": 26000000,
                                   "traffic": None, "achieved": 1101.7, "peak": 8000.0, "unit": "GB/s", "frac": 0.1377, "workload": "C3 x 256"}
    out["cpu_baseline"]["label"] = "C restatement, not the Go reference"
    out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"] * 20   # (an over-long sentence is cut, never the line lost)
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < bench.COMPACT_LIMIT < 4097, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "value_wall", "ms_per_step_wall", "value_int64", "headline_bit_exact", "roofline_feasibility"):
        assert k in line, k
    assert set(line["config"]) >= {"workload", "batch_per_gpu", "streams"} and "model" not in line["config"]
    assert set(line["roofline"]) >= {"bound", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "traffic", "achieved", "peak", "unit", "frac", "issue_roofline"}
    assert set(line["roofline"]["issue_roofline"]) >= {"bound", "frac"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "sims_per_s"} and line["cpu_baseline"]["kind"] in ("port", "reference")
    assert abs(line["value"] - out["value"]) / out["value"] < 1e-5 and abs(line["roofline"]["frac"] - out["roofline"]["frac"]) < 1e-6
    # emit(): C stdio output buffered BEFORE the line must not land behind it
    libc = ctypes.CDLL(None)
    libc.printf(b"RCCL version : banner printed by C code\n")
    bench.emit(out, side_path=str(tmp_path / "side.json"))
    cap = capfd.readouterr()
    lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert json.loads(lines[-1]) == json.loads(json.dumps(bench.compact_line(out, "side.json"))), lines[-1][:200]
    assert any("banner" in ln for ln in lines[:-1])
    assert json.load(open(tmp_path / "side.json"))["configs"] == out["configs"]          # the side tables are all there, in the file ...
    assert "bench side tables: " in cap.err and '"configs"' in cap.err                   # ... and on stderr
