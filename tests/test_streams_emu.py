"""casim_options.n_streams (csrc/casim_streams.h): a batch of simulations cut by simulation into sub-batches that the product runs on
internal HIP streams of ONE context.  Here the cutting (table views, re-based candidate ranges and simulation offsets), the
per-part expander queries and the merge of the results (group arrays, CSR offsets, PEG ids back in the whole batch's numbering)
run under the emulator: results must equal the uncut batch, for every cut — with the parts one after the other and with the parts as tasks
of the host pool (CASIM_EMU_THREADS=1: casim_streams.h with `threads` on, the way the product runs a streamed call on the device — upload
turns, list bases handed from part to part under a mutex, every part fetched by its own worker; the emulator keeps its running block per
thread).  tests/tools/sanitize_cpu.sh runs this module under ThreadSanitizer."""
import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from harness import GroupSpec, Scenario, encode_batch, run_emu_streams, run_emu_tables

KINDS = [[_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_LEAST_WASTE], [_abi.EXPANDER_MOST_PODS, _abi.EXPANDER_LEAST_NODES]]


@pytest.fixture(autouse=True, params=["parts-in-turn", "parts-on-the-pool"])
def parts_mode(request, monkeypatch):
    monkeypatch.setenv("CASIM_EMU_THREADS", "1" if request.param == "parts-on-the-pool" else "0")
    return request.param


FIELDS = ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum", "offsets")


def _scenario(seed, **kw):
    w = workloads.fuzz(seed, max_groups=5, max_pegs=14, **kw)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)


def _same(a, b, what):
    for f in FIELDS:
        assert list(getattr(a, f)) == list(getattr(b, f)), (what, f)
    nnz = int(a.offsets[-1])
    assert list(a.order[:nnz]) == list(b.order[:nnz]), (what, "order")
    assert list(a.placed[:nnz]) == list(b.placed[:nnz]), (what, "placed")


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("k", [2, 3, 4, 16])
def test_streamed_parts_equal_the_uncut_batch(seed, k):
    n = 2 + (seed * 3) % 9
    scs = [_scenario(4000 + 31 * seed + i, rich=(seed % 2 == 0)) for i in range(n)]
    enc, ts, _ = encode_batch(scs)
    kinds = KINDS[seed % len(KINDS)]
    whole, wexp = run_emu_tables(ts, kinds=kinds)
    res, exp, parts = run_emu_streams(ts, k, kinds=kinds)
    assert parts == min(k, n)
    _same(res, whole, f"seed {seed} k {k}")
    for f in ("best", "n_best", "packed"):
        assert list(exp[f]) == list(wexp[f]), f
    assert list(exp["best_set"]) == list(wexp["best_set"]) and exp["keys"].tolist() == wexp["keys"].tolist()
    enc.close()


def test_validity_mask_group_id_base_and_the_int64_packer_through_the_parts():
    scs = [_scenario(7100 + i) for i in range(6)]
    enc, ts, _ = encode_batch(scs)
    rng = np.random.default_rng(5)
    valid = (rng.random(ts.n_groups) < 0.7).astype(np.uint8)
    ts.global_id = None   # keys then carry group_id_base + index inside the WHOLE batch, whichever part a group lands in
    for generic in (False, True):
        whole, wexp = run_emu_tables(ts, kinds=KINDS[0], valid=valid, generic=generic)
        for k in (2, 5):
            res, exp, parts = run_emu_streams(ts, k, kinds=KINDS[0], valid=valid, generic=generic)
            assert parts == k
            _same(res, whole, f"valid k {k}")
            assert list(exp["best"]) == list(wexp["best"]) and list(exp["packed"]) == list(wexp["packed"])
    enc.close()


def test_batches_that_cannot_be_cut_run_as_one_part():
    enc, ts, _ = encode_batch([_scenario(7300)])           # ONE simulation
    res, _, parts = run_emu_streams(ts, 4, kinds=KINDS[0])
    assert parts == 1
    whole, _ = run_emu_tables(ts, kinds=KINDS[0])
    _same(res, whole, "one simulation")
    enc.close()


def test_simulations_of_very_different_sizes_and_empty_ones():
    """parts whose PEG ranges start far into the table, a simulation without node groups, one without PEGs"""
    from harness import mixed_list_simulations
    scs = mixed_list_simulations()[:3] + [_scenario(7400), _scenario(7401)]
    enc, ts, _ = encode_batch(scs)
    whole, wexp = run_emu_tables(ts, kinds=KINDS[0])
    for k in (2, 3, 5):
        res, exp, parts = run_emu_streams(ts, k, kinds=KINDS[0])
        _same(res, whole, f"mixed k {k}")
        assert list(exp["packed"]) == list(wexp["packed"])
    enc.close()


@pytest.mark.parametrize("seed", range(8))
def test_parts_that_take_the_link_in_turn_and_columns_sent_from_where_they_lie(seed, monkeypatch):
    """round 6: the parts of an enter -> return call can upload IN TURN (CASIM_UPLOAD_FIFO=1: UploadGate in issue order + an event chain on the
    device — a part's tables are enqueued when the part in front has enqueued all of its own), page-locked columns are LISTED and go out at the next flush
    that finds the link free for the part (they used to be copied the moment ProblemT::up saw them), and every part is fetched by its own
    worker (list bases handed from part to part).  The emulated backend plays the turn order (CASIM_EMU_FIFO), calls every column
    page-locked (CASIM_EMU_PINNED) and the test knobs make the small tables travel in many pieces: same results as the uncut batch —
    int64 and caller-narrowed requests, winners only, chained."""
    for k, v in (("CASIM_EMU_FIFO", "1"), ("CASIM_EMU_PINNED", "1"), ("CASIM_TEST_UPLOAD_CHUNK", "512"), ("CASIM_TEST_DIRECT_MIN", "256")):
        monkeypatch.setenv(k, v)
    n = 3 + seed % 5
    scs = [_scenario(5200 + 17 * seed + i, rich=(seed % 2 == 1)) for i in range(n)]
    enc, ts, _ = encode_batch(scs)
    kinds = KINDS[seed % len(KINDS)]
    for narrow in (False, True):
        whole, wexp = run_emu_tables(ts, kinds=kinds, narrow_requests=narrow)
        for k in (2, 4):
            res, exp, parts = run_emu_streams(ts, k, kinds=kinds, narrow_requests=narrow)
            assert parts == min(k, n)
            _same(res, whole, f"seed {seed} k {k} narrow {narrow}")
            assert list(exp["best"]) == list(wexp["best"]) and list(exp["packed"]) == list(wexp["packed"])
    wo, woexp = run_emu_tables(ts, kinds=kinds, winners_only=True)
    res, exp, _ = run_emu_streams(ts, 3, kinds=kinds, winners_only=True)
    assert list(res.node_count) == list(wo.node_count) and list(exp["best"]) == list(woexp["best"])
    nw = int(sum(int(wo.offsets[b + 1] - wo.offsets[b]) for b in woexp["best"] if b >= 0))
    assert list(res.order[:nw]) == list(wo.order[:nw]) and list(res.placed[:nw]) == list(wo.placed[:nw])
    ch, _ = run_emu_tables(ts, chain=True)
    res, _, _ = run_emu_streams(ts, 3, chain=True)
    _same(res, ch, f"seed {seed} chained")
    enc.close()
