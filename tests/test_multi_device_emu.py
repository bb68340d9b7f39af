"""casim_estimate_batch_multi (casim_multi.h): one batch over several devices behind ONE caller — node groups of every
simulation block-partitioned, PEG table replicated, one min over packed keys for the expander.  CPU: n emulated devices
vs the single-device run of the same batch (which the other tests tie to the oracle)."""
import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from harness import GroupSpec, Scenario, assert_matches_oracle, encode, encode_batch, run_emu_multi, run_emu_tables, run_oracle
from kubernetes_autoscaler_amd.tables import TableSet

KINDS = [[_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_MOST_PODS], [_abi.EXPANDER_LEAST_WASTE], [_abi.EXPANDER_LEAST_NODES, _abi.EXPANDER_LEAST_WASTE],
         [_abi.EXPANDER_LEAST_WASTE, _abi.EXPANDER_MOST_PODS]]


def _scenario(seed, groups=6):
    w = workloads.fuzz(seed, max_groups=groups, max_pegs=14)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)


def _same_results(a, b):
    for f in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum", "offsets",
              "order", "placed"):
        assert list(getattr(a, f)) == list(getattr(b, f)), f


@pytest.mark.parametrize("n_devices", [1, 2, 3, 8])
@pytest.mark.parametrize("seed", range(6))
def test_multi_device_batch_equals_single_device(n_devices, seed):
    scs = [_scenario(3100 + 10 * seed + k) for k in range(1 + seed % 4)]
    enc, ts, bases = encode_batch(scs)
    one, _ = run_emu_tables(ts)
    got, _, info = run_emu_multi(ts, n_devices)
    _same_results(got, one)
    assert sum(info[1:1 + n_devices]) == ts.n_groups
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc)])
    assert_matches_oracle(got, want, f"multi {n_devices} devices seed {seed}")
    enc.close()


@pytest.mark.parametrize("kinds", KINDS)
@pytest.mark.parametrize("n_devices", [2, 5])
@pytest.mark.parametrize("use_hook", [True, False])
def test_expander_choice_across_devices(kinds, n_devices, use_hook):
    scs = [_scenario(5200 + k, groups=7) for k in range(5)]
    enc, ts, bases = encode_batch(scs)
    _, one = run_emu_tables(ts, kinds=kinds)
    _, exp, info = run_emu_multi(ts, n_devices, kinds=kinds, use_hook=use_hook)
    integer_only = len(kinds) == 1 and kinds[0] != _abi.EXPANDER_LEAST_WASTE
    assert info[0] == (1 if (use_hook and integer_only) else 0)     # the collective handles integer single-filter chains
    assert list(exp["best"]) == list(one["best"])                   # same winner (caller's index), every simulation
    assert list(exp["packed"]) == list(one["packed"])
    enc.close()


def test_explicit_peg_lists_and_validity_mask_across_devices():
    w = workloads.fuzz(77, max_groups=6, max_pegs=10)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups])   # host-side subsets
    enc = encode(sc)
    ts = TableSet.from_encoder(enc)
    ts.sim_offsets = np.array([0, ts.n_groups], np.int32)
    kinds = [_abi.EXPANDER_LEAST_NODES]
    one, e1 = run_emu_tables(ts, kinds=kinds)
    got, e2, _ = run_emu_multi(ts, 3, kinds=kinds)
    _same_results(got, one)
    assert list(e2["best"]) == list(e1["best"])
    if e1["best"][0] >= 0:
        valid = np.ones(ts.n_groups, np.uint8); valid[e1["best"][0]] = 0
        _, m1 = run_emu_tables(ts, kinds=kinds, valid=valid)
        _, m2, _ = run_emu_multi(ts, 3, kinds=kinds, valid=valid)
        assert list(m2["best"]) == list(m1["best"]) and m2["best"][0] != e1["best"][0]
    enc.close()


def test_more_devices_than_groups():
    sc = _scenario(9, groups=2)
    enc, ts, _ = encode_batch([sc])
    one, e1 = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES])
    got, e2, info = run_emu_multi(ts, 8, kinds=[_abi.EXPANDER_LEAST_NODES])
    _same_results(got, one)
    assert list(e2["best"]) == list(e1["best"]) and 0 in info[1:9]
    _, w1 = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_WASTE, _abi.EXPANDER_MOST_PODS])
    _, w2, _ = run_emu_multi(ts, 8, kinds=[_abi.EXPANDER_LEAST_WASTE, _abi.EXPANDER_MOST_PODS])     # empty shards carry no capacity columns
    assert list(w2["best"]) == list(w1["best"])
    enc.close()


@pytest.mark.parametrize("use_hook", [True, False])
@pytest.mark.parametrize("base", [0, 100, 70000])
def test_group_id_base_without_global_ids(base, use_hook):
    """casim_groups.global_id == NULL and q->group_id_base != 0: the keys carry base + the caller's index on every
    device, and the winner lookup finds it again (it used to answer "no option" for every simulation)."""
    scs = [_scenario(6400 + k, groups=6) for k in range(4)]
    enc, ts, _ = encode_batch(scs)
    ts.global_id = None
    kinds = [_abi.EXPANDER_LEAST_NODES]
    _, one = run_emu_tables(ts, kinds=kinds)
    _, exp, _ = run_emu_multi(ts, 3, kinds=kinds, use_hook=use_hook, group_id_base=base)
    assert list(exp["best"]) == list(one["best"]) and any(b >= 0 for b in exp["best"])
    for s_, b in enumerate(exp["best"]):
        if b >= 0:
            assert int(exp["packed"][s_]) & 0xFFFFF == base + b
            assert int(exp["keys"][s_][9]) == base + b and int(exp["keys"][s_][0]) == int(exp["packed"][s_])
    enc.close()


def test_group_ids_beyond_the_key_field_are_rejected():
    scs = [_scenario(6500, groups=3)]
    enc, ts, _ = encode_batch(scs)
    ts.global_id = None
    run_emu_multi(ts, 2, kinds=[_abi.EXPANDER_LEAST_NODES], group_id_base=(1 << 20) - 1, expect_rc=_abi.ERR_INVALID)
    enc.close()


@pytest.mark.parametrize("n_devices", [1, 2, 3])
@pytest.mark.parametrize("seed", range(12))
def test_singleton_runs_across_devices(seed, n_devices):
    """SingletonRuns (casim_pipeline.h) under the multi-device path: every device merges the runs of its own view and writes them out
    member by member; one simulation (merging is off for batches of simulations), groups block-partitioned over the devices."""
    w = workloads.fuzz_singleton_runs(300 + seed, max_groups=5)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=True)
    enc = encode(sc)
    ts = TableSet.from_encoder(enc).as_one_simulation()
    got, _, info = run_emu_multi(ts, n_devices)
    assert_matches_oracle(got, run_oracle(sc), f"singleton runs over {n_devices} devices, seed {seed}")
    enc.close()
