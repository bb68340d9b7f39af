"""The orderer's ranks once per (simulation, allocatable pair) — rank_shapes_kernel + order_ranked_kernel (csrc/casim_kernels.h) — against the per-group
sort it replaces (order_strided_kernel) and against the oracle: same lists, same records, same estimates.  CASIM_RANK_ONCE=1 forces the path on
batches it would not pay for (short lists, a pair per group); the automatic rule takes it for C3-shaped batches.  CPU: product code under the emulator."""
import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.tables import TableSet
from harness import GroupSpec, Scenario, assert_matches_oracle, encode_batch, run_emu_streams, run_emu_tables, run_oracle

FIELDS = ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum", "order", "placed")


def _scenario(seed, max_groups=6, max_pegs=14):
    w = workloads.fuzz(seed, max_groups=max_groups, max_pegs=max_pegs)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True, lanes=w.lanes)


def _both(ts, monkeypatch, **kw):
    monkeypatch.setenv("CASIM_RANK_ONCE", "0")
    a = run_emu_tables(ts, **kw)
    monkeypatch.setenv("CASIM_RANK_ONCE", "1")
    b = run_emu_tables(ts, **kw)
    monkeypatch.delenv("CASIM_RANK_ONCE")
    for f in FIELDS:
        assert list(getattr(a[0], f)) == list(getattr(b[0], f)), f
    if a[1] is not None:
        assert list(a[1]["packed"]) == list(b[1]["packed"]) and list(a[1]["best"]) == list(b[1]["best"])
    return b


@pytest.mark.parametrize("seed", range(40))
def test_ranked_lists_equal_sorted_lists_and_the_oracle(seed, monkeypatch):
    scs = [_scenario(7000 + 10 * seed + k) for k in range(2 + seed % 4)]
    if len({sc.lanes for sc in scs}) > 1:
        scs = [scs[0], scs[0]]
    enc, ts, bases = encode_batch(scs)
    for generic in (0, 1, 2):
        res, _ = _both(ts, monkeypatch, kinds=[_abi.EXPANDER_LEAST_WASTE], generic=generic, chain=bool(seed % 2))
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc, chain=bool(seed % 2))])
    assert_matches_oracle(res, want, f"ranked seed {seed}")
    enc.close()


def test_equal_scores_keep_the_canonical_tie_order(monkeypatch):
    """PEGs with identical requests (equal scores): ascending PEG id inside a tie, the rule the per-group sort applies to its ascending lists"""
    from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, PodEquivalenceGroup
    GiB = 1 << 30
    pegs = []
    for i in range(70):
        c, m = [(500, GiB), (250, 2 * GiB), (500, GiB), (1000, GiB // 2)][i % 4]
        sel = {"pool": "a"} if i % 3 == 0 else {}
        pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"p{i}-{j}", labels={"app": f"a{i}"}, requests={"cpu": c, "memory": m}, node_selector=dict(sel), controller_uid=f"rs{i}") for j in range(2)]))
    groups = []
    for gi in range(6):
        cap = {"cpu": [4000, 8000][gi % 2], "memory": [16 * GiB, 32 * GiB][gi % 2], "pods": 110}
        groups.append(GroupSpec(NodeInfo(Node(name=f"t{gi}", labels={"pool": "a" if gi < 3 else "b"}, taints=[], capacity=dict(cap), allocatable=dict(cap))), 8, 0, None))
    sc = Scenario(pegs=pegs, groups=groups, device_csr=True)
    enc, ts, _ = encode_batch([sc, sc])
    res, _ = _both(ts, monkeypatch, kinds=[_abi.EXPANDER_LEAST_NODES])
    assert_matches_oracle(res, [(e, ids) for e, ids in run_oracle(sc)] + [(e, [len(pegs) + i for i in ids]) for e, ids in run_oracle(sc)], "ties")
    enc.close()


def test_long_lists_and_streamed_parts(monkeypatch):
    """lists beyond the one-wave networks (the general LDS network in order_strided_kernel) and the batch cut into parts"""
    from harness import mixed_list_simulations
    enc, ts, _ = encode_batch(mixed_list_simulations())
    _both(ts, monkeypatch, kinds=[_abi.EXPANDER_LEAST_NODES])
    monkeypatch.setenv("CASIM_RANK_ONCE", "0")
    a, ea, _ = run_emu_streams(ts, 3, kinds=[_abi.EXPANDER_LEAST_NODES], winners_only=True)
    monkeypatch.setenv("CASIM_RANK_ONCE", "1")
    b, eb, _ = run_emu_streams(ts, 3, kinds=[_abi.EXPANDER_LEAST_NODES], winners_only=True)
    monkeypatch.delenv("CASIM_RANK_ONCE")
    for f in ("node_count", "pods_scheduled", "placed", "order"):
        assert list(getattr(a, f)) == list(getattr(b, f)), f
    assert list(ea["packed"]) == list(eb["packed"])
    enc.close()


def test_the_automatic_rule_takes_c3_shaped_batches(monkeypatch):
    """two C3-like simulations (long candidate ranges, a pair serves a dozen groups): the default takes the ranked path — results equal the forced-off run"""
    monkeypatch.delenv("CASIM_RANK_ONCE", raising=False)
    sets = []
    from kubernetes_autoscaler_amd import Encoder
    for s in range(2):
        w = workloads.config_c3(seed_offset=s, n_groups=24, n_pegs=600, pods_per_peg=4)
        enc = Encoder(lanes=w.lanes)
        for pg in w.pegs:
            enc.add_peg(pg)
        for g in w.groups:
            enc.add_group(g.template, max_nodes=g.max_nodes, existing_nodes=0, last_index=g.last_index, pegs=None)
        enc.finalize()
        sets.append(TableSet.from_encoder(enc).as_one_simulation())
        enc.close()
    ts = TableSet.concat(sets)
    auto = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES])
    monkeypatch.setenv("CASIM_RANK_ONCE", "0")
    off = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES])
    for f in FIELDS:
        assert list(getattr(auto[0], f)) == list(getattr(off[0], f)), f


def test_proportional_pairs_share_a_ranking_only_when_the_check_passes(monkeypatch):
    """allocatable pairs k x (4000 m, 16 GiB): the pair with k = 2 scales every score exactly (its check passes, it reads the base ranking); under
    k = 3 two PEGs that TIE under the base pair (ranked by id) get different scores by rounding — the one with the higher id now comes first: the
    check fails and the pair sorts for itself.  Either way the lists are the per-group sort's and the oracle's."""
    from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, PodEquivalenceGroup
    GiB, MiB = 1 << 30, 1 << 20
    reqs = [(3650, 1207959552), (900, 13019119616), (650, 14092861440), (3900, 134217728), (1400, 10871635968), (2750, 1006632960), (1500, 6375342080)]
    assert reqs[0][0] / 4000.0 + reqs[0][1] / (16.0 * GiB) == reqs[1][0] / 4000.0 + reqs[1][1] / (16.0 * GiB)
    assert reqs[0][0] / 12000.0 + reqs[0][1] / (48.0 * GiB) < reqs[1][0] / 12000.0 + reqs[1][1] / (48.0 * GiB)
    reqs += [(50 * (1 + (7 * i) % 60), 64 * MiB * (1 + (11 * i) % 200)) for i in range(40)]
    pegs = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}-{j}", labels={"app": f"a{i}"}, requests={"cpu": c, "memory": m}, controller_uid=f"rs{i}") for j in range(2)]) for i, (c, m) in enumerate(reqs)]
    groups = []
    for gi in range(9):
        k = [1, 2, 3][gi % 3]
        cap = {"cpu": 4000 * k, "memory": 16 * GiB * k, "pods": 110}
        groups.append(GroupSpec(NodeInfo(Node(name=f"t{gi}", labels={}, taints=[], capacity=dict(cap), allocatable=dict(cap))), 6, 0, None))
    sc = Scenario(pegs=pegs, groups=groups, device_csr=True)
    enc, ts, _ = encode_batch([sc, sc, sc])
    monkeypatch.setenv("CASIM_RANK_SHARE", "1")   # (sharing a ranking between proportional pairs is off by default: measured no faster)
    res, _ = _both(ts, monkeypatch, kinds=[_abi.EXPANDER_LEAST_WASTE])
    monkeypatch.delenv("CASIM_RANK_SHARE")
    _both(ts, monkeypatch, kinds=[_abi.EXPANDER_LEAST_WASTE])
    one = run_oracle(sc)
    want = []
    for k in range(3):
        want += [(e, [k * len(pegs) + i for i in ids]) for e, ids in one]
    assert_matches_oracle(res, want, "proportional pairs")
    # the two PEGs really swap between the pairs (the case is what it claims to be)
    off = res.offsets
    first_of = lambda g: [int(x) for x in res.order[off[g]:off[g + 1]]]
    a, b = first_of(0), first_of(2)
    assert a.index(0) < a.index(1) and b.index(1) < b.index(0)
    enc.close()
