"""front_kernel (csrc/casim_kernels.h): a call of <= 1024 groups with device-derived lists runs feasibility, list offsets, lists
and PEG order as ONE launch — every block publishes its count and waits for the counts in front of it.  Same results as the four
separate launches a batch uses (casim_options.no_front_kernel), and as the oracle.  CPU: product kernels under the wave emulator."""
import numpy as np
import pytest

from harness import (GroupSpec, Scenario, assert_matches_oracle, emu_lib, encode, encode_batch, run_emu, run_emu_feasibility,
                     run_emu_tables, run_oracle)
from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.objects import NodeInfo, Pod, PodEquivalenceGroup

FIELDS = ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "req_cpu_sum", "req_mem_sum", "order", "placed", "offsets")


def _scenario(w, fastpath=False):
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups],
                    existing=w.existing, lanes=w.lanes, fastpath=fastpath, device_csr=True)


def _both(enc, **kw):
    L = emu_lib()
    fused, bf = run_emu(enc, front=True, **kw)
    assert L.emu_last_front() == 1, "the fused launch was not taken"
    split, bs = run_emu(enc, front=False, **kw)
    assert L.emu_last_front() == 0
    for f in FIELDS:
        a, b = getattr(fused, f, None), getattr(split, f, None)
        if a is None and b is None:
            continue
        assert np.array_equal(np.asarray(a), np.asarray(b)), f
    return fused, bf, bs


@pytest.mark.parametrize("seed", range(120))
def test_fuzz_feature_mix(seed):
    """taints, selectors, ports, (anti-)affinity, preloaded pods, every limiter sign; a third with the fastpath, a third through the
    generic packer, a few with the HBM-scratch orderer (front_kernel<false>)"""
    fast = seed % 3 == 1
    sc = _scenario(workloads.fuzz(3000 + seed), fastpath=fast)
    kw = dict(fastpath=fast, generic=seed % 3 == 2, kinds=[0, 1] if seed % 4 == 0 else None)
    if seed % 10 == 9:
        kw["lds_budget"] = 256
    fused, bf, bs = _both(encode(sc), **kw)
    assert_matches_oracle(fused, run_oracle(sc), f"seed {seed}")
    if bf is not None:
        assert bf[0] == bs[0] and bf[1] == bs[1] and (bf[2] == bs[2]).all() and (bf[3] == bs[3]).all()


@pytest.mark.parametrize("seed", range(30))
def test_fuzz_long_rows(seed):
    """up to 260 PEGs per group: several ballot words per row, one per wave of the block and more"""
    sc = _scenario(workloads.fuzz(7000 + seed, max_groups=3, max_pegs=260, rich=seed % 2 == 0))
    fused, _, _ = _both(encode(sc))
    assert_matches_oracle(fused, run_oracle(sc), f"seed {seed}")


def test_many_groups_wait_for_more_than_one_wave_of_tickets():
    """200 groups: the blocks behind the 64th collect the counts in front of them in several rounds of 64 lanes; rows of 3 words,
    many of them empty (selectors nobody satisfies) or full"""
    from kubernetes_autoscaler_amd.workloads import _node, SplitMix64
    rng = SplitMix64(0xF207)
    pegs = []
    for i in range(150):
        sel = {"pool": f"p{rng.below(5)}"} if rng.chance(2, 3) else {}
        pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": 100 * (1 + rng.below(8)), "memory": (128 << 20) * (1 + rng.below(6))},
                                                      node_selector=sel)] * (1 + rng.below(5))))
    groups = []
    for k in range(200):
        labels = {"pool": f"p{rng.below(7)}"} if rng.chance(3, 4) else {}
        groups.append(GroupSpec(NodeInfo(_node(f"g{k}", 50 if k % 9 == 4 else 1000 * (1 + rng.below(8)), (1 + rng.below(16)) << 30, 30, labels)),
                                max_nodes=rng.pick([0, 2, 5, 20]),
                                last_index=0, pegs=None))
    sc = Scenario(pegs=pegs, groups=groups, device_csr=True)
    enc = encode(sc)
    fused, _, _ = _both(enc)
    assert_matches_oracle(fused, run_oracle(sc), "200 groups")
    off = np.asarray(fused.offsets)
    assert (np.diff(off) == 0).any() and off[-1] > 0
    enc.close()


def test_a_few_simulations_in_one_call_take_the_fused_launch_too(monkeypatch):
    """a TableSet of 6 simulations (peg_lo / peg_hi per simulation, 18 groups): below the batch geometry, one launch.  (Since round 4 a batch of
    >= 2 simulations prefers fixed-stride lists — emu_last_front() == 2, the same comparison below — and takes the ticket kernel only when those
    are switched off.)"""
    scs = [Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], existing=[], lanes=w.lanes,
                    device_csr=True) for w in (workloads.fuzz(5100 + i, max_groups=3, max_pegs=40, rich=False) for i in range(6))]
    enc, ts, bases = encode_batch(scs)
    L = emu_lib()
    strided, estr = run_emu_tables(ts, kinds=[0])
    assert L.emu_last_front() == 2
    monkeypatch.setenv("CASIM_NO_STRIDED", "1")
    fused, ef = run_emu_tables(ts, kinds=[0])
    assert L.emu_last_front() == 1
    for f in FIELDS:
        a, b = getattr(fused, f, None), getattr(strided, f, None)
        if a is not None or b is not None:
            assert np.array_equal(np.asarray(a), np.asarray(b)), ("fixed-stride lists", f)
    for k in ("best", "n_best", "best_set", "keys", "packed"):
        assert np.array_equal(ef[k], estr[k]), k
    split, es = run_emu_tables(ts, kinds=[0], front=False)
    assert L.emu_last_front() == 0
    for f in FIELDS:
        a, b = getattr(fused, f, None), getattr(split, f, None)
        if a is not None or b is not None:
            assert np.array_equal(np.asarray(a), np.asarray(b)), f
    for k in ("best", "n_best", "best_set", "keys", "packed"):
        assert np.array_equal(ef[k], es[k]), k
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend((est, [pb + i for i in ids]) for est, ids in run_oracle(sc))
    assert_matches_oracle(fused, want, "6 simulations")
    enc.close()


def test_feasibility_bits_are_still_written():
    """casim_feasibility reads the bit matrix the row pass leaves in HBM"""
    sc = _scenario(workloads.fuzz(3007))
    enc = encode(sc)
    bits = run_emu_feasibility(enc)
    fused, _, _ = _both(enc)
    off = np.asarray(fused.offsets)
    for g in range(len(sc.groups)):
        assert int(off[g + 1] - off[g]) == int(sum(bin(int(x)).count("1") for x in np.atleast_1d(bits[g])))
    enc.close()


@pytest.mark.parametrize("seed", range(40))
def test_blocks_that_do_not_wait_count_the_rows_in_front_themselves(seed, monkeypatch):
    """CASIM_FRONT_SPIN=0: no ticket is ever waited for — every block recounts the rows of all the groups in front of it (the path a
    block takes on the device when a predecessor has not published within the poll limit: nobody waits without bound)."""
    monkeypatch.setenv("CASIM_FRONT_SPIN", "0")
    fast = seed % 3 == 1
    sc = _scenario(workloads.fuzz(3000 + seed) if seed % 2 else workloads.fuzz(7000 + seed, max_groups=4, max_pegs=200), fastpath=fast)
    fused, _, _ = _both(encode(sc), fastpath=fast)
    assert_matches_oracle(fused, run_oracle(sc), f"seed {seed}")


def test_recount_with_many_groups(monkeypatch):
    """80 groups, rows of up to 3 words, every block recounting everything in front of it (two chunks of 64 predecessors)"""
    from kubernetes_autoscaler_amd.workloads import _node, SplitMix64
    monkeypatch.setenv("CASIM_FRONT_SPIN", "0")
    rng = SplitMix64(0xF208)
    pegs = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": 100 * (1 + rng.below(8)), "memory": (128 << 20) * (1 + rng.below(6))},
                                          node_selector=({"pool": f"p{rng.below(4)}"} if rng.chance(2, 3) else {}))] * (1 + rng.below(4))) for i in range(150)]
    groups = [GroupSpec(NodeInfo(_node(f"g{k}", 50 if k % 7 == 3 else 1000 * (1 + rng.below(8)), (1 + rng.below(16)) << 30, 30,
                                       {"pool": f"p{rng.below(5)}"} if rng.chance(3, 4) else {})), max_nodes=rng.pick([0, 2, 5]), last_index=0, pegs=None) for k in range(80)]
    sc = Scenario(pegs=pegs, groups=groups, device_csr=True)
    enc = encode(sc)
    fused, _, _ = _both(enc)
    assert_matches_oracle(fused, run_oracle(sc), "80 groups, recount")
    enc.close()


def test_one_call_with_long_lists_and_the_expander():
    """lists beyond 16 384 entries (order / placed outside the results slab, fetched through the staging buffer in two pieces) next to an
    expander answer that rides in the slab: 6 groups x 3 000 PEGs in ONE call"""
    from kubernetes_autoscaler_amd.workloads import _node, SplitMix64
    rng = SplitMix64(0xF209)
    pegs = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": 10 * (1 + rng.below(300)), "memory": (16 << 20) * (1 + rng.below(200))})] * (1 + rng.below(3)))
            for i in range(3000)]
    groups = [GroupSpec(NodeInfo(_node(f"g{k}", 4000 * (1 + k), (8 + 8 * k) << 30, 110, {})), max_nodes=rng.pick([5, 20, 60]), last_index=0, pegs=None) for k in range(6)]
    sc = Scenario(pegs=pegs, groups=groups, device_csr=True)
    enc, ts, bases = encode_batch([sc])
    res, exp = run_emu_tables(ts, kinds=[0, 1], per_sim=False)
    want = run_oracle(sc)
    assert_matches_oracle(res, want, "6 x 3000")
    assert int(np.asarray(res.offsets)[-1]) > 16384
    nodes = [int(x) for x in res.node_count]
    best = int(exp["best"][0])
    assert best >= 0 and nodes[best] == min(n for n in nodes if n > 0)
    enc.close()
