"""Batches of independent simulations in one launch (casim_groups.peg_lo / peg_hi / sim_offsets, casim_best_option_sims):
every group only sees its own simulation's PEGs, the expander chain runs once per simulation, node groups of every
simulation can be sharded over GPUs and the per-simulation winners recombined with one min over packed keys.
CPU: product kernels under the wave emulator vs the oracle (per simulation) and vs un-batched runs."""
import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.tables import TableSet
from harness import (GroupSpec, Scenario, assert_matches_oracle, encode, encode_batch, run_emu, run_emu_tables, run_oracle)

KINDS = [[_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_MOST_PODS, _abi.EXPANDER_LEAST_NODES],
         [_abi.EXPANDER_LEAST_WASTE], [_abi.EXPANDER_LEAST_NODES, _abi.EXPANDER_LEAST_WASTE]]


def _scenario(seed):
    w = workloads.fuzz(seed, max_groups=5, max_pegs=14)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)


def _shift(oracle, base):
    return [(est, [base + i for i in ids]) for est, ids in oracle]


@pytest.mark.parametrize("seed", range(12))
def test_batched_simulations_match_the_oracle_per_simulation(seed):
    scs = [_scenario(1000 * seed + k) for k in range(1 + seed % 5)]
    enc, ts, bases = encode_batch(scs)
    res, _ = run_emu_tables(ts)
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend(_shift(run_oracle(sc), pb))
    assert_matches_oracle(res, want, f"batch seed {seed}")
    enc.close()


def test_slices_of_a_batch_are_batches_of_their_own():
    """TableSet.sim_slice: what bench.py hands to its HIP streams.  Every slice gives exactly its simulations' part of
    the whole batch's results."""
    scs = [_scenario(500 + k) for k in range(7)]
    enc, ts, bases = encode_batch(scs)
    whole, wexp = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES])
    for a, b in ((0, 3), (3, 4), (4, 7), (0, 7), (6, 7)):
        part = ts.sim_slice(a, b)
        res, exp = run_emu_tables(part, kinds=[_abi.EXPANDER_LEAST_NODES])
        g0, g1 = int(ts.sim_offsets[a]), int(ts.sim_offsets[b])
        assert list(res.node_count) == list(whole.node_count[g0:g1]) and list(res.pods_scheduled) == list(whole.pods_scheduled[g0:g1])
        assert list(res.last_index_out) == list(whole.last_index_out[g0:g1])
        z0, z1 = int(whole.offsets[g0]), int(whole.offsets[g1])
        assert list(res.placed) [:z1 - z0] == list(whole.placed[z0:z1])
        p0 = int(ts.peg_lo[g0])
        assert [o + p0 for o in res.order[:z1 - z0]] == list(whole.order[z0:z1])
        assert list(exp["packed"]) == list(wexp["packed"][a:b])   # (keys carry simulation-wide group ids)
    enc.close()


def test_lists_of_very_different_lengths_in_one_batch():
    """40 / 300 / 700 / 120 PEGs per simulation: several record chunks per group in the packer, most PEGs behind a dry
    limiter, lists beyond the orderer's one-wave networks (the GPU test tiles this batch to 3072 groups)."""
    from harness import mixed_list_simulations
    scs = mixed_list_simulations()
    enc, ts, bases = encode_batch(scs)
    res, _ = run_emu_tables(ts)
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend(_shift(run_oracle(sc), pb))
    assert_matches_oracle(res, want, "mixed list lengths")
    enc.close()


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("kinds", KINDS)
def test_expander_runs_once_per_simulation(seed, kinds):
    scs = [_scenario(500 + 97 * seed + k) for k in range(4)]
    enc, ts, bases = encode_batch(scs)
    res, exp = run_emu_tables(ts, kinds=kinds, per_sim=True)
    for s, (sc, (pb, gb)) in enumerate(zip(scs, bases)):
        e1 = encode(sc)
        _, best1 = run_emu(e1, kinds=kinds)
        want_best, want_n, want_set, want_key = best1
        assert int(exp["best"][s]) == (gb + want_best if want_best >= 0 else -1), (seed, s)
        assert int(exp["n_best"][s]) == want_n
        assert list(exp["best_set"][gb:gb + len(sc.groups)]) == list(want_set)
        assert list(exp["keys"][s]) == list(want_key)          # global ids are simulation-local: same key block
        assert int(exp["packed"][s]) == int(want_key[0])
        e1.close()
    # one reduce over every group still available on the same batch
    _, flat = run_emu_tables(ts, kinds=kinds, per_sim=False)
    assert len(flat["best"]) == 1
    enc.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_batch_recombines_with_one_min_over_packed_keys(world):
    """Node groups of every simulation block-partitioned (rotated) over `world` shards; each shard's per-simulation packed
    key, min over shards == the single-shard winner (integer first filter)."""
    scs = [_scenario(7000 + k) for k in range(6)]
    enc, ts, bases = encode_batch(scs)
    for kinds in ([_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_MOST_PODS]):
        _, whole = run_emu_tables(ts, kinds=kinds)
        packed = np.full((world, ts.n_sims), np.iinfo(np.int64).max, np.int64)
        n_groups = 0
        for r in range(world):
            sh = ts.shard(r, world)
            n_groups += sh.n_groups
            if sh.n_groups == 0:
                continue
            res, exp = run_emu_tables(sh, kinds=kinds)
            packed[r] = exp["packed"]
        assert n_groups == ts.n_groups
        got = packed.min(axis=0)
        # the chain's first filter decides up to ties; ties resolve to the lowest global id in both forms
        assert list(got) == list(whole["packed"])
    enc.close()


def test_validity_mask_is_applied_before_the_first_filter():
    """all-or-nothing (orchestrator.go:1057-1063): an option that leaves pods behind never reaches the expander."""
    scs = [_scenario(4242 + k) for k in range(3)]
    enc, ts, bases = encode_batch(scs)
    res, free = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES])
    valid = np.ones(ts.n_groups, np.uint8)
    for s in range(ts.n_sims):
        if free["best"][s] >= 0:
            valid[free["best"][s]] = 0
    _, masked = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES], valid=valid)
    for s in range(ts.n_sims):
        if free["best"][s] >= 0:
            assert masked["best"][s] != free["best"][s]
            assert masked["best_set"][free["best"][s]] == 0
    enc.close()


def test_tiled_batch_repeats_its_results():
    sc = _scenario(31337)
    enc, ts, _ = encode_batch([sc])
    big = ts.tile(5)
    res, exp = run_emu_tables(big, kinds=[_abi.EXPANDER_LEAST_NODES])
    ng = ts.n_groups
    for k in range(1, 5):
        assert list(res.node_count[k * ng:(k + 1) * ng]) == list(res.node_count[:ng])
        assert list(res.pods_scheduled[k * ng:(k + 1) * ng]) == list(res.pods_scheduled[:ng])
        assert int(exp["packed"][k]) == int(exp["packed"][0])
    enc.close()


def test_bad_ranges_are_rejected():
    sc = _scenario(5)
    enc, ts, _ = encode_batch([sc])
    ts.peg_hi = ts.peg_hi + 1
    from harness import emu_lib
    import ctypes as C
    pegs, groups = ts.structs()
    L = emu_lib()
    st_rc = L.emu_feasibility(C.byref(pegs), C.byref(groups), np.zeros(64, np.uint64).ctypes.data_as(_abi.u64p))
    assert st_rc == _abi.ERR_INVALID
    enc.close()


def test_large_batch_takes_the_multi_block_scan_and_wave_per_group_order():
    """> 2048 groups: the CSR scan runs in several 1024-thread blocks, the orderer with one wave per group."""
    sc = _scenario(424242)
    enc, ts, _ = encode_batch([sc, _scenario(17)])
    ng = ts.n_groups
    times = 2100 // ng + 1
    big = ts.tile(times)
    assert big.n_groups >= 2048
    res, exp = run_emu_tables(big, kinds=[_abi.EXPANDER_LEAST_NODES])
    base, _ = run_emu_tables(ts, kinds=[_abi.EXPANDER_LEAST_NODES])
    nnz = int(base.offsets[-1])
    for k in range(times):
        assert list(res.offsets[k * ng:(k + 1) * ng + 1] - res.offsets[k * ng]) == list(base.offsets)
        assert list(res.node_count[k * ng:(k + 1) * ng]) == list(base.node_count)
        assert list(res.pods_scheduled[k * ng:(k + 1) * ng]) == list(base.pods_scheduled)
        assert list(res.placed[k * nnz:(k + 1) * nnz]) == list(base.placed)
        assert list(res.order[k * nnz:(k + 1) * nnz] - k * ts.n_pegs) == list(base.order)
    enc.close()


@pytest.mark.parametrize("n_pegs", [20, 100, 220])
def test_one_wave_orderer_networks_of_64_128_256_entries(n_pegs, monkeypatch):
    """Launches of >= 2048 groups sort every list of <= 256 PEGs in ONE wave, in registers (order_group<NPAD>: 64 / 128 / 256
    entries = 1 / 2 / 4 slots per lane, partners through the LDS crossbar): scores with many ties (same requests, different
    counts) so that the position tie-break is exercised in every stage; 2 simulations tiled to the batch geometry vs the oracle
    (CASIM_TEST_BATCH_GROUPS lowers the 2048 to 60 groups: 2 100 emulated groups took two minutes of the CPU suite; the MI355X runs the
    real threshold in tests/test_gpu_round2.py and in the bench's headline check)."""
    monkeypatch.setenv("CASIM_TEST_BATCH_GROUPS", "60")
    from kubernetes_autoscaler_amd.objects import NodeInfo, Pod, PodEquivalenceGroup
    from kubernetes_autoscaler_amd.workloads import _node, SplitMix64
    rng = SplitMix64(0x0DE7 + n_pegs)
    scs = []
    for s in range(2):
        pegs = []
        for i in range(n_pegs):
            req = {"cpu": 50 * (1 + rng.below(6)), "memory": (64 << 20) * (1 + rng.below(4))}   # 24 distinct scores: ties everywhere
            pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"s{s}p{i}", requests=req)] * (1 + rng.below(3))))
        groups = [GroupSpec(NodeInfo(_node(f"t{s}-{k}", 1000 * (2 + k), (2 + 2 * k) << 30, 30, {})), max_nodes=3 + k, last_index=0, pegs=None) for k in range(3)]
        scs.append(Scenario(pegs=pegs, groups=groups, device_csr=True))
    enc, ts, bases = encode_batch(scs)
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend(_shift(run_oracle(sc), pb))
    times = 12    # 2 simulations x 3 groups x 12 = 72 groups >= 60: the batch geometry (one wave per group)
    big = ts.tile(times)
    res, _ = run_emu_tables(big)
    G, NG = ts.n_pegs, ts.n_groups
    for t in (0, 1, times // 2, times - 1):
        for gi, (est, ids) in enumerate(want):
            g = t * NG + gi
            order, placed = res.group(g)
            assert [int(o) - t * G for o in order] == [ids[k] for k in est.order], f"copy {t} group {gi}: PEG order"
            assert list(placed) == list(est.placed) and int(res.node_count[g]) == est.node_count
    enc.close()


@pytest.mark.parametrize("seed", range(10))
def test_tables_uploaded_in_pieces(seed, monkeypatch):
    """A big batch sends its staged columns in pieces of 4 MB while the host stages the rest (ProblemT::flush_uploads_early);
    CASIM_TEST_UPLOAD_CHUNK=48 makes a few hundred bytes of tables travel in a dozen pieces: same results."""
    scs = [_scenario(4000 * seed + k) for k in range(2 + seed % 3)]
    enc, ts, bases = encode_batch(scs)
    base, _ = run_emu_tables(ts, kinds=[0])
    monkeypatch.setenv("CASIM_TEST_UPLOAD_CHUNK", "48")
    res, _ = run_emu_tables(ts, kinds=[0])
    for f in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "order", "placed", "offsets"):
        assert np.array_equal(np.asarray(getattr(res, f)), np.asarray(getattr(base, f))), f
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend(_shift(run_oracle(sc), pb))
    assert_matches_oracle(res, want, f"pieces seed {seed}")
    enc.close()


def test_simulations_with_dictionaries_of_their_own_widen_to_one_batch():
    """TableSet.concat of sets whose mask widths differ (every simulation encoded by an encoder of its own: taints only in one, pairwise
    anti-affinity bits only in another): the columns are padded with zero words and every group still equals the oracle's estimate of its
    own simulation — what bench.py's batched C4 row relies on (seeds whose exclusion dictionaries differ in size)"""
    import bench
    import kubernetes_autoscaler_amd as kaa
    makes = [lambda seed_offset=0: workloads.config_c2(seed_offset, n_groups=4, n_pegs=30, pods_per_peg=4, cap=6),
             lambda seed_offset=0: workloads.config_c4(seed_offset, n_groups=5, n_pegs=40, pods_per_peg=4, cap=8),
             lambda seed_offset=0: workloads.config_c1(seed_offset, n_pegs=20, pods_per_peg=5, cap=12)]
    sets = [bench.simulation_tables(m, range(2), kaa.Encoder, TableSet) for m in makes]
    assert len({tuple(sorted(s.dims.items())) for s in sets}) > 1          # the widths really differ
    ts = TableSet.concat(sets)
    assert ts.n_sims == 6 and ts.dims["w_taint"] == max(s.dims["w_taint"] for s in sets) and ts.dims["w_excl"] == max(s.dims["w_excl"] for s in sets)
    res, _ = run_emu_tables(ts)
    g = 0
    for m, s in zip(makes, sets):
        part = ts.sim_slice(g, g + 2)
        r, _ = run_emu_tables(part)
        chk = bench.verify_headline(workloads, m, 2, part, r)
        assert chk["headline_bit_exact"], chk
        g0, g1 = int(ts.sim_offsets[g]), int(ts.sim_offsets[g + 2])
        assert list(res.node_count[g0:g1]) == list(r.node_count) and list(res.pods_scheduled[g0:g1]) == list(r.pods_scheduled)
        g += 2


def test_simulations_that_share_their_peg_rows():
    """TableSet.tile_groups: the simulations of every copy point (casim_groups.peg_lo / peg_hi) into ONE copy of the PEG tables — a sweep of limiter /
    template variants over the same pending pods ships the pods once.  Same answers as the batch that carries a copy per simulation, PEG ids in the
    lists are the shared table's; cut into streamed parts as well (every part uploads the rows its simulations can see)."""
    from harness import run_emu_streams
    scs = [_scenario(700 + k) for k in range(5)]
    enc, ts, _ = encode_batch(scs)
    big, shared = ts.tile(3), ts.tile_groups(3)
    assert shared.n_pegs == ts.n_pegs and shared.n_groups == big.n_groups and shared.n_sims == big.n_sims
    for kinds in ([_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_LEAST_WASTE]):
        a, ea = run_emu_tables(big, kinds=kinds)
        b, eb = run_emu_tables(shared, kinds=kinds)
        for f in ("node_count", "pods_scheduled", "nodes_added", "last_index_out", "status", "placed", "offsets"):
            assert list(getattr(a, f)) == list(getattr(b, f)), f
        assert [o % ts.n_pegs for o in a.order] == list(b.order) and list(ea["packed"]) == list(eb["packed"]) and list(ea["best"]) == list(eb["best"])
    for wo in (False, True):
        a, ea, pa = run_emu_streams(big, 3, kinds=[_abi.EXPANDER_LEAST_NODES], winners_only=wo)
        b, eb, pb = run_emu_streams(shared, 3, kinds=[_abi.EXPANDER_LEAST_NODES], winners_only=wo)
        assert pa == pb == 3 and list(a.node_count) == list(b.node_count) and list(a.placed) == list(b.placed)
        assert [o % ts.n_pegs for o in a.order] == list(b.order) and list(ea["packed"]) == list(eb["packed"])
    # a head of it is a batch of its own
    h = shared.head(7)
    c, _ = run_emu_tables(h, kinds=None)
    assert list(c.node_count) == list(b.node_count[:h.n_groups])
    enc.close()


def _c4_batch(n_seeds, spoil=False):
    """C4-shaped simulations whose hostname anti-affinity bits need TWO exclusion words (400 PEGs, 60 % with a term).  spoil: one template of
    the second simulation carries a DaemonSet pod that matches a term — its init_excl word is no longer zero."""
    import copy
    import bench
    import kubernetes_autoscaler_amd as kaa
    from kubernetes_autoscaler_amd.objects import Pod

    def make(seed_offset=0):
        w = workloads.config_c4(seed_offset, n_groups=4, n_pegs=400, pods_per_peg=2, cap=6)
        if spoil and seed_offset == 1:
            w = copy.deepcopy(w)
            victim = next(pg.pods[0] for pg in w.pegs if pg.pods[0].anti_affinity)
            w.groups[2].template.pods.append(Pod(name="ds", requests={"cpu": 10, "memory": 1 << 20}, labels=dict(victim.labels)))
        return w
    return make, bench.simulation_tables(make, range(n_seeds), kaa.Encoder, TableSet)


@pytest.mark.parametrize("spoil", [False, True], ids=["vacuous", "a-template-holds-a-marked-pod"])
def test_exclusion_words_that_say_nothing_about_a_fresh_node_take_the_simulation_major_path(spoil, monkeypatch):
    """round 6: exclusion words speak about a FRESH node only through what the template already holds ((block & init_excl) != (block & polarity),
    fits_fresh_node).  Pod anti-affinity between PENDING pods alone — BASELINE config C4 — leaves every init_excl word zero: the batch's cells
    are decided by requests, taints and selectors, and the simulation-major kernels (feas_sim / feas_stream + strided lists, one word per mask
    kind) take it although the dictionary is two words wide (the batched C4 row spent 35 % of its step in the dense feas_kernel + scan + fill
    before).  One marked pod on one template and the batch is back on the general path.  Either way: the oracle's answers, and the same
    answers as with the shortcut switched off."""
    import bench
    from harness import emu_lib
    make, ts = _c4_batch(3, spoil)
    assert ts.dims["w_excl"] == 2, ts.dims
    res, _ = run_emu_tables(ts)
    path = emu_lib().emu_last_front()
    assert (path == 2) == (not spoil), path           # 2 = strided lists (simulation-major feasibility); 0 / 1 = dense kernel + CSR, or the fused front kernel of small calls
    chk = bench.verify_headline(workloads, make, 3, ts, res)
    assert chk["headline_bit_exact"], chk
    monkeypatch.setenv("CASIM_NO_VACUOUS_EXCL", "1")
    ref, _ = run_emu_tables(ts)
    assert emu_lib().emu_last_front() != 2
    for f in ("node_count", "pods_scheduled", "nodes_added", "last_index_out", "status", "order", "placed", "offsets"):
        assert list(getattr(res, f)) == list(getattr(ref, f)), f
    res64, _ = run_emu_tables(ts, generic=True)
    for f in ("node_count", "pods_scheduled", "order", "placed"):
        assert list(getattr(res64, f)) == list(getattr(ref, f)), f


@pytest.mark.parametrize("shape", ["zeros-and-ties", "forty-binades", "narrow"])
def test_one_wave_networks_on_extreme_scores(shape, monkeypatch):
    """the one-wave (key, position) networks of the batch geometry on score shapes the corpus does not hold: "zeros-and-ties" — requests of zero
    (score +0.0) next to many equal scores (the position decides); "forty-binades" — a one-milli request on a node of 10^12 milli next to
    requests of most of a node (2^-40 against 2^-1); "narrow" — the usual shape.  Written for round 6's experiment with ONE 64-bit key per
    element (exponent code, mantissa, position: exact when the list's scores lie within 31 binades; measured 1 % of the orderer, not kept —
    DESIGN section 8); the cases stay as a check of the general network against the oracle's order."""
    from kubernetes_autoscaler_amd.objects import NodeInfo, Pod, PodEquivalenceGroup
    from kubernetes_autoscaler_amd.workloads import _node, SplitMix64
    monkeypatch.setenv("CASIM_TEST_BATCH_GROUPS", "8")
    rng = SplitMix64(0x1E7 + len(shape))
    scs = []
    for s in range(2):
        pegs = []
        for i in range(90 + 20 * s):
            if shape == "zeros-and-ties":
                req = {"cpu": 0 if i % 7 == 0 else 100 * (1 + rng.below(3)), "memory": 0 if i % 5 == 0 else (128 << 20) * (1 + rng.below(2))}
            elif shape == "forty-binades":
                req = {"cpu": 1 if i % 3 == 0 else 10 ** 6 * (1 + rng.below(900000)), "memory": 1 if i % 4 == 0 else (1 << 20) * (1 + rng.below(4000))}
            else:
                req = {"cpu": 50 + rng.below(4000), "memory": (1 << 20) * (16 + rng.below(8000))}
            pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"{shape}{s}p{i}", requests=req)] * (1 + rng.below(2))))
        big = shape == "forty-binades"
        groups = [GroupSpec(NodeInfo(_node(f"t{s}-{k}", (10 ** 12 if big else 4000) * (1 + k), ((1 << 42) if big else (16 << 30)) * (1 + k), 110, {})),
                            max_nodes=4 + k, last_index=0, pegs=None) for k in range(4)]
        scs.append(Scenario(pegs=pegs, groups=groups, device_csr=True))
    enc, ts, bases = encode_batch(scs)
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend(_shift(run_oracle(sc), pb))
    for generic in (False, True):
        res, _ = run_emu_tables(ts, generic=generic)
        assert_matches_oracle(res, want, f"{shape} generic={generic}")
    enc.close()
