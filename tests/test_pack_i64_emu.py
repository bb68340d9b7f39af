"""The register packer on int64 lanes (RegStore<2, NPT, WX, int64_t>, pack_fast64_kernel; VERDICT r3 weak #6 / next #4b): batches whose
lanes do not narrow to int32 after the gcd no longer fall on the LDS store's generic packer.  Same kernel source under the wave emulator:
   * casim_options.force_generic_packer == 2 takes it for ANY eligible batch (R <= 2, no negative request): every fuzz family against the
     oracle and against the int32 store;
   * byte-granular co-prime memory amounts beyond 2^31 select it by themselves (emu_last_packer() == 8xx), amounts beyond 2^53 take the
     real division;
   * three lanes, negative requests: still the generic packer."""
import numpy as np
import pytest

from harness import GroupSpec, Scenario, assert_matches_oracle, emu_lib, encode, run_emu, run_oracle
from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, PodEquivalenceGroup

FIELDS = ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "req_cpu_sum", "req_mem_sum", "order", "placed", "status")


def same(a, b, what=""):
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), (what, f)


@pytest.mark.parametrize("seed", range(60))
def test_forced_int64_register_packer_matches_the_oracle_and_the_int32_store(seed):
    import test_kernels_emu_fuzz as F
    sc = F.scenario_of(workloads.fuzz(seed, rich=seed % 3 == 0))
    enc = encode(sc)
    r32, _ = run_emu(enc)
    r64, _ = run_emu(enc, generic=2)
    lanes64 = emu_lib().emu_last_packer() // 100
    assert_matches_oracle(r64, run_oracle(sc), f"seed {seed}")
    same(r32, r64, f"seed {seed}")
    assert lanes64 in (0, 8)
    if len(sc.lanes) <= 2 and not (seed % 3 == 0):
        assert lanes64 == 8   # resource-only two-lane batches are always eligible


@pytest.mark.parametrize("seed", range(40))
def test_forced_int64_fastpath_device_lists_and_long_lists(seed):
    import test_kernels_emu_fuzz as F
    kind = seed % 4
    if kind == 0:
        sc = F.scenario_of(workloads.fuzz(2000 + seed), fastpath=True)
        res, _ = run_emu(encode(sc), fastpath=True, generic=2)
    elif kind == 1:
        sc = F.scenario_of(workloads.fuzz(3000 + seed), device_csr=True)
        res, _ = run_emu(encode(sc), generic=2)
    elif kind == 2:
        sc = F.scenario_of(workloads.fuzz(5000 + seed, max_groups=3, max_pegs=48))
        res, _ = run_emu(encode(sc), generic=2)
    else:
        sc = F.scenario_of(workloads.fuzz(4000 + seed))
        res, _ = run_emu(encode(sc), lds_budget=256, generic=2)
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed} kind {kind}")


def shape_scenario(seed, mem_unit, mem_scale, lanes=("cpu", "memory"), negative=False):
    rng = workloads.SplitMix64(9100 + seed)
    cap_nodes = rng.pick([5, 64, 65, 200, 256, 257, 700, 1024, 0])
    alloc = {"cpu": 1000 * rng.pick([4, 16, 64]), "memory": mem_unit * mem_scale * rng.pick([64, 1000, 4096]) + rng.pick([0, 1, 12345]), "pods": rng.pick([8, 30, 110])}
    if len(lanes) == 3:
        alloc["ephemeral-storage"] = 100 << 30
    node = Node(name=f"wide{seed}", labels={}, allocatable=dict(alloc), capacity=dict(alloc))
    pegs = []
    for i in range(1 + rng.below(40)):
        # co-prime byte amounts: the lane's gcd is 1 and the allocatable does not fit 31 bits
        req = {"cpu": 50 * rng.below(60), "memory": mem_unit * rng.below(40) * mem_scale + (rng.below(1000) * 2 + 1 if rng.below(3) else 0)}
        if len(lanes) == 3:
            req["ephemeral-storage"] = rng.below(5) << 30
        if negative and i == 2:
            req["cpu"] = -100
        pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests=req)] * rng.pick([1, 3, 50, 400])))
    return Scenario(pegs=pegs, groups=[GroupSpec(NodeInfo(node), max_nodes=cap_nodes, last_index=rng.below(5))], existing=[], lanes=lanes)


@pytest.mark.parametrize("seed", range(40))
def test_lanes_that_do_not_narrow_select_the_int64_register_store(seed):
    """memory in bytes with odd amounts: gcd 1, allocatable 64 GiB .. 4 TiB > 2^31: round 3 sent these to the LDS store"""
    sc = shape_scenario(seed, 1 << 20, 1024)
    enc = encode(sc)
    res, _ = run_emu(enc)
    packer = emu_lib().emu_last_packer()
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")
    gen, _ = run_emu(enc, generic=True)
    assert emu_lib().emu_last_packer() == 0
    same(res, gen, f"seed {seed}")
    assert packer // 100 == 8, packer


@pytest.mark.parametrize("seed", range(20))
def test_amounts_beyond_2_to_53_take_the_division(seed):
    sc = shape_scenario(seed, 1 << 40, 1 << 9)   # 2^55 .. 2^61 "bytes"
    enc = encode(sc)
    res, _ = run_emu(enc)
    assert emu_lib().emu_last_packer() // 100 == 8
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")


@pytest.mark.parametrize("seed", range(10))
def test_three_wide_lanes_and_negative_requests_stay_on_the_generic_packer(seed):
    sc = shape_scenario(seed, 1 << 20, 1024, lanes=("cpu", "memory", "ephemeral-storage"))
    res, _ = run_emu(encode(sc))
    assert emu_lib().emu_last_packer() == 0
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")
    sc = shape_scenario(seed, 1 << 20, 1024, negative=True)
    res, _ = run_emu(encode(sc), generic=2)
    assert emu_lib().emu_last_packer() == 0
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed} negative")


@pytest.mark.parametrize("seed", range(12))
def test_batches_of_simulations_on_the_int64_store(seed):
    """a TableSet of simulations (device-derived fixed-stride lists, the expander per simulation, the cut into stream parts): the int64
    register store against the int32 one, field by field"""
    from kubernetes_autoscaler_amd import _abi
    from harness import encode_batch, run_emu_streams, run_emu_tables
    n = 2 + (seed * 3) % 7
    scs = []
    for i in range(n):
        w = workloads.fuzz(7100 + 31 * seed + i, max_groups=5, max_pegs=20, rich=(seed % 2 == 0))
        scs.append(Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True))
    enc, ts, _ = encode_batch(scs)
    kinds = [_abi.EXPANDER_LEAST_WASTE]
    a, ea = run_emu_tables(ts, kinds=kinds)
    b, eb = run_emu_tables(ts, kinds=kinds, generic=2)
    lanes = emu_lib().emu_last_packer() // 100
    assert lanes in (0, 8)
    if seed % 2 == 1:
        assert lanes == 8
    c, ec, parts = run_emu_streams(ts, 3, kinds=kinds, generic=2)
    for f in FIELDS + ("offsets",):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert np.array_equal(getattr(a, f), getattr(c, f)), ("streams", f)
    for f in ("best", "n_best", "packed"):
        assert list(ea[f]) == list(eb[f]) == list(ec[f]), f
    d, ed, _ = run_emu_streams(ts, 2, kinds=kinds, generic=2, winners_only=True)
    assert list(ed["best"]) == list(ea["best"])
