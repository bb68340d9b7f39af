"""feas_stream_kernel (round 5): the streaming form of the SchedulablePodGroups matrix for batches — group records as scalar loads, the
static Filters of a cell as ONE accumulated word, gates folded into the data (csrc/casim_kernels.h).  Every instantiation against the
oracle AND against the LDS-staged feas_sim_kernel it replaces (CASIM_NO_FEAS_STREAM=1), on batches that exercise what the folding could
get wrong: groups without a free pod slot, unschedulable templates next to tolerating PEGs, PEGs without requests, dictionaries
beyond 32 entries (upper mask halves), exclusion words with NEED polarity, four narrowed lanes, ragged simulations (rows that end inside
a word, simulations with more than 64 groups).  CPU: product kernels under the wave emulator."""
import os

import numpy as np
import pytest

from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, PodEquivalenceGroup, Taint, Toleration
from harness import GroupSpec, Scenario, assert_matches_oracle, encode_batch, run_emu_tables, run_oracle

GiB = 1 << 30


def _run_both(ts, capfd, resident=True, **kw):
    """resident: the casim_problem_create form (init waits for the device and learns which mask bits the batch uses: the instantiations
    without upper-half terms / with NodeUnschedulable on a spare bit); else the one-shot call of casim_estimate_batch_query (general one)"""
    os.environ["CASIM_FEAS_TRACE"] = "1"
    os.environ.pop("CASIM_NO_FEAS_STREAM", None)
    if resident:
        os.environ["CASIM_EMU_RESIDENT"] = "1"
    try:
        new, _ = run_emu_tables(ts, **kw)
        trace = capfd.readouterr().err
        os.environ["CASIM_NO_FEAS_STREAM"] = "1"
        old, _ = run_emu_tables(ts, **kw)
    finally:
        os.environ.pop("CASIM_NO_FEAS_STREAM", None)
        os.environ.pop("CASIM_FEAS_TRACE", None)
        os.environ.pop("CASIM_EMU_RESIDENT", None)
    assert "[feas] feas_stream_kernel" not in capfd.readouterr().err      # the A/B switch really switches
    for f in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "offsets", "order", "placed"):
        assert np.array_equal(getattr(new, f), getattr(old, f)), f
    return new, trace


def _want(scs, bases):
    out = []
    for sc, (pb, _) in zip(scs, bases):
        out.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc)])
    return out


def _sc(seed, **kw):
    w = workloads.fuzz(seed, **kw)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True, lanes=w.lanes)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_batches_on_the_streaming_kernel(seed, capfd):
    """rich fuzz simulations (taints, tolerations, selectors, unschedulable templates, ports, anti-affinity: exclusion words): the full
    instantiation; plain ones: the lean one"""
    rich = seed % 2 == 0
    scs = [_sc(52000 + 97 * seed + k, max_groups=6, max_pegs=90 if seed % 3 == 0 else 14, rich=rich) for k in range(2 + seed % 5)]
    enc, ts, bases = encode_batch(scs)
    res, trace = _run_both(ts, capfd, resident=seed % 4 < 2)
    if "[feas] feas_stream_kernel<" not in trace:
        pytest.skip("this batch does not take the simulation-major kernel (mask widths > 1 word or ragged PEG ranges)")
    assert_matches_oracle(res, _want(scs, bases), f"seed {seed}")
    enc.close()


def _tmpl(name, cpu=8000, mem=32 * GiB, pods=110, taints=(), labels=None, unschedulable=False, pre=()):
    cap = {"cpu": cpu, "memory": mem, "pods": pods}
    node = Node(name=name, labels=dict(labels or {}), taints=list(taints), allocatable=dict(cap), capacity=dict(cap))
    node.unschedulable = unschedulable
    return NodeInfo(node, list(pre))


def test_the_gates_folded_into_the_group_record(capfd):
    """a template whose pod slots are all taken (INT32_MIN in lane 0: even a PEG without requests fails), an unschedulable template (only the
    PEG that tolerates node.kubernetes.io/unschedulable passes), a PEG with no requests at all, a PEG that asks for more than any template has"""
    filler = [Pod(name=f"ds{i}", namespace="kube-system", requests={"cpu": 10, "memory": 1 << 20}) for i in range(3)]
    groups = [GroupSpec(_tmpl("roomy"), 0, 0, None), GroupSpec(_tmpl("no-slot-left", pods=3, pre=filler), 0, 0, None),
              GroupSpec(_tmpl("cordoned", unschedulable=True), 0, 0, None), GroupSpec(_tmpl("small", cpu=500, mem=1 * GiB), 0, 0, None)]
    tol = [Toleration(key="node.kubernetes.io/unschedulable", operator="Exists", effect="NoSchedule")]
    pegs = [PodEquivalenceGroup(pods=[Pod(name="plain", requests={"cpu": 1000, "memory": 2 * GiB})] * 5),
            PodEquivalenceGroup(pods=[Pod(name="nothing", requests={})] * 4),
            PodEquivalenceGroup(pods=[Pod(name="tolerates-cordon", requests={"cpu": 100, "memory": 1 << 28}, tolerations=tol)] * 3),
            PodEquivalenceGroup(pods=[Pod(name="too-big", requests={"cpu": 64000, "memory": 1 * GiB})] * 2),
            PodEquivalenceGroup(pods=[Pod(name="memory-only", requests={"memory": 512 << 20})] * 6)]
    scs = [Scenario(pegs=pegs, groups=groups, device_csr=True), Scenario(pegs=pegs[::-1], groups=groups[::-1], device_csr=True)]
    enc, ts, bases = encode_batch(scs)
    res, trace = _run_both(ts, capfd)
    assert "feas_stream_kernel<lean" in trace
    want = _want(scs, bases)
    assert_matches_oracle(res, want, "gates")
    lists = {g.template.node.name: sorted(ids) for g, (_, ids) in zip(groups, want[:4])}
    assert lists == {"roomy": [0, 1, 2, 4], "no-slot-left": [], "cordoned": [2], "small": [1, 2, 4]}, lists
    enc.close()


def test_dictionaries_beyond_31_entries_use_the_upper_mask_halves(capfd):
    """40 distinct taints and 40 selector pairs: bits 32-39 of the words decide cells — the instantiation without upper-half terms must NOT be
    chosen, and a batch whose dictionaries stay in the lower halves takes it; both find a spare bit for NodeUnschedulable"""
    def batch(n_keys):
        groups = [GroupSpec(_tmpl(f"t{i}", taints=[Taint(f"k{i}", "v", "NoSchedule")], labels={f"l{i}": "x"}), 0, 0, None) for i in range(n_keys)]
        pegs = []
        for i in range(n_keys):
            pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": 100, "memory": 1 << 28}, node_selector={f"l{i}": "x"},
                                                      tolerations=[Toleration(key=f"k{i}", operator="Exists")])] * 2))
        return [Scenario(pegs=pegs, groups=groups, device_csr=True), Scenario(pegs=pegs[5:], groups=groups[3:], device_csr=True)]
    for n_keys, tag in ((40, "hi, bit>"), (12, "lo, bit>")):
        scs = batch(n_keys)
        enc, ts, bases = encode_batch(scs)
        res, trace = _run_both(ts, capfd)
        assert f"feas_stream_kernel<lean, {tag}" in trace, trace[-300:]
        want = _want(scs, bases)
        assert_matches_oracle(res, want, tag)
        assert [ids for _, ids in want[:n_keys]] == [[i] for i in range(n_keys)]      # PEG i fits template i only (its taint, its label)
        enc.close()


def test_more_than_64_groups_and_rows_that_end_inside_a_word(capfd):
    """a simulation of 70 node groups (two rounds of the 64-lane word store) next to one of 3; 130 PEGs (rows of three words, the last one
    two bits long)"""
    rng = np.random.default_rng(7)
    pegs = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": int(rng.choice([100, 500, 2000, 9000])), "memory": int(rng.choice([1, 4, 40])) << 28})] * int(rng.integers(1, 4)))
            for i in range(130)]
    groups = [GroupSpec(_tmpl(f"g{i}", cpu=int(rng.choice([1000, 4000, 16000])), mem=int(rng.choice([2, 16, 64])) * GiB), 3, 0, None) for i in range(70)]
    scs = [Scenario(pegs=pegs, groups=groups, device_csr=True), Scenario(pegs=pegs[:130], groups=groups[:3], device_csr=True)]
    enc, ts, bases = encode_batch(scs)
    res, trace = _run_both(ts, capfd)
    assert "feas_stream_kernel<" in trace
    assert_matches_oracle(res, _want(scs, bases), "70 groups")
    enc.close()


def test_no_spare_bit_and_an_unschedulable_template(capfd):
    """64 taints AND 64 label requirements in use: no spare bit anywhere — NodeUnschedulable stays a term of its own (<.., hi, term>), and with
    one bit free it rides there; an unschedulable template among the groups makes the bit matter"""
    from kubernetes_autoscaler_amd.objects import Taint, Toleration
    tol_unsched = Toleration(key="node.kubernetes.io/unschedulable", operator="Exists", effect="NoSchedule")
    for n_keys, tag in ((64, "hi, term>"), (63, "hi, bit>")):
        groups = [GroupSpec(_tmpl(f"t{i}", taints=[Taint(f"k{i}", "v", "NoSchedule")], labels={f"l{i}": "x"}, unschedulable=i % 5 == 0), 0, 0, None) for i in range(n_keys)]
        pegs = [PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests={"cpu": 100, "memory": 1 << 28}, node_selector={f"l{i}": "x"},
                                              tolerations=[Toleration(key=f"k{i}", operator="Exists")] + ([tol_unsched] if i % 10 == 0 else []))] * 2) for i in range(n_keys)]
        scs = [Scenario(pegs=pegs, groups=groups, device_csr=True), Scenario(pegs=pegs[7:], groups=groups[2:], device_csr=True)]
        enc, ts, bases = encode_batch(scs)
        if enc.pegs.w_taint > 1 or enc.pegs.w_label > 1:
            enc.close()
            pytest.skip("the encoder spent a second mask word")
        res, trace = _run_both(ts, capfd)
        assert f"feas_stream_kernel<lean, {tag}" in trace, trace[-300:]
        want = _want(scs, bases)
        assert_matches_oracle(res, want, tag)
        # PEG i fits template i only — and not even that one when the template is cordoned and the PEG does not tolerate it
        assert [ids for _, ids in want[:n_keys]] == [[i] if (i % 5 != 0 or i % 10 == 0) else [] for i in range(n_keys)]
        enc.close()
