"""Resources BY NAME behind the C ABI (ABI 9; VERDICT r4 missing #2).  The reference keeps every scalar / extended resource by name in
Resource.ScalarResources and checks each entry of the pod's request map (V/.../noderesources/fit.go:731-763; accumulation
framework/types.go:444-448).  Round 4's Go binding copied cpu / memory / ephemeral-storage only: a pod asking for nvidia.com/gpu was
estimated WITHOUT the request, silently.  Now a binding hands every entry over under its Kubernetes name
(casim_enc_pod_set_request / casim_enc_group_set_allocatable) and the encoder owns the lane numbers; a name that finds no lane DELEGATES
the pod (CASIM_PEG_UNSUPPORTED) — a request is never dropped.  `Encoder(named_lanes=True)` makes exactly the calls
integration/go/gpubinpacking/encode.go makes."""
import ctypes as C

import numpy as np
import pytest

from kubernetes_autoscaler_amd import _abi, trace
from kubernetes_autoscaler_amd._ffi import lib
from kubernetes_autoscaler_amd.encoder import Encoder
from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, PodEquivalenceGroup
from harness import GroupSpec, Scenario, assert_matches_oracle, encode, run_emu, run_oracle
from orchestrator_rows import ROWS, Row, per_group_of_batch

GPU_ROWS = [r for r in ROWS if any(p.get("gpu", 0) > 0 for p in r["extra_pods"]) or any(n.get("gpu", 0) > 0 for n in r["nodes"])]
GiB = 1 << 30


def test_the_reference_holds_three_gpu_pool_rows():
    assert len(GPU_ROWS) == 3, [r["name"] for r in GPU_ROWS]


@pytest.mark.parametrize("generic", [False, True], ids=["register-packer", "int64-packer"])
@pytest.mark.parametrize("row", GPU_ROWS, ids=[r["name"] for r in GPU_ROWS])
def test_gpu_pool_rows_through_the_call_sequence_of_the_go_binding(row, generic):
    """orchestrator_test.go's GPU-pool rows (a NoSchedule taint + an Exists toleration + nvidia.com/gpu inside a scale-up) with the extended
    resource handed over BY NAME, three positional lanes — the emulator must find the reference's option"""
    r = Row(row)
    sc = r.scenario()
    enc = encode(sc, named_lanes=True)
    assert enc.lanes == ("cpu", "memory", "ephemeral-storage", "nvidia.com/gpu"), enc.lanes
    res, _ = run_emu(enc, generic=generic)
    assert all(int(s) == 0 for s in res.status)
    r.check(r.decide(per_group_of_batch(res)), "emulator, named lanes:")
    assert_matches_oracle(res, run_oracle(sc), row["name"])
    enc.close()


def _tmpl(name, extra=None, cpu=8000, mem=32 * GiB):
    cap = {"cpu": cpu, "memory": mem, "pods": 110}
    cap.update(extra or {})
    return NodeInfo(Node(name=name, labels={}, allocatable=dict(cap), capacity=dict(cap)), [])


def test_a_hugepages_pod_is_encoded_never_dropped():
    """hugepages-2Mi is a scalar resource of the scheduler (schedutil.IsScalarResourceName): 3 pods x 1 GiB of huge pages on nodes that hold 2 GiB
    need two nodes — an estimate that ignored the request would say one"""
    huge = "hugepages-2Mi"
    pods = [Pod(name="hp", requests={"cpu": 100, "memory": 64 << 20, huge: 1 * GiB})] * 3
    sc = Scenario(pegs=[PodEquivalenceGroup(pods=pods)], groups=[GroupSpec(_tmpl("with-hugepages", {huge: 2 * GiB}), 0, 0, None),
                                                                GroupSpec(_tmpl("without"), 0, 0, None)],
                  existing=[], lanes=("cpu", "memory", huge), device_csr=True)
    enc = encode(sc, named_lanes=True)
    assert huge in enc.lanes and enc.pegs.n_res == 4
    res, _ = run_emu(enc)
    want = run_oracle(sc)
    assert_matches_oracle(res, want, "hugepages")
    assert int(res.node_count[0]) == 2 and int(res.pods_scheduled[0]) == 3       # two nodes, not one
    assert int(res.pods_scheduled[1]) == 0 and want[1][1] == []                    # a template without huge pages takes none of them
    enc.close()


def test_a_name_without_a_lane_delegates_the_pod():
    """CASIM_MAX_RES lanes: cpu, memory, ephemeral-storage + five names.  The sixth name finds none: casim_enc_pod_set_request answers
    CASIM_ENC_DELEGATED, the pod spec is CASIM_PEG_UNSUPPORTED, every group that lists it comes back CASIM_NG_UNSUPPORTED — and a pod that
    asks for ZERO of the name stays inside (fit.go:733 skips zero quantities)."""
    names = [f"example.com/dev{i}" for i in range(6)]
    enc = Encoder(named_lanes=True)
    h = enc._h
    vec = (C.c_int64 * _abi.MAX_RES)(100, 1 << 20, 0)
    a = lib.casim_enc_add_pod_spec(h, b"default", vec)
    for i, n in enumerate(names[:5]):
        assert lib.casim_enc_pod_set_request(h, a, n.encode(), 1) == _abi.OK
        assert lib.casim_enc_lane(h, n.encode()) == 3 + i
    b = lib.casim_enc_add_pod_spec(h, b"default", vec)
    assert lib.casim_enc_lane(h, names[5].encode()) == _abi.ERR_NO_LANE
    assert lib.casim_enc_pod_set_request(h, b, names[5].encode(), 2) == _abi.ENC_DELEGATED
    c = lib.casim_enc_add_pod_spec(h, b"default", vec)
    assert lib.casim_enc_pod_set_request(h, c, names[5].encode(), 0) == _abi.OK
    for spec in (a, b, c):
        assert lib.casim_enc_add_peg(h, spec, 3) >= 0
    alloc = (C.c_int64 * _abi.MAX_RES)(4000, 16 << 30, 0)
    g = lib.casim_enc_add_group(h, b"tmpl", alloc, 110, 4000, 16 << 30, 0)
    for n in names[:5]:
        assert lib.casim_enc_group_set_allocatable(h, g, n.encode(), 8) == _abi.OK
    assert lib.casim_enc_group_set_allocatable(h, g, names[5].encode(), 8) == _abi.OK   # (kept aside: no pod request opened a lane for it)
    assert lib.casim_enc_group_set_allocatable(h, g, b"pods", 17) == _abi.OK
    assert lib.casim_enc_lane(h, b"pods") == _abi.ERR_INVALID and lib.casim_enc_lane(h, b"") == _abi.ERR_INVALID
    assert lib.casim_enc_group_set_limits(h, g, 0, 0, 0) == _abi.OK
    pegs, groups = enc.finalize()
    assert pegs.n_res == _abi.MAX_RES == lib.casim_enc_lane_count(h) and enc.lanes[3:] == tuple(names[:5])
    flags = [int(pegs.flags[i]) for i in range(3)]
    assert not flags[0] & _abi.PEG_UNSUPPORTED and flags[1] & _abi.PEG_UNSUPPORTED and not flags[2] & _abi.PEG_UNSUPPORTED
    assert int(groups.allowed_pods[0]) == 17
    req = np.ctypeslib.as_array(pegs.req, shape=(3, _abi.MAX_RES))
    assert list(req[0]) == [100, 1 << 20, 0, 1, 1, 1, 1, 1] and list(req[1][3:]) == [0] * 5
    res, _ = run_emu(enc)
    assert int(res.status[0]) == _abi.NG_UNSUPPORTED     # the group is delegated as a whole: the shim runs the reference path for it
    # new names after finalize find no lane either (update sessions re-encode rows of fixed width)
    assert lib.casim_enc_lane(h, b"example.com/late") == _abi.ERR_NO_LANE and lib.casim_enc_lane(h, names[2].encode()) == 5
    enc.close()


@pytest.mark.parametrize("seed", range(40))
def test_named_lanes_give_the_tables_of_positional_lanes(seed):
    """fuzz: requests on 0-3 extended resources per pod / template; the named encoder's tables equal the positional encoder's with the
    same lane order, and both equal the oracle"""
    rng = np.random.default_rng(900 + seed)
    ext = [f"vendor.io/r{i}" for i in range(int(rng.integers(1, 4)))]
    pegs = []
    for i in range(int(rng.integers(2, 9))):
        rq = {"cpu": int(rng.choice([100, 250, 1000])), "memory": int(rng.choice([1, 2, 4])) << 28}
        for n in ext:
            if rng.integers(0, 3) == 0:
                rq[n] = int(rng.integers(0, 4))
        if rng.integers(0, 4) == 0:
            rq["ephemeral-storage"] = int(rng.integers(1, 5)) << 30
        pegs.append(PodEquivalenceGroup(pods=[Pod(name=f"p{i}", requests=rq)] * int(rng.integers(1, 30))))
    groups = []
    for gi in range(int(rng.integers(1, 5))):
        extra = {n: int(rng.integers(1, 9)) for n in ext if rng.integers(0, 3) > 0}
        extra["ephemeral-storage"] = int(rng.integers(8, 64)) << 30
        groups.append(GroupSpec(_tmpl(f"t{gi}", extra), int(rng.choice([0, 3, 20])), 0, None))
    # the encoder's rule: a name gets a lane when a pod asks for a NON-ZERO amount of it, in order of first such request (PEG order);
    # a request of zero and a template's Allocatable open none (a column no pod reads is not a column)
    seen = []
    for pg in pegs:
        for n, v in pg.pods[0].requests.items():
            if n in ext and v > 0 and n not in seen:
                seen.append(n)
    lanes = ("cpu", "memory", "ephemeral-storage", *seen)
    sc = Scenario(pegs=pegs, groups=groups, existing=[], lanes=lanes, device_csr=True)
    named, positional = encode(sc, named_lanes=True), encode(sc)
    assert named.lanes == lanes
    for col, n in (("req", named.pegs.n_pegs * named.pegs.n_res),):
        assert list(np.ctypeslib.as_array(getattr(named.pegs, col), shape=(n,))) == list(np.ctypeslib.as_array(getattr(positional.pegs, col), shape=(n,)))
    n = named.groups.n_groups * named.pegs.n_res
    assert list(np.ctypeslib.as_array(named.groups.alloc, shape=(n,))) == list(np.ctypeslib.as_array(positional.groups.alloc, shape=(n,)))
    res, _ = run_emu(named, generic=seed % 2 == 1)
    assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")
    named.close(); positional.close()


def test_the_named_calls_replay_natively(tmp_path):
    """the new entry points travel through the call trace (kubernetes_autoscaler_amd/trace.py) that tools/casim_native replays from plain
    C++: a recorded GPU-pool row carries casim_enc_pod_set_request / casim_enc_group_set_allocatable lines with the resource name"""
    with trace.recording() as tr:
        enc = encode(Row(GPU_ROWS[0]).scenario(), named_lanes=True)
        enc.close()
    lines = tr.lines
    assert any(ln.startswith("casim_enc_pod_set_request\t") and "nvidia.com/gpu" in ln for ln in lines)
    assert any(ln.startswith("casim_enc_group_set_allocatable\t") and "nvidia.com/gpu" in ln for ln in lines)
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import native_trace as nt
    path = str(tmp_path / "gpu_row.trace")
    tr.save(path)
    enc = encode(Row(GPU_ROWS[0]).scenario(), named_lanes=True)
    rc, out = nt.run_native(path, repeat=1)
    assert out["pegs"] == enc.pegs.n_pegs and out["groups"] == enc.groups.n_groups
    assert out["tables_fnv"] == nt.tables_fnv(enc.pegs, enc.groups)     # the C++ replay built the very tables, extended-resource lane included
    enc.close()


def test_allocatable_nobody_asks_for_opens_no_lane():
    """real nodes list hugepages-1Gi: 0, hugepages-2Mi: 0 and attachable-volumes-*; encode.go hands every
    IsScalarResourceName entry of Allocatable over.  A lane per listed name made the tables five or more lanes wide — past the four the
    register packer, feas_stream_kernel and the lean removal kernel take.  Allocatable opens no lane, a request of zero opens none
    (fit.go:733 skips zero quantities); only a pod's non-zero request does, and then the group's value is found whatever the order of the calls."""
    node_extras = {"hugepages-1Gi": 0, "hugepages-2Mi": 0, "attachable-volumes-aws-ebs": 25, "example.com/fpga": 2}
    enc = Encoder(named_lanes=True)
    h = enc._h
    vec = (C.c_int64 * _abi.MAX_RES)(100, 1 << 20, 0)
    alloc = (C.c_int64 * _abi.MAX_RES)(4000, 16 << 30, 100 << 30)
    g0 = lib.casim_enc_add_group(h, b"before-the-pods", alloc, 110, 4000, 16 << 30, 0)
    for n, v in node_extras.items():
        assert lib.casim_enc_group_set_allocatable(h, g0, n.encode(), v) == _abi.OK
    assert lib.casim_enc_lane_count(h) == 3
    a = lib.casim_enc_add_pod_spec(h, b"default", vec)
    assert lib.casim_enc_pod_set_request(h, a, b"hugepages-2Mi", 0) == _abi.OK      # PodRequests lists the name with a zero quantity
    assert lib.casim_enc_lane_count(h) == 3
    b = lib.casim_enc_add_pod_spec(h, b"default", vec)
    assert lib.casim_enc_pod_set_request(h, b, b"example.com/fpga", 1) == _abi.OK   # the one name a pod needs
    assert lib.casim_enc_lane_count(h) == 4
    g1 = lib.casim_enc_add_group(h, b"after-the-pods", alloc, 110, 4000, 16 << 30, 0)
    for n, v in node_extras.items():
        assert lib.casim_enc_group_set_allocatable(h, g1, n.encode(), v + (1 if n == "example.com/fpga" else 0)) == _abi.OK
    g2 = lib.casim_enc_add_group(h, b"without-the-device", alloc, 110, 4000, 16 << 30, 0)
    for spec in (a, b):
        assert lib.casim_enc_add_peg(h, spec, 5) >= 0
    for g in (g0, g1, g2):
        assert lib.casim_enc_group_set_limits(h, g, 0, 0, 0) == _abi.OK
    pegs, groups = enc.finalize()
    assert pegs.n_res == 4 and enc.lanes == ("cpu", "memory", "ephemeral-storage", "example.com/fpga")
    al = np.ctypeslib.as_array(groups.alloc, shape=(3, 4))
    assert list(al[:, 3]) == [2, 3, 0]
    req = np.ctypeslib.as_array(pegs.req, shape=(2, 4))
    assert list(req[:, 3]) == [0, 1]
    res, _ = run_emu(enc)
    assert all(int(s) == 0 for s in res.status)
    # 5 plain pods fit one node everywhere; 5 fpga pods need ceil(5 / 2) = 3 and ceil(5 / 3) = 2 nodes more, and none fit a node without the device
    assert [int(x) for x in res.pods_scheduled] == [10, 10, 5]
    assert [int(x) for x in res.node_count] == [3, 2, 1]
    enc.close()


def test_an_invalid_request_is_reported_not_recorded():
    """a negative quantity answers CASIM_ERR_INVALID and the request is NOT recorded — the binding must fail closed on rc < 0
    (encode.go marks the pod unsupported; tests/test_go_shim_symbols.py pins that)"""
    enc = Encoder(named_lanes=True)
    h = enc._h
    a = lib.casim_enc_add_pod_spec(h, b"default", (C.c_int64 * _abi.MAX_RES)(100, 1 << 20, 0))
    assert lib.casim_enc_pod_set_request(h, a, b"example.com/dev", -1) == _abi.ERR_INVALID
    assert lib.casim_enc_pod_set_request(h, a, b"pods", 1) == _abi.ERR_INVALID
    assert lib.casim_enc_lane_count(h) == 3
    enc.close()
