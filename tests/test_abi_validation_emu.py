"""Argument validation of the host orchestration behind the newer C ABI entries (casim_try_schedule_pods,
casim_simulate_node_removals, casim_estimate_on_cluster), exercised through the emulator build of the same
casim_pipeline.h code: malformed tables come back as error codes, never as a crash or a silent result."""
import ctypes as C

import numpy as np
import pytest

from harness import emu_lib, removal_encode, sched_encode, RemovalCase, SchedCase
from kubernetes_autoscaler_amd import _abi
from kubernetes_autoscaler_amd.engine import (alloc_removal_results, make_cluster_estimate, make_pod_sequence,
                                              make_removal_candidates)
from kubernetes_autoscaler_amd.objects import NodeInfo, build_test_pod
from kubernetes_autoscaler_amd.workloads import _node


def _lib():
    L = emu_lib()
    L.emu_try_schedule_pods.restype = C.c_int32
    L.emu_try_schedule_pods.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.PodSequence), C.c_int64,
                                        _abi.i32p, _abi.i32p, _abi.i32p, _abi.i32p]
    L.emu_simulate_node_removals.restype = C.c_int32
    L.emu_simulate_node_removals.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.RemovalCandidates), C.c_int64,
                                             C.POINTER(_abi.RemovalResults)]
    L.emu_estimate_on_cluster.restype = C.c_int32
    L.emu_estimate_on_cluster.argtypes = [C.POINTER(_abi.Pegs), C.POINTER(_abi.Groups), C.POINTER(_abi.ClusterEstimate), C.c_int64,
                                          C.POINTER(_abi.ClusterEstimateResult)]
    return L


def _cluster():
    nodes = [NodeInfo(_node(f"n{i}", 1000, 10**9, 10)) for i in range(3)]
    nodes[0].pods.append(build_test_pod("r", 100, 1))
    return nodes


def _try(seq, enc):
    out = np.full(8, -1, np.int32)
    li, ns = C.c_int32(0), C.c_int32(0)
    return _lib().emu_try_schedule_pods(C.byref(enc.pegs), C.byref(enc.groups), C.byref(seq), 0, out.ctypes.data_as(_abi.i32p), C.byref(li),
                                        C.byref(ns), None)


def test_try_schedule_pods_rejects_bad_tables():
    enc, pod_class = sched_encode(SchedCase(nodes=_cluster(), pods=[build_test_pod("p", 100, 1)]))
    seq, keep = make_pod_sequence([5])                      # class out of range
    assert _try(seq, enc) == _abi.ERR_INVALID
    seq, keep = make_pod_sequence([-1])
    assert _try(seq, enc) == _abi.ERR_INVALID
    seq, keep = make_pod_sequence(pod_class, hint_node=[99])   # a hint beyond the cluster is "node gone", not an error
    assert _try(seq, enc) == 0
    rules = _abi.DomainRules(n_rules=1, n_nodes=7, n_classes=1)   # rules built for other tables
    seq, keep = make_pod_sequence(pod_class, rules=rules)
    assert _try(seq, enc) == _abi.ERR_INVALID
    enc.close()


def test_removals_reject_bad_tables():
    case = RemovalCase(nodes=_cluster(), candidates=[0])
    enc, pod_class, off = removal_encode(case)

    def run(cands, offsets, classes):
        st, keep = make_removal_candidates(cands, offsets, classes)
        res, packed = alloc_removal_results(st)
        return _lib().emu_simulate_node_removals(C.byref(enc.pegs), C.byref(enc.groups), C.byref(st), 0, C.byref(res))
    assert run([0], off, pod_class) == 0
    assert run([3], off, pod_class) == _abi.ERR_INVALID           # candidate beyond the node table
    assert run([-1], off, pod_class) == _abi.ERR_INVALID
    assert run([0, 1], [0, 1, 0], pod_class) == _abi.ERR_INVALID  # offsets not monotone
    assert run([0], [1, 1], pod_class) == _abi.ERR_INVALID        # offsets must start at 0
    assert run([0], off, [9]) == _abi.ERR_INVALID                 # class out of range
    enc.close()


def test_cluster_estimate_rejects_bad_tables():
    from kubernetes_autoscaler_amd.estimator import encode_cluster_estimate
    from kubernetes_autoscaler_amd.objects import PodEquivalenceGroup
    nodes = _cluster()
    enc = encode_cluster_estimate(("cpu", "memory"), [PodEquivalenceGroup(pods=[build_test_pod("p", 100, 1)] * 3)], nodes[:2], nodes[2], 2)

    def run(n_existing):
        params, res, arrs = make_cluster_estimate(enc.pegs, n_existing, 2, 0, enc.rules, enc.port_block)
        return _lib().emu_estimate_on_cluster(C.byref(enc.pegs), C.byref(enc.groups), C.byref(params), 0, C.byref(res))
    assert run(2) == 0
    assert run(4) == _abi.ERR_INVALID     # no template clone left in the table
    assert run(-1) == _abi.ERR_INVALID
    enc.close()


def test_reserved_peg_flag_bits_are_refused():
    """bit 0x40 of casim_pegs.flags is the library's own (CASIM_KFLAG_SINGLETON_RUN: the packer then applies the lastIndex
    rule of merged singleton rows) and the bits above it are record bits — a caller that sets one gets CASIM_ERR_INVALID, not a changed
    last_index_out."""
    from harness import GroupSpec, Scenario, alloc_results, encode
    from kubernetes_autoscaler_amd import workloads
    w = workloads.config_c0()
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes)
    enc = encode(sc)
    L = emu_lib()

    def run():
        st, arrs = alloc_results(enc.groups.n_groups, enc.pegs.n_pegs * enc.groups.n_groups)
        opts = _abi.Options()
        nnz = C.c_int32(0)
        off = np.zeros(enc.groups.n_groups + 1, np.int32)
        return L.emu_estimate_batch(C.byref(enc.pegs), C.byref(enc.groups), C.byref(opts), C.byref(st), 0, C.byref(nnz), off.ctypes.data_as(_abi.i32p),
                                    None, -1, 0, None, None, None)
    assert run() == 0
    flags = np.ctypeslib.as_array(enc.pegs.flags, shape=(enc.pegs.n_pegs,))
    for bit in (0x40, 0x80, 0x100, 0x20000000, 0x40000000, 0x80000000):
        keep = int(flags[1])
        flags[1] = keep | bit
        assert run() == _abi.ERR_INVALID, hex(bit)
        flags[1] = keep
    assert run() == 0
    enc.close()
