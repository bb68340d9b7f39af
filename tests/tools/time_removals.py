#!/usr/bin/env python3
"""SURVEY §8 row f4 measurement: the planner's SimulateNodeRemoval loop on the MI355X (one resident call over all
candidates, HIP-event time of fill + K_sched_static + K_sched with transactions) next to the CPU oracle."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import workloads  # noqa: E402
from harness import RemovalCase, assert_removal_matches, removal_encode, removal_oracle  # noqa: E402

ORACLE_NODE_LIMIT = int(os.environ.get("CASIM_ORACLE_NODE_LIMIT", 5000))
ctx = kaa.Context(0)
for n_nodes, frac in ((100, 0.3), (1000, 0.3), (5000, 0.3), (15000, 0.2)):
    w = workloads.removal_scale(n_nodes, pods_per_node=12, frac_candidates=frac, seed=1)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    t0 = time.perf_counter(); enc, pod_class, off = removal_encode(case); t_enc = time.perf_counter() - t0
    args = (enc.pegs, enc.groups, case.candidates, off, pod_class)
    t0 = time.perf_counter(); res = ctx.simulate_node_removals(*args); t_cold = time.perf_counter() - t0
    t0 = time.perf_counter(); res = ctx.simulate_node_removals(*args); t_call = time.perf_counter() - t0
    _, ms = ctx.simulate_node_removals(*args, time_iters=10)
    rec = {"workload": w.name, "nodes": n_nodes, "candidates": len(w.candidates), "pods_listed": int(off[-1]), "classes": int(enc.pegs.n_pegs),
           "removable": int((res.removable == 1).sum()), "no_place": int((res.removable == 0).sum()), "pods_listed_again": int(len(res.ext_pod)),
           "candidates_processed": res.n_processed, "gpu_kernels_ms": ms, "gpu_call_ms": t_call * 1e3, "gpu_call_ms_cold": t_cold * 1e3,
           "encode_ms": t_enc * 1e3, "candidates_per_s_kernels": res.n_processed / (ms * 1e-3)}
    if n_nodes <= ORACLE_NODE_LIMIT:
        t0 = time.perf_counter(); want = removal_oracle(case); t_orc = time.perf_counter() - t0
        assert_removal_matches(res, want, w.name)
        rec.update(oracle_ms=t_orc * 1e3, parity="bit-exact")
    print(json.dumps(rec), flush=True)
    enc.close()
ctx.close()
