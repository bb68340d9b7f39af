#!/usr/bin/env python3
"""R3 = the reference's own benchmark of the scale-down loop, BenchmarkRunOnceScaleDown (CA/core/bench/benchmark_runonce_test.go:505-521: 400 nodes
at 40 %, every node a candidate, verifyToBeDeleted(240)), on the device: the kernel the library picks (the one-wave kernel with an LDS log sized to
what fits), the one-wave kernel refused an optimistic log (CASIM_NO_OPTIMISTIC_LOG=1: K_sched, as before), K_sched forced — HIP-event time of the
resident pass, every result against the oracle's native call.  Usage on the GPU box: python tests/tools/time_runonce_scale_down.py [nodes ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import workloads  # noqa: E402
from harness import RemovalCase, removal_encode  # noqa: E402
from oracle_driver import OracleScenario  # noqa: E402

ctx = kaa.Context(0)
rows = []
for n in [int(x) for x in sys.argv[1:]] or [400, 1000]:
    w = workloads.runonce_scale_down(n)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    enc, pc, off = removal_encode(case)
    s = OracleScenario()
    for info in w.nodes:
        s.add_existing(info)
    lists = case.pod_lists()
    for lst in lists:
        for p in lst:
            s.pod(p)
    ext_cap = 2 * len(pc) + 64      # (the Python mirror's default)
    want = s.simulate_node_removals(w.candidates, lists, None, None, True, 0, None, ext_cap, 0, None)
    row = {"nodes": n, "candidates": n, "pods": int(len(pc)), "pods_listed_again": len(want["ext"]), "removable_oracle": int(sum(1 for r in want["removable"] if r == 1)),
           "reference_answer": int(0.6 * n), "oracle_ms": s.last_native_s * 1e3, "ext_capacity": ext_cap}
    s.close()
    for name, env in (("default", {}), ("log_in_hbm", {"CASIM_LEAN_HBM_LOG": "1"}), ("no_optimistic_log", {"CASIM_NO_OPTIMISTIC_LOG": "1", "CASIM_LEAN_HBM_LOG": "0"}),
                      ("k_sched", {"CASIM_NO_LEAN_REMOVALS": "1"})):
        for k in ("CASIM_NO_OPTIMISTIC_LOG", "CASIM_NO_LEAN_REMOVALS", "CASIM_LEAN_HBM_LOG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        r = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc, ext_capacity=ext_cap)
        info = ctx.last_removals_info()
        _, ms = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc, time_iters=5, ext_capacity=ext_cap)
        ext = list(zip(r.ext_candidate.tolist(), r.ext_pod.tolist(), r.ext_node.tolist()))
        exact = bool(np.array_equal(np.asarray(r.removable), want["removable"]) and np.array_equal(np.asarray(r.node_out), want["node_out"]) and
                     int(r.last_index) == want["last_index"] and int(r.n_processed) == want["n_processed"] and ext == [tuple(x) for x in want["ext"]])
        row[name] = {"kernels_ms": ms, "ran_lean": bool(info["lean"]), "removable": int((r.removable == 1).sum()), "bit_exact": exact,
                     "us_per_candidate": ms * 1e3 / n, "speedup_vs_oracle": row["oracle_ms"] / ms}
    for k in ("CASIM_NO_OPTIMISTIC_LOG", "CASIM_NO_LEAN_REMOVALS", "CASIM_LEAN_HBM_LOG"):
        os.environ.pop(k, None)
    rows.append(row)
    enc.close()
ctx.close()
print(json.dumps(rows))
