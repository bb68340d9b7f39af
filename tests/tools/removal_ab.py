#!/usr/bin/env python3
"""The removal loop of bench.py's node_removals row (5000 nodes, 1500 candidates) and of the 15 000-node case through both kernels:
removals_lean_kernel (default where its shape allows) and K_sched's transaction loop (CASIM_NO_LEAN_REMOVALS=1) — HIP-event time of the
resident pass, results compared with each other.  Usage on the GPU box: python tests/tools/removal_ab.py [nodes [iters]]"""
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

ctx = kaa.Context(0)
rows = []
sizes = ((5000, 0.3), (15000, 0.2), (1000, 0.3))
if len(sys.argv) > 1:   # one size only (counter passes): removal_ab.py 5000 [iters]
    sizes = tuple(x for x in sizes if x[0] == int(sys.argv[1]))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for n, frac in sizes:
    w = workloads.removal_scale(n, pods_per_node=12, frac_candidates=frac, seed=1)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    enc, pc, off = removal_encode(case)
    row = {"nodes": n, "candidates": len(case.candidates), "pods": int(len(pc))}
    res = {}
    for name, env in (("lean", None), ("k_sched", "1")):
        if env:
            os.environ["CASIM_NO_LEAN_REMOVALS"] = env
        else:
            os.environ.pop("CASIM_NO_LEAN_REMOVALS", None)
        r = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc)
        info = ctx.last_removals_info()
        _, ms = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc, time_iters=iters)
        res[name] = r
        row[name] = {"kernels_ms": ms, "ran_lean": info["lean"], "us_per_candidate": ms * 1e3 / len(case.candidates), "us_per_pod": ms * 1e3 / max(len(pc), 1)}
    os.environ.pop("CASIM_NO_LEAN_REMOVALS", None)
    row["same_results"] = bool(np.array_equal(res["lean"].removable, res["k_sched"].removable) and np.array_equal(res["lean"].node_out, res["k_sched"].node_out) and
                               res["lean"].last_index == res["k_sched"].last_index and res["lean"].n_processed == res["k_sched"].n_processed)
    rows.append(row)
    enc.close()
ctx.close()
print(json.dumps(rows))
