#!/usr/bin/env python3
"""What a link of the removal chain is made of (removals_lean_kernel): clusters whose candidates move 0, ~1.5 and ~6 pods each, 5000 nodes / 1500
candidates, HIP-event time of the resident pass.  Usage on the GPU box: python tests/tools/removal_parts.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import workloads
from harness import RemovalCase, removal_encode
ctx = kaa.Context(0)
rows = []
for ppn, frac in ((0, 0.3), (12, 0.3), (12, 0.8), (40, 0.3)):
    w = workloads.removal_scale(5000, pods_per_node=ppn, frac_candidates=frac, seed=1)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    enc, pc, off = removal_encode(case)
    r = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc)
    info = ctx.last_removals_info()
    _, ms = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc, time_iters=10)
    rows.append({"pods_per_node_max": ppn, "candidates": len(case.candidates), "pods": int(len(pc)), "removable": int((r.removable == 1).sum()), "ext": int(len(r.ext_pod)),
                 "lean": info["lean"], "kernels_ms": ms, "us_per_candidate": ms * 1e3 / len(case.candidates)})
    enc.close()
ctx.close()
print(json.dumps(rows))
