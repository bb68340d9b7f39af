#!/usr/bin/env python3
"""Kernel times of ONE simulation (a lone wave per node group: latency, not throughput) for the given workloads, on the MI355X.
    python tests/tools/time_latency.py [C1 C2 R1 ...]      (CASIM_LIB_PATH selects the library: same-box A/B of two builds)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import workloads  # noqa: E402
from harness import GroupSpec, Scenario, encode  # noqa: E402

names = sys.argv[1:] or ["C1", "C2", "C3", "R1", "R2"]
ctx = kaa.Context(0)
for name in names:
    w = workloads.CONFIGS[name]()
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing,
                  lanes=w.lanes, device_csr=name in ("C2", "C3", "C4"))
    enc = encode(sc)
    with kaa.Problem(ctx, enc.pegs, enc.groups) as p:
        p.run(); res = p.fetch()
        best = None
        for _ in range(3):
            tot, k = p.time(iters=20)
            if best is None or tot < best[0]:
                best = (tot, k)
        steps = max(int(x) for x in [max(len(ids) if ids is not None else len(w.pegs) for ids in [g.pegs for g in w.groups])])
        print(json.dumps({"lib": os.environ.get("CASIM_LIB_PATH", "libcasim.so"), "config": name, "pipeline_ms": round(best[0], 4),
                          **{a: round(b, 4) for a, b in best[1].items()}, "max_pegs_per_group": steps,
                          "pack_us_per_peg_step": round(best[1].get("pack_ms", 0) * 1e3 / max(steps, 1), 4), "nodes": int(res.node_count.sum())}))
ctx.close()
