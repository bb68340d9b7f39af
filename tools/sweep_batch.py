#!/usr/bin/env python3
"""Per-kernel HIP-event times of the hot path for several batch sizes (C1-shaped simulations)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import workloads  # noqa: E402

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,1024,2048,4096,8192,16384").split(",")]
pegs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = kaa.Context(0)
for B in sizes:
    enc, checks, _ = bench.build_batch(workloads, kaa.Encoder, B, 0, pegs, 50, 256)
    with kaa.Problem(ctx, enc.pegs, enc.groups) as p:
        p.run(); p.fetch()
        tot, k = p.time(iters=10)
    print(json.dumps({"B": B, "pegs": pegs, "total_ms": tot, **k, "sims_per_s": B / (tot * 1e-3), "us_per_peg_step_per_wave": k["pack_ms"] * 1e3 / pegs}))
ctx.close()
