#!/usr/bin/env python3
"""Where a step of the resident headline loop goes: host enqueue time vs device time, for the in-library streams (one casim_ctx,
casim_options.n_streams = K) and for K separate contexts driven from Python (round 2's form).
    python tests/tools/step_probe.py [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import _abi, workloads  # noqa: E402
from kubernetes_autoscaler_amd.tables import TableSet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, S = 4096, 64
kinds = [_abi.EXPANDER_LEAST_NODES]
seed_set = bench.simulation_tables(workloads.CONFIGS["C2"], range(S), kaa.Encoder, TableSet)
full = seed_set.tile((B + S - 1) // S).head(B)


def loop(step, n):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return {"enqueue_ms_per_step": round((t1 - t0) / n * 1e3, 4), "step_ms": round((t2 - t0) / n * 1e3, 4)}


for K in (1, 2, 4, 8):
    side = torch.cuda.Stream(device=0)
    with torch.cuda.stream(side), kaa.StreamedBatch(0, full, n_streams=K, stream=side.cuda_stream) as b:
        def step():
            b.run(); b.best_option_sims(kinds, fetch=False)
        r = loop(step, steps)
        def step_run_only():
            b.run()
        r2 = loop(step_run_only, steps)
        print(json.dumps({"form": "in-library streams", "K": K, **r, "run_only": r2}), flush=True)

for K in (4,):
    ctxs = [kaa.Context(0) for _ in range(K)]
    n = full.n_sims
    parts = [full.sim_slice((n * i) // K, (n * (i + 1)) // K) for i in range(K)]
    structs = [p.structs() for p in parts]
    probs = [kaa.Problem(c, *s) for c, s in zip(ctxs, structs)]
    def step4():
        for p in probs:
            p.run()
        for p, pt in zip(probs, parts):
            p.best_option_sims(kinds, per_sim=True, fetch=False, n_sims=pt.n_sims)
    r = loop(step4, steps)
    print(json.dumps({"form": "K contexts from Python", "K": K, **r}), flush=True)
    for p in probs:
        p.close()
    for c in ctxs:
        c.close()
