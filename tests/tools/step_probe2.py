#!/usr/bin/env python3
"""step_probe variants: python tests/tools/step_probe2.py <K> <own|torch> [steps]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import _abi, workloads  # noqa: E402
from kubernetes_autoscaler_amd.tables import TableSet  # noqa: E402
K = int(sys.argv[1]); mode = sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
B, S = 4096, 64
kinds = [_abi.EXPANDER_LEAST_NODES]
seed_set = bench.simulation_tables(workloads.CONFIGS["C2"], range(S), kaa.Encoder, TableSet)
full = seed_set.tile((B + S - 1) // S).head(B)
def loop(step, n):
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return {"enqueue_ms_per_step": round((t1 - t0) / n * 1e3, 4), "step_ms": round((t2 - t0) / n * 1e3, 4)}
if mode == "torch":
    side = torch.cuda.Stream(device=0); torch.cuda.set_stream(side); h = side.cuda_stream
else:
    h = None
with kaa.StreamedBatch(0, full, n_streams=K, stream=h) as b:
    def step():
        b.run(); b.best_option_sims(kinds, fetch=False)
    r = loop(step, steps)
    print(json.dumps({"form": "in-library", "ctx_stream": mode, "K": K, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "probe": os.environ.get("CASIM_LANE_PROBE"), "info": b.prob.info(), **r}), flush=True)
