"""enter -> return of the headline batch, three hand-overs side by side: int64 requests (the library's device gcd pass), casim_pegs.req32 + req_unit,
req32 + PEG rows shared between the tiles.  Mean of `reps` calls after warm-up, then ONE call of each with CASIM_INIT_TIMING=1 (stages on stderr).
Usage on the GPU box: python tests/tools/enter_return_variants.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import bench
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import BatchCall
from kubernetes_autoscaler_amd.tables import TableSet
seed_set = bench.simulation_tables(workloads.CONFIGS["C2"], range(64), kaa.Encoder, TableSet)
full = seed_set.tile(64).head(4096)
shared = seed_set.tile_groups(64).head(4096)
ctx = kaa.Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None   # one variant only (a process started with CASIM_INIT_TIMING=1 prints every init's stages)
variants = [("int64", full, False), ("req32", full, True), ("shared+req32", shared, True), ("shared-int64", shared, False)]
if only:
    variants = [v for v in variants if v[0] == only]
calls = {}
for name, ts, narrow in variants:
    pegs, groups = ts.structs(narrow_requests=narrow)
    calls[name] = (BatchCall(ctx, pegs, groups, kinds=[_abi.EXPANDER_LEAST_NODES], n_streams=4, winners_only=True), ts)
for rnd in range(2):
    for name, (call, _) in calls.items():
        for _ in range(5):
            call.call_raw()
        seq = []
        for _ in range(reps):
            t0 = time.perf_counter(); call.call_raw(); seq.append((time.perf_counter() - t0) * 1e3)
        seq.sort()
        print(f"round {rnd} {name:14s}: mean {sum(seq) / len(seq):.3f} ms  median {seq[len(seq) // 2]:.3f}  min {seq[0]:.3f}", flush=True)
for name, (call, _) in calls.items():
    print(f"--- stages of one {name} call", file=sys.stderr, flush=True)
    call.call_raw()
