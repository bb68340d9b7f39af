import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import bench
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import BatchCall
from kubernetes_autoscaler_amd.tables import TableSet
seed_set = bench.simulation_tables(workloads.CONFIGS["C2"], range(64), kaa.Encoder, TableSet)
full = seed_set.tile(64).head(4096)
ctx = kaa.Context(0)
pegs, groups = full.structs()
for K in (1, 4):
    call = BatchCall(ctx, pegs, groups, kinds=[_abi.EXPANDER_LEAST_NODES], n_streams=K)
    for _ in range(3): call.call_raw()
    t0 = time.perf_counter()
    for _ in range(5): call.call_raw()
    print("K", K, "ms per call", (time.perf_counter() - t0) / 5 * 1e3, flush=True)
os.environ["X"] = "1"
