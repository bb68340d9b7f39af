import gc, os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch  # noqa
import bench
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import BatchCall
from kubernetes_autoscaler_amd.tables import TableSet
full = bench.simulation_tables(workloads.CONFIGS["C2"], range(64), kaa.Encoder, TableSet).tile(64).head(4096)
ctx = kaa.Context(0)
gc.collect(); gc.disable()
call = BatchCall(ctx, *full.structs(), kinds=[_abi.EXPANDER_LEAST_NODES], n_streams=int(sys.argv[1]) if len(sys.argv) > 1 else 4, winners_only=True)
for _ in range(10): call.call_raw()
seq = []
for i in range(100):
    t0 = time.perf_counter(); call.call_raw(); seq.append((time.perf_counter() - t0) * 1e3)
s = sorted(seq)
print("n", len(seq), "median %.3f p90 %.3f p99 %.3f max %.3f mean %.3f" % (s[50], s[90], s[98], s[-1], sum(seq) / len(seq)))
print("outliers (>6 ms) at", [(i, round(v, 1)) for i, v in enumerate(seq) if v > 6])
