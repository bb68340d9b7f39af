"""A handful of enter -> return calls of the headline batch for a rocprofv3 kernel + memory-copy trace (the GPU-side timeline of one call):
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python tests/tools/enter_return_trace.py [pinned] [req32] [shared] [K]
Prints the host wall time of every call; the last call starts after a 5 ms pause (a gap in the trace to find it by)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import bench
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import BatchCall
from kubernetes_autoscaler_amd.tables import TableSet
args = sys.argv[1:]
seeds = bench.simulation_tables(workloads.CONFIGS["C2"], range(64), kaa.Encoder, TableSet)
full = (seeds.tile_groups(64) if "shared" in args else seeds.tile(64)).head(4096)
ts = full.pinned() if "pinned" in args else full
K = next((int(a) for a in args if a.isdigit()), 4)
ctx = kaa.Context(0)
call = BatchCall(ctx, *ts.structs(narrow_requests="req32" in args), kinds=[_abi.EXPANDER_LEAST_NODES], n_streams=K, winners_only=True)
for i in range(8):
    if i == 7:
        time.sleep(0.005)
    t0 = time.perf_counter(); call.call_raw(); print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
