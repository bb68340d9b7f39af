"""enter -> return of the headline batch with the caller's tables in page-locked memory (casim_host_alloc) next to pageable ones;
CASIM_INIT_TIMING=1 prints the stages of ProblemT::init of every part (stderr)."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import bench
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.engine import BatchCall
from kubernetes_autoscaler_amd.tables import TableSet
full = bench.simulation_tables(workloads.CONFIGS["C2"], range(64), kaa.Encoder, TableSet).tile(64).head(4096)
pin = full.pinned()
ctx = kaa.Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
gc.collect(); gc.disable()
for rnd in range(2):
    for name, ts in (("pageable", full), ("pinned  ", pin)):
        for K in (1, 4):
            call = BatchCall(ctx, *ts.structs(), kinds=[_abi.EXPANDER_LEAST_NODES], n_streams=K, winners_only=True)
            for _ in range(3):
                call.call_raw()
            seq = []
            for _ in range(reps):
                t0 = time.perf_counter(); call.call_raw(); seq.append((time.perf_counter() - t0) * 1e3)
            print(f"round {rnd} tables {name} streams {K}: mean {sum(seq) / reps:.3f} ms  min {min(seq):.3f}  max {max(seq):.3f}", flush=True)
if os.environ.get("CASIM_INIT_TIMING"):
    print("--- one more pinned call, stages on stderr", flush=True)
    BatchCall(ctx, *pin.structs(), kinds=[_abi.EXPANDER_LEAST_NODES], n_streams=4, winners_only=True).call_raw()
