"""enter -> return of the headline batch (casim_estimate_batch_query, 4096 C2 simulations), every form side by side:
   every list / winners only  x  host gcd pass / device gcd pass  x  1 / 4 internal streams.
CASIM_INIT_TIMING=1 prints the stages of ProblemT::init of every part (stderr)."""
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
ctx = kaa.Context(0)
pegs, groups = full.structs()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rnd in range(2):
    for dev_gcd in (False, True):
        os.environ["CASIM_DEV_GCD_MIN"] = "1" if dev_gcd else str(1 << 40)
        for winners in (False, True):
            for K in (1, 4):
                call = BatchCall(ctx, pegs, groups, kinds=[_abi.EXPANDER_LEAST_NODES], n_streams=K, winners_only=winners)
                for _ in range(3):
                    call.call_raw()
                t0 = time.perf_counter()
                for _ in range(reps):
                    call.call_raw()
                print(f"round {rnd} gcd {'device' if dev_gcd else 'host  '} lists {'winners' if winners else 'every  '} streams {K}: "
                      f"{(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call", flush=True)
