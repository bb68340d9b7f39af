"""enter -> return of the headline batch over (tables pageable / page-locked) x (int64 / req32 requests) x (parts K) — one line per cell; run it
under CASIM_UPLOAD_GATE=0 / 1 to see what taking the link in turn does in each cell (the gate is read once per process).
Usage on the GPU box: [CASIM_UPLOAD_GATE=1] python tests/tools/enter_return_matrix.py [reps]"""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
gc.collect(); gc.disable()
gate = os.environ.get("CASIM_UPLOAD_GATE", "0")
for name, ts in (("pageable", full), ("pinned", pin)):
    for narrow in (False, True):
        for K in (4, 6, 8):
            call = BatchCall(ctx, *ts.structs(narrow_requests=narrow), kinds=[_abi.EXPANDER_LEAST_NODES], n_streams=K, winners_only=True)
            for _ in range(4):
                call.call_raw()
            seq = []
            for _ in range(reps):
                t0 = time.perf_counter(); call.call_raw(); seq.append((time.perf_counter() - t0) * 1e3)
            seq.sort()
            print(f"gate {gate} tables {name:8s} requests {'req32' if narrow else 'int64'} K {K}: median {seq[len(seq) // 2]:.3f} ms  min {seq[0]:.3f}  max {seq[-1]:.3f}", flush=True)
