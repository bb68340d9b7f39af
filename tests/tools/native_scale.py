"""Native (plain C++, no Python between the calls) end-to-end figures for the callers either side of the path at cluster scale:
encode (casim_enc_* calls + finalize), the whole call enter -> return, kernels — tools/casim_native on recorded traces."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import native_trace as nt  # noqa: E402
from kubernetes_autoscaler_amd import workloads  # noqa: E402

out = []
for n_nodes, n_pods, n_cls, seed in ((1000, 12000, 32, 1), (5000, 50000, 64, 2), (15000, 150000, 128, 3)):
    w = workloads.pending_scale(n_nodes, n_pods, n_cls, seed)
    path = f"/tmp/pending_{n_nodes}.trace"
    t0 = time.perf_counter(); enc, _ = nt.trace_pending(w, path, iters=5); t_py = time.perf_counter() - t0
    enc.close()
    rc, r = nt.run_native(path, repeat=3)
    out.append({"entry": "casim_try_schedule_pods", "nodes": n_nodes, "pending_pods": n_pods, "classes": n_cls, "python_mirror_encode_ms": t_py * 1e3,
                **{k: r.get(k) for k in ("enc_calls", "encode_calls_ms", "finalize_ms", "encode_ms", "wall_ms", "kernels_ms", "scheduled", "status", "engine_error")}})
    print(json.dumps(out[-1]), flush=True)
for n_nodes in (1000, 5000, 15000):
    w = workloads.removal_scale(n_nodes, pods_per_node=12, frac_candidates=0.3 if n_nodes < 15000 else 0.2, seed=1)
    path = f"/tmp/removal_{n_nodes}.trace"
    t0 = time.perf_counter(); enc, _, _ = nt.trace_removals(w, path, iters=5); t_py = time.perf_counter() - t0
    enc.close()
    rc, r = nt.run_native(path, repeat=3)
    out.append({"entry": "casim_simulate_node_removals", "nodes": n_nodes, "candidates": len(w.candidates), "python_mirror_encode_ms": t_py * 1e3,
                **{k: r.get(k) for k in ("enc_calls", "encode_calls_ms", "finalize_ms", "encode_ms", "wall_ms", "kernels_ms", "removable", "status", "engine_error")}})
    print(json.dumps(out[-1]), flush=True)
