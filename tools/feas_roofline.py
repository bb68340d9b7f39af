#!/usr/bin/env python3
"""The batched feasibility launch alone (bench.py's roofline_feasibility row) — the command the rocprofv3 passes of tools/gpu_round5.sh wrap:
   python tools/feas_roofline.py [--config C2] [--sims 16384] [--seeds 64] [--iters 50] [--probes]
Builds `sims` simulations of the config (seeds distinct ones, tiled) as ONE resident problem, times its feasibility launch
(casim_problem_time_feasibility) and prints the row as JSON.  --probes also runs the known-byte-count read streams (casim_stream_probe:
4 B / lane, 16 B / lane, 32-byte scalar loads of 1 GiB) so that a --pmc FETCH_SIZE pass carries its own calibration."""
import argparse
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--sims", type=int, default=16384)
    ap.add_argument("--seeds", type=int, default=64)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--probes", action="store_true")
    ap.add_argument("--verify", action="store_true")
    a = ap.parse_args()
    import kubernetes_autoscaler_amd as kaa
    from kubernetes_autoscaler_amd import workloads
    from kubernetes_autoscaler_amd.tables import TableSet
    import bench
    ctx = kaa.Context(0)
    row = bench.feasibility_roofline(kaa, ctx, workloads, TableSet, a.config, a.sims, min(a.seeds, a.sims), iters=a.iters, verify=a.verify)
    if a.probes:
        row["read_stream_gbps"] = {"4B_per_lane": ctx.stream_probe_gbps(1 << 30, 4, 3), "16B_per_lane": ctx.stream_probe_gbps(1 << 30, 16, 3),
                                   "32B_scalar_load_per_wave": ctx.stream_probe_gbps(1 << 30, 0, 3)}
    ctx.close()
    print(json.dumps(row))


if __name__ == "__main__":
    main()
