#!/usr/bin/env python3
"""per-call wall times of the first calls of casim_try_schedule_pods for a sequence of workload sizes (warm-up effects of pools / staging)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.scheduling import encode_pending_pods
ctx = kaa.Context(0)
for spec in ((200, 200, 60000), (1000, 1000, 12000), (1000, 1000, 12000), (100, 3000, 1000), (1000, 1000, 12000)):
    w = workloads.filter_out_schedulable_benchmark(*spec)
    enc, pod_class = encode_pending_pods(w.nodes, w.pods)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        ctx.try_schedule_pods(enc.pegs, enc.groups, pod_class, w.hints, w.acceptable, w.break_on_failure, w.last_index)
        ts.append(round((time.perf_counter() - t0) * 1e3, 3))
    print(os.environ.get("CASIM_LIB_PATH", "in-tree")[-16:], w.name, ts, flush=True)
