#!/usr/bin/env python3
"""enter -> return of ONE casim_estimate_batch_query call (tables H2D, kernels, expander, results D2H) for single simulations, on
the MI355X: median / best wall time over many calls, next to the oracle's time for the same simulation.
    python tests/tools/time_call.py [C0 C2 R2 ...]      (CASIM_LIB_PATH selects the library: same-box A/B; CASIM_NO_FRONT=1 = separate launches)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import _abi, workloads  # noqa: E402
from kubernetes_autoscaler_amd.engine import BatchCall  # noqa: E402
from harness import GroupSpec, Scenario, encode, run_oracle  # noqa: E402

names = sys.argv[1:] or ["C0", "C1", "C2", "C4", "R2"]
ctx = kaa.Context(0)
for name in names:
    w = workloads.CONFIGS[name]()
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing,
                  lanes=w.lanes, device_csr=True)
    enc = encode(sc)
    for kinds in (None, [_abi.EXPANDER_LEAST_NODES]):
        bc = BatchCall(ctx, enc.pegs, enc.groups, kinds=kinds)
        for _ in range(20):
            bc.call_raw()
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            bc.call_raw()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        t0 = time.perf_counter(); run_oracle(sc); t_or = time.perf_counter() - t0
        print(json.dumps({"lib": os.environ.get("CASIM_LIB_PATH", "libcasim.so"), "no_front": bool(os.environ.get("CASIM_NO_FRONT")), "config": name,
                          "expander": kinds is not None, "groups": len(w.groups), "pegs": len(w.pegs), "call_us_median": round(ts[len(ts) // 2] * 1e6, 1),
                          "call_us_best": round(ts[0] * 1e6, 1), "call_us_p90": round(ts[int(len(ts) * 0.9)] * 1e6, 1),
                          "oracle_us_incl_binding": round(t_or * 1e6, 1)}))
    enc.close()
ctx.close()
