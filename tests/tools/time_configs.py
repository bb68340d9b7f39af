#!/usr/bin/env python3
"""Latency of one full scale-up simulation for each BASELINE.json config on the MI355X (resident
tables; feasibility + CSR + order + pack + expander), next to the CPU oracle on the same inputs."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import workloads  # noqa: E402
from harness import GroupSpec, Scenario, assert_matches_oracle, encode, run_oracle  # noqa: E402

ctx = kaa.Context(0)
for name in ("C0", "C1", "C2", "C3", "C4"):
    w = workloads.CONFIGS[name]()
    dcsr = name in ("C2", "C3", "C4")
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing,
                  lanes=w.lanes, device_csr=dcsr)
    t0 = time.perf_counter(); enc = encode(sc); t_enc = time.perf_counter() - t0
    with kaa.Problem(ctx, enc.pegs, enc.groups) as p:
        p.run(); res = p.fetch()
        tot, k = p.time(iters=20)
        info = p.info()
        t0 = time.perf_counter()
        for _ in range(20):
            p.run(); p.best_option([1])
        wall = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter(); orc = run_oracle(sc); t_orc = time.perf_counter() - t0
    assert_matches_oracle(res, orc, name)
    checks = sum(sum(len(w.pegs[i].pods) for i in ids) * max(g.max_nodes, 0) for (_, ids), g in zip(orc, w.groups))
    print(json.dumps({"config": name, "pods": w.n_pods, "groups": len(w.groups), "pegs": len(w.pegs), "checks": checks,
                      "gpu_pipeline_ms": tot, **k, "gpu_run_plus_expander_wall_ms": wall * 1e3, "oracle_ms": t_orc * 1e3,
                      "encode_ms": t_enc * 1e3, "packer": info, "filter_runs_oracle": sum(o.filter_runs for o, _ in orc)}))
ctx.close()
