#!/usr/bin/env python3
"""SURVEY §8 row f3 measurement: BinpackingNodeEstimator.Estimate on the whole snapshot (K_est) for a node group whose
PEGs carry spread constraints, next to the CPU oracle: wall time of one casim_estimate_on_cluster call (upload + kernels
+ fetch; the path is a fallback for rule-carrying groups, a resident variant does not exist)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import workloads  # noqa: E402
from kubernetes_autoscaler_amd.estimator import encode_cluster_estimate  # noqa: E402
from kubernetes_autoscaler_amd.objects import LABEL_ZONE, NodeInfo, Pod, TopologySpreadConstraint  # noqa: E402
from harness import GroupSpec, Scenario, assert_cluster_estimate_matches, run_oracle  # noqa: E402

ctx = kaa.Context(0)
for n_existing, n_pegs, pods_per_peg, cap in ((100, 20, 25, 64), (1000, 40, 50, 256), (5000, 40, 50, 256)):
    w = workloads.config_c1(n_pegs=n_pegs, pods_per_peg=pods_per_peg, cap=cap)
    tmpl = w.groups[0].template
    tmpl.node.labels[LABEL_ZONE] = "zone-new"
    for i, pg in enumerate(w.pegs):   # every fourth PEG spreads over hostnames, every fourth over zones
        pod = pg.pods[0]
        if i % 4 == 0:
            pod.spread_constraints = [TopologySpreadConstraint(3, "kubernetes.io/hostname", 0, dict(pod.labels))]
        elif i % 4 == 1:
            pod.spread_constraints = [TopologySpreadConstraint(40, LABEL_ZONE, 0, dict(pod.labels))]
        pod.topology_spread = bool(pod.spread_constraints)
    existing = []
    for i in range(n_existing):
        info = NodeInfo(workloads._node(f"old-{i}", 8000, 32 * workloads.GiB, 110, {LABEL_ZONE: f"zone-{i % 3}"}))
        info.pods.append(Pod(name=f"r{i}", labels={"app": "running"}, requests={"cpu": 7900, "memory": workloads.GiB}))   # nearly full
        existing.append(info)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(tmpl, cap, 0, None)], existing=existing)
    t0 = time.perf_counter(); enc = encode_cluster_estimate(sc.lanes, sc.pegs, existing, tmpl, cap); t_enc = time.perf_counter() - t0
    ctx.estimate_on_cluster(enc.pegs, enc.groups, len(existing), cap, 0, enc.rules, enc.port_block)
    t0 = time.perf_counter(); rc, out = ctx.estimate_on_cluster(enc.pegs, enc.groups, len(existing), cap, 0, enc.rules, enc.port_block); t_call = time.perf_counter() - t0
    t0 = time.perf_counter(); est, ids = run_oracle(sc)[0]; t_orc = time.perf_counter() - t0
    assert_cluster_estimate_matches((rc, out, ids), est, ids, "cluster estimate")
    print(json.dumps({"existing_nodes": n_existing, "pegs": n_pegs, "pods": n_pegs * pods_per_peg, "node_cap": cap, "nodes_added": out["nodes_added"],
                      "pods_scheduled": out["pods_scheduled"], "gpu_call_ms": t_call * 1e3, "oracle_ms": t_orc * 1e3, "encode_ms": t_enc * 1e3,
                      "parity": "bit-exact"}), flush=True)
    enc.close()
ctx.close()
