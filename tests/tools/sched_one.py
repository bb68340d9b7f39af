#!/usr/bin/env python3
"""One f1 and one f4 call at the 15 000-node scale (for counter passes: rocprofv3 --pmc ... -- python tests/tools/sched_one.py [pending|removal]).
`bench` = the two workloads of bench.py's try_schedule_pods / node_removals rows (5000 nodes): tools/sched_counters.sh turns the counters of
that run into profiles/sched_counters.json, which bench.py reads for the rows' issue rooflines."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import workloads  # noqa: E402
from kubernetes_autoscaler_amd.scheduling import encode_pending_pods  # noqa: E402
from harness import RemovalCase, removal_encode  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = kaa.Context(0)
if which == "bench":
    w = workloads.pending_scale(5000, 50000, 64, 2)
    enc, pc = encode_pending_pods(w.nodes, w.pods)
    for _ in range(reps):
        rc, node_out, li, ns = ctx.try_schedule_pods(enc.pegs, enc.groups, pc)
    print("bench pending scheduled", int(ns), "last_index", int(li))
    w = workloads.removal_scale(5000, pods_per_node=12, frac_candidates=0.3, seed=1)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    enc, pc, off = removal_encode(case)
    for _ in range(reps):
        r = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc)
    print("bench removal candidates", len(case.candidates))
if which in ("pending", "both"):
    w = workloads.pending_scale(15000, 150000, 128, 3)
    enc, pc = encode_pending_pods(w.nodes, w.pods)
    for _ in range(reps):
        rc, node_out, li, ns = ctx.try_schedule_pods(enc.pegs, enc.groups, pc)
    print("pending scheduled", int(ns), "last_index", int(li))
if which in ("removal", "both"):
    w = workloads.removal_scale(15000, pods_per_node=12, frac_candidates=0.2, seed=1)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    enc, pc, off = removal_encode(case)
    for _ in range(reps):
        r = ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc)
    print("removal candidates", len(case.candidates))
ctx.close()
