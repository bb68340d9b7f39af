#!/bin/bash
# Phase profile of K_sched (s_memtime ticks of thread 0) on the f1 / f4 workloads: needs the profiling build
#   make -C kubernetes_autoscaler_amd/csrc OUT=../libcasim_prof.so EXTRA=-DCASIM_PACK_PROF
export CASIM_LIB_PATH=$PWD/kubernetes_autoscaler_amd/libcasim_prof.so CASIM_PACK_PROF_DUMP=1
python - <<'PY' 2>&1 | grep -E "prof|workload" 
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.scheduling import encode_pending_pods
from harness import RemovalCase, removal_encode
ctx = kaa.Context(0)
for w in (workloads.pending_scale(1000, 12000, 32, 1), workloads.pending_scale(15000, 150000, 128, 3)):
    enc, pc = encode_pending_pods(w.nodes, w.pods)
    ctx.try_schedule_pods(enc.pegs, enc.groups, pc)
    print("workload", w.name, file=sys.stderr); sys.stderr.flush()
    ctx.try_schedule_pods(enc.pegs, enc.groups, pc)
for n in (1000, 15000):
    w = workloads.removal_scale(n, pods_per_node=12, frac_candidates=0.3 if n < 10000 else 0.2, seed=1)
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates)
    enc, pc, off = removal_encode(case)
    ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc)
    print("workload", w.name, file=sys.stderr); sys.stderr.flush()
    ctx.simulate_node_removals(enc.pegs, enc.groups, case.candidates, off, pc)
PY
