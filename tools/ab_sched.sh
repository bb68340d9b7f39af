#!/bin/bash
# A/B of two libcasim builds on the f1 / f4 / K_est timing tools.  Usage: bash tools/ab_sched.sh <tag> <libA> <libB>
TAG=$1; A=$2; B=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for V in A B; do
  L=$A; [ $V = B ] && L=$B
  for T in time_pending time_removals time_cluster_estimate; do
    CASIM_ORACLE_CHECK_LIMIT=0 CASIM_LIB_PATH=$PWD/$L timeout 600 python tests/tools/$T.py > $OUT/${T}_$V.jsonl 2> $OUT/${T}_$V.err
    echo "$T $V ($L) exit $?"
  done
done
python - <<PY
import json,glob
for t in ("time_pending","time_removals","time_cluster_estimate"):
    a=[json.loads(l) for l in open("$OUT/%s_A.jsonl"%t) if l.startswith("{")]
    b=[json.loads(l) for l in open("$OUT/%s_B.jsonl"%t) if l.startswith("{")]
    for x,y in zip(a,b):
        k=[k for k in x if k.endswith("kernels_ms") or k=="gpu_ms" or k=="call_ms"]
        print(t, x.get("workload") or x.get("cluster") or x.get("name"), {kk:(round(x[kk],3),round(y[kk],3)) for kk in k})
PY
