#!/bin/bash
# same-box A/B of the single-call latency: tools/ab_call.sh <tag> [base_lib.so]  -> in-tree (fused front kernel), in-tree with CASIM_NO_FRONT=1, base library
TAG=$1; BASE=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  echo "== in-tree (round $round)" | tee -a $OUT/ab_call.txt
  timeout 300 python tests/tools/time_call.py ${CONFIGS:-} 2>&1 | tail -12 | tee -a $OUT/ab_call.txt
  echo "== in-tree, CASIM_NO_FRONT=1 (round $round)" | tee -a $OUT/ab_call.txt
  CASIM_NO_FRONT=1 timeout 300 python tests/tools/time_call.py ${CONFIGS:-} 2>&1 | tail -12 | tee -a $OUT/ab_call.txt
  if [ -n "$BASE" ]; then
    echo "== $BASE (round $round)" | tee -a $OUT/ab_call.txt
    CASIM_LIB_PATH=$PWD/$BASE timeout 300 python tests/tools/time_call.py ${CONFIGS:-} 2>&1 | tail -12 | tee -a $OUT/ab_call.txt
  fi
done
