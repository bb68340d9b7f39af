#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
for P in 1 0; do for K in 4 3 8; do
  CASIM_LANE_PROBE=$P timeout 120 python tests/tools/step_probe2.py $K torch 2>/dev/null | tee -a $OUT/probe2.txt
done; done
for P in 1 0; do for K in 4 8; do
  GPU_MAX_HW_QUEUES=4 CASIM_LANE_PROBE=$P timeout 120 python tests/tools/step_probe2.py $K torch 2>/dev/null | tee -a $OUT/probe2.txt
done; done
