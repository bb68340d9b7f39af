#!/bin/bash
# Same-box A/B of two builds of libcasim on the contract bench (boxes of the pool differ by several percent):
#   tools/ab_lib.sh <other_lib.so> [rounds=3]    -> ms_per_step of the in-tree library and of the other one, alternating
OTHER=$1; N=${2:-3}
for i in $(seq $N); do
  for L in "" "$OTHER"; do
    CASIM_LIB_PATH=$L timeout 300 python bench.py --steps 600 --no-cpu-baseline --no-configs --no-next-rows --no-c3 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().split('\n')[0]); print('${L:-in-tree}', round(d['ms_per_step'],4), d['roofline']['device_to_itself']['kernel_ms_all'])"
  done
done
