#!/bin/bash
# same-box A/B: tools/ab_latency.sh <tag> <libA.so> <libB.so> ... — single-simulation kernel times + a short headline bench for every library
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  for LIB in "$@"; do
    echo "== $LIB (round $round)" | tee -a $OUT/ab.txt
    CASIM_LIB_PATH=$PWD/$LIB timeout 300 python tests/tools/time_latency.py ${CONFIGS:-C1 C2 C3 R1} 2>&1 | tail -8 | tee -a $OUT/ab.txt
    CASIM_LIB_PATH=$PWD/$LIB timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-configs --no-next-rows --no-c3 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_ms %.4f  sims/s %.4g  pack_ms %.4f alone %.4f' % (d['ms_per_step'], d.get('sims_per_s'), d['roofline']['kernel_ms'], d['roofline']['device_to_itself']['kernel_ms']))" 2>&1 | tee -a $OUT/ab.txt
  done
done
