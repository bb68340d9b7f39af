#!/bin/bash
# tools/ab_fork.sh <tag>: the resident headline loop under the three fork policies of casim_streams.h (CASIM_FORK_MODE 0 never / 1 when busy / 2 always)
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  for M in 1 0 2; do
    echo "== CASIM_FORK_MODE=$M (round $round)" | tee -a $OUT/fork.txt
    CASIM_FORK_MODE=$M timeout 300 python bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-configs --no-next-rows --no-c3 ${BENCH_EXTRA:-} 2>$OUT/err_$M.txt | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_ms %.4f  sims/s %.4g  pack_ms %.4f alone %.4f  forks %s' % (d['ms_per_step'], d.get('sims_per_s'), d['roofline']['kernel_ms'], d['roofline']['device_to_itself']['kernel_ms'], d['config'].get('forks_from_the_context_stream')))" 2>&1 | tee -a $OUT/fork.txt
  done
done
