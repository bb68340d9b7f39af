#!/bin/bash
# tools/ab_batch.sh <tag>: the resident headline loop at several batch sizes and stream counts
OUT=gpurun_out/$1; mkdir -p $OUT
for B in 2048 4096 8192 16384; do for K in 4 6; do
  timeout 300 python bench.py --steps 200 --warmup 10 --batch $B --streams $K --no-cpu-baseline --no-configs --no-next-rows --no-c3 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B K=$K parts=%s step_ms %.4f  sims/s %.4g  checks/s %.4g' % (d['config'].get('streams'), d['ms_per_step'], d.get('sims_per_s'), d['value']))" 2>&1 | tee -a $OUT/batch.txt
done; done
