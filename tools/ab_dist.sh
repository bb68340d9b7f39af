#!/bin/bash
# 1-rank RCCL group (torch), full batch: the N > 1 control flow of bench.py on a 1-GPU box
OUT=gpurun_out/$1; mkdir -p $OUT
for B in bench_old_tmp.py bench.py; do for r in 1 2; do
  [ -f $B ] || continue
  CASIM_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
      $B --gpus 1 --steps 300 --warmup 10 --no-cpu-baseline --no-configs --no-next-rows --no-c3 2>$OUT/err.txt | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$B step_ms %.4f sims/s %.4g allreduce_ms %s forks %s' % (d['ms_per_step'], d.get('sims_per_s'), d['multi_gpu']['all_reduce_ms'], d['config'].get('forks_from_the_context_stream')))" 2>&1 | tee -a $OUT/dist.txt
done; done
