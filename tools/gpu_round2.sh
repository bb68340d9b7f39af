#!/bin/bash
# One GPU-box session of round 2: parity tests, smoke, the contract bench (C2 batch), the 2-rank RCCL run on one GPU,
# issue-rate microbench, rocprofv3 kernel trace + PMC passes.
# Usage (repo root on the GPU box):  bash tools/gpu_round2.sh <tag> [quick|full|prof]
set -u
TAG=${1:-r02}
MODE=${2:-full}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
S="$OUT/summary.txt"
echo "== device ==" | tee "$S"
(rocminfo | grep -E "Marketing Name|gfx9" | head -4; nproc; free -g | head -2) 2>&1 | tee -a "$S"

if [ "$MODE" != "prof" ]; then
  echo "== pytest -m gpu ==" | tee -a "$S"
  timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" | tee -a "$S"
  tail -5 "$OUT/pytest_gpu.log" | tee -a "$S"

  echo "== smoke ==" | tee -a "$S"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
  echo "smoke exit $?" | tee -a "$S"
  tail -3 "$OUT/smoke.log" | tee -a "$S"

  echo "== bench (driver form, defaults) ==" | tee -a "$S"
  T0=$(date +%s.%N)
  timeout 1200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $? in $(echo "$(date +%s.%N) - $T0" | bc) s" | tee -a "$S"
  tail -c 9000 "$OUT/bench.json" | tee -a "$S"
  tail -5 "$OUT/bench.err" | tee -a "$S"

  # RCCL refuses two ranks on one device ("Duplicate GPU detected"): on a 1-GPU box the N > 1 control flow (self-launch,
  # sharded tables, per-step all-reduce of the packed keys, C3 sharded) runs over gloo with both ranks on cuda:0, and
  # the RCCL calls themselves run in a 1-rank group (CASIM_BENCH_FORCE_DIST=1: init nccl, all-reduce every step).
  echo "== bench --gpus 2, both ranks on cuda:0, gloo ==" | tee -a "$S"
  CASIM_BENCH_ONE_GPU=1 CASIM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 100 --warmup 5 --batch 1024 --no-cpu-baseline --no-configs --no-dense --no-next-rows \
      > "$OUT/bench_2ranks_one_gpu_gloo.json" 2> "$OUT/bench_2ranks_one_gpu_gloo.err"
  echo "2-rank bench exit $?" | tee -a "$S"
  tail -c 2500 "$OUT/bench_2ranks_one_gpu_gloo.json" | tee -a "$S"
  tail -3 "$OUT/bench_2ranks_one_gpu_gloo.err" | tee -a "$S"
  echo "== bench, 1 rank, RCCL group forced ==" | tee -a "$S"
  CASIM_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 1 --steps 200 --warmup 5 --batch 1024 --no-cpu-baseline --no-configs --no-dense --no-next-rows \
      > "$OUT/bench_1rank_rccl.json" 2> "$OUT/bench_1rank_rccl.err"
  echo "1-rank RCCL bench exit $?" | tee -a "$S"
  tail -c 2500 "$OUT/bench_1rank_rccl.json" | tee -a "$S"
  tail -3 "$OUT/bench_1rank_rccl.err" | tee -a "$S"

  echo "== VALU issue-rate microbench ==" | tee -a "$S"
  (make -s -C tools/ubench 2>/dev/null; timeout 120 tools/ubench/valu_rates) > "$OUT/valu_rates.txt" 2>&1
  cat "$OUT/valu_rates.txt" | tee -a "$S"
fi

if [ "$MODE" != "quick" ]; then
  BARGS="--steps 300 --warmup 5 --no-cpu-baseline --no-configs --no-next-rows --no-c3"   # (the timed region dominates the trace: its kernels overlap across streams)
  echo "== rocprofv3 kernel trace ==" | tee -a "$S"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --truncate-kernels -d "$OLDPWD/$OUT/prof_trace" -o trace -- \
      python "$OLDPWD/bench.py" $BARGS > "$OLDPWD/$OUT/prof_trace.log" 2>&1)
  echo "trace exit $?" | tee -a "$S"
  find "$OUT/prof_trace" -name "*kernel_stats*" | head -3 | while read f; do echo "--- $f"; head -15 "$f"; done | tee -a "$S"
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    echo "== rocprofv3 pmc $C ==" | tee -a "$S"
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/prof_pmc_$N" -o pmc -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-next-rows --no-c3 > "$OLDPWD/$OUT/prof_pmc_$N.log" 2>&1)
    echo "pmc $N exit $?" | tee -a "$S"
  done
  python tools/rocpd_summary.py "$OUT"/prof_trace "$OUT"/prof_pmc_* > "$OUT/rocpd_summary.txt" 2>&1
  python tools/pmc_traffic.py "$OUT" "$OUT/pack_traffic.json" 2>&1 | tee -a "$S"
  find "$OUT" -name "*.csv" -size +8M -delete
  find "$OUT" -name "*.db" -size +24M -delete
fi
echo "== done ==" | tee -a "$S"
