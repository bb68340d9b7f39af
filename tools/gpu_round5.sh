#!/bin/bash
# One GPU-box session of round 5.  Usage (repo root on the GPU box):  bash tools/gpu_round5.sh <tag> [tests|bench|feas|feasprof|sched|prof ...]   (modes joined by |)
#   tests     pytest -m gpu + smoke                         bench   the contract bench with the DRIVER's arguments (--steps 20 --warmup 5) and with the defaults
#   feas      tools/feas_roofline.py (C2 x 16384, C3 x 1024): the batched feasibility launch alone
#   feasprof  rocprofv3 kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) of the same command -> feas_traffic.json, rocpd summary
#   prof      tools/gpu_round3.sh prof (trace + counter passes of the bench command)        sched   K_sched counter passes
set -u
TAG=${1:-r10}
MODE=${2:-tests|bench}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
S="$OUT/summary.txt"
has() { [[ "|$MODE|" == *"|$1|"* ]]; }
echo "== device ==" | tee "$S"
(rocminfo | grep -E "Marketing Name|gfx9" | head -4; nproc) 2>&1 | tee -a "$S"
if has tests; then
  echo "== pytest -m gpu ==" | tee -a "$S"
  timeout 1800 python -m pytest tests -q -m gpu --durations=12 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" | tee -a "$S"
  tail -25 "$OUT/pytest_gpu.log" | tee -a "$S"
  grep -E "^(FAILED|ERROR)|roofline_feasibility|self-check" "$OUT/pytest_gpu.log" | head -40 | tee -a "$S"
  echo "== smoke ==" | tee -a "$S"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
  echo "smoke exit $?" | tee -a "$S"
  tail -3 "$OUT/smoke.log" | tee -a "$S"
fi
if has bench; then
  echo "== bench, the driver's command line ==" | tee -a "$S"
  T0=$SECONDS
  timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.out" 2> "$OUT/bench_driver.err"
  echo "bench exit $? in $((SECONDS - T0)) s; stdout lines: $(wc -l < "$OUT/bench_driver.out"), last line bytes: $(tail -1 "$OUT/bench_driver.out" | wc -c)" | tee -a "$S"
  tail -1 "$OUT/bench_driver.out" | tee -a "$S"
  cp bench_side.json "$OUT/bench_driver_side.json" 2>/dev/null
  echo "== bench, defaults ==" | tee -a "$S"
  T0=$SECONDS
  timeout 1500 python3 bench.py > "$OUT/bench.out" 2> "$OUT/bench.err"
  echo "bench exit $? in $((SECONDS - T0)) s" | tee -a "$S"
  tail -1 "$OUT/bench.out" | tee -a "$S"
  cp bench_side.json "$OUT/bench_side.json" 2>/dev/null
  tail -3 "$OUT/bench.err" | cut -c1-400 | tee -a "$S"
fi
if has feas; then
  echo "== feasibility launch alone ==" | tee -a "$S"
  timeout 600 python tools/feas_roofline.py --config C2 --sims 16384 --seeds 64 --probes --verify 2>&1 | tail -1 | tee "$OUT/feas_c2.json" | cut -c1-1500 | tee -a "$S"
  timeout 600 python tools/feas_roofline.py --config C2 --sims 4096 --seeds 64 2>&1 | tail -1 | tee "$OUT/feas_c2_4096.json" | cut -c1-900 | tee -a "$S"
  timeout 600 python tools/feas_roofline.py --config C3 --sims 1024 --seeds 8 --verify 2>&1 | tail -1 | tee "$OUT/feas_c3.json" | cut -c1-900 | tee -a "$S"
  CASIM_NO_FEAS_STREAM=1 timeout 600 python tools/feas_roofline.py --config C2 --sims 16384 --seeds 64 2>&1 | tail -1 | tee "$OUT/feas_c2_round4_kernel.json" | cut -c1-900 | tee -a "$S"
fi
if has feasprof; then
  FARGS="--config C2 --sims 16384 --seeds 64 --iters 20 --probes"
  echo "== rocprofv3 kernel trace of the feasibility launch ==" | tee -a "$S"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/feas_trace" -o trace -- python "$OLDPWD/tools/feas_roofline.py" $FARGS > "$OLDPWD/$OUT/feas_trace.log" 2>&1)
  echo "trace exit $?" | tee -a "$S"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/feas_trace_c3" -o trace -- python "$OLDPWD/tools/feas_roofline.py" --config C3 --sims 1024 --seeds 8 --iters 20 > "$OLDPWD/$OUT/feas_trace_c3.log" 2>&1)
  find "$OUT/feas_trace" "$OUT/feas_trace_c3" -name "*kernel_stats*" | head -4 | while read f; do echo "--- $f"; head -8 "$f"; done | cut -c1-260 | tee -a "$S"
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/feas_pmc_$N" -o pmc -- python "$OLDPWD/tools/feas_roofline.py" $FARGS > "$OLDPWD/$OUT/feas_pmc_$N.log" 2>&1)
    echo "pmc $N exit $?" | tee -a "$S"
    (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/feas_pmc_c3_$N" -o pmc -- python "$OLDPWD/tools/feas_roofline.py" --config C3 --sims 1024 --seeds 8 --iters 20 > "$OLDPWD/$OUT/feas_pmc_c3_$N.log" 2>&1)
  done
  python tools/rocpd_summary.py "$OUT"/feas_trace "$OUT"/feas_trace_c3 "$OUT"/feas_pmc_* > "$OUT/feas_rocpd_summary.txt" 2>&1
  python tools/feas_traffic.py "$OUT" "$OUT/feas_traffic.json" 2>&1 | cut -c1-3000 | tee -a "$S"
  find "$OUT" -name "*.csv" -size +8M -delete
  find "$OUT" -name "*.db" -size +24M -delete
fi
if has sched; then bash tools/sched_counters.sh "$TAG" 2>&1 | tail -40 | tee -a "$S"; fi
if has prof; then bash tools/gpu_round3.sh "$TAG" prof 2>&1 | tail -60 | tee -a "$S"; fi
echo "== done ==" | tee -a "$S"
