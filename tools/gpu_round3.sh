#!/bin/bash
# One GPU-box session of round 3.  Usage (repo root on the GPU box):  bash tools/gpu_round3.sh <tag> [tests|bench|quick|full|prof|dist]
#   tests  pytest -m gpu + smoke          bench  the contract bench only         quick  tests + bench
#   dist   the N > 1 control flow on one GPU (gloo, 2 ranks) + RCCL in a 1-rank group
#   prof   rocprofv3 kernel trace + PMC passes of the bench command          full  everything
set -u
TAG=${1:-r03}
MODE=${2:-quick}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
S="$OUT/summary.txt"
echo "== device ==" | tee "$S"
(rocminfo | grep -E "Marketing Name|gfx9" | head -4; nproc; free -g | head -2) 2>&1 | tee -a "$S"
has() { [[ "|$1|" == *"|$MODE|"* ]]; }

if has "tests|quick|full"; then
  echo "== pytest -m gpu ==" | tee -a "$S"
  timeout 1800 python -m pytest tests -x -q -m gpu --durations=12 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" | tee -a "$S"
  tail -18 "$OUT/pytest_gpu.log" | tee -a "$S"
  echo "== smoke ==" | tee -a "$S"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
  echo "smoke exit $?" | tee -a "$S"
  tail -3 "$OUT/smoke.log" | tee -a "$S"
fi
if has "bench|quick|full"; then
  echo "== bench (driver form, defaults) ==" | tee -a "$S"
  T0=$(date +%s.%N)
  timeout 1500 python bench.py ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $? in $(echo "$(date +%s.%N) - $T0" | bc) s" | tee -a "$S"
  python - "$OUT/bench.json" <<'PY' | tee -a "$S"
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print("no bench line:", e); sys.exit(0)
keep = {k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "timed_region_s", "sims_per_s", "kernel_ms", "headline_rows", "multi_gpu")}
keep["roofline"] = {k: v for k, v in (d.get("roofline") or {}).items() if k in ("kernel", "achieved", "frac", "kernel_ms", "traffic", "device_to_itself", "issue_roofline")}
keep["cpu_baseline"] = d.get("cpu_baseline")
keep["configs"] = [{k: r.get(k) for k in ("config", "wall_ms", "oracle_ms", "bit_exact", "phases_ms", "pack_us_per_peg_step", "got", "error", "native")} for r in d.get("configs", [])]
for k in ("c3_sharded", "try_schedule_pods", "node_removals", "c3_in_process_multi_device", "group_pods", "incremental_encode"):
    keep[k] = d.get(k)
print(json.dumps(keep, indent=1)[:14000])
PY
  tail -5 "$OUT/bench.err" | tee -a "$S"
fi
if has "dist|full"; then
  echo "== bench --gpus 2, both ranks on cuda:0, gloo ==" | tee -a "$S"
  CASIM_BENCH_ONE_GPU=1 CASIM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 100 --warmup 5 --batch 1024 --no-cpu-baseline --no-configs --no-next-rows \
      > "$OUT/bench_2ranks_one_gpu_gloo.json" 2> "$OUT/bench_2ranks_one_gpu_gloo.err"
  echo "2-rank bench exit $?" | tee -a "$S"
  tail -c 3000 "$OUT/bench_2ranks_one_gpu_gloo.json" | tee -a "$S"
  tail -3 "$OUT/bench_2ranks_one_gpu_gloo.err" | tee -a "$S"
  echo "== bench, 1 rank, RCCL group forced ==" | tee -a "$S"
  CASIM_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 1 --steps 200 --warmup 5 --batch 1024 --no-cpu-baseline --no-configs --no-next-rows \
      > "$OUT/bench_1rank_rccl.json" 2> "$OUT/bench_1rank_rccl.err"
  echo "1-rank RCCL bench exit $?" | tee -a "$S"
  tail -c 3000 "$OUT/bench_1rank_rccl.json" | tee -a "$S"
  tail -3 "$OUT/bench_1rank_rccl.err" | tee -a "$S"
fi
if has "prof|full"; then
  BARGS="--steps 300 --warmup 5 --no-cpu-baseline --no-configs --no-next-rows --no-c3"   # (the timed region dominates the trace: its kernels overlap across streams)
  echo "== rocprofv3 kernel trace ==" | tee -a "$S"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --truncate-kernels -d "$OLDPWD/$OUT/prof_trace" -o trace -- \
      python "$OLDPWD/bench.py" $BARGS > "$OLDPWD/$OUT/prof_trace.log" 2>&1)
  echo "trace exit $?" | tee -a "$S"
  find "$OUT/prof_trace" -name "*kernel_stats*" | head -3 | while read f; do echo "--- $f"; head -15 "$f"; done | tee -a "$S"
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    echo "== rocprofv3 pmc $C ==" | tee -a "$S"
    # (counters are device-global, sampled around a dispatch: kernels of other streams that overlap it pollute them.  The counter passes
    # therefore run ONE sub-batch of the timed loop as the whole batch on one stream — 1024 simulations = the same 20480-wave launches,
    # device to themselves)
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/prof_pmc_$N" -o pmc -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 1 --batch 1024 --streams 1 --no-cpu-baseline --no-configs --no-next-rows --no-c3 > "$OLDPWD/$OUT/prof_pmc_$N.log" 2>&1)
    echo "pmc $N exit $?" | tee -a "$S"
  done
  python tools/rocpd_summary.py "$OUT"/prof_trace "$OUT"/prof_pmc_* > "$OUT/rocpd_summary.txt" 2>&1
  python tools/pmc_traffic.py "$OUT" "$OUT/pack_traffic.json" 2>&1 | tee -a "$S"
  find "$OUT" -name "*.csv" -size +8M -delete
  find "$OUT" -name "*.db" -size +24M -delete
fi
echo "== done ==" | tee -a "$S"
