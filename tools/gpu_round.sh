#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh <tag> [quick]
# Everything it writes goes to gpurun_out/<tag>/ (merged back by gpurun).
set -u
TAG=${1:-r01}
MODE=${2:-full}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== device ==" | tee "$OUT/summary.txt"
(rocminfo | grep -E "Marketing Name|gfx9" | head -4; nproc; free -g | head -2) 2>&1 | tee -a "$OUT/summary.txt"

echo "== pytest -m gpu ==" | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"

echo "== smoke ==" | tee -a "$OUT/summary.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"

echo "== bench ==" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"
tail -c 3000 "$OUT/bench.json" | tee -a "$OUT/summary.txt"
tail -5 "$OUT/bench.err" | tee -a "$OUT/summary.txt"

if [ "$MODE" = "full" ]; then
  echo "== rocprofv3 kernel trace ==" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --truncate-kernels -d "$OLDPWD/$OUT/prof_trace" -o trace -- \
      python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-next-rows > "$OLDPWD/$OUT/prof_trace.log" 2>&1)
  echo "trace exit $?" | tee -a "$OUT/summary.txt"
  find "$OUT/prof_trace" -name "*kernel_stats*" | head -3 | while read f; do echo "--- $f"; head -15 "$f"; done | tee -a "$OUT/summary.txt"
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    echo "== rocprofv3 pmc $C ==" | tee -a "$OUT/summary.txt"
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/prof_pmc_$N" -o pmc -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-next-rows > "$OLDPWD/$OUT/prof_pmc_$N.log" 2>&1)
    echo "pmc $N exit $?" | tee -a "$OUT/summary.txt"
  done
  python tools/summarize_pmc.py "$OUT" 2>&1 | tee -a "$OUT/summary.txt"
  python tools/rocpd_summary.py "$OUT"/prof_trace "$OUT"/prof_pmc_* > "$OUT/rocpd_summary.txt" 2>&1
  python tools/pmc_traffic.py "$OUT" "$OUT/pack_traffic.json" 2>&1 | tee -a "$OUT/summary.txt"
  # keep the merged payload small: raw per-dispatch CSVs can be large
  find "$OUT" -name "*.csv" -size +8M -delete
fi
echo "== done ==" | tee -a "$OUT/summary.txt"
