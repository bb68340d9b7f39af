#!/bin/bash
# GPU-box session for SURVEY §8 row f1 (filter-out-schedulable): parity tests, timing, rocprofv3 kernel trace.
# Usage (repo root on the GPU box): bash tools/sched_round.sh [tag]; writes gpurun_out/<tag>/.
set -u
TAG=${1:-r01g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sched.py -x -q -m gpu > $OUT/pytest_sched.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -15 $OUT/pytest_sched.log | tee -a $OUT/summary.txt
timeout 900 python tests/tools/time_pending.py > $OUT/time_pending.jsonl 2> $OUT/time_pending.err; echo "time exit $?" | tee -a $OUT/summary.txt
cat $OUT/time_pending.jsonl | tee -a $OUT/summary.txt; tail -5 $OUT/time_pending.err | tee -a $OUT/summary.txt
(cd /tmp && CASIM_ORACLE_CHECK_LIMIT=0 timeout 900 rocprofv3 --kernel-trace --stats --truncate-kernels -d "$OLDPWD/$OUT/prof_trace" -o trace -- \
    python "$OLDPWD/tests/tools/time_pending.py" > "$OLDPWD/$OUT/prof_trace.log" 2>&1)
echo "trace exit $?" | tee -a $OUT/summary.txt
python tools/rocpd_summary.py $OUT/prof_trace 2>&1 | tee $OUT/prof_trace_summary.txt | head -30 | tee -a $OUT/summary.txt
find "$OUT" -name "*.csv" -size +8M -delete
timeout 900 python tests/tools/time_removals.py > $OUT/time_removals.jsonl 2> $OUT/time_removals.err; echo "removals exit $?" | tee -a $OUT/summary.txt
cat $OUT/time_removals.jsonl | tee -a $OUT/summary.txt; tail -5 $OUT/time_removals.err | tee -a $OUT/summary.txt
