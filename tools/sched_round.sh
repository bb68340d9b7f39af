set -u
OUT=gpurun_out/r01g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sched.py -x -q -m gpu > $OUT/pytest_sched.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -15 $OUT/pytest_sched.log | tee -a $OUT/summary.txt
timeout 900 python tools/time_pending.py > $OUT/time_pending.jsonl 2> $OUT/time_pending.err; echo "time exit $?" | tee -a $OUT/summary.txt
cat $OUT/time_pending.jsonl | tee -a $OUT/summary.txt; tail -5 $OUT/time_pending.err | tee -a $OUT/summary.txt
