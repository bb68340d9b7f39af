# K_sched workgroup sizes on the f1 / f4 timing tools: SWEEP_T="64 128 256 512"
for T in ${SWEEP_T:-64 128 256 512}; do
  echo "== T=$T"
  CASIM_SCHED_THREADS=$T CASIM_ORACLE_CHECK_LIMIT=0 python tests/tools/time_pending.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], round(r['gpu_kernels_ms'],4))"
  CASIM_SCHED_THREADS=$T CASIM_ORACLE_NODE_LIMIT=0 python tests/tools/time_removals.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], round(r['gpu_kernels_ms'],4))"
done
