#!/bin/bash
# Instruction counters of K_sched on the two workloads of bench.py's try_schedule_pods / node_removals rows (separate --pmc passes with
# --kernel-trace only) -> gpurun_out/<tag>/sched_counters.json; copy it to profiles/sched_counters.json (bench.py reads it there).
# Usage on the GPU box: bash tools/sched_counters.sh <tag>
set -u
TAG=${1:-sched_counters}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$i" -o pmc -- \
      python "$OLDPWD/tests/tools/sched_one.py" bench 3 > "$OUT/pmc_$i.log" 2>&1)
  echo "pass $i exit $?"
done
python tools/sched_counters.py "$OUT" "$OUT/sched_counters.json"
python tools/rocpd_summary.py "$OUT"/pmc_* 2>&1 | grep -E "sched_kernel|sched_static|removals_lean|lean_fit0|^kernel," > "$OUT/sched_counters.txt"
find "$OUT" -name "*.csv" -size +4M -delete
find "$OUT" -name "*.db" -size +16M -delete
