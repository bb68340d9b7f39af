#!/bin/bash
# Instruction / wait counters of K_sched on the 15 000-node f1 and f4 workloads (counters only + --kernel-trace).
# Usage on the GPU box: bash tools/pmc_sched.sh <tag> [lib]   -> gpurun_out/<tag>/pmc_sched.txt
set -u
TAG=${1:-pmc_sched}; LIB=${2:-}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
[ -n "$LIB" ] && export CASIM_LIB_PATH=$PWD/$LIB
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$i" -o pmc -- \
      python "$OLDPWD/tests/tools/sched_one.py" both 2 > "$OUT/pmc_$i.log" 2>&1)
  echo "pass $i exit $?"
done
python tools/rocpd_summary.py "$OUT"/pmc_* 2>&1 | grep -E "sched_kernel|^kernel," > "$OUT/pmc_sched.txt"
cat "$OUT/pmc_sched.txt"
find "$OUT" -name "*.csv" -size +4M -delete
