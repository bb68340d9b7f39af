#!/bin/bash
# Instruction-mix PMC passes of the bench's pack kernel (one rocprofv3 run per counter group; counters only, no tracing
# domains besides --kernel-trace).  Usage on the GPU box: bash tools/pmc_insts.sh <tag>   -> gpurun_out/<tag>/pmc_insts.txt
set -u
TAG=${1:-pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$i" -o pmc -- \
      python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-next-rows --no-c3 > "$OUT/pmc_$i.log" 2>&1)
  echo "pass $i exit $?"
done
python tools/rocpd_summary.py "$OUT"/pmc_* 2>&1 | grep -E "pack_fast|^kernel," > "$OUT/pmc_insts.txt"
cat "$OUT/pmc_insts.txt"
find "$OUT" -name "*.csv" -size +4M -delete
