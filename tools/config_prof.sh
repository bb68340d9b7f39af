#!/bin/bash
# rocprofv3 kernel trace + one instruction-counter pass of the resident step of a batched BASELINE config other than the headline's:
#   bash tools/config_prof.sh <tag> <config> <batch> <seeds>      e.g.  r11k C4 2048 32
# trace: the timed loop as bench.py runs it (4 streams); counters: one sub-batch as the whole batch on one stream (device to itself).
set -u
TAG=$1; CFG=$2; B=$3; S=$4
OUT=$PWD/gpurun_out/$TAG/$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
COMMON="--config $CFG --seeds $S --no-cpu-baseline --no-configs --no-next-rows --no-c3 --no-feasibility-row --no-verify"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --truncate-kernels -d "$OUT/trace" -o trace -- python "$OLDPWD/bench.py" $COMMON --batch $B --steps 200 --warmup 5 > "$OUT/trace.log" 2>&1)
echo "trace exit $?"; tail -1 "$OUT/trace.log" | cut -c1-400
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d "$OUT/pmc" -o pmc -- \
    python "$OLDPWD/bench.py" $COMMON --batch $((B / 4)) --streams 1 --steps 3 --warmup 1 > "$OUT/pmc.log" 2>&1)
echo "pmc exit $?"
python tools/rocpd_summary.py "$OUT/trace" "$OUT/pmc" > "$OUT/rocpd_summary.txt" 2>&1
head -12 "$OUT/rocpd_summary.txt" | cut -c1-220
python tools/config_counters.py "$PWD/gpurun_out/$TAG" "$PWD/gpurun_out/$TAG/config_counters.json" | head -5
find "$OUT" -name "*.csv" -size +4M -delete
find "$OUT" -name "*.db" -size +16M -delete
