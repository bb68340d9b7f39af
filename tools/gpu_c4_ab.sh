#!/bin/bash
# C4 / C2 / C3 / C1 resident rows of the bench (other_configs) + GPU tests, for A/B runs of the packer.  Usage: bash tools/gpu_c4_ab.sh <tag>
set -u
TAG=${1:-r15d}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests -q -m gpu -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?"; tail -3 "$OUT/pytest_gpu.log"
timeout 600 python bench.py --steps 300 --no-cpu-baseline --no-configs --no-next-rows --no-c3 --no-feasibility-row > "$OUT/bench.out" 2> "$OUT/bench.err"; echo "bench exit $?"
tail -1 "$OUT/bench.out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 ms_per_step', d['ms_per_step'], 'bit_exact', d.get('headline_bit_exact')); print(json.dumps(d.get('other_configs')))"
timeout 600 python bench.py --config C4 --batch 2048 --steps 200 --no-cpu-baseline --no-configs --no-next-rows --no-c3 --no-feasibility-row > "$OUT/bench_c4.out" 2> "$OUT/bench_c4.err"; echo "bench C4 exit $?"
tail -1 "$OUT/bench_c4.out" | cut -c1-1500
