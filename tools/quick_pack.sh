#!/bin/bash
# Quick GPU check of a packer change: parity tests of the main modules + the contract bench without its side legs.
# usage: bash tools/quick_pack.sh <tag>
T=${1:-quick}; O=gpurun_out/$T; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_ab_structurizer.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-configs --no-next-rows --no-c3 > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.4f kernels %s" % (d["value"], d["ms_per_step"], d.get("kernel_ms")))
PY
tail -2 $O/bench.err
