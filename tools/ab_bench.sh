#!/bin/bash
# A/B of packer builds on one GPU box: tools/ab_bench.sh <lib.so>...   (libs built with `make -C kubernetes_autoscaler_amd/csrc OUT=... EXTRA=...`)
# For every library: the GPU parity tests of the estimator path, then two short bench runs (sims/s, pack kernel ms).
for LIB in "$@"; do
  echo "== $LIB"
  CASIM_LIB_PATH=$PWD/$LIB timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --deselect tests/test_gpu_parity.py::test_native_library_is_loaded 2>&1 | tail -1
  for i in 1 2; do
    CASIM_LIB_PATH=$PWD/$LIB timeout 200 python bench.py --steps 30 --warmup 5 --no-dense --no-cpu-baseline --no-next-rows 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sims/s %.4g  pack_ms %.4f' % (d.get('sims_per_s'), d['roofline']['kernel_ms']))"
  done
done
