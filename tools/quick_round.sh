set -u
OUT=gpurun_out/r01j
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/summary.txt
python bench.py --steps 20 --warmup 3 --no-dense 2>$OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-700 $OUT/bench.json | tee -a $OUT/summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/prof_pmc_$C" -o pmc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-next-rows > "$OLDPWD/$OUT/prof_pmc_$C.log" 2>&1)
done
python tools/pmc_traffic.py $OUT $OUT/pack_traffic.json | tee -a $OUT/summary.txt
