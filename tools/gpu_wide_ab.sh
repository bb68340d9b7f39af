set -x
export CASIM_SET_HW_QUEUES=1
O=gpurun_out/r15a; mkdir -p $O
F="--no-cpu-baseline --no-configs --no-next-rows --no-c3 --no-feasibility-row"
for s in 4 2 8; do
  timeout 300 python bench.py --steps 400 --streams $s $F > $O/wide_s$s.out 2> $O/wide_s$s.err; echo "wide streams $s rc $?"; tail -1 $O/wide_s$s.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'bit_exact', d.get('headline_bit_exact'), 'pack', d['roofline'].get('kernel'), d['roofline'].get('kernel_ms'))"
done
CASIM_NO_WIDE=1 timeout 300 python bench.py --steps 400 $F --no-verify > $O/old_s4.out 2> $O/old_s4.err; tail -1 $O/old_s4.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OLD ms_per_step', d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -- python /root/repo/bench.py --steps 200 $F --no-verify > /root/repo/$O/prof.out 2>&1; cd /root/repo
find $O/prof -name "*kernel_stats*" | head -3
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
