#!/bin/bash
# Instruction / wait counters of the removal loop's two kernels on bench.py's node_removals workload (5000 nodes, 1500 candidates): separate
# rocprofv3 --pmc passes (kernel trace only) of tests/tools/removal_ab.py -> gpurun_out/<tag>/removal_counters.json.
# Usage on the GPU box: bash tools/removal_counters.sh <tag> [runonce]     runonce: the passes run tests/tools/time_runonce_scale_down.py 400 instead
# (R3 = BenchmarkRunOnceScaleDown: removals_lean_kernel<2, true> — runs a word of nodes at a time, the log in eight parts — next to K_sched)
set -u
TAG=${1:-removal_counters}
WHAT=${2:-ab}
CMD="tests/tools/removal_ab.py 5000 2"
if [ "$WHAT" = "runonce" ]; then CMD="tests/tools/time_runonce_scale_down.py 400"; fi
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
         "SQ_WAVES SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$i" -o pmc -- \
      python "$OLDPWD/"$CMD > "$OUT/pmc_$i.log" 2>&1)
  echo "pass $i exit $?"
done
python - "$OUT" <<'PY'
import glob, json, os, sqlite3, sys
root = sys.argv[1]
rec = {}
for db in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(db).cursor()
    try:
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        kt = "kernels" if "kernels" in tabs else None
        rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%sched_kernel%' or kernel_name like '%removals_lean_kernel%' group by kernel_name, counter_name").fetchall()
        durs = cur.execute("select name, avg(duration), count(*) from kernels where name like '%sched_kernel%' or name like '%removals_lean_kernel%' group by name").fetchall() if kt else []
    except sqlite3.Error as e:
        print("skip", db, e); continue
    for name, counter, val, n in rows:
        key = "lean" if "removals_lean" in name else "k_sched"
        rec.setdefault(key, {"kernel": name[:100], "counters": {}, "kernel_ns_in_counter_passes": []})["counters"][counter] = float(val)
    for name, dur, n in durs:
        key = "lean" if "removals_lean" in name else "k_sched"
        rec.setdefault(key, {"kernel": name[:100], "counters": {}, "kernel_ns_in_counter_passes": []})["kernel_ns_in_counter_passes"].append(float(dur))
json.dump(rec, open(os.path.join(root, "removal_counters.json"), "w"), indent=1)
print(json.dumps(rec, indent=1)[:4000])
PY
find "$OUT" -name "*.csv" -size +4M -delete
find "$OUT" -name "*.db" -size +16M -delete
