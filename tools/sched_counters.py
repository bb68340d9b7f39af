#!/usr/bin/env python3
"""rocprofv3 PMC passes of tests/tools/sched_one.py bench (tools/sched_counters.sh) -> sched_counters.json: per sched_kernel
instantiation (the TrySchedulePods pass and the removal loop are different template instances) the wave-instructions by issue port, the
waves, the busy cycles and the dispatch duration.  bench.py turns them into the issue roofline of its try_schedule_pods /
node_removals rows.  Usage: sched_counters.py <gpurun_out/TAG> <out.json>"""
import glob
import json
import os
import sqlite3
import sys

root, out_path = sys.argv[1], sys.argv[2]
rec = {}
for db_path in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(db_path).cursor()
    try:
        kern = cur.execute("select name, grid_x, workgroup_x, avg(duration), count(*) from kernels where name like '%sched_kernel%' or name like '%removals_lean_kernel%' group by name, grid_x").fetchall()
        cnt = cur.execute("select kernel_name, grid_size_x, counter_name, avg(value) from counters_collection where kernel_name like '%sched_kernel%' or kernel_name like '%removals_lean_kernel%' "
                          "group by kernel_name, grid_size_x, counter_name").fetchall()
    except sqlite3.Error as e:
        print("skip", db_path, e)
        continue
    for name, gx, wx, dur, n in kern:
        # template arguments <kLds, kRemoval, ...>: the second one tells the removal loop from the TrySchedulePods pass
        args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",") if "<" in name else []
        row = "node_removals" if ("removals_lean" in name or (len(args) > 1 and args[1] in ("true", "1"))) else "try_schedule_pods"
        r = rec.setdefault(row, {"kernel": name[:120], "workgroup_threads": int(wx), "grid_x": int(gx), "counters": {}, "kernel_ns_in_counter_passes": []})
        r["kernel_ns_in_counter_passes"].append(float(dur))
    for name, gx, counter, val in cnt:
        args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",") if "<" in name else []
        row = "node_removals" if ("removals_lean" in name or (len(args) > 1 and args[1] in ("true", "1"))) else "try_schedule_pods"
        if row in rec:
            rec[row]["counters"][counter] = float(val)
for row, r in rec.items():
    c = r["counters"]
    r["run"] = os.path.basename(os.path.normpath(root))
    if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        r["cycles_per_valu"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(rec, indent=1)[:3000])
