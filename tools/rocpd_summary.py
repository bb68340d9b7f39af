#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd (.db) outputs: per-kernel stats of a kernel trace, and per-kernel
mean counter values of PMC passes.  Usage: rocpd_summary.py <dir-with-db-files>..."""
import glob
import os
import sqlite3
import sys


def summarize(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"## {path}")
    try:
        # one row per (kernel, grid size): a batch launch and a single-simulation launch of the same kernel are different
        # workloads and must not be averaged together
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                           "grid_x, max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count), max(scratch_size) "
                           "from kernels group by name, grid_x order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct,grid_x,wg_x,lds,vgpr,sgpr,scratch")
        for r in rows:
            print(f"{r[0][:70]},{r[1]},{r[2]},{r[3]:.0f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]}")
    except sqlite3.Error as e:
        print("no kernel table:", e)
    try:
        cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
        if cols:
            namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
            ccol = "counter_name" if "counter_name" in cols else None
            vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
            if namecol and ccol and vcol:
                gcol = "grid_size_x" if "grid_size_x" in cols else "0"
                rows = cur.execute(f"select {namecol}, {ccol}, count(*), avg({vcol}), sum({vcol}), {gcol} from counters_collection "
                                   f"group by {namecol}, {gcol}, {ccol} order by {namecol}, {gcol}, {ccol}").fetchall()
                if rows:
                    print("kernel,grid_x,counter,dispatches,mean_per_dispatch,sum")
                    for r in rows:
                        print(f"{str(r[0])[:70]},{r[5]},{r[1]},{r[2]},{r[3]:.6g},{r[4]:.6g}")
            else:
                print("counters_collection columns:", cols)
    except sqlite3.Error as e:
        print("no counters:", e)
    db.close()


for d in sys.argv[1:]:
    files = [d] if d.endswith(".db") else sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    for f in files:
        summarize(f)
