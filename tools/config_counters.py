#!/usr/bin/env python3
"""tools/config_prof.sh output (gpurun_out/<tag>/<config>/trace + pmc) -> profiles/config_counters.json: per batched BASELINE config the dominant
kernel of the resident step — calls and mean duration inside the timed loop (4 streams), mean duration and wave-instructions by issue port with the
device to itself (one sub-batch on one stream).  bench.py annotates headline_rows.c3_resident / c4_resident with it.
Usage: config_counters.py <gpurun_out/TAG> <out.json>"""
import glob
import json
import os
import sqlite3
import sys

root, out_path = sys.argv[1], sys.argv[2]
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
for cdir in sorted(glob.glob(os.path.join(root, "C*"))):
    cfg = os.path.basename(cdir)
    def db(sub):
        f = glob.glob(os.path.join(cdir, sub, "**", "*.db"), recursive=True)
        return sqlite3.connect(f[0]).cursor() if f else None
    tr, pm = db("trace"), db("pmc")
    if not tr or not pm:
        continue
    # the step's kernels: those launched about once per timed step and stream (>= 400 dispatches in a 200-step trace)
    loop = tr.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name, grid_x having count(*) >= 400 order by sum(duration) desc").fetchall()
    total = sum(r[3] for r in loop) or 1.0
    rec = {"run": os.path.basename(os.path.normpath(root)), "step_kernels_in_loop": [{"kernel": n[:60], "calls": c, "mean_us": a / 1e3, "share_of_loop_kernel_time": s / total} for n, c, a, s in loop[:6]]}
    top = loop[0][0] if loop else None
    if top:
        like = top.split("(")[0].split("<")[0]
        rows = pm.execute("select kernel_name, grid_size_x, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, grid_size_x, counter_name",
                          (f"%{like}%",)).fetchall()
        best = {}
        for name, gx, counter, val, n in rows:
            best.setdefault((name, gx), {"n": n})[counter] = val
        if best:
            (name, gx), c = max(best.items(), key=lambda kv: (kv[1]["n"], kv[0][1]))
            dur = pm.execute("select avg(duration) from kernels where name = ? and grid_x = ?", (name, gx)).fetchone()[0]
            rec["dominant"] = {"kernel": name[:80], "grid_x": gx, "kernel_us_alone": (dur or 0) / 1e3, "valu_insts": c.get("SQ_INSTS_VALU"), "salu_insts": c.get("SQ_INSTS_SALU"),
                               "lds_insts": c.get("SQ_INSTS_LDS"), "waves": c.get("SQ_WAVES"),
                               "cycles_per_valu": (4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]) if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU") else None,
                               "wait_share_of_wave_cycles": (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_INST_ANY") else None}
    out[cfg] = rec
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
