#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of
bench.py, see tools/gpu_round.sh) -> profiles/pack_traffic.json, which bench.py reports as roofline.traffic.

Per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced streaming read, so the corrected figure doubles it (an upper bound for this kernel's mixed
4-byte record loads; the uncorrected figure is kept next to it); WRITE_SIZE is taken as reported.
Usage: pmc_traffic.py <gpurun_out/TAG> <out.json>"""
import glob
import json
import os
import sqlite3
import sys

root, out_path = sys.argv[1], sys.argv[2]


def per_launch(counter):
    best = None
    for db_path in glob.glob(os.path.join(root, f"prof_pmc_{counter}*", "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db_path).cursor()
        rows = cur.execute("select kernel_name, grid_size_x, avg(value), count(*) from counters_collection "
                           "where counter_name = ? and kernel_name like '%pack%' group by kernel_name, grid_size_x "
                           "order by grid_size_x desc", (counter,)).fetchall()
        if rows:
            best = dict(kernel=rows[0][0], grid_x=int(rows[0][1]), kib=float(rows[0][2]), dispatches=int(rows[0][3]), db=os.path.relpath(db_path, root))
    return best


f, w = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE")
if not f or not w:
    sys.exit(f"no pack-kernel counters under {root}")
rec = {"kernel": f["kernel"], "waves_per_launch": f["grid_x"] // 64, "fetch_kib_reported": f["kib"], "write_kib_reported": w["kib"],
       "fetch_bytes_corrected_x2": f["kib"] * 1024 * 2, "write_bytes": w["kib"] * 1024,
       "traffic_bytes_per_launch": f["kib"] * 1024 * 2 + w["kib"] * 1024,
       "traffic_bytes_per_launch_uncorrected": (f["kib"] + w["kib"]) * 1024,
       "dispatches_averaged": [f["dispatches"], w["dispatches"]], "source": [f["db"], w["db"]], "run": os.path.basename(os.path.normpath(root))}
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(rec))
