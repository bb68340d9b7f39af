#!/usr/bin/env python3
"""HBM traffic and instruction counts of feas_stream_kernel from rocprofv3 PMC passes of tools/feas_roofline.py (tools/gpu_round5.sh feas)
-> <out.json>, committed as profiles/feas_traffic.json, which bench.py reports as roofline_feasibility.traffic.

MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE come in KiB; on gfx950 FETCH_SIZE reports half the bytes of a coalesced vector
read stream — calibrated here on the 4 B / lane probe of the SAME pass when it is there (the kernel's PEG columns are 4 / 8-byte lane
loads), else the guide's x2; scalar-load streams (the group records) calibrate at 1.0 and are a few per cent of this kernel's bytes.
WRITE_SIZE is taken as reported.  Usage: feas_traffic.py <gpurun_out/TAG> <out.json>"""
import glob
import json
import os
import sqlite3
import sys

root, out_path = sys.argv[1], sys.argv[2]


def rows_of(counter, like):
    out = {}
    for db_path in glob.glob(os.path.join(root, "feas_pmc_*", "**", "*.db"), recursive=True):
        try:
            cur = sqlite3.connect(db_path).cursor()
            # (grid_size_x counts threads; the kernel's blocks are 256 threads)
            for name, gx, val, n in cur.execute("select kernel_name, grid_size_x, avg(value), count(*) from counters_collection where counter_name = ? "
                                                "and kernel_name like ? group by kernel_name, grid_size_x", (counter, like)).fetchall():
                out[(name, int(gx) // 256)] = (float(val), int(n), os.path.relpath(db_path, root))
        except sqlite3.Error:
            pass
    return out


def durations(like):
    out = {}
    for db_path in glob.glob(os.path.join(root, "feas_trace*", "**", "*.db"), recursive=True):
        try:
            cur = sqlite3.connect(db_path).cursor()
            tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
            t = next((x for x in tables if x == "kernels"), None)
            if t is None:
                continue
            for name, gx, ns, n in cur.execute("select name, grid_x, avg(duration), count(*) from kernels where name like ? group by name, grid_x", (like,)).fetchall():
                out[(name, int(gx) // 256)] = (float(ns), int(n))
        except sqlite3.Error:
            pass
    return out


def tag_of(name):
    """the tag bench.py builds from casim_problem_time_feasibility's info: feas_stream_kernel<lean|full, lo|hi, bit|term>"""
    s = name.replace(" ", "")
    i = s.find("feas_stream_kernel<")
    if i < 0:
        return name
    args = s[i + len("feas_stream_kernel<"):].split(">")[0].split(",")
    if len(args) < 3:
        return name
    return "feas_stream_kernel<%s, %s, %s>" % ("lean" if args[0] == "true" else "full", "hi" if args[1] == "true" else "lo", "term" if args[2] == "true" else "bit")


fetch, write = rows_of("FETCH_SIZE", "%feas_stream_kernel%"), rows_of("WRITE_SIZE", "%feas_stream_kernel%")
if not fetch or not write:
    sys.exit(f"no feas_stream_kernel counters under {root}")
cal = None
for (name, _), (kib, n, _) in rows_of("FETCH_SIZE", "%stream_probe_kernel<4>%").items():
    cal = kib * 1024 / float(1 << 30)
dur = durations("%feas_stream_kernel%")
valu, salu = rows_of("SQ_INSTS_VALU", "%feas_stream_kernel%"), rows_of("SQ_INSTS_SALU", "%feas_stream_kernel%")
rows = []
for key, (kib, n, db) in sorted(fetch.items()):
    if key not in write:
        continue
    factor = 1.0 / cal if cal else 2.0
    r = {"kernel": key[0], "kernel_tag": tag_of(key[0]), "workgroups": key[1], "fetch_kib_reported": kib, "write_kib_reported": write[key][0],
         "fetch_correction": factor, "fetch_correction_source": "4 B / lane read probe of the same pass" if cal else "MI355X_MICROARCH.md: x2 for coalesced vector streams",
         "fetch_bytes": kib * 1024 * factor, "write_bytes": write[key][0] * 1024, "traffic_bytes_per_launch": kib * 1024 * factor + write[key][0] * 1024,
         "dispatches_averaged": [n, write[key][1]], "source": [db, write[key][2]]}
    if key in dur:
        r["kernel_ms_rocprof"] = dur[key][0] * 1e-6; r["dispatches_in_the_trace"] = dur[key][1]
    if key in valu:
        r["valu_insts_per_launch"] = valu[key][0]
    if key in salu:
        r["salu_insts_per_launch"] = salu[key][0]
    rows.append(r)
json.dump({"run": os.path.basename(os.path.normpath(root)), "rows": rows, "fetch_size_reported_over_known_4B_probe": cal}, open(out_path, "w"), indent=1)
print(json.dumps(rows, indent=1)[:3000])
