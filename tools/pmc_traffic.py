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


def per_launch(counter, like="%pack_fast%"):
    best = None
    for db_path in glob.glob(os.path.join(root, "prof_pmc_*", "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db_path).cursor()
        rows = cur.execute("select kernel_name, grid_size_x, avg(value), count(*) from counters_collection "
                           "where counter_name = ? and kernel_name like ? group by kernel_name, grid_size_x "
                           "order by count(*) desc, grid_size_x desc", (counter, like)).fetchall()   # (the launch size of the timed loop: the most frequent one)
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
# instruction issue (the packer's actual bound): wave-instructions per launch of the same dispatch size
for key, counter in (("valu_insts_per_launch", "SQ_INSTS_VALU"), ("salu_insts_per_launch", "SQ_INSTS_SALU"), ("waves", "SQ_WAVES"),
                     ("active_inst_valu_quadcycles", "SQ_ACTIVE_INST_VALU"), ("wave_quadcycles", "SQ_WAVE_CYCLES"), ("busy_cycles", "SQ_BUSY_CYCLES")):
    r = per_launch(counter)
    if r and r["grid_x"] == f["grid_x"]:
        rec[key] = r["kib"]
if rec.get("valu_insts_per_launch") and rec.get("active_inst_valu_quadcycles"):
    rec["cycles_per_valu"] = 4.0 * rec["active_inst_valu_quadcycles"] / rec["valu_insts_per_launch"]
# effective shader clock during the kernel: GRBM_GUI_ACTIVE cycles of the dispatch / its duration (same pass, same dispatches)
g = per_launch("GRBM_GUI_ACTIVE")
if g and g["grid_x"] == f["grid_x"]:
    for db_path in glob.glob(os.path.join(root, "prof_pmc_SQ*", "**", "*.db"), recursive=True):
        try:
            cur = sqlite3.connect(db_path).cursor()
            row = cur.execute("select avg(duration) from kernels where name like '%pack%' and grid_x = ?", (g["grid_x"],)).fetchone()
            if row and row[0]:
                rec["gui_active_cycles_per_launch"] = g["kib"]
                rec["kernel_ns_in_the_counter_pass"] = float(row[0])
                # the counter comes back summed over the 8 XCDs of the device (19.4 "GHz" otherwise)
                rec["gui_active_xcds"] = 8
                ghz = g["kib"] / 8.0 / float(row[0])
                # (only a pass in which the kernel ran at its usual speed says something about the clock: with several streams' launches
                # overlapping under the profiler the dispatch lasts 1.4x longer and the quotient drops below any real clock of the part)
                if 2.0 <= ghz <= 2.6:
                    rec["effective_clock_ghz"] = ghz
                else:
                    rec["effective_clock_ghz_rejected"] = ghz
        except sqlite3.Error:
            pass
# calibration of the FETCH_SIZE rule on known streams (casim_stream_probe: 4 B / lane and 16 B / lane reads of 1 GiB)
cal = {}
for width in (4, 16):
    r = per_launch("FETCH_SIZE", f"%stream_probe_kernel<{width}>%")
    if r:
        cal[f"read_{width}B_per_lane"] = {"fetch_kib_reported": r["kib"], "known_bytes": 1 << 30, "reported_over_known": r["kib"] * 1024 / float(1 << 30)}
r = per_launch("FETCH_SIZE", "%stream_probe_scalar_kernel%")
if r:   # (the probe reads 65536 regions of whole 32-byte records: 1 GiB exactly for a 1 GiB request)
    cal["read_32B_scalar_loads"] = {"fetch_kib_reported": r["kib"], "known_bytes": 1 << 30, "reported_over_known": r["kib"] * 1024 / float(1 << 30)}
if cal:
    rec["fetch_size_calibration"] = cal
    c4 = cal.get("read_4B_per_lane")
    if c4 and c4["reported_over_known"] > 0:
        rec["fetch_bytes_corrected_by_4B_probe"] = f["kib"] * 1024 / c4["reported_over_known"]
        rec["traffic_bytes_per_launch_by_4B_probe"] = rec["fetch_bytes_corrected_by_4B_probe"] + w["kib"] * 1024
    cs_ = cal.get("read_32B_scalar_loads")
    if cs_ and cs_["reported_over_known"] > 0 and "pack_fast" in rec.get("kernel", ""):
        # the register packer fetches its records with scalar loads: the counter is corrected by the probe of THAT access path
        rec["fetch_bytes_corrected_by_scalar_probe"] = f["kib"] * 1024 / cs_["reported_over_known"]
        rec["traffic_bytes_per_launch_by_scalar_probe"] = rec["fetch_bytes_corrected_by_scalar_probe"] + w["kib"] * 1024
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(rec))
