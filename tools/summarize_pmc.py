#!/usr/bin/env python3
"""Aggregates rocprofv3 counter_collection CSVs per kernel name (mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "prof_pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")[:60]
                c = row.get("Counter_Name", "?")
                try:
                    v = float(row.get("Counter_Value", "0"))
                except ValueError:
                    continue
                agg[k][c] += v
                cnt[k][c] += 1
        print(f"--- {f}")
        for k in sorted(agg):
            parts = [f"{c}={agg[k][c] / max(cnt[k][c], 1):.6g} (n={cnt[k][c]})" for c in sorted(agg[k])]
            print(f"{k}: " + "  ".join(parts))
