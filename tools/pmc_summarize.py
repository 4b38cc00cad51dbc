#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per-counter mean over the dispatches of one kernel.  usage: pmc_summarize.py <dir> [kernel substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "wavefront_tile_kernel<false"
acc = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} mean {sum(v)/len(v):16.1f}   n={len(v)}")
