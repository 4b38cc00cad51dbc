#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per-counter mean over the dispatches of one kernel.  usage: pmc_summarize.py <dir> [kernel substr]
The first line records a digest of the kernel sources the passes were taken on: bench.py drops the figures when the sources have changed."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "wavefront_tile_kernel<false"
extra = (" " + os.environ["BENCH_ARGS"]) if os.environ.get("BENCH_ARGS") else ""
acc = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"# kernel_source_digest {bench.kernel_source_digest()}")
print(f"# separate rocprofv3 --pmc passes of `python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-orbit --no-configs --no-graph{extra}` (tools/pmc_profile.sh); per-dispatch means over the launches of {pat}")
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} mean {sum(v)/len(v):16.1f}   n={len(v)}")
