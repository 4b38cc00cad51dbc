#!/usr/bin/env python3
"""profiles/rNN_kernel_stats.json from a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py: the average begin-to-end
duration of the product kernel, stamped with a digest of the kernel sources (bench.py drops the figure when the sources have changed).
usage: kernel_stats_summarize.py <dir with *kernel_stats.csv> <out.json> [kernel substr]"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d, out = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "wavefront_tile_kernel<false"
best = None
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row["Name"] and (best is None or int(row["Calls"]) > best["calls"]):
                best = {"kernel": row["Name"].split("(")[0].replace("void ", ""), "calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]),
                        "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]), "source_csv": os.path.relpath(f, ROOT)}
if best is None:
    raise SystemExit(f"no kernel matching {pat!r} under {d}")
best["kernel_source_digest"] = bench.kernel_source_digest()
best["command"] = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-orbit --no-configs"
with open(out, "w") as fh:
    json.dump(best, fh, indent=1)
    fh.write("\n")
print(json.dumps(best))
