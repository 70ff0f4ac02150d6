#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile_counters.sh: per-kernel
time from the kernel trace and per-launch PMC counter values of the render kernel."""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "renderKernel"
out = {}
ks = glob.glob(os.path.join(d, "trace", "*kernel_stats.csv"))
if ks:
    for row in csv.DictReader(open(ks[0])):
        if kernel in row["Name"]:
            out["kernel_stats"] = {k: row[k] for k in row}
vals = defaultdict(list)
for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if kernel in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
out["counters_per_launch_mean"] = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
out["launches_seen"] = {k: len(v) for k, v in sorted(vals.items())}
print(json.dumps(out, indent=1))
