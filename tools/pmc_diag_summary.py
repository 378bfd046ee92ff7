#!/usr/bin/env python3
"""Summarise the passes of tools/r03_pmc_diag.sh: per kernel, the mean of every counter over its dispatches, plus the
kernel-trace duration, in one table (counters of all passes side by side)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in agg.items():
    n = max(len(v) for v in c.values())
    if n < 3:
        continue
    ds = sorted(dur.get(k, [0.0]))
    print("%s\n   dispatches/pass ~%d   duration under the profiler: median %.1f us" % (k, n, ds[len(ds) // 2]))
    for name, v in sorted(c.items()):
        print("   %-40s mean=%.5g" % (name, sum(v) / len(v)))
