#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 kernel-trace CSV over STEADY-STATE launches only: the first `--drop` calls of every kernel (warm-up steps,
first-touch, clock ramp) are left out (the --stats file averages the warm-up launches in). Prints a CSV: kernel, calls, steady calls,
mean us, median us, p10, p90, total ms.   usage: steady_stats.py <dir or kernel_trace.csv> [--drop 6]"""
import collections
import csv
import glob
import os
import sys

import numpy as np

src = sys.argv[1]
drop = int(sys.argv[sys.argv.index("--drop") + 1]) if "--drop" in sys.argv else 6
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "steady_calls", "mean_us", "median_us", "p10_us", "p90_us", "steady_total_ms"])
out = []
for k, v in rows.items():
    v.sort()
    d = np.array([x[1] for x in v[drop:]]) if len(v) > drop else np.array([x[1] for x in v])
    out.append((d.sum(), k, len(v), len(d), d.mean(), np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
for tot, k, n, ns, mean, med, p10, p90 in sorted(out, reverse=True):
    w.writerow([k[:160], n, ns, "%.2f" % mean, "%.2f" % med, "%.2f" % p10, "%.2f" % p90, "%.3f" % (tot / 1e3)])
