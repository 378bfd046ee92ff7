#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv and prints how much kernels of different queues overlapped in time."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "copyBuffer" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                      # steady state: second half
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("queues:", sorted(set(r["Queue_Id"] for r in rows)), " kernels:", len(rows), " span %.3f ms" % ((t1 - t0) / 1e6))
for k in sorted(hist): print("  %d kernels running: %5.1f %%" % (k, 100.0 * hist[k] / tot))
print("  sum of kernel durations / span = %.2f" % (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / (t1 - t0)))
