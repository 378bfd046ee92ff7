#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (counter_collection.csv): per kernel name, mean of each counter over dispatches."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, c in agg.items():
            if "pw_gemm" in k or "dw3x3" in k or "conv" in k:
                print(k)
                for name, v in sorted(c.items()):
                    print("   %-34s n=%-3d mean=%.4g" % (name, len(v), sum(v) / len(v)))
