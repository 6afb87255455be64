#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output directories: per kernel, average counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%-4d avg=%.6g" % (c, len(v), sum(v) / len(v)))
