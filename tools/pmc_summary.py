#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, mean over dispatches."""
import csv, glob, sys, collections
for path in sys.argv[1:]:
    for f in glob.glob(path + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "trace_tiles" not in k and "ct_" not in k:
                continue
            print(k)
            for c, v in sorted(cs.items()):
                print("   %-32s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
