#!/usr/bin/env python3
"""Print the ct_* rows of a rocprofv3 kernel_stats.csv: tools/kstats.py DIR [label]"""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
print("== " + label)
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "ct_" in n:
            print("  %-62s calls %3s avg %9.3f us  min %9.3f" % (n.replace("void ", "").split("(")[0][:62], r["Calls"],
                  float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
