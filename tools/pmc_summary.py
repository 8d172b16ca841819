#!/usr/bin/env python3
"""summarise rocprofv3 --pmc csv output: per kernel name, mean of each counter per dispatch"""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = "row" if "rowpass" in k else "col" if "colpass" in k else None
        if not short: continue
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
