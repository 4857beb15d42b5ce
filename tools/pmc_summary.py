#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes written by tools/pmc.sh: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and pat not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k[:110])
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
