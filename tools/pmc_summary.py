#!/usr/bin/env python3
"""Average PMC counters per kernel name from rocprofv3 counter_collection CSVs."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
