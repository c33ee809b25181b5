#!/usr/bin/env python3
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print(f'{r["Name"][:44]:46s} calls={r["Calls"]:>4s} avg={float(r["AverageNs"])/1e6:8.4f} ms  min={float(r["MinNs"])/1e6:8.4f}  {r["Percentage"]}%')
