#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each
counter per dispatch.  usage: pmc_summary.py dir [dir...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
