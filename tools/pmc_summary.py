#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel name: dispatches, and per counter sum / per-dispatch mean."""
import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0][-40:]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[name].add(r["Dispatch_Id"])
for name in sorted(agg):
    n = len(cnt[name])
    print("%-42s dispatches=%d" % (name, n))
    for c, v in sorted(agg[name].items()):
        print("    %-22s sum=%.6g  per_dispatch=%.6g" % (c, v, v / max(n, 1)))
