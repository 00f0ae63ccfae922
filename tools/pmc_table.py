#!/usr/bin/env python3
"""Pivot rocprofv3 --pmc counter_collection.csv files: one row per kernel (mean over dispatches)."""
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "svb::" not in k: continue
        k = re.sub(r"\(.*", "", k).replace("void svb::", "").replace("svb::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc):
    print(k)
    for c in names:
        v = acc[k].get(c)
        if v: print(f"    {c:24s} {sum(v)/len(v):16.0f}   (n={len(v)})")
