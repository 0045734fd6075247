#!/usr/bin/env python3
"""Per-kernel mean of each counter in a rocprofv3 counter_collection.csv (value per dispatch)."""
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-48:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    if not k.startswith(("pm::", "void pm::")):
        continue
    n = len(disp[k])
    print(k, "dispatches", n, {a: round(b / n) for a, b in sorted(v.items())})
