#!/usr/bin/env python3
"""Dev tool: one rocprofv3 --pmc pass (a directory with *counter_collection.csv) -> JSON {kernel name: {"launches": n,
counter: SUM over its dispatches}}.  Usage: pmc_table.py <dir>"""
import collections, csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    d = (r.get("Dispatch_Id"), k)
    if d not in seen:
        seen.add(d); acc[k]["launches"] += 1
json.dump({k: dict(v) for k, v in acc.items()}, sys.stdout, indent=0)
