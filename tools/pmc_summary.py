#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, the largest dispatch's counters."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
by = collections.defaultdict(dict)
for r in rows:
    key = (r["Kernel_Name"].split("(")[0][-40:], r["Dispatch_Id"])
    by[key][r["Counter_Name"]] = float(r["Counter_Value"])
    by[key]["_grid"] = int(r.get("Grid_Size", 0) or 0)
best = {}
for (k, d), v in by.items():
    if k not in best or v["_grid"] > best[k]["_grid"]:
        best[k] = v
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None  # kernel name fragments to keep
for k, v in sorted(best.items(), key=lambda kv: -kv[1]["_grid"]):
    if only is None or any(o in k for o in only):
        print(k, {a: b for a, b in v.items()})
