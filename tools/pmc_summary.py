#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", "?").split("(")[0][:70]
        rows[name][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0) or 0))
for name, ctrs in sorted(rows.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    parts = [f"{c}: n={len(v)} mean={sum(v) / len(v):.4g}" for c, v in sorted(ctrs.items())]
    print(f"{name:70s} " + " | ".join(parts))
