#!/usr/bin/env python3
"""Summarise PMC counters from a rocprofv3 rocpd sqlite database: per kernel, per counter,
number of dispatches and the mean value per dispatch.
usage: tools/pmc_summary.py results.db"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for name, ctr, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    short = re.sub(r'^void ', '', name.replace('(anonymous namespace)::', ''))
    rows[re.sub(r'\(.*$', '', short)[:60]][ctr].append(float(val))
for name, ctrs in sorted(rows.items(), key=lambda kv: -max(sum(v) for v in kv[1].values())):
    parts = [f"{c}: n={len(v)} mean={sum(v) / len(v):.5g}" for c, v in sorted(ctrs.items())]
    print(f"{name:60s} " + " | ".join(parts))
