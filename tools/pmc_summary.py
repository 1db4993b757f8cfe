#!/usr/bin/env python3
"""Summarise PMC counters from a rocprofv3 rocpd sqlite database: per kernel, per counter,
number of dispatches and the mean value per dispatch.
usage: tools/pmc_summary.py results.db [--tail N]     (--tail N: only the last N dispatches of each kernel)"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = collections.defaultdict(lambda: collections.defaultdict(list))
try:        # dispatch order (needed by --tail)
    query = cur.execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id")
except sqlite3.OperationalError:
    query = cur.execute("select kernel_name, counter_name, value from counters_collection")
for name, ctr, val in query:
    short = re.sub(r'^void ', '', name.replace('(anonymous namespace)::', ''))
    rows[re.sub(r'\(.*$', '', short)[:60]][ctr].append(float(val))
tail = int(sys.argv[sys.argv.index("--tail") + 1]) if "--tail" in sys.argv else 0
for name, ctrs in sorted(rows.items(), key=lambda kv: -max(sum(v) for v in kv[1].values())):
    if tail:
        ctrs = {c: v[-tail:] for c, v in ctrs.items()}
    parts = [f"{c}: n={len(v)} mean={sum(v) / len(v):.5g}" for c, v in sorted(ctrs.items())]
    print(f"{name:60s} " + " | ".join(parts))
