#!/usr/bin/env python3
"""Where the GPU idles around an epoch boundary: from a rocprofv3 --kernel-trace database of tools/host_cost_probe.py's last
part (or any run), the gaps between consecutive kernel dispatches that are longer than a threshold, with the kernels on
either side.   usage: tools/boundary_gaps.py trace_results.db [min_gap_us] [last_n_dispatches]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute(f"select {name_col}, start, end from kernels order by start"))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
last = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
rows = rows[-last:]


def short(n):
    n = re.sub(r'^void ', '', n.replace('(anonymous namespace)::', ''))
    return re.sub(r'\(.*$', '', n)[:60]


t_end = rows[0][2]
base = rows[0][1]
for k in range(1, len(rows)):
    gap = (rows[k][1] - t_end) / 1e3
    if gap >= min_gap:
        print(f"t = {(rows[k][1] - base) / 1e6:9.3f} ms: gap {gap:8.1f} us   after {short(rows[k - 1][0]):45s} ({(rows[k - 1][2] - rows[k - 1][1]) / 1e3:6.1f} us)"
              f"  before {short(rows[k][0])}")
    t_end = max(t_end, rows[k][2])
print(f"{len(rows)} dispatches over {(rows[-1][2] - base) / 1e6:.2f} ms")
