#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd sqlite database (the default output of
`rocprofv3 --kernel-trace --stats` on ROCm 7.2): calls, total / mean / min / max duration.
usage: tools/rocpd_stats.py trace_results.db [> profiles/xxx_kernel_stats.txt]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
q = f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc"
rows = list(cur.execute(q))
total = sum(r[2] for r in rows) or 1
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for name, n, tot, avg, mn, mx in rows:
    short = re.sub(r'^void ', '', name.replace('(anonymous namespace)::', ''))
    short = re.sub(r'\(.*$', '', short)[:72]
    print(f"{short:72s} {n:7d} {tot / 1e6:10.3f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f}")
print(f"{'TOTAL':72s} {sum(r[1] for r in rows):7d} {total / 1e6:10.3f}")
