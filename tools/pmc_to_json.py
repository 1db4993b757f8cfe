#!/usr/bin/env python3
"""The three counter passes of `tools/gpu_session.sh pmc` (rocprofv3 --pmc over tools/spmm_pmc.py: FETCH_SIZE |
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | SQ_*) -> the record bench.py quotes as `roofline.traffic`
(profiles/spmm_dense_traffic[_<shape>_d<emb>].json), stamped with the git blob id of the csrc/spmm.hip it was measured
on so that bench.py can refuse it once the kernel source changes.

    tools/pmc_to_json.py --out profiles/spmm_dense_traffic.json --summary profiles/r03_a_pmc_dense_spmm.txt \
        --what "..." gpurun_out/pmc_*/**.db

Counter units and the gfx950 correction are MI355X_MICROARCH.md's: FETCH_SIZE / WRITE_SIZE count kilobytes; on gfx950
FETCH_SIZE counts a 128-byte request as 64 bytes, so fetched bytes = 2 x FETCH_SIZE x 1024 (cross-check: TCC_MISS x 128 B);
WRITE_SIZE is taken as reported."""
import argparse
import collections
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("dbs", nargs="+")
ap.add_argument("--out", required=True)
ap.add_argument("--summary", required=True, help="the committed text summary this record points at")
ap.add_argument("--what", required=True, help="kernel / workload description")
ap.add_argument("--kernel", default="spmm_rows_kernel")
ap.add_argument("--tail", type=int, default=40)
a = ap.parse_args()
vals = collections.defaultdict(list)
for path in a.dbs:
    cur = sqlite3.connect(path).cursor()
    try:
        q = cur.execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id")
    except sqlite3.OperationalError:
        q = cur.execute("select kernel_name, counter_name, value from counters_collection")
    per = collections.defaultdict(list)
    for name, ctr, val in q:
        if a.kernel in name:
            per[ctr].append(float(val))
    for ctr, v in per.items():
        vals[ctr] = v[-a.tail:]
mean = {c: sum(v) / len(v) for c, v in vals.items() if v}
need = ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")
missing = [c for c in need if c not in mean]
if missing:
    raise SystemExit(f"counters missing from the passes: {missing} (have {sorted(mean)})")
fetch, write = 2.0 * mean["FETCH_SIZE"] * 1024.0, mean["WRITE_SIZE"] * 1024.0
hit = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rec = {"kernel": a.what,
       "FETCH_SIZE_KB_per_launch": round(mean["FETCH_SIZE"], 1), "WRITE_SIZE_KB_per_launch": round(mean["WRITE_SIZE"], 1),
       "TCC_HIT_per_launch": round(mean["TCC_HIT_sum"], 1), "TCC_MISS_per_launch": round(mean["TCC_MISS_sum"], 1),
       "traffic_bytes_per_launch": int(fetch + write),
       "l2_hit_rate": round(hit, 4),
       "other_counters_per_launch": {c: round(v, 1) for c, v in sorted(mean.items()) if c not in need},
       "how": (f"rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python tools/spmm_pmc.py (tools/gpu_session.sh "
               f"pmc); mean over the last {a.tail} dispatches of {a.kernel}; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
               f"counts 128-B requests at 64 B; cross-check: TCC_MISS x 128 B = {mean['TCC_MISS_sum'] * 128 / 1e6:.1f} MB vs "
               f"{fetch / 1e6:.1f} MB fetched); WRITE_SIZE as reported; L2 hit rate {hit * 100:.1f} %"),
       "summary": a.summary,
       "spmm_hip_blob": bench.git_blob_hash(os.path.join(repo, "selfrec_amd", "csrc", "spmm.hip"))}
with open(a.out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec, indent=1))
