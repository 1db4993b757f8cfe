#!/usr/bin/env python3
"""gpurun_out/eval_pmc.txt (tools/gpu_session.sh evalpmc: per-dispatch means of the counters per kernel) -> profiles/eval_mfma_busy.json,
stamped with the git blob of csrc/eval.hip so that bench.py quotes it only for the kernels it was measured on.

  python tools/eval_pmc_record.py [gpurun_out/eval_pmc.txt] [summary file under profiles/] [kernel stats file]
"""
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "eval_pmc.txt")
summary = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04_j_eval_pmc.txt"
kernels = {}
for line in open(src):
    m = re.match(r"^(\S.*?)\s{2,}(\w+: n=.*)$", line.rstrip())
    if not m or "filter16_kernel" not in m.group(1):
        continue
    c = {k: float(v) for k, v in re.findall(r"(\w+): n=\d+ mean=([0-9.e+]+)", m.group(2))}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        c = {"mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 4), **c}
        kernels[m.group(1).strip()] = c
# per-kernel durations of the same probe under rocprofv3 --kernel-trace --stats (gpurun_out/eval_kernel_stats.txt or the
# committed summary: lines "<kernel>  calls total_ms avg_us ...")
kernel_us = {}
stats = sys.argv[3] if len(sys.argv) > 3 else None
if stats and os.path.exists(stats):
    for line in open(stats):
        m = re.match(r"^(\S.*?)\s{2,}(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line.rstrip())
        if m and any(k in m.group(1) for k in ("filter16", "rescore", "bound_rows", "gemm_nt", "topk", "split_rows")):
            kernel_us[m.group(1).strip()] = {"calls": int(m.group(2)), "avg_us": float(m.group(4))}
main = next((v for k, v in kernels.items() if "false" in k and k.startswith("filter16_kernel<64")), None)
blob = subprocess.check_output(["git", "hash-object", os.path.join(REPO, "selfrec_amd", "csrc", "eval.hip")], text=True).strip()
rec = {"kernels": kernels, "mfma_busy_filter16": main["mfma_busy"] if main else None,
       "how": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -- python tools/eval_probe.py "
              "(tools/gpu_session.sh evalpmc); per-dispatch means; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): "
              "the fraction of SIMD-cycles with the matrix pipe busy while the kernel runs",
       "summary": summary, "eval_hip_blob": blob, "kernel_us": kernel_us, "kernel_us_source": stats}
with open(os.path.join(REPO, "profiles", "eval_mfma_busy.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec)[:400])
