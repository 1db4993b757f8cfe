#!/usr/bin/env python3
"""The fused XSimGCL step with the fixed-order batch-gradient reduction (rows_finish: one writer per gradient row, no float
atomics -- engine.det_scatter, the default) against the atomic scatter (nce_finish_bpr2), A/B in ONE process: alternating
fenced regions of the same trainer (the graph is re-captured at every switch), so that box-to-box drift (+-3 % between
bench.py runs) cancels.  Prints ms/step per region and the paired difference."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, args.emb, model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1,
                  batch_size=2048, use_graph=True, nce_precision="f32")
assert tr.det_scatter
r = bench.Runner(tr, args.seed)
r.run(50); r.fence()
N = int(os.environ.get("AB_STEPS", 600))
out = {"fixed-order": [], "atomics": []}
for rep in range(int(os.environ.get("AB_REPS", 5))):
    for mode in ("fixed-order", "atomics"):
        tr._seg_off = mode == "atomics"
        tr.reset_graph()
        r.run(30); r.fence()
        dt, bounds, _ = r.timed(N, mode)
        out[mode].append(dt / N * 1e3)
        print(f"rep {rep} {mode:11s}: {dt / N * 1e3:.4f} ms/step  ({bounds} epoch boundaries inside)", flush=True)
f, s = np.array(out["fixed-order"]), np.array(out["atomics"])
print(f"fixed-order median {np.median(f):.4f} ms/step = {2048 / np.median(f) / 1e3:.3f} M pairs/s")
print(f"atomics     median {np.median(s):.4f} ms/step = {2048 / np.median(s) / 1e3:.3f} M pairs/s")
print(f"paired difference fixed-order - atomics: median {np.median(f - s) * 1e3:.1f} us per step (min {np.min(f - s) * 1e3:.1f}, max {np.max(f - s) * 1e3:.1f})")
