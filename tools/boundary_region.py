#!/usr/bin/env python3
"""Five real epoch boundaries, each inside ten steps on either side, between fences -- for a kernel trace
(tools/gpu_session.sh boundarytrace -> tools/boundary_gaps.py): which kernels border the idle time a boundary leaves."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, args.emb, model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1,
                  batch_size=2048, use_graph=True, nce_precision="f32")
r = bench.Runner(tr, args.seed)
r.run(100); r.fence()
for rep in range(5):
    r.run(r.left - 10)                       # ten batches of this epoch left
    time.sleep(0.15)                         # (the sampler thread is done: the boundary below does not wait for it)
    r.fence()
    t0 = time.perf_counter()
    r.run(20)
    r.fence()
    print(f"rep {rep}: 20 steps over a boundary {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per step, host ms at the boundary "
          f"{r.boundary_ms[-1]}", flush=True)
    time.sleep(0.05)
