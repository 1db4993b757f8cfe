#!/usr/bin/env python3
"""The fused XSimGCL step with its row-masked product (the last forward layer: batch rows only) on the batch's live-task list
(engine.live_stride, the default) against the launch over the plan's whole list, A/B in ONE process: alternating fenced
regions of the same trainer (the graph is re-captured at every switch).  Also: what the host thread spends per epoch on
sampling, the row -> slot lists and the live-task lists (the epoch itself lasts ~0.18 s on the device at this shape)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, args.emb, model=os.environ.get("AB_MODEL", "XSimGCL"), n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2,
                  tau=0.2, layer_cl=1, batch_size=2048, use_graph=True, nce_precision="f32")
assert tr.live_stride > 0, "no live-task lists at this shape"
print(f"live_stride {tr.live_stride} records per batch (the plan runs {tr.ops.spmm_plan_run_tasks(tr.adj, tr.d)}); "
      f"host threads available {os.cpu_count()}")
tr.sampler.seed(3)
for rep in range(3):
    t0 = time.perf_counter()
    ep = tr.sampler.epoch(tr.B, 1, with_unique=True, slot=0)
    t1 = time.perf_counter()
    ep = tr.sampler.epoch(tr.B, 1, with_unique=True, slot=1, with_segments=tr.rows.segment_row_offsets())
    t2 = time.perf_counter()
    host = tr._sample_epoch_host_now(slot=0)
    t3 = time.perf_counter()
    print(f"host per epoch: sampling {t1 - t0:.3f} s | sampling + row -> slot lists {t2 - t1:.3f} | + live-task lists {t3 - t2:.3f} "
          f"(mean {host['n_live'].mean():.0f} records per batch, max {host['n_live'].max()})")
r = bench.Runner(tr, args.seed)
r.run(50); r.fence()
N = int(os.environ.get("AB_STEPS", 600))
stride = tr.live_stride
out = {"live": [], "whole": []}
for rep in range(int(os.environ.get("AB_REPS", 5))):
    for mode in ("live", "whole"):
        tr._live_off = mode != "live"
        tr.reset_graph()
        r.run(30); r.fence()
        dt, bounds, _ = r.timed(N, mode)
        out[mode].append(dt / N * 1e3)
        print(f"rep {rep} {mode:5s}: {dt / N * 1e3:.4f} ms/step  ({bounds} epoch boundaries inside)", flush=True)
f, s = np.array(out["live"]), np.array(out["whole"])
print(f"live-task lists median {np.median(f):.4f} ms/step = {2048 / np.median(f) / 1e3:.3f} M pairs/s")
print(f"whole list      median {np.median(s):.4f} ms/step = {2048 / np.median(s) / 1e3:.3f} M pairs/s")
print(f"paired difference live - whole: median {np.median(f - s) * 1e3:.1f} us per step (min {np.min(f - s) * 1e3:.1f}, max {np.max(f - s) * 1e3:.1f})")
