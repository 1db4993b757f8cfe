#!/usr/bin/env python3
"""Why does the driver's 20-step region (bench.py --steps 20 --warmup 5: 0.295 ms/step in round 3) read 4 % slower than the
steady state (0.284)?  On bench.py's trainer, after the graph is captured:
  A. ten back-to-back fenced regions of 20 steps (what the driver times, repeated): a ramp (clocks, caches) shows as a trend,
     a fixed cost per region as a constant offset against C;
  B. the GPU time of each of the first 80 replays (HIP events between replays);
  C. one fenced region of 2,000 steps;
  D. the host cost of trainer.step() (python + hipGraphLaunch): enqueue 300 steps without a fence, against their GPU time;
  E. region of 20 steps with the launches issued back to back from a pre-built list (no python work between replays)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
from selfrec_amd.engine import FusedTrainer  # noqa: E402

t0 = time.perf_counter()
tr = FusedTrainer(data, args.emb, model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1,
                  batch_size=2048, use_graph=True)
print(f"trainer constructed in {time.perf_counter() - t0:.3f} s (plan, XCD calibration)")
r = bench.Runner(tr, args.seed)
t0 = time.perf_counter(); r.run(1); r.fence()
print(f"first step (epoch upload + capture + replay): {time.perf_counter() - t0:.3f} s")
r.run(4); r.fence()
print("A. fenced regions of 20 steps, ms/step:", end=" ")
for k in range(10):
    dt, _ = r.timed(20, "a")
    print(f"{dt / 20 * 1e3:.4f}", end=" ")
print()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
r.fence()
ev[0].record()
for k in range(80):
    r.run(1)
    ev[k + 1].record()
r.fence()
per = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(80)]
print("B. GPU us per replay, first 80 after a fence:", " ".join(f"{v:.0f}" for v in per))
dt, b = r.timed(2000, "c")
print(f"C. fenced region of 2000 steps: {dt / 2000 * 1e3:.4f} ms/step ({b} epoch boundaries inside)")
r.fence()
t0 = time.perf_counter(); r.run(300); t_enq = time.perf_counter() - t0
r.fence(); t_all = time.perf_counter() - t0
print(f"D. enqueue of 300 steps: {t_enq / 300 * 1e6:.1f} us/step on the host; with the fence {t_all / 300 * 1e6:.1f} us/step")
g = tr._graph
for k in range(5):
    r.fence()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"E. 20 bare graph.replay() + synchronize: {dt / 20 * 1e3:.4f} ms/step")
    tr.step_count += 20

# F. no sampler thread alive, GPU idle for 0.3 s first: a slow first region here is the GPU's own ramp (clocks), not the host
if r.pre._thread is not None:
    r.pre._thread.join()
for idle in (0.3, 0.0, 1.0):
    r.fence()
    time.sleep(idle)
    out = []
    for k in range(4):
        dt, _ = r.timed(20, "f")
        out.append(f"{dt / 20 * 1e3:.4f}")
    print(f"F. after {idle:.1f} s idle, sampler thread finished: fenced regions of 20 steps, ms/step:", " ".join(out))
# G. a sampler thread working on the host next to the launches (what the first 46 ms after an epoch boundary look like)
import threading  # noqa: E402
for rep in range(2):
    r.fence()
    th = threading.Thread(target=tr.sample_epoch_host, daemon=True)
    th.start()
    out = []
    for k in range(4):
        dt, _ = r.timed(20, "g")
        out.append(f"{dt / 20 * 1e3:.4f}")
    alive = th.is_alive()
    th.join()
    print(f"G. with a sampler thread running (still alive after the 4 regions: {alive}): ms/step:", " ".join(out))
