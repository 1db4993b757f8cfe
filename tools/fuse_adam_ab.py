#!/usr/bin/env python3
"""The fused XSimGCL step with Adam inside the last backward product's row epilogue (SRH_EPI_ADAM, the default) against
the separate optimiser pass, A/B in ONE process: alternating fenced regions of the same trainer (the graph is re-captured
at every switch), so that box-to-box and minute-to-minute drift cancels.  Prints ms/step per region and the paired
difference, then checks that both forms leave the same parameters from the same state."""
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
assert tr.fuse_adam, "fuse_adam is off (SRH_FUSE_ADAM=0?)"
r = bench.Runner(tr, args.seed)
r.run(50); r.fence()


def switch(on):
    tr.fuse_adam, tr._graph = on, None


N = int(os.environ.get("AB_STEPS", 600))
out = {True: [], False: []}
for rep in range(int(os.environ.get("AB_REPS", 5))):
    for on in (True, False):
        switch(on)
        r.run(30); r.fence()
        dt, bounds, _ = r.timed(N, "fused" if on else "separate")
        out[on].append(dt / N * 1e3)
        print(f"rep {rep} {'epilogue' if on else 'separate'}: {dt / N * 1e3:.4f} ms/step  ({bounds} epoch boundaries inside)", flush=True)
f, s = np.array(out[True]), np.array(out[False])
print(f"Adam in the epilogue median {np.median(f):.4f} ms/step = {2048 / np.median(f) / 1e3:.3f} M pairs/s")
print(f"Adam as its own pass median {np.median(s):.4f} ms/step = {2048 / np.median(s) / 1e3:.3f} M pairs/s")
print(f"paired difference epilogue - separate: median {np.median(f - s) * 1e3:.1f} us per step (min {np.min(f - s) * 1e3:.1f}, max {np.max(f - s) * 1e3:.1f})")

# same state, same batches, same noise counters: the two forms against each other, and the separate pass against itself
# (the step is not bitwise reproducible: float atomics in the BPR scatter -- tools/determinism_probe.py -- so the second
# pair is the noise floor the first is read against; the bit-for-bit statement is the kernel test,
# tests/test_gpu_kernels.py::test_adam_in_the_product_epilogue_equals_product_then_adam)
if r.left < 40:
    r.run(r.left + 1)                 # (keep the replays of the five batches inside one epoch)
r.fence()
snap = (tr.E0.clone(), tr.m.clone(), tr.v.clone(), tr.cursor.clone())
res = []
for on in (True, False, False):
    switch(on)
    tr.E0.copy_(snap[0]); tr.m.copy_(snap[1]); tr.v.copy_(snap[2]); tr.cursor.copy_(snap[3])
    r.run(5); r.fence()
    res.append((tr.E0.clone(), tr.m.clone(), tr.v.clone(), tr.cursor.clone()))
for name, (x, y) in (("epilogue vs separate", (res[0], res[1])), ("separate vs separate", (res[1], res[2]))):
    print(f"{name}: after 5 steps from the same state max |dE0| = {float((x[0] - y[0]).abs().max()):.3e}, "
          f"max |dm| = {float((x[1] - y[1]).abs().max()):.3e}, max |dv| = {float((x[2] - y[2]).abs().max()):.3e}, "
          f"cursor equal: {torch.equal(x[3], y[3])}")
