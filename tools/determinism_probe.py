#!/usr/bin/env python3
"""Run-to-run determinism of the fused XSimGCL step at the Yelp2018 shape: the same trainer configuration twice on the
canonical XCD dealing, and once on calibrated shares -- bitwise / max relative difference of losses, of the gradient before
the first Adam step, and of the final embeddings after 3 steps.  (The loss section scatters its gradients with atomics:
rows that several pairs of a batch share receive their addends in an order that varies from run to run.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd import engine, ops  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

os.environ["SRH_FUSE_ADAM"] = "0"     # the gradient before Adam exists in memory only when the optimiser is a pass of its own
args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
kw = dict(model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1, batch_size=2048)


def run(calibrate):
    os.environ["SRH_XCD_CALIBRATE"] = "1" if calibrate else "0"
    data.device_graph().adj.__dict__.pop("_xcd_calibrated", None)
    ops.spmm_set_xcd_shares(data.device_graph().adj, 64, None)
    torch.manual_seed(5)
    tr = FusedTrainer(data, 64, **kw)
    grads = []
    real = engine.ops.adam_step

    def adam_step(param, grad, *a, **k):
        if not grads:
            grads.append(grad.clone())
        return real(param, grad, *a, **k)
    engine.ops.adam_step = adam_step
    try:
        tr.sampler.seed(11)
        tr.begin_epoch()
        losses = []
        for _ in range(3):
            tr.step()
            losses.append(tr.read_losses())
    finally:
        engine.ops.adam_step = real
    fu, fi = tr.embeddings()
    return np.asarray(losses), grads[0].cpu().numpy(), torch.cat([fu, fi]).cpu().numpy(), tr.E0.cpu().numpy(), tr.xcd_shares


def cmp(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return f"max|diff| {d.max():.2e} (rel to max {d.max() / np.abs(b).max():.1e}), elements differing {int((d > 0).sum())} of {d.size}"


r1, r2, r3 = run(False), run(False), run(True)
for name, x, y in (("canonical vs canonical", r1, r2), ("calibrated vs canonical", r3, r1)):
    print(f"== {name} (shares {x[4] if x[4] is None else list(x[4])})")
    print("  losses        ", cmp(x[0], y[0]))
    print("  gE0 before Adam", cmp(x[1], y[1]))
    print("  E0 after 3 steps", cmp(x[3], y[3]))
    print("  final embeddings", cmp(x[2], y[2]))
