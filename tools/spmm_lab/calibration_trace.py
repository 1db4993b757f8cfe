#!/usr/bin/env python3
"""What did the engine's start-up calibration of the XCD shares do on this box?  Builds the bench's trainer, prints the
shares it settled on, and probes / times the step's dominant launch on the calibrated and on the canonical list."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from selfrec_amd import ops  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, args.emb, model=args.model, n_layers=args.layers, batch_size=args.batch, use_graph=False)
print("shares the engine settled on:", None if tr.xcd_shares is None else list(map(int, tr.xcd_shares)))
kw = dict(perturb_eps=tr.eps, rng_seed=1, rng_offset=0)
if tr.vfree:
    kw.update(row_scale=tr.dinv, scale_in=True, scale_out=True)
ep = ops.make_epilogue(**kw)
pat = {"pattern": True} if tr.vfree else {}


def timed(n=200):
    for _ in range(10):
        ops.spmm(tr.adj, tr.E0, out=tr.Ha, epilogue=ep, **pat)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        ops.spmm(tr.adj, tr.E0, out=tr.Ha, epilogue=ep, **pat)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for label, shares in (("calibrated", tr.xcd_shares), ("canonical", None), ("calibrated", tr.xcd_shares)):
    ops.spmm_set_xcd_shares(tr.adj, tr.d, shares)
    fin = np.median(np.stack([ops.spmm_probe(tr.adj, tr.E0, tr.Ha, epilogue=ep, pattern=bool(tr.vfree))[0] for _ in range(5)]), axis=0)
    print(f"{label:<11} launch {timed():6.2f} us   per-XCD finish (probe): " + " ".join(f"{v:6.2f}" for v in fin))
