#!/usr/bin/env python3
"""A/B the SpMM kernel variants on the Yelp2018-shaped adjacency, interleaved in one process
(cdna_hip_programming.md rule 24).  Variants: SRH_SPMM_FLAGS bits (1 nt loads, 2 skip zero
gathers, 4 in-kernel split-row finish) x xcd_split on/off x split_len."""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import ops, synth  # noqa: E402
from selfrec_amd.data.ui_graph import Interaction  # noqa: E402

tu, ti, su, si, U, I = synth.make_dataset("yelp2018")
data = Interaction({}, synth.as_triples(tu, ti), [])
g = data.device_graph()
N, d = U + I, 64
x = torch.randn((N, d), device="cuda")
y = torch.empty_like(x)
h_indptr, idx, vals = g.adj.h_indptr, g.adj.indices, g.adj.vals
variants = {}
for flags, split, slen in itertools.product(range(8), (0, U), (256, 1024)):
    if slen == 1024 and flags not in (0, 4, 7):
        continue
    os.environ["SRH_SPMM_FLAGS"] = str(flags)
    csr = ops.DeviceCSR(h_indptr, idx.cpu().numpy(), vals, (N, N), split_len=slen, xcd_split_row=split)
    variants[(flags, "xcd" if split else "mix", slen)] = csr
ref = None
for k, csr in variants.items():
    out = ops.spmm(csr, x)
    if ref is None:
        ref = out
    assert (out - ref).abs().max().item() < 1e-4, k
ep = ops.make_epilogue(perturb_eps=0.2, philox_seed=1)
times = {k: [] for k in variants}
for rnd in range(7):
    for k, csr in variants.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            ops.spmm(csr, x, out=y, epilogue=ep)
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 20 * 1e3)
print(f"{'flags':>5s} {'order':>5s} {'split':>5s} {'median_us':>10s} {'min_us':>8s}")
for k, v in sorted(times.items(), key=lambda kv: np.median(kv[1])):
    print(f"{k[0]:5d} {k[1]:>5s} {k[2]:5d} {np.median(v):10.2f} {min(v):8.2f}")
