#!/usr/bin/env python3
"""Does the ORDER of the table rows matter to the d = 64 propagation launch?  Node ids are opaque to the model
(ui_graph.py:29-45 hands them out in first-appearance order), so the engine may keep its tables in any order.
A/B on the Yelp2018-shaped adjacency: columns as they come, relabelled by descending / ascending degree inside the
user block and the item block, and shuffled -- same row lengths, same schedule, only the gathered addresses move."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import ops, synth  # noqa: E402
from selfrec_amd.data.ui_graph import Interaction  # noqa: E402

tu, ti, su, si, U, I = synth.make_dataset("yelp2018")
data = Interaction({}, synth.as_triples(tu, ti), [])
g = data.device_graph(column_classes=False)
N, d = U + I, 64
h_indptr = g.adj.h_indptr.astype(np.int64)
h_idx = g.adj.indices.cpu().numpy().astype(np.int64)
vals = g.adj.vals.cpu().numpy()
deg = np.diff(h_indptr)
rng = np.random.default_rng(0)


def relabel(kind):
    new = np.arange(N)
    for lo, hi in ((0, U), (U, N)):
        ids = np.arange(lo, hi)
        if kind == "desc":
            order = ids[np.argsort(-deg[lo:hi], kind="stable")]
        elif kind == "asc":
            order = ids[np.argsort(deg[lo:hi], kind="stable")]
        elif kind == "shuffle":
            order = rng.permutation(ids)
        else:
            order = ids
        new[order] = ids                      # node order[k] moves to row lo + k
    return new


variants = {}
for kind in ("as-is", "desc", "asc", "shuffle"):
    cols = relabel(kind)[h_idx]
    perm, row_mid = ops.column_class_order(h_indptr, cols, 64)
    variants[kind] = ops.DeviceCSR(h_indptr.astype(np.int32), cols[perm].astype(np.int32), vals[perm], (N, N),
                                   xcd_split_row=U, row_mid=row_mid)
x = torch.randn((N, d), device="cuda")
y = torch.empty_like(x)
ep = ops.make_epilogue(perturb_eps=0.2, rng_seed=1)
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
print(f"{'column order':12s} {'median_us':>10s} {'min_us':>8s}")
for k, v in times.items():
    print(f"{k:12s} {np.median(v):10.2f} {min(v):8.2f}")
