#!/usr/bin/env python3
"""Segment length of the SpMM plan (rows longer than split_len are cut into cooperative tasks) against the launch time
of each flavour a training step issues, Yelp2018 shape, bench.py's node ids.  The masked flavours run few, long tasks:
does a finer plan shorten their critical path?

    python tools/split_probe.py            (on the GPU box; prints one table)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import _lib, ops, synth  # noqa: E402
from selfrec_amd.data.device_graph import DeviceGraph  # noqa: E402
from selfrec_amd.data.ui_graph import Interaction  # noqa: E402


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def graph_timed(fn, reps=20, iters=20):
    """us per launch with the launches replayed from a hipGraph (20 per graph): launch-rate effects of the python
    caller out of the picture -- a 10 us kernel timed through ctypes measures ctypes."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    return timed(gr.replay, iters) / reps


def main():
    _lib.require_gpu()
    dev = torch.device("cuda", 0)
    tu, ti, su, si, U, I = synth.make_dataset("yelp2018", seed=2024)
    data = Interaction({}, synth.as_triples(tu, ti), [])
    tu, ti = data.train_u.astype(np.int64), data.train_i.astype(np.int64)
    N, d = U + I, 64
    rng = np.random.default_rng(3)
    pick = rng.choice(len(tu), size=2048, replace=False)
    marked = np.unique(np.concatenate([tu[pick], ti[pick] + U, rng.integers(0, I, 2048) + U]))
    mark = torch.zeros(N, dtype=torch.int32, device=dev)
    mark[torch.from_numpy(marked).to(dev)] = 7
    stamp = torch.tensor([7], dtype=torch.int64, device=dev)
    x = (torch.randn((N, d), generator=torch.Generator().manual_seed(1)) * 0.1).to(dev)
    y = torch.zeros((N, d), device=dev)
    print(f"# yelp2018 shape, N = {N}, d = {d}, {len(marked)} marked nodes; us per launch")
    print(f"{'split_len':>10}{'tasks':>9}{'dense':>10}{'dense v-free':>14}{'row-masked':>12}{'col-masked':>12}")
    notes = []
    for split in (384, 512):
        g = DeviceGraph(data.interaction_mat, device=dev, split_len=split)
        adj = g.adj
        pat = adj.with_values(torch.ones_like(adj.vals))
        pk = dict(perturb_eps=0.2, rng_seed=1, rng_offset=0)
        eps = [ops.make_epilogue(**pk), ops.make_epilogue(**pk, row_scale=g.dinv, scale_out=True),
               ops.make_epilogue(**pk, row_mark=mark, mark_stamp=stamp), ops.make_epilogue(col_mark=mark, mark_stamp=stamp)]
        cells = [
            graph_timed(lambda: ops.spmm(adj, x, out=y, epilogue=eps[0])),
            graph_timed(lambda: ops.spmm(pat, x, out=y, pattern=True, epilogue=eps[1])),
            graph_timed(lambda: ops.spmm(adj, x, out=y, epilogue=eps[2])),
            graph_timed(lambda: ops.spmm(adj, x, out=y, epilogue=eps[3])),
        ]
        if split == 512:
            dead = torch.tensor([9], dtype=torch.int64, device=dev)          # a stamp no node carries: every row / column dead
            de = [ops.make_epilogue(**pk, row_mark=mark, mark_stamp=dead), ops.make_epilogue(col_mark=mark, mark_stamp=dead)]
            t_r = graph_timed(lambda: ops.spmm(adj, x, out=y, epilogue=de[0]))
            t_c = graph_timed(lambda: ops.spmm(adj, x, out=y, epilogue=de[1]))
            notes.append(f"# split 512 with NO live node (the scan alone): row-masked {t_r:.2f} us, col-masked {t_c:.2f} us")
        n_tasks = int(np.sum(np.maximum(1, -(-np.diff(adj.h_indptr) // split))))
        print(f"{split:>10}{n_tasks:>9}" + "".join(f"{c:>{w}.2f}" for c, w in zip(cells, (10, 14, 12, 12))))
        del g, adj, pat
    print("\n".join(notes))


if __name__ == "__main__":
    main()
