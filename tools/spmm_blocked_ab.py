#!/usr/bin/env python3
"""measure, do not argue, the out-of-cache regime of the propagation product (BASELINE.json config 4 on one
GPU: 1 M users x 500 k items, d = 128, 100 M stored non-zeros; round 4: frac 0.049 of HBM on algorithmic bytes, 17 x the
algorithmic traffic through the fabric, L2 hit 14.6 %).

A/B of ONE propagation product y = A x on the normalised adjacency, all through the product's own kernel (ops.spmm):

  full        one launch over the whole matrix (what the engine runs)
  blocks K    x cut into K blocks of consecutive rows, each sized for the 256 MiB Infinity Cache when K >= 4
              (768 MB of x / K); A cut into the K matching column blocks (K CSR matrices over the same rows); K launches,
              the first writes y, the others accumulate through the AXPY epilogue (y = A_k x + y).  Trades K - 1 extra
              passes over y (768 MB read + written each) for gathers that stay inside one cache-sized slice of x.
  relabel     nodes renumbered by degree, descending (users and items each among themselves; ids are opaque:
              data/ui_graph.py:29-38), so the popular rows of x are contiguous -- then `full` and `blocks K` again.

Prints us per product for each, the result check against `full`, and (under rocprofv3 --pmc, SPMM_AB_ONLY=<variant>) runs
one variant alone so that FETCH_SIZE / TCC_HIT / TCC_MISS can be read per variant (tools/gpu_session.sh blockedab)."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import ops, synth  # noqa: E402

SHAPE = os.environ.get("SPMM_AB_SHAPE", "1m-500k")
D = int(os.environ.get("SPMM_AB_D", 128))
E = int(os.environ.get("SPMM_AB_EDGES", 50_000_000))
KS = [int(k) for k in os.environ.get("SPMM_AB_BLOCKS", "2,4,6,8,12").split(",") if k]
ONLY = os.environ.get("SPMM_AB_ONLY", "")
ITERS = int(os.environ.get("SPMM_AB_ITERS", 10))
DEV = torch.device("cuda", 0)


def normalised_adjacency(tu, ti, U, I):
    """D^-1/2 [[0, R], [R^T, 0]] D^-1/2 as scipy CSR (data/graph.py:10-24), fp32 values"""
    n = U + I
    rows = np.concatenate([tu, ti + U])
    cols = np.concatenate([ti + U, tu])
    a = sp.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)), shape=(n, n))
    deg = np.asarray(a.sum(axis=1)).ravel()
    dinv = np.zeros(n, dtype=np.float32)
    dinv[deg > 0] = np.power(deg[deg > 0], -0.5).astype(np.float32)
    a = sp.diags(dinv) @ a @ sp.diags(dinv)
    a = a.tocsr().astype(np.float32)
    a.sort_indices()
    return a


def timed(fn, iters=ITERS):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def column_blocks(a, k):
    """K CSR matrices over the same rows, block b holding the columns [b n / K, (b + 1) n / K)"""
    n = a.shape[1]
    edges = [n * b // k for b in range(k + 1)]
    csc = a.tocsc()
    out = []
    for b in range(k):
        sub = sp.csc_matrix((csc.data[csc.indptr[edges[b]]:csc.indptr[edges[b + 1]]],
                             csc.indices[csc.indptr[edges[b]]:csc.indptr[edges[b + 1]]],
                             csc.indptr[edges[b]:edges[b + 1] + 1] - csc.indptr[edges[b]]),
                            shape=(a.shape[0], edges[b + 1] - edges[b]))
        m = sub.tocsr()
        m.sort_indices()
        # columns keep their GLOBAL ids (x is passed whole; only rows of the block are ever fetched)
        out.append((m.indptr, (m.indices + edges[b]).astype(np.int32), m.data))
    return out


def run(label, a, x, ref=None):
    n = a.shape[0]
    results = {}
    if not ONLY or ONLY == f"{label}:full":
        full = ops.DeviceCSR(a.indptr, a.indices, a.data, a.shape, device=DEV)
        y = torch.empty((n, D), device=DEV)
        us = timed(lambda: ops.spmm(full, x, out=y))
        results["full"] = (us, y.clone())
        print(f"{label:8s} full: {us:9.1f} us per product   ({a.nnz} non-zeros, {a.nnz * D * 4 / 1e9:.1f} GB of row gathers)", flush=True)
        del full
    for k in KS:
        if ONLY and ONLY != f"{label}:blocks{k}":
            continue
        t0 = time.time()
        blocks = [ops.DeviceCSR(ip, ix, v, a.shape, device=DEV) for ip, ix, v in column_blocks(a, k)]
        y = torch.empty((n, D), device=DEV)
        acc = ops.make_epilogue(add=[y], add_scale=[1.0])

        def product():
            ops.spmm(blocks[0], x, out=y)
            for b in blocks[1:]:
                ops.spmm(b, x, out=y, epilogue=acc)
        us = timed(product)
        note = ""
        base = results.get("full", (None, ref))[1]
        if base is not None:
            err = float((y - base).abs().max() / base.abs().max())
            note = f"max |diff| / max |y| vs full = {err:.2e}"
        print(f"{label:8s} blocks {k:2d} ({n * D * 4 / k / 2**20:6.0f} MiB of x each, plans built in {time.time() - t0:.0f} s): "
              f"{us:9.1f} us per product   {note}", flush=True)
        del blocks
        torch.cuda.empty_cache()
    return results


def main():
    t0 = time.time()
    kw = {"n_edges": E} if SHAPE == "1m-500k" else {}
    tu, ti, su, si, U, I = synth.make_dataset(SHAPE, **kw)
    a = normalised_adjacency(tu, ti, U, I)
    print(f"# {SHAPE}: {U} x {I}, {a.nnz} stored non-zeros, d = {D}; built in {time.time() - t0:.0f} s", flush=True)
    rng = np.random.default_rng(0)
    x_host = (rng.standard_normal((U + I, D)) * 0.1).astype(np.float32)
    x = torch.from_numpy(x_host).to(DEV)
    if not ONLY or ONLY.startswith("ids:"):
        run("ids", a, x)
    if not ONLY or ONLY.startswith("relabel:"):
        # degree-descending relabel, users and items each among themselves (a symmetric permutation of A; x rows follow)
        deg = np.diff(a.indptr)
        order = np.concatenate([np.argsort(-deg[:U], kind="stable"), U + np.argsort(-deg[U:], kind="stable")])
        new_of_old = np.empty_like(order)
        new_of_old[order] = np.arange(order.size)
        pa = a[order][:, order].tocsr()
        pa.sort_indices()
        px = torch.from_numpy(x_host[order]).to(DEV)
        run("relabel", pa, px)


if __name__ == "__main__":
    main()
