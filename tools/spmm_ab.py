#!/usr/bin/env python3
"""A/B the SpMM kernel variants on the Yelp2018-shaped adjacency, interleaved in one process
(cdna_hip_programming.md rule 24).  Variants: SRH_SPMM_FLAGS bits (1 nt loads, 2 skip zero
gathers, 4 in-kernel split-row finish) x xcd_split on/off x split_len, plus two diagnostic
matrices: column ids folded into 4096 rows (x stays L2-resident: the kernel's speed without
L2 misses) and a degree-sorted relabelling."""
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
h_indptr, h_idx, vals = g.adj.h_indptr, g.adj.indices.cpu().numpy(), g.adj.vals
variants = {}


def add(name, flags, split, slen, idx=None):
    os.environ["SRH_SPMM_FLAGS"] = str(flags)
    variants[name] = ops.DeviceCSR(h_indptr, h_idx if idx is None else idx, vals, (N, N), split_len=slen,
                                   xcd_split_row=split)


for short in (16, 32, 64, 128, 256, 512):
    os.environ["SRH_SPMM_SHORT"] = str(short)
    add(f"rows+finish f20 xcd split512 short{short}", 20, U, 512)
os.environ["SRH_SPMM_SHORT"] = "64"
add("rows+finish f20 xcd split384 short64", 20, U, 384)
add("rows+finish f20 xcd split768 short64", 20, U, 768)
ref = None
for k, csr in variants.items():
    if "cols%" in k:
        continue
    out = ops.spmm(csr, x)
    if ref is None:
        ref = out
    err = (out - ref).abs().max().item()
    assert err < 1e-4, (k, err)
ep = ops.make_epilogue(perturb_eps=0.2, rng_seed=1)
plain = {}
times = {k: [] for k in list(variants) + list(plain)}
for rnd in range(7):
    for k, csr in plain.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            ops.spmm(csr, x, out=y)
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 20 * 1e3)
    for k, csr in variants.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            ops.spmm(csr, x, out=y, epilogue=ep)
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 20 * 1e3)
# reference points: a 56 MB device copy and the plain (no-epilogue) product
src = torch.empty(56 * 2**20 // 4, device="cuda"); dst = torch.empty_like(src)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    dst.copy_(src)
a.record()
for _ in range(20):
    dst.copy_(src)
b.record(); torch.cuda.synchronize()
print(f"copy of 56 MiB (read+write 112 MiB): {a.elapsed_time(b) / 20 * 1e3:.2f} us")
print(f"{'variant':52s} {'median_us':>10s} {'min_us':>8s}")
for k, v in sorted(times.items(), key=lambda kv: np.median(kv[1])):
    print(f"{k:52s} {np.median(v):10.2f} {min(v):8.2f}")
