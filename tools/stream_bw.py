#!/usr/bin/env python3
"""Measured streaming bandwidth of this GPU with the library's own elementwise kernels, footprints far
beyond the 256 MiB Infinity Cache: axpby (2 reads + 1 write) and Adam (4 reads + 3 writes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import ops  # noqa: E402


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


for mib in (64, 256, 1024):
    n = mib * (1 << 20) // 4
    x, y = torch.ones(n, device="cuda"), torch.ones(n, device="cuda")
    t = timed(lambda: ops.axpby(0.5, x, 0.5, y))
    print(f"axpby  {mib:5d} MiB arrays: {3 * n * 4 / t / 1e9:8.1f} GB/s")
    p, g, m, v = (torch.full((n,), 0.01, device="cuda") for _ in range(4))
    t = timed(lambda: ops.adam_step(p, g, m, v, step=3, lr=1e-3))
    print(f"adam   {mib:5d} MiB arrays: {7 * n * 4 / t / 1e9:8.1f} GB/s")
    del x, y, p, g, m, v
