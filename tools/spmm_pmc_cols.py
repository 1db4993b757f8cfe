#!/usr/bin/env python3
"""Launch ONLY the propagation SpMM of the column-sharded layout -- 40 dense launches per width on (N, w) tables,
w = 32 / 16 / 8 (spmm_slice_kernel<8>, <4>, spmm_pair_kernel), plain graph order as that layout builds it -- so that
`rocprofv3 --pmc ...` passes over this script give those launches' counters (tools/gpu_session.sh pmccols)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd import ops  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
g = data.device_graph(column_classes=False)
N = g.n_nodes
torch.manual_seed(0)
x64 = torch.randn((N, 64), device="cuda")
step = torch.tensor([3], dtype=torch.int64, device="cuda")
for w in (32, 16, 8):
    xs = x64[:, :w].contiguous()
    ys = torch.empty_like(xs)
    ep = ops.make_epilogue(perturb_eps=0.2, rng_seed=1, rng_step=step, rng_stride=16 * N, d_full=64, col0=0)
    for _ in range(40):
        ops.spmm(g.adj, xs, out=ys, epilogue=ep)
torch.cuda.synchronize()
print("launched 40 dense propagation SpMMs per width (32, 16, 8 columns)")
