#!/usr/bin/env python3
"""Launch ONLY the dense flavour of the propagation SpMM (the launch bench.py's roofline block is about)
so that a `rocprofv3 --pmc ...` pass over this script gives that launch's counters undiluted by the
masked flavours: tools/gpu_session.sh pmcdense -> profiles/spmm_dense_traffic.json (read by bench.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd import ops  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, args.emb, model=args.model, n_layers=args.layers, batch_size=args.batch, use_graph=False)
tr.sampler.seed(args.seed)
tr.begin_epoch()
for _ in range(3):
    tr.step()
ep = ops.make_epilogue(perturb_eps=tr.eps, rng_seed=1, rng_offset=0)
for _ in range(40):
    ops.spmm(tr.graph.adj, tr.E0, out=tr.Ha, epilogue=ep)
torch.cuda.synchronize()
print("launched 40 dense propagation SpMMs after 3 training steps")
