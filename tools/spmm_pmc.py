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

args = bench.parse([a for a in sys.argv[1:]])        # --shape 1m-500k --emb 128: BASELINE.json configs[3] on one GPU
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, args.emb, model=args.model, n_layers=args.layers, batch_size=args.batch, use_graph=False)
tr.sampler.seed(args.seed)
tr.begin_epoch()
for _ in range(3):
    tr.step()
# the step's dominant launch: the value-free dense product when the engine uses it, else the dense product with values
vf = bool(getattr(tr, "vfree", False)) and not os.environ.get("SPMM_PMC_VALUES")
kw = dict(row_scale=tr.dinv, scale_in=True, scale_out=True) if vf else {}
ep = ops.make_epilogue(perturb_eps=tr.eps, rng_seed=1, rng_offset=0, **kw)
for _ in range(int(os.environ.get("SPMM_PMC_LAUNCHES", "40"))):
    ops.spmm(tr.graph.adj, tr.E0, out=tr.Ha, epilogue=ep, **({"pattern": True} if vf else {}))
torch.cuda.synchronize()
print(f"launched dense propagation SpMMs at {args.shape} d={args.emb} ({'value-free' if vf else 'with values'}) after 3 training steps")
