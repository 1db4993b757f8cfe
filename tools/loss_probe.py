#!/usr/bin/env python3
"""Launch the batch-loss entry points in isolation (BPR alone, InfoNCE alone, the fused call) on realistic index lists
(one sampled batch of the Yelp2018-shape graph) so that `rocprofv3 --kernel-trace --stats -- python tools/loss_probe.py`
attributes the loss section's time kernel by kernel."""
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
tr = FusedTrainer(data, args.emb, model="XSimGCL", n_layers=3, batch_size=2048, use_graph=False)
tr.sampler.seed(args.seed)
tr.begin_epoch()
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
st, B = tr.stage, tr.B
rows_dev, nuu, nui = tr.meta[0:1], tr.meta[1:2], tr.meta[2:3]
F, CL = tr.F, tr.Y[0]
bpr = dict(batch=B, n_rows_dev=rows_dev, reg_coef=1e-4, reg_include_neg=False, loss_scale=1.0, g_user=tr.gF, g_item=tr.gF,
           greg_user=tr.gF, greg_item=tr.gF, losses=tr.losses[0:2])
bpr_in = (F, F, F, F, st["u"], st["i"], st["j"])
problems = [(F, CL, st["uniq_u"], B, nuu, tr.gF, tr.gCL), (F, CL, st["uniq_i"], B, nui, tr.gF, tr.gCL)]
PREC = os.environ.get("LOSS_PROBE_PRECISION") or None     # 'split' | 'f32' | unset = the process default
if PREC:
    ops.set_infonce_precision(PREC)
for _ in range(int(os.environ.get("LOSS_PROBE_ITERS", "200"))):
    ops.bpr_l2_fwd_bwd(*bpr_in, **bpr, ws=tr.bpr_ws)                                     # bpr_phase1 + bpr_phase2
    ops.infonce_multi(problems, d=64, tau=0.2, loss_scale=0.2, loss=tr.losses[2:3], ws=tr.nce_ws)   # prep, 2 tiles, finish_both
    ops.bpr_infonce(*bpr_in, **bpr, bpr_ws=tr.bpr_ws, problems=problems, tau=0.2, cl_scale=0.2, cl_loss=tr.losses[2:3],
                    nce_ws=tr.nce_ws)                                                       # the fused 4-launch form
    if tr.det_scatter:                                                                     # ... and its fixed-order form: rows_finish
        ed = tr._epoch_dev
        seg = dict(n_uniq_u=nuu, n_uniq_i=nui, n_uniq_n=ed["n_uniq_n"], seg_rows=ed["seg_rows"], seg_end=ed["seg_end"],
                   seg=ed["seg"], seg_a=ed["seg_a"], seg_b=ed["seg_b"], batch_no=tr.meta[3:4],
                   rows_are_zero=os.environ.get("LOSS_PROBE_RMW") is None)
        ops.bpr_infonce(*bpr_in, **bpr, bpr_ws=tr.bpr_ws, problems=problems, tau=0.2, cl_scale=0.2, cl_loss=tr.losses[2:3],
                        nce_ws=tr.nce_ws, seg=seg, nce_rows=1)
torch.cuda.synchronize()
print("rows", int(tr.meta[0]), "uniq users", int(tr.meta[1]), "uniq items", int(tr.meta[2]))
