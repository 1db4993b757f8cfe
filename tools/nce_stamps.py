#!/usr/bin/env python3
"""Where the all-f32 InfoNCE passes spend their time, per wave, on the shader clock -- needs the laboratory build of the
library (ALT_SRC=losses tools/spmm_lab/build_alt.sh stamps "-DSRH_NCEF32_STAMPS", copied over selfrec_amd/lib/):
every wave of nce_tile_f32 leaves {start, plan made, first chunk landed, key loop done, stores done} + its block count."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd import _lib, ops  # noqa: E402
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
nuu, nui = tr.meta[1:2], tr.meta[2:3]
F, CL = tr.F, tr.Y[0]
problems = [(F, CL, st["uniq_u"], B, nuu, tr.gF, tr.gCL), (F, CL, st["uniq_i"], B, nui, tr.gF, tr.gCL)]
lib = C.CDLL(_lib.LIB_PATH)
for rep in range(3):
    ops.infonce_multi(problems, d=64, tau=0.2, loss_scale=0.2, loss=tr.losses[2:3], ws=tr.nce_ws, precision="f32")
torch.cuda.synchronize()
buf = np.zeros((2, 512, 8, 8), dtype=np.uint64)
assert lib.srh_debug_nce_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
print("uniq users", int(tr.meta[1]), "uniq items", int(tr.meta[2]))
for p in (0, 1):
    s = buf[p].astype(np.int64)
    live = s[:, :, 5] > 0                          # waves that had a task
    t0 = s[:, :, 0][live].min()
    rel = lambda k: (s[:, :, k][live] - t0)        # noqa: E731
    print(f"pass {p + 1}: {int(live.sum())} waves with a task, {len(np.unique(s[:, :, 6][live]))} tasks; blocks per task: "
          f"{np.bincount(s[:, 0, 5][live[:, 0]].astype(int)).tolist()}")
    for k, name in ((0, "start"), (1, "plan made"), (2, "first chunk landed"), (3, "key loop done"), (4, "stores done")):
        v = rel(k)
        v = v[s[:, :, k][live] > 0]
        print(f"   {name:20s} min {v.min():8d}  median {int(np.median(v)):8d}  max {v.max():8d}   cycles after the first wave's start")
    loop = (s[:, :, 3] - s[:, :, 2])[live]
    nb = s[:, :, 5][live]
    print(f"   key loop: {np.median(loop / nb):.0f} cycles per block (median over waves; 2 waves per SIMD -> 4096 = matrix-pipe bound)")
