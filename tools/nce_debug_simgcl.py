#!/usr/bin/env python3
"""Debug aid: InfoNCE f32 vs split on SimGCL-like inputs (two strongly correlated views) at the Yelp2018 batch shape."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
rng = np.random.default_rng(3)
rows, d = 70000, 64
base = (rng.standard_normal((rows, d)) * 0.05).astype(np.float32)
for corr, tag in ((0.02, "views = base + 2 % noise"), (0.5, "views = base + 50 % noise"), (3.0, "nearly independent views")):
    t1 = torch.from_numpy(base + corr * 0.05 * rng.standard_normal((rows, d)).astype(np.float32)).to(DEV)
    t2 = torch.from_numpy(base + corr * 0.05 * rng.standard_normal((rows, d)).astype(np.float32)).to(DEV)
    for n in (1640, 1877, 2048, 700):
        idx = torch.from_numpy(np.sort(rng.choice(rows, n, replace=False)).astype(np.int32)).to(DEV)
        pad = torch.zeros(2048, dtype=torch.int32, device=DEV); pad[:n] = idx
        nd = torch.tensor([n], dtype=torch.int32, device=DEV)
        res = {}
        for mode in ("split", "f32"):
            g = torch.zeros((rows, d), device=DEV)
            loss = torch.zeros(1, dtype=torch.float64, device=DEV)
            ws = ops.infonce_ws(2048, d, DEV)
            ops.infonce_multi([(t1, t2, pad, 2048, nd, g, g)], d=d, tau=0.2, loss_scale=0.2, loss=loss, ws=ws, precision=mode)
            torch.cuda.synchronize()
            res[mode] = (loss.item(), g.clone())
        a = t1.double()[idx.long()].requires_grad_(True); b = t2.double()[idx.long()].requires_grad_(True)
        an, bn = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(b, dim=1)
        ref = 0.2 * (-(torch.log_softmax(an @ bn.T / 0.2, dim=1).diag()).mean())
        ref.backward()
        gref = torch.zeros((rows, d), dtype=torch.float64, device=DEV)
        gref[idx.long()] = a.grad + b.grad
        for mode in ("split", "f32"):
            l, g = res[mode]
            err = float((g.double() - gref).abs().max() / gref.abs().max())
            print(f"{tag:28s} n={n:5d} {mode:5s} loss {l:.9f} (ref {ref.item():.9f}, rel {abs(l - ref.item()) / abs(ref.item()):.1e})  grad max err / max {err:.2e}")
