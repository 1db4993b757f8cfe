#!/usr/bin/env python3
"""Parameter / embedding error of the fused engine against the reference-run goldens at the BASELINE shapes
(tests/golden/shapes.npz) under both InfoNCE arithmetic modes.  Adam divides by sqrt(v) + 1e-8: where |g| ~ 1e-8 or
less (rows far from the batch), an ABSOLUTE gradient error of 1e-11 becomes a parameter error of 1e-6."""
import json
import os
import random
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from selfrec_amd import ops, synth  # noqa: E402
from selfrec_amd.data.ui_graph import Interaction  # noqa: E402
from tests.test_gpu_shapes import trainer_for  # noqa: E402
from tests.test_shapes_cpu import seeded_init  # noqa: E402

G = os.path.join(REPO, "tests", "golden")
shapes = np.load(os.path.join(G, "shapes.npz"))
meta = json.load(open(os.path.join(G, "shapes_meta.json")))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max()), float(np.abs(a - b).max())


for tag, shape in (("Y_XSimGCL", "yelp2018"), ("F_SGL", "ifashion")):
    tu, ti, su, si, U, I = synth.make_dataset(shape, seed=2024)
    data = Interaction({}, synth.as_triples(tu, ti), [])
    info = meta[tag]
    for mode in ops.NCE_PRECISIONS:
        ue, ie = seeded_init(info)
        tr = trainer_for(info, data, ue, ie)
        tr.set_nce_precision(mode)
        random.seed(info["sampler_seed"])
        tr.seed_sampler_from_python()
        tr.begin_epoch()
        losses = []
        for _ in range(info["n_steps"]):
            tr.step()
            losses.append(tr.read_losses())
        ru = torch.from_numpy(shapes[f"{tag}_rows_user"].astype(np.int64)).cuda()
        ri = torch.from_numpy(shapes[f"{tag}_rows_item"].astype(np.int64)).cuda()
        fu, fi = tr.embeddings()
        nce = shapes[f"{tag}_loss_nce"].reshape(info["n_steps"], -1).sum(1) * tr.cl_rate
        print(f"{tag} [{mode}]: cl rel diff {np.abs(np.asarray(losses)[:, 2] / nce - 1).max():.1e}; "
              f"param user (rel, abs) {rel(tr.user_emb[ru].cpu().numpy(), shapes[f'{tag}_param_user'])} item "
              f"{rel(tr.item_emb[ri].cpu().numpy(), shapes[f'{tag}_param_item'])}; final user "
              f"{rel(fu[ru].cpu().numpy(), shapes[f'{tag}_final_user'])} item {rel(fi[ri].cpu().numpy(), shapes[f'{tag}_final_item'])}")
        del tr
