#!/usr/bin/env python3
"""Diagnostic for tests/test_gpu_shapes.py::test_1m_500k_xsimgcl_step_matches_reference_run: where do the post-Adam
parameters of the fused step differ from the reference run's, and what gradient do the two sides imply there?
(first Adam step: delta = -lr g / (|g| + eps)  =>  |g| = eps |t| / (1 - |t|), t = delta / lr)"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_shapes as T  # noqa: E402
from test_shapes_cpu import seeded_init  # noqa: E402
from selfrec_amd import synth  # noqa: E402
from selfrec_amd.data.ui_graph import Interaction  # noqa: E402
import json, random  # noqa: E402

shapes = np.load(os.path.join(REPO, "tests", "golden", "shapes.npz"))
smeta = json.load(open(os.path.join(REPO, "tests", "golden", "shapes_meta.json")))
info = smeta["B_XSimGCL"]
tu, ti, su, si, U, I = synth.make_dataset("1m-500k", seed=2024)
data = Interaction.from_id_arrays({}, T.first_appearance_ids(tu), T.first_appearance_ids(ti), np.zeros(0, np.int64),
                                  np.zeros(0, np.int64), U, I)
ue, ie = seeded_init(info)
kept = {}
for precision in ("bf16x3", "with-values"):
    tr = T.trainer_for(info, data, ue, ie)
    if precision == "with-values":
        tr.vfree = False          # same arithmetic, different rounding order: pattern products + row scaling -> values
    random.seed(info["sampler_seed"])
    tr.seed_sampler_from_python()
    tr.begin_epoch()
    tr.step()
    print(precision, "losses", tr.read_losses(), "want", shapes["B_XSimGCL_loss_bpr"], shapes["B_XSimGCL_loss_reg"], shapes["B_XSimGCL_loss_nce"])
    lr, eps = info["lr"], 1e-8
    for side, emb, init in (("user", tr.user_emb, ue), ("item", tr.item_emb, ie)):
        rows = shapes[f"B_XSimGCL_rows_{side}"].astype(np.int64)
        got = emb[torch.from_numpy(rows).to(emb.device)].cpu().numpy().astype(np.float64)
        if side in kept:
            d2 = np.abs(got - kept[side])
            print(f"{side}: THIS RUN vs THE ENGINE'S OWN value-free run: max |diff| {d2.max():.3e}; > 1e-6: {(d2 > 1e-6).sum()}; "
                  f"> 1e-5: {(d2 > 1e-5).sum()}; columns of the 12 largest: {sorted(set((np.argsort(d2.ravel())[::-1][:12] % d2.shape[1]).tolist()))}")
        else:
            kept[side] = got
        want = shapes[f"B_XSimGCL_param_{side}"].astype(np.float64)
        ini = np.asarray(init)[rows].astype(np.float64)
        tw, tg = (want - ini) / lr, (got - ini) / lr
        diff = np.abs(got - want)
        deg = np.diff(np.asarray(data.interaction_mat.indptr if side == "user" else data.interaction_mat.T.tocsr().indptr))[rows]
        gw = eps * np.abs(tw) / np.maximum(1 - np.abs(tw), 1e-12)
        gg = eps * np.abs(tg) / np.maximum(1 - np.abs(tg), 1e-12)
        print(f"{side}: max |diff| {diff.max():.3e}; elements with |diff| > 1e-6: {(diff > 1e-6).sum()} of {diff.size}; > 1e-5: {(diff > 1e-5).sum()}")
        print("  quantiles of |t_want| (|delta| / lr):", np.quantile(np.abs(tw), [0.01, 0.1, 0.5, 0.9, 0.99]).round(4))
        order = np.argsort(diff.ravel())[::-1][:12]
        for k in order:
            r, c = divmod(int(k), diff.shape[1])
            print(f"  row {rows[r]:>8} deg {deg[r]:>5} col {c:>3}: t_want {tw[r, c]:+.5f} t_got {tg[r, c]:+.5f}  implied |g| want {gw[r, c]:.3e} got {gg[r, c]:.3e}"
                  f"  rel {abs(gg[r, c] - gw[r, c]) / max(gw[r, c], 1e-30):.2e}")
        # how the relative gradient error distributes over |g|
        rel = np.abs(gg - gw) / np.maximum(gw, 1e-30)
        for lo, hi in ((0, 1e-10), (1e-10, 1e-9), (1e-9, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6)):
            m = (gw >= lo) & (gw < hi)
            if m.any():
                print(f"  implied |g| in [{lo:.0e}, {hi:.0e}): {m.sum():>6} elements, median rel diff {np.median(rel[m]):.2e}, max {rel[m].max():.2e}")
    del tr
