"""Throughput of the drop-in tiers: a model file written the reference's way, and the reference's own file fused."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3


def dropin_throughput(args, raw, steps=60, warmup=8):
    """pairs/s of the OP-LEVEL drop-in tier: what a SELFRec user gets without switching to selfrec_amd's model
    classes -- a model file written the reference's way (torch.sparse.mm on the uploaded adjacency, rand_like /
    normalize / sign perturbation, stack + mean, fancy-index gathers, util.loss_torch losses, torch.optim.Adam,
    the next_batch_pairwise generator pulled synchronously: XSimGCL.py:23-50,83-101) running on this package's
    sampler, SpMM handle and loss kernels.  The unmodified reference files were run the same way in a gpurun
    session (profiles/r02_a_dropin_reference_models.txt); /root/reference does not exist where bench.py runs."""
    import random
    import torch.nn.functional as F
    from selfrec_amd import dropin, synth
    dropin.install(fuse=False)         # the mirrors under the reference's module names + the host-side fast paths (util/fastpath.py)
    from selfrec_amd.base.torch_interface import TorchGraphInterface
    from selfrec_amd.data.ui_graph import Interaction
    from selfrec_amd.util.loss_torch import InfoNCE, bpr_loss, l2_reg_loss
    from selfrec_amd.util.sampler import next_batch_pairwise
    tu, ti, su, si, U, I = raw
    data = Interaction({}, synth.as_triples(tu, ti), [])
    adj = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj).cuda()
    torch.manual_seed(args.seed)
    emb = torch.nn.ParameterDict({
        "user_emb": torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(U, args.emb))),
        "item_emb": torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(I, args.emb)))}).cuda()
    opt = torch.optim.Adam(emb.parameters(), lr=1e-3)
    eps, lam, tau, l_star, reg = 0.2, 0.2, args.tau, 1, 1e-4

    def encode(perturbed):
        h = torch.cat([emb["user_emb"], emb["item_emb"]], 0)
        layers, view = [], None
        for k in range(args.layers):
            h = torch.sparse.mm(adj, h)
            if perturbed:
                h = h + torch.sign(h) * F.normalize(torch.rand_like(h), dim=-1) * eps
            layers.append(h)
            if k == l_star - 1:
                view = h
        out = torch.stack(layers, dim=1).mean(dim=1)
        return torch.split(out, [U, I]) + torch.split(view, [U, I])

    random.seed(args.seed)
    done, t0 = 0, None
    for u_idx, i_idx, j_idx in next_batch_pairwise(data, args.batch):
        if done == warmup:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        ue, ie, cu, ci = encode(True)
        u, p, n = ue[u_idx], ie[i_idx], ie[j_idx]
        uu = torch.unique(torch.Tensor(u_idx).type(torch.long)).cuda()          # (XSimGCL.py:46-47, as the file spells it)
        ui = torch.unique(torch.Tensor(i_idx).type(torch.long)).cuda()
        cl = InfoNCE(ue[uu], cu[uu], tau) + InfoNCE(ie[ui], ci[ui], tau)
        loss = bpr_loss(u, p, n) + l2_reg_loss(reg, u, p) + lam * cl
        opt.zero_grad(); loss.backward(); opt.step()
        done += 1
        if done == warmup + steps:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dropin.uninstall()
    return {"pairs_per_s": round(steps * args.batch / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "what": "XSimGCL written against SELFRec's API (raw torch.sparse.mm on the handle, torch autograd, torch.optim.Adam, "
                    "python generator sampler pulled synchronously, table[list] gathers, torch.unique(torch.Tensor(list))) under "
                    "dropin.install(): HIP SpMM / loss / sampler / Adam kernels underneath, ids uploaded once per batch",
            "final_loss": float(loss.item())}


def dropin_fused_throughput(args, raw, epochs=3):
    """pairs/s of the THIRD tier: the reference's own, unmodified model/graph/XSimGCL.py with dropin.install(fuse=True)
    -- the file's SHA-256 is checked and its train() is served by engine.FusedTrainer (selfrec_amd/dropin.py).  Needs
    the reference's model/ directory: it is staged untracked under _refstage/ for a GPU session (reference sources are
    never committed; /root/reference does not exist on the bench box) -- without it this returns the committed
    measurement's location instead of a number."""
    import importlib
    import random
    import tempfile
    stage = os.path.join(REPO, "_refstage")
    if not os.path.isfile(os.path.join(stage, "model", "graph", "XSimGCL.py")):
        return {"pairs_per_s": None, "note": "no staged reference checkout on this box (_refstage/model/graph/XSimGCL.py); "
                "measured in a gpurun session: profiles/r03_b_dropin_fused_reference_models.txt (6.31 M pairs/s over 5 epochs)"}
    from selfrec_amd import dropin, synth
    from selfrec_amd.util.conf import ModelConf
    tu, ti, su, si, U, I = raw
    dropin.install(fuse=True)
    sys.path.insert(0, stage)
    cwd = os.getcwd()
    try:
        mod = importlib.import_module("model.graph.XSimGCL")
        if "XSimGCL" not in dropin._state["fused"]:
            return {"pairs_per_s": None, "note": "the staged XSimGCL.py is not byte-for-byte the reference's: not fused"}
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            conf = ModelConf({"model": {"name": "XSimGCL", "type": "graph"}, "item.ranking.topN": [10, 20],
                              "embedding.size": args.emb, "max.epoch": epochs, "batch.size": args.batch, "learning.rate": 0.001,
                              "reg.lambda": 0.0001, "output": "./results/", "training.set": "x", "test.set": "y",
                              "XSimGCL": {"n_layer": args.layers, "l_star": 1, "lambda": 0.2, "eps": 0.2, "tau": args.tau}})
            torch.manual_seed(args.seed); random.seed(args.seed)
            model = mod.XSimGCL(conf, [list(t) for t in synth.as_triples(tu, ti)], [list(t) for t in synth.as_triples(su, si)])
            t_eval = [0.0]
            real_eval = model.fast_evaluation

            def timed_eval(epoch):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = real_eval(epoch)
                torch.cuda.synchronize(); t_eval[0] += time.perf_counter() - t0
                return r
            model.fast_evaluation = timed_eval
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.train()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0 - t_eval[0]
    finally:
        os.chdir(cwd)
        sys.path.remove(stage)
        dropin.uninstall()
    return {"pairs_per_s": round(epochs * len(tu) / dt, 1), "epochs": epochs, "seconds": round(dt, 3),
            "fast_evaluation_seconds": round(t_eval[0], 3),
            "what": "model/graph/XSimGCL.py of the reference, unmodified (SHA-256 checked), dropin.install(fuse=True): "
                    "train() on engine.FusedTrainer; engine construction, XCD calibration, graph capture, sampling included"}
