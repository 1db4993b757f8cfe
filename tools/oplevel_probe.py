"""Where does an op-level model's step go?  Runs N training steps of BUIR / MixGCF / DirectAU / SelfCF on the synthetic
Yelp2018-shaped graph with a device synchronisation after every statement of interest and prints the mean time of each
(GPU box only; tools/gpu_session.sh stage ``oplevel``).

    python tools/oplevel_probe.py BUIR 40
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from selfrec_amd import synth                                              # noqa: E402
from selfrec_amd.data.loader import FileIO                                 # noqa: E402,F401
from selfrec_amd.util import torch_rng                                     # noqa: E402
from selfrec_amd.util.conf import ModelConf                                # noqa: E402

ATEN_RAND = bool(int(__import__("os").environ.get("PROBE_ATEN_RAND", "0")))


def build(model_name):
    import importlib
    import os
    conf = ModelConf(f"./conf/{model_name}.yaml")
    if not os.path.exists(conf["training.set"]):
        tu, ti, su, si, _, _ = synth.make_dataset("yelp2018")
        os.makedirs(os.path.dirname(conf["training.set"]) or ".", exist_ok=True)
        synth.write_text(conf["training.set"], tu, ti)
        synth.write_text(conf["test.set"], su, si)
    train = FileIO.load_data_set(conf["training.set"], conf["model"]["type"])
    test = FileIO.load_data_set(conf["test.set"], conf["model"]["type"])
    cls = getattr(importlib.import_module(f"selfrec_amd.model.graph.{model_name}"), model_name)
    return cls(conf, train, test)


class Clock:
    def __init__(self):
        self.t, self.acc, self.n = time.perf_counter(), {}, {}

    def lap(self, name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.acc[name] = self.acc.get(name, 0.0) + now - self.t
        self.n[name] = self.n.get(name, 0) + 1
        self.t = time.perf_counter()

    def report(self, steps):
        for k, v in self.acc.items():
            print(f"  {k:34s} {1e3 * v / steps:9.3f} ms/step  ({self.n[k] // steps} per step)")
        print(f"  {'total':34s} {1e3 * sum(self.acc.values()) / steps:9.3f} ms/step")


def probe_buir(rec, steps):
    from selfrec_amd.util.sampler import next_batch_pairwise
    model = rec.model.cuda()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=rec.lRate)
    ck = Clock()

    def enc_forward(enc, inputs):
        rate = np.random.random() * enc.drop_ratio
        if ATEN_RAND:
            keep = torch.floor(1 - rate + torch.rand(enc.sparse_norm_adj._nnz())).type(torch.bool)
        else:
            keep = torch_rng.keep_mask(enc.sparse_norm_adj._nnz(), 1 - rate)
        ck.lap("keep mask (host: %s)" % ("ATen torch.rand" if ATEN_RAND else "util/torch_rng replay"))
        adj = enc.sparse_norm_adj.dropout(keep, 1.0 / (1 - rate))
        ck.lap("SparseAdjHandle.dropout")
        hops = enc.hops(adj)
        ck.lap("hops (SpMM)")
        mean = torch.stack(hops, dim=1).mean(dim=1)
        ck.lap("stack + mean")
        out = mean[:enc.data.user_num][inputs["user"]], mean[enc.data.user_num:][inputs["item"]]
        ck.lap("row gathers")
        return out

    it = next_batch_pairwise(rec.data, rec.batch_size, 1, as_arrays=True)
    for step in range(steps + 3):
        if step == 3:
            ck = Clock()
        u, i, _ = (torch.from_numpy(a).cuda() for a in next(it))
        ck.lap("sampler + H2D")
        inputs = {"user": u, "item": i}
        u_on, i_on = enc_forward(model.online_encoder, inputs)
        u_tg, i_tg = enc_forward(model.target_encoder, inputs)
        loss = model.get_loss((model.predictor(u_on), u_tg, model.predictor(i_on), i_tg))
        ck.lap("predictor + loss")
        opt.zero_grad()
        loss.backward()
        ck.lap("backward")
        opt.step()
        ck.lap("Adam")
        model.update_target(u, i)
        ck.lap("update_target")
    ck.report(steps)


def probe_generic(rec, steps):
    from selfrec_amd.util.sampler import next_batch_pairwise
    model = rec.model.cuda()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=rec.lRate)
    ck = Clock()
    it = next_batch_pairwise(rec.data, rec.batch_size, rec.n_negs, as_arrays=True)
    for step in range(steps + 3):
        if step == 3:
            ck = Clock()
        u, i, j = (torch.from_numpy(a).cuda() for a in next(it))
        ck.lap("sampler + H2D")
        loss = rec.batch_loss(u, i, j)
        ck.lap("batch_loss (forward)")
        opt.zero_grad()
        loss.backward()
        ck.lap("backward")
        opt.step()
        ck.lap("Adam")
        rec.after_step(u, i, j)
        ck.lap("after_step")
    ck.report(steps)


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "BUIR"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rec = build(name)
    print(f"{name}: {steps} steps, every statement synchronised")
    (probe_buir if name == "BUIR" else probe_generic)(rec, steps)
