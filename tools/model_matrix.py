#!/usr/bin/env python3
"""Step time of every engine-backed model at the BASELINE.json shapes (1 GPU), plus the scoring GEMM
and top-K kernels on their own.  Prints one line per case; used for the tables in DESIGN.md."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import ops, synth  # noqa: E402
from selfrec_amd.data.ui_graph import Interaction  # noqa: E402
from selfrec_amd.engine import EpochPrefetcher, FusedTrainer  # noqa: E402

CASES = [("yelp2018", "MF", {}), ("yelp2018", "LightGCN", dict(n_layers=3)),
         ("yelp2018", "XSimGCL", dict(n_layers=3, layer_cl=1, eps=0.2, cl_rate=0.2, tau=0.2)),
         ("yelp2018", "SimGCL", dict(n_layers=3, eps=0.1, cl_rate=0.5)),
         ("yelp2018", "SGL", dict(n_layers=3, drop_rate=0.1, cl_rate=0.1, tau=0.2)),
         ("ifashion", "SGL", dict(n_layers=3, drop_rate=0.1, cl_rate=0.1, tau=0.2)),
         ("ifashion", "XSimGCL", dict(n_layers=3, layer_cl=1, eps=0.2, cl_rate=0.2, tau=0.2))]
cache = {}
for shape, model, kw in CASES:
    if shape not in cache:
        t0 = time.time()
        tu, ti, su, si, U, I = synth.make_dataset(shape)
        cache[shape] = Interaction({}, synth.as_triples(tu, ti), synth.as_triples(su, si))
        print(f"# {shape}: {U} x {I}, {len(tu)} train edges, built in {time.time() - t0:.1f} s", flush=True)
    data = cache[shape]
    torch.manual_seed(0)
    tr = FusedTrainer(data, 64, model=model, batch_size=2048, use_graph=True, **kw)
    tr.sampler.seed(1)
    pre = EpochPrefetcher(tr)
    pre.start()
    tr.upload_epoch(pre.take())
    steps = min(300, tr.epoch_batches - 40)
    for _ in range(30):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    losses = tr.read_losses()
    ok = all(x == x and abs(x) < 1e6 for x in losses) and bool(torch.isfinite(tr.E0).all())
    print(f"{shape:9s} {model:8s} L={tr.L}  {dt * 1e3:7.3f} ms/step  {2048 / dt / 1e6:6.2f} M pairs/s  losses={tuple(round(x, 5) for x in losses)} finite={ok}", flush=True)
    del tr

# evaluation kernels alone (Yelp shape)
data = cache["yelp2018"]
g = data.device_graph()
d = 64
ue = torch.randn((data.user_num, d), device="cuda")
ie = torch.randn((data.item_num, d), device="cuda")
for m in (2048, 8192):
    out = torch.empty((m, data.item_num), device="cuda")
    for fn, name, flops in ((lambda: ops.gemm_nt(ue[:m], ie, out=out), "gemm_nt (fp32 MFMA)", 2.0 * m * data.item_num * d),
                            (lambda: ops.topk_rows(out, 20), "topk_rows k=20", 0.0)):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        extra = f"{flops / us / 1e6:7.1f} TFLOP/s" if flops else f"{m * data.item_num * 4 / us / 1e6:7.2f} TB/s read"
        print(f"eval {name:22s} m={m:5d} n={data.item_num}: {us:8.1f} us  {extra}", flush=True)
