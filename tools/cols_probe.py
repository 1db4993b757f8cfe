#!/usr/bin/env python3
"""One rank's share of a column-sharded step on ONE GPU (everything except the wire time of the
all-gather): rank 0 of a G-rank job with a stand-in communicator that fills every slot of the receive
buffer with this rank's own send buffer (same bytes moved by the unpack / loss / scatter kernels).
Also times the thin SpMM launch per width against the d = 64 launch.  Yelp2018 shape, XSimGCL L=3."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import ops, synth  # noqa: E402
from selfrec_amd.data.ui_graph import Interaction  # noqa: E402
from selfrec_amd.dist import ShardedTrainer  # noqa: E402
from selfrec_amd.engine import EpochPrefetcher, FusedTrainer  # noqa: E402


class SelfComm:
    def __init__(self, world):
        self.world, self.rank = world, 0

    def all_gather(self, out, inp):
        out.view(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shape = sys.argv[1] if len(sys.argv) > 1 else "yelp2018"
tu, ti, su, si, U, I = synth.make_dataset(shape)
data = Interaction({}, synth.as_triples(tu, ti), synth.as_triples(su, si))
g = data.device_graph()
N = U + I
print(f"# {shape}: {U} x {I}, {len(tu)} train edges, nnz {g.adj.nnz}")
x64 = torch.randn((N, 64), device="cuda")
y64 = torch.empty_like(x64)
step = torch.tensor([3], dtype=torch.int64, device="cuda")
ep64 = ops.make_epilogue(perturb_eps=0.2, rng_seed=1, rng_step=step, rng_stride=16 * N)
print(f"spmm d=64 (perturb)          {timed(lambda: ops.spmm(g.adj, x64, out=y64, epilogue=ep64)):8.2f} us")
for w in (32, 16, 8):
    xs = x64[:, :w].contiguous()
    ys = torch.empty_like(xs)
    ep = ops.make_epilogue(perturb_eps=0.2, rng_seed=1, rng_step=step, rng_stride=16 * N, d_full=64, col0=0)
    plain = ops.make_epilogue(d_full=64, col0=0)
    t_p = timed(lambda: ops.spmm(g.adj, xs, out=ys, epilogue=ep))
    t_0 = timed(lambda: ops.spmm(g.adj, xs, out=ys, epilogue=plain))
    alg = g.adj.nnz * 8 + (N + 1) * 4 + 2 * N * w * 4
    print(f"spmm thin w={w:2d}  perturb {t_p:8.2f} us   plain {t_0:8.2f} us   algorithmic {alg / 1e6:6.1f} MB -> {alg / t_0 / 1e6:6.2f} TB/s")
if False:
    xs = x64[:, :32].contiguous(); ys = torch.empty_like(xs)
    print(f"spmm d=32 (row-group kernel) plain {timed(lambda: ops.spmm(g.adj, xs, out=ys)):8.2f} us")

if os.environ.get("COLS_PROBE_KERNELS_ONLY"):
    sys.exit(0)
model = os.environ.get("COLS_PROBE_MODEL", "XSimGCL")
kw = dict(model=model, n_layers=3, layer_cl=1, eps=0.2, cl_rate=0.2 if model != "SGL" else 0.1, tau=0.2, drop_rate=0.1,
          batch_size=2048)
print(f"# {model} L=3, B=2048")
print("# TIMING ONLY at world > 1: one rank of G on one GPU, stand-in communicator (same kernels and bytes, NO wire time, and the "
      "all-gather returns copies of this rank's own columns: the printed losses are meaningless there)")
worlds = [int(w) for w in os.environ.get("COLS_PROBE_WORLDS", "1,2,4,8").split(",")]
modes = [m == "graph" for m in os.environ.get("COLS_PROBE_MODES", "graph,eager").split(",")]
for world in worlds:
    for use_graph in modes:
        torch.manual_seed(0)
        if world == 1:
            tr = FusedTrainer(data, 64, use_graph=use_graph, **kw)
        else:
            tr = ShardedTrainer(data, 64, layout="cols", comm=SelfComm(world), use_graph=use_graph, **kw)
        tr.sampler.seed(1)
        pre = EpochPrefetcher(tr)
        pre.start()
        tr.upload_epoch(pre.take())
        for _ in range(30):
            tr.step()
        torch.cuda.synchronize()
        steps = 300
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        losses = tr.read_losses()
        print(f"world {world}  {'graph' if use_graph else 'eager'}  {dt * 1e6:8.1f} us/step (no wire time)  "
              f"-> {2048 / dt / 1e6:6.2f} M pairs/s   losses {tuple(round(v, 4) for v in losses) if world == 1 else '(timing only)'}", flush=True)
        del tr
