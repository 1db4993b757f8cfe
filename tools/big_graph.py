#!/usr/bin/env python3
"""BASELINE.json config 4 on ONE MI355X: XSimGCL, synthetic 1 M users x 500 k items, d=128, E = 50 M
interactions (avg user degree 50; the config leaves E open).  The config is specified for 8 GPUs with the
table sharded; on 288 GB it also fits one GPU, which is what this measures -- followed by ONE rank's share of
the column-sharded step at 4 and 8 ranks (rank 0 with a stand-in communicator: same kernels and bytes as on
an 8-GPU node, no wire time; BIG_COLS_WORLDS=4,8) and of the 2-D grid's step (BIG_2D=4x2: 4 column blocks x 2 row parts,
the layout this shape takes on 8 GPUs -- DESIGN.md 6.2; the row all-gathers are device copies of the same size)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import synth  # noqa: E402
from selfrec_amd.data.ui_graph import Interaction  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

E = int(os.environ.get("BIG_EDGES", 50_000_000))
t0 = time.time()
tu, ti, su, si, U, I = synth.make_dataset("1m-500k", n_edges=E)
print(f"generated {U} x {I}, {len(tu)} train / {len(su)} test edges in {time.time() - t0:.0f} s", flush=True)
t0 = time.time()
data = Interaction.from_id_arrays({}, tu, ti, su[:1000], si[:1000], U, I)
SINGLE = not os.environ.get("BIG_SKIP_SINGLE")
host = None
if SINGLE:
    tr = FusedTrainer(data, 128, model="XSimGCL", n_layers=3, layer_cl=1, eps=0.2, cl_rate=0.2, tau=0.2, batch_size=2048,
                      use_graph=True)
    torch.cuda.synchronize()
    print(f"device graph + plan + trainer in {time.time() - t0:.0f} s; HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    tr.sampler.seed(1)
    t0 = time.time()
    host = tr.sample_epoch_host()
    print(f"sampled one epoch ({len(tu)} pairs) in {time.time() - t0:.1f} s", flush=True)
    tr.upload_epoch(host)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"XSimGCL 1M x 500k d=128 L=3: {dt * 1e3:.2f} ms/step  {2048 / dt / 1e3:.1f} k pairs/s  losses={tr.read_losses()} "
          f"finite={bool(torch.isfinite(tr.E0).all())}", flush=True)

    del tr
    torch.cuda.empty_cache()


class SelfComm:
    def __init__(self, world):
        self.world, self.rank = world, 0

    def all_gather(self, out, inp):
        out.view(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))


class SelfRows:
    """table-row all-gather of one column block with the partners' parts stood in by this rank's own (same bytes)"""

    def __init__(self, world):
        self.world, self.rank = world, 0

    def all_gather(self, out, inp):
        flat, n = out.view(-1), inp.numel()
        for r in range(1, self.world):
            flat[r * n:(r + 1) * n].copy_(inp.reshape(-1))


from selfrec_amd.dist import ShardedTrainer  # noqa: E402
steps = 20
for world in [int(w) for w in os.environ.get("BIG_COLS_WORLDS", "4,8").split(",") if w]:
    t0 = time.time()
    tr = ShardedTrainer(data, 128, layout="cols", comm=SelfComm(world), model="XSimGCL", n_layers=3, layer_cl=1, eps=0.2,
                        cl_rate=0.2, tau=0.2, batch_size=2048, use_graph=True)
    tr.sampler.seed(1)
    tr.upload_epoch(host if host is not None and not tr.sharded else tr.sample_epoch_host())
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"column-sharded, rank 0 of {world} (w = {tr.w} columns, no wire time): {dt * 1e3:.2f} ms/step  "
          f"{2048 / dt / 1e3:.1f} k pairs/s  losses={tr.read_losses()} finite={bool(torch.isfinite(tr.E0).all())}  "
          f"HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    del tr
    torch.cuda.empty_cache()

for grid in [g for g in os.environ.get("BIG_2D", "4x2").split(",") if g]:
    gc, gr = (int(v) for v in grid.split("x"))
    tr = ShardedTrainer(data, 128, layout=f"2d:{gc}x{gr}", comm=(SelfComm(gc), SelfRows(gr)), model="XSimGCL", n_layers=3,
                        layer_cl=1, eps=0.2, cl_rate=0.2, tau=0.2, batch_size=2048, use_graph=False)
    tr.sampler.seed(1)
    tr.upload_epoch(tr.sample_epoch_host())
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    slab = tr.n_pad * tr.w * 4
    # wire-time model of the 7 exchanges (2L products + E0): direct = one xGMI link carries the whole slab per partner;
    # two-hop = every link carries 2 / G of it per partner (engine.TwoHopRows); ~75 GB/s per link and direction
    direct_ms = (2 * tr.L + 1) * (gr - 1) * slab / 75e9 * 1e3
    twohop_ms = (2 * tr.L + 1) * (gr - 1) * 2 * slab / (gc * gr) / 75e9 * 1e3
    print(f"2-D grid {gc} x {gr}, rank 0 of {gc * gr} (w = {tr.w} columns, {tr.n_pad} of {tr.N} rows, {tr.adj.nnz} non-zeros; "
          f"row exchanges as local copies of {slab / 2**20:.0f} MiB, no wire time): {dt * 1e3:.2f} ms/step  "
          f"{2048 / dt / 1e3:.1f} k pairs/s  losses={tr.read_losses()} finite={bool(torch.isfinite(tr.E0).all())}  "
          f"HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB;  modelled wire time of the "
          f"{2 * tr.L + 1} row exchanges: direct {direct_ms:.2f} ms, two-hop {twohop_ms:.2f} ms", flush=True)
    del tr
    torch.cuda.empty_cache()
