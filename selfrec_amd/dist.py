"""(e) Multi-GPU: the fused step over the GPUs of one node -- one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI).  Two LAYOUTS of ``engine.FusedTrainer`` (not second engines):

**"cols" -- column-sharded tables (the default whenever d / G is 8, 16, 32, 64 or 128).**  Rank r keeps columns
[r.w, (r+1).w), w = d / G, of every (N, d) table -- parameters, Adam moments, layer outputs, gradient
buffers -- and the whole graph (Yelp2018 shape: 26 MB; the 1 M x 500 k graph: 1 GB of 288 GB).

    forward / backward layers   Y[:, own] = A . X[:, own]: a sparse product is independent per column, so
                                the 2L + 1 products of a step need NO exchange (thin kernel, csrc/spmm.hip)
    perturbation                the unit vector is normalised over the whole row (XSimGCL.py:90): the
                                counter RNG regenerates the other ranks' columns -- hash only, no memory
    losses                      read whole rows, but only the O(batch) rows the staged lists name:
                                pack -> ONE all-gather per step (a few hundred KB per rank) -> compact
                                (5B, d) tables -> the unchanged BPR / InfoNCE kernels (replicated, O(batch))
                                -> each rank scatters its columns of the batch-row gradients (csrc/exchange.hip)
    optimiser                   Adam on the rank's columns
    hipGraph                    two captured graphs per step with the all-gather between them

xGMI is point-to-point (7 links per GPU): a per-layer exchange of (N, d) tables costs 2L + 1 all-gathers of
N.d.4 bytes per step (17.8 MB each at the Yelp2018 shape against a 0.32 ms step); the column layout moves
< 1 MB per step instead, and every rank still streams the (col, val) arrays -- the price: the index stream
is read G times in total, and gathered x rows are 32 .. 128 bytes.

**"rows" -- row-sharded graph and tables (for d / G outside those widths, e.g. 3 ranks).**
Nodes are dealt round-robin (node p -> rank p % G, local row p // G: power-law rows balance
without a partitioner), every (.., d) table is kept in all-gather order, each rank owns one slice of
the parameters, the Adam moments and every layer output, and computes it with the same kernels from
its CSR rows:

    forward layer k    Y_k[own] = A[own, :] . Y_(k-1)           then all-gather Y_k
    backward layer k   H_k[own] = A[own, :] . H_(k+1) + ...      then all-gather H_k   (A is symmetric:
                       the transpose product is the same local row product, so there is no
                       reduce-scatter -- 2L all-gathers of N.d.4 bytes per step, plus one for E0)
    batch level        with whole tables on every rank the O(batch) losses and their gradients are simply
                       recomputed everywhere -- no collective
    optimiser          Adam on the owned rows only
    graph              each rank normalises its own CSR rows on the device (degrees of its rows, one all-gather of
                       the D^-1/2 vector, then the values: data/device_graph.ShardedDeviceGraph), which is also
                       how SGL's edge-dropped views are rebuilt every epoch

Both: the sampler is replicated (same seed, same MT19937 stream => identical batches on every rank) and the
step keeps its device-side cursor.

**"2d:GCxGR"** -- both at once on a Gc x Gr grid (engine.py), for gather-bound graphs.

**"dp" -- data parallel (round 3).**  The layouts above divide ONE batch's work (strong scaling).  At the Yelp2018 shape a
step is six latency-bound launches plus an O(batch) loss section, and no table split makes those shorter.  What the
interconnect can pay for is more batches per step: every rank keeps the whole graph and tables (26 MB + 5 x 17.8 MB at
this shape), runs the single-GPU step -- hipGraph, value-free products, calibrated plan, all of it -- on ITS OWN batch,
and the ranks meet once per step in an all-reduce of the dense gradient before Adam takes the mean: synchronous
data-parallel SGD with a global batch of G x B pairs, the semantics torch's DistributedDataParallel gives the reference's
model file.  bench.py takes it for N > 1 on graphs below GATHER_BOUND_NNZ and says `"scaling": "weak"`.
"""
from __future__ import annotations

import os

import torch.distributed as _dist

from .engine import SLICE_WIDTHS, FusedTrainer, parse_grid, shard_adjacency  # noqa: F401  (public helpers)

# Above this many stored non-zeros a rank's propagation launch is bound by its GATHERS (one 128-byte line per
# non-zero whatever the slice width: DESIGN.md 6.2, measured at 80.5 M), and column blocks narrower than 32
# columns stop paying: "auto" then takes the 2-D grid with 32-column blocks.
GATHER_BOUND_NNZ = 30_000_000


def pick_layout(emb_size: int, world: int, layout: str | None = None, nnz: int | None = None) -> str:
    """"cols" whenever the column slice d / world is a width the SpMM kernels serve, else "rows"; "2d:GCxGR" for
    gather-bound graphs (``nnz`` stored non-zeros of the (N x N) adjacency) once d / world < 32.
    ``layout`` / ``SRH_SHARD_LAYOUT`` = rows | cols | 2d | 2d:GCxGR | auto overrides."""
    layout = (layout or os.environ.get("SRH_SHARD_LAYOUT") or "auto").lower()
    if layout in ("rows", "cols", "dp"):
        return layout
    if layout.startswith("2d"):
        gc, gr = parse_grid(layout, world, emb_size)
        return f"2d:{gc}x{gr}"
    if layout != "auto":
        raise ValueError(f"shard layout {layout!r}: rows, cols, 2d[:GCxGR], dp or auto")
    # (a single rank has nothing to split: "auto" keeps it on the row layout's code path, "cols" can still be asked for)
    cols_ok = world > 1 and emb_size % world == 0 and emb_size // world in SLICE_WIDTHS
    if cols_ok and nnz is not None and nnz >= GATHER_BOUND_NNZ and emb_size // world < 32 and world % 2 == 0:
        gc = max(1, emb_size // 32)
        if world % gc == 0 and gc < world:
            return f"2d:{gc}x{world // gc}"
    return "cols" if cols_ok else "rows"


def describe_layout(emb_size: int, world: int, layout: str | None = None, nnz: int | None = None) -> str:
    """One line for bench.py's ``config.parallelism``: the layout this world size takes and what it exchanges."""
    lay = pick_layout(emb_size, world, layout, nnz)
    if lay == "dp":
        return (f"data parallel x{world}: whole graph + tables on every rank, each rank its own batches, one all-reduce of "
                f"the dense gradient (N x d floats) per step, Adam on the mean gradient")
    if lay == "cols":
        return (f"column-sharded tables x{world} (w = {emb_size // max(world, 1)} columns per rank, graph replicated, "
                f"one all-gather of the batch rows per step)")
    if lay == "rows":
        return f"row-sharded graph + tables x{world} (an all-gather of (N, d) after every product)"
    gc, gr = parse_grid(lay, world, emb_size)
    return (f"2-D grid x{world}: {gc} column blocks (w = {emb_size // gc}) x {gr} row parts (per product: an all-gather "
            f"of (N, w) over the {gr} ranks of a column block; per step: one all-gather of the batch rows over the "
            f"{gc} ranks of a row part)")


class ShardedTrainer(FusedTrainer):
    """``FusedTrainer`` over the default process group (all five models).  Same constructor, same
    ``begin_epoch / step / read_losses / embeddings``; every rank must be seeded identically."""

    def __init__(self, data, emb_size, layout=None, **kw):
        kw.pop("backend", None)
        comm = kw.get("comm")
        if isinstance(comm, tuple):                      # (2-D stand-in communicators: (batch-row comm, table-row comm))
            world = comm[0].world * comm[1].world
        else:
            world = comm.world if comm is not None else _dist.get_world_size()
        nnz = 2 * int(data.interaction_mat.nnz) if hasattr(data, "interaction_mat") else None
        super().__init__(data, emb_size, shard=pick_layout(int(emb_size), int(world), layout, nnz), **kw)

    def parameters_full(self):
        return self.user_emb, self.item_emb


def deal_users(user_ids, rank: int, world: int):
    """Evaluation over N ranks: test users are dealt round-robin -- rank r ranks users r, r + N, ...
    against the replicated item table.  Returns (this rank's users, padded share size)."""
    mine = user_ids[rank::world]
    return mine, (len(user_ids) + world - 1) // world


def gather_ranked(ids_local, n_users: int, rank: int, world: int, device, all_gather=None):
    """Reassemble the (n_users, k) ranked-id table from every rank's share (the inverse of ``deal_users``): one
    all-gather of the padded shares, then user u's row is row u // N of rank u % N's share.  ``all_gather(out, inp)``
    defaults to torch.distributed's; returns an int32 tensor on ``device``."""
    import numpy as np
    import torch
    k = int(ids_local.shape[1])
    n_max = (n_users + world - 1) // world
    pad = torch.full((n_max, k), -1, dtype=torch.int32, device=device)
    if len(ids_local):
        pad[:len(ids_local)] = torch.as_tensor(np.ascontiguousarray(ids_local), dtype=torch.int32).to(device)
    everyone = torch.empty((world * n_max, k), dtype=torch.int32, device=device)
    (all_gather or _dist.all_gather_into_tensor)(everyone, pad)
    users = torch.arange(n_users, device=device)
    return everyone[(users % world) * n_max + users // world]
