"""Where the tables of the fused step live, and what a step exchanges: the multi-GPU layouts as COMPONENTS of a placement.

``engine.FusedTrainer`` computes the step (encoder passes, losses, backward chain, optimiser) against three small
interfaces and never asks which layout it runs in:

    rows      who owns which ROWS of the graph and of every (.., d) table        RowsWhole | RowParts
              (node -> table row, this rank's slice, the all-gather that makes a freshly written table whole again,
              the graph object whose CSR rows the products run on)
    colx      which COLUMNS of the tables this rank keeps                          ColumnsWhole | ColumnBlocks
              (width, offset, and the step's one batch-row exchange: pack -> all-gather -> unpack -> scatter)
    sync      what happens to the dense gradient between backward and Adam        NoGradSync | GradientAllReduce

A layout is a choice of the three (``make_placement``):

    single    RowsWhole  + ColumnsWhole  + NoGradSync          one GPU: the step of DESIGN.md section 5
    dp        RowsWhole  + ColumnsWhole  + GradientAllReduce   every rank its own batches, one all-reduce per step (6.4)
    cols      RowsWhole  + ColumnBlocks  + NoGradSync          columns of every table split, graph replicated (6.1)
    rows      RowParts   + ColumnsWhole  + NoGradSync          rows dealt round-robin, an all-gather per product (6.2)
    2d:GCxGR  RowParts   + ColumnBlocks  + NoGradSync          both, on a Gc x Gr grid of ranks (6.3)

``Placement`` also answers the questions that depend on the combination (may the step use the single-GPU-only kernels --
value-free products, the fetch rider, the fused Adam reset, the calibrated plan; how is it captured; which phases does a
step have).  Communicators are ``comm.TorchComm`` objects or test stand-ins with the same two methods."""
from __future__ import annotations

import os

import numpy as np
import torch

from ._lib import SelfrecHipError

SLICE_WIDTHS = (8, 16, 32, 64, 128)   # column slices the SpMM kernels serve (csrc/spmm.hip: pair, slice<4/8>, rows<16/32>)


# ---------------------------------------------------------------------------------------------------------------------
# rows
# ---------------------------------------------------------------------------------------------------------------------
class RowsWhole:
    """Every rank holds every row: node p lives at table row p."""
    parts, part, comm = 1, 0, None
    dealt = False

    def bind(self, n_nodes, n_users, dev):
        self.N, self.U = int(n_nodes), int(n_users)
        self.n_pad = self.P = self.N
        self.pos = np.arange(self.N, dtype=np.int32)
        self.pos_dev = torch.arange(self.N, dtype=torch.int64, device=dev)
        self.own = slice(0, self.P)
        return self

    def build_graph(self, data, dev, column_classes):
        return data.device_graph(dev, column_classes=column_classes)

    def mine(self, t):
        """the rows of a table this rank computes (the whole table)"""
        return t

    def make_whole(self, t):
        """a table whose owned rows were just written is whole again on every rank (nothing to do)"""

    def rows_of_users(self, t):
        return t[:self.U]

    def rows_of_items(self, t):
        return t[self.U:]

    def epoch_to_table_rows(self, ep):
        for k in ("i", "j", "uniq_i"):                  # items follow the users
            ep[k] += self.U
        return ep

    def segment_row_offsets(self):
        """(user_row0, item_row0) of the row -> slot lists (srh_sampler_epoch_segments): ids -> table rows."""
        return 0, self.U

    def epoch_node_ids(self, host, n_rows):
        return host["u"], host["i"] - self.U, host["j"] - self.U

    def place_noise(self, t, d):
        return t.contiguous()

    def rng_row_offset(self):
        return 0


class RowParts(RowsWhole):
    """Nodes dealt round-robin over ``parts`` ranks (node p -> rank p % parts, local row p // parts: power-law rows balance
    without a partitioner); every table is kept in all-gather order (row = owner * n_pad + local row), so
    ``all_gather_into_tensor`` of the owners' slices IS the table."""
    dealt = True

    def __init__(self, comm):
        self.comm = comm
        self.parts, self.part = int(comm.world), int(comm.rank)

    def bind(self, n_nodes, n_users, dev):
        self.N, self.U = int(n_nodes), int(n_users)
        G = self.parts
        self.n_pad = (self.N + G - 1) // G
        self.P = G * self.n_pad
        nodes = np.arange(self.N, dtype=np.int64)
        self.pos = ((nodes % G) * self.n_pad + nodes // G).astype(np.int32)
        self.pos_dev = torch.from_numpy(self.pos.astype(np.int64)).to(dev)
        self.own = slice(self.part * self.n_pad, (self.part + 1) * self.n_pad)
        return self

    def build_graph(self, data, dev, column_classes):
        from .data import device_graph as _dg
        return _dg.ShardedDeviceGraph(data.interaction_mat, self.part, self.parts, dev, self.make_whole)

    def mine(self, t):
        return t if t is None else t[self.own]

    def make_whole(self, t):
        mine = t[self.own]
        self.comm.all_gather(t, mine if t.device.type == "cuda" else mine.clone())

    def rows_of_users(self, t):
        return t[self.pos_dev[:self.U]]

    def rows_of_items(self, t):
        return t[self.pos_dev[self.U:]]

    def epoch_to_table_rows(self, ep):
        pos_u, pos_i = self.pos[:self.U], self.pos[self.U:]
        for k, table in (("u", pos_u), ("i", pos_i), ("j", pos_i), ("uniq_u", pos_u), ("uniq_i", pos_i)):
            ep[k] = table[ep[k]]
        return ep

    def epoch_node_ids(self, host, n_rows):
        node_of_row = np.full(self.P, -1, dtype=np.int64)
        node_of_row[self.pos] = np.arange(self.N)
        return node_of_row[host["u"]], node_of_row[host["i"]] - self.U, node_of_row[host["j"]] - self.U

    def place_noise(self, t, d):
        full = torch.zeros((self.P, d), dtype=torch.float32, device=t.device)
        full[self.pos_dev] = t
        return full[self.own].contiguous()

    def rng_row_offset(self):
        return self.part * self.n_pad


# ---------------------------------------------------------------------------------------------------------------------
# columns
# ---------------------------------------------------------------------------------------------------------------------
class ColumnsWhole:
    """Every rank holds whole rows: the losses read the tables directly, nothing is exchanged."""
    blocks, block, comm = 1, 0, None
    split = False

    def bind(self, d, d_valid):
        self.d, self.w, self.col0 = int(d), int(d), 0
        return self

    def slice_kw(self):
        return {}

    def full(self, t):
        return t

    # the batch-row exchange: nothing to move -- the loss kernels index the tables themselves
    def init_exchange(self, tr):
        pass

    def pack(self, tr):
        pass

    def exchange(self):
        pass

    def loss_views(self, tr):
        ident = lambda t: t                              # noqa: E731
        return ident, ident, tr.stage, tr.stage_cat

    def scatter(self, tr):
        pass


class ColumnBlocks(ColumnsWhole):
    """Rank r keeps columns [r w, (r + 1) w), w = d / blocks, of EVERY table.  A sparse product is independent per column,
    so the products of a step need no exchange; the losses read whole rows, but only the O(batch) rows the staged lists
    name: those are all-gathered once per step into compact (5B, d) tables (csrc/exchange.hip) on which the unchanged loss
    kernels run, and each rank scatters its columns of the batch-row gradients back."""
    split = True

    def __init__(self, comm):
        self.comm = comm
        self.blocks, self.block = int(comm.world), int(comm.rank)

    def bind(self, d, d_valid):
        from . import ops
        G = self.blocks
        if d_valid != d and G > 1:
            raise SelfrecHipError(f"embedding.size = {d_valid} is stored padded to {d} columns: column-sharded layouts need "
                                  f"one of {ops.ROW_WIDTHS} (use SRH_SHARD_LAYOUT=rows)")
        # (a single block keeps whole rows: the layout then only adds the batch-row exchange -- a way to run this code
        # path, collective included, on one GPU)
        if d % G or ((d // G) not in SLICE_WIDTHS and G > 1):
            raise SelfrecHipError(f"column-sharded layout: d / column blocks = {d}/{G} must be one of {SLICE_WIDTHS}")
        self.d, self.w = int(d), int(d) // G
        self.col0 = self.block * self.w
        return self

    def slice_kw(self):
        """PERTURB on a column slice: tell the kernel where the slice sits in the whole row"""
        return dict(d_full=self.d, col0=self.col0) if self.w != self.d else {}

    def full(self, t):
        """(rows, w) slice on every rank -> the whole (rows, d) table (a collective; plumbing, not per step)."""
        recv = torch.empty((self.blocks,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.comm.all_gather(recv, t.contiguous())
        return recv.permute(1, 0, 2).reshape(t.shape[0], self.d).contiguous()

    def init_exchange(self, tr):
        from . import ops
        m, B, d, dev = tr.model, tr.B, tr.d, tr.dev
        a, b = (tr.views + [None, None])[:2]
        if m == "MF":
            tables, grads = [tr.E0], [tr.gF]
        elif m == "LightGCN":
            tables, grads = [tr.F, tr.E0], [tr.gF, tr.gReg]
        elif m == "XSimGCL":
            cl = tr.E0 if tr.layer_cl == 0 else tr.Y[tr.layer_cl - 1]
            tables, grads = [tr.F, cl], [tr.gF, tr.gCL]
        elif m == "SimGCL":
            tables, grads = [tr.F, a["F"], b["F"]], [tr.gF]       # (the views' gradients join gF: one chain)
        else:
            tables, grads = [tr.F, a["F"], b["F"]], [tr.gF, a["gF"], b["gF"]]
        rows = 5 * B

        def compact():
            return torch.zeros((rows, d), dtype=torch.float32, device=dev)
        self.tables = tables
        self.compact = {id(t): compact() for t in tables}
        self.cgrad = {id(g): compact() for g in grads}
        self.pairs = [(self.cgrad[id(g)], g) for g in grads]
        self.send = torch.zeros((len(tables), rows, self.w), dtype=torch.float32, device=dev)
        self.recv = torch.zeros((self.blocks, len(tables), rows, self.w), dtype=torch.float32, device=dev)
        slots = torch.arange(rows, dtype=torch.int32, device=dev)
        self.idx = {k: slots[s * B:(s + 1) * B] for s, k in enumerate(("u", "i", "j", "uniq_u", "uniq_i"))}
        self.cat = torch.zeros(2 * B, dtype=torch.int32, device=dev) if m == "SGL" else None
        self.lists = tr.ops.batch_lists(tr.stage, tr.meta, B)

    def pack(self, tr):
        tr.ops.batch_pack(self.lists, self.tables, self.send, cat_idx=self.cat)

    def exchange(self):
        """The step's one collective: every rank's slices of the batch rows."""
        self.comm.all_gather(self.recv, self.send)

    def loss_views(self, tr):
        """whole rows exist only for the batch: compact (5B, d) tables, slot -> slot index lists"""
        tr.ops.batch_unpack(self.lists, self.recv, self.blocks, self.w, [self.compact[id(t)] for t in self.tables],
                            [c for c, _ in self.pairs])
        return (lambda t: self.compact[id(t)]), (lambda g: self.cgrad[id(g)]), self.idx, self.cat

    def scatter(self, tr):
        """this rank's columns of the batch-row gradients, added to the nodes' rows of the local tables"""
        tr.ops.batch_scatter(self.lists, self.pairs, self.d, self.col0, self.w)


# ---------------------------------------------------------------------------------------------------------------------
# gradient synchronisation
# ---------------------------------------------------------------------------------------------------------------------
class NoGradSync:
    """One batch per step: the gradient Adam takes is this rank's."""
    world, rank, comm = 1, 0, None
    active = False

    def rng_seed(self, seed):
        return seed

    def sampler_seed(self, seed):
        return seed

    def reduce(self, tr):
        pass


class GradientAllReduce(NoGradSync):
    """Data parallel: whole graph and tables on every rank, every rank trains on ITS OWN batches, and the ranks meet once
    per step in an all-reduce of the dense gradient (N x d floats) between the backward chain and Adam, which takes the
    MEAN -- synchronous data-parallel SGD with a global batch of world x B pairs (InfoNCE's negatives stay inside a rank's
    batch).  Every rank perturbs with its own noise: the counter RNG's seed is offset by the rank; rank 0 keeps the
    single-GPU stream, so a one-rank job is the single-GPU run."""
    active = True

    def __init__(self, comm):
        self.comm = comm
        self.world, self.rank = int(comm.world), int(comm.rank)

    def rng_seed(self, seed):
        return (seed + 0x9E3779B97F4A7C15 * self.rank) & 0xFFFFFFFFFFFFFFFF

    def sampler_seed(self, seed):
        return int(seed) + self.rank

    def reduce(self, tr):
        self.comm.all_reduce_sum(tr.gE0)
        if self.world > 1:
            tr.ops.axpby(1.0 / self.world, tr.gE0, 0.0, tr.gE0)


# ---------------------------------------------------------------------------------------------------------------------
# a placement = rows + columns + gradient sync
# ---------------------------------------------------------------------------------------------------------------------
class Placement:
    def __init__(self, name, rows, colx, sync):
        self.name, self.rows, self.colx, self.sync = name, rows, colx, sync
        self.G = rows.parts * colx.blocks * sync.world
        self.rank = (colx.block * rows.parts + rows.part) if not sync.active else sync.rank

    # ---- what the combination decides ----
    @property
    def single_gpu_step(self):
        """whole rows AND whole columns on this rank: the step may use the kernels that assume it -- value-free
        products, the fetch rider on the first product, Adam's fused row reset, the class-free plan of the column-masked
        launch, the calibrated XCD shares (single and dp)"""
        return not self.rows.dealt and not self.colx.split

    @property
    def replicated_batches(self):
        """more than one rank working on the SAME batches (rows / cols / 2-D): seeds and sampled epochs are checked"""
        return self.G > 1 and not self.sync.active

    @property
    def collective_in_step(self):
        """the step has ONE collective between two halves (column blocks: the batch rows; dp: the gradient): it runs as
        (before, collective, after) -- a test can drive virtual ranks phase by phase -- and is captured as two hipGraphs with
        the collective issued eagerly between them"""
        return self.colx.split or self.sync.active

    def comms(self):
        return [c for c in {id(c): c for c in (self.colx.comm, self.rows.comm, self.sync.comm) if c is not None}.values()]

    def graph_by_default(self, env):
        """single: captured; cols / dp: two graphs unless SRH_SHARDED_GRAPH=0; dealt rows (a collective after every product:
        rows, 2-D): eager unless SRH_SHARDED_GRAPH=1 asks for RCCL inside the capture"""
        if self.single_gpu_step and not self.sync.active:
            return True
        if env == "1":
            return True
        return self.collective_in_step and not self.rows.dealt and env != "0"


def parse_grid(spec, world, emb_size):
    """"2d" or "2d:GCxGR" -> (column blocks, row parts) with GC * GR = world.  Unspecified: two row parts (the exchange a
    rank waits for per layer is (Gr - 1) / Gr of an (N, w) slice -- DESIGN.md 6.2 -- so rows are split as little as
    the column widths allow), more only while d / GC would fall below the narrowest slice the kernels serve."""
    world, d = int(world), int(emb_size)
    if ":" in str(spec):
        try:
            gc, gr = (int(v) for v in str(spec).split(":", 1)[1].lower().split("x"))
        except ValueError:
            raise SelfrecHipError(f"shard layout {spec!r}: expected 2d:GCxGR, e.g. 2d:4x2") from None
        if gc * gr != world:
            raise SelfrecHipError(f"shard layout {spec!r}: {gc} x {gr} != world size {world}")
        return gc, gr
    gr = 2 if world % 2 == 0 else 1
    while world % gr or d % (world // gr) or (world // gr > 1 and d // (world // gr) not in SLICE_WIDTHS):
        gr += 1
        if gr > world:
            raise SelfrecHipError(f"no 2-D grid for {world} ranks at d = {d}")
    return world // gr, gr


def make_placement(shard, comm, emb_size, make_comm, make_grid):
    """``shard``: False / None (one GPU), True / "rows", "cols", "2d" / "2d:GCxGR", "dp".  ``comm``: a communicator (a
    pair for 2-D: (batch-row comm over the column blocks, table-row comm over the row parts)) or None = the default process
    group (``make_comm()`` / ``make_grid(gc, gr)`` build it)."""
    grid = None
    if isinstance(shard, str) and shard.startswith("2d"):
        grid, shard = shard, "2d"
    if shard not in (False, True, None, "rows", "cols", "2d", "dp"):
        raise SelfrecHipError(f"FusedTrainer: unknown shard layout {shard!r} (False, 'rows', 'cols', '2d[:GCxGR]' or 'dp')")
    if shard in (False, None):
        return Placement(False, RowsWhole(), ColumnsWhole(), NoGradSync())
    if shard == "2d":
        if comm is None:
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_initialized() else 1
            gc, gr = parse_grid(grid, world, int(emb_size))
            comm = make_grid(gc, gr)
        c_cols, c_rows = comm
        p = Placement("2d", RowParts(c_rows), ColumnBlocks(c_cols), NoGradSync())
        p.name = f"2d:{p.colx.blocks}x{p.rows.parts}"
        return p
    comm = comm if comm is not None else make_comm()
    if shard == "dp":
        return Placement("dp", RowsWhole(), ColumnsWhole(), GradientAllReduce(comm))
    if shard == "cols":
        return Placement("cols", RowsWhole(), ColumnBlocks(comm), NoGradSync())
    return Placement("rows", RowParts(comm), ColumnsWhole(), NoGradSync())


def calibration_enabled():
    """The ONE switch of the start-up calibration of the dense plan's XCD shares: SRH_XCD_CALIBRATE=0 keeps the equal dealing."""
    return os.environ.get("SRH_XCD_CALIBRATE", "1") != "0"
