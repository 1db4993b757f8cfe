"""Communicators of the multi-GPU layouts: a ``torch.distributed`` process group (or a sub-group of it) as the three
things a placement component needs from it -- ``all_gather``, ``all_reduce_sum``, ``assert_replicated`` -- and the two-hop
form of the 2-D layout's table-row exchange over every xGMI link.  Test stand-ins implement the same
methods (tests/test_gpu_cols.py drives virtual ranks through them)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as _dist

from ._lib import SelfrecHipError


class TorchComm:
    """A process group (the default one, or a sub-group of it) as the two things the trainer needs from it."""

    def __init__(self, group=None, ranks=None):
        if not _dist.is_initialized():
            raise SelfrecHipError("sharded training needs an initialised torch.distributed process group")
        self.group = group
        if group is None:
            self.world, self.rank = _dist.get_world_size(), _dist.get_rank()
        else:
            self.world, self.rank = len(ranks), list(ranks).index(_dist.get_rank())

    @classmethod
    def grid(cls, n_col_groups, n_row_parts):
        """The two communicators of the 2-D layout on the default group of G = Gc * Gr ranks.  Rank q holds column
        block q // Gr and row part q % Gr; it exchanges BATCH rows with the ranks of its row part (one per column
        block: `cols`) and TABLE rows with the ranks of its column block (`rows`).  Every rank creates every group,
        in the same order (torch.distributed's rule)."""
        world, me = _dist.get_world_size(), _dist.get_rank()
        if world != n_col_groups * n_row_parts:
            raise SelfrecHipError(f"2-D layout {n_col_groups} x {n_row_parts} needs {n_col_groups * n_row_parts} ranks, not {world}")
        mine = {}
        for r in range(n_row_parts):
            ranks = [c * n_row_parts + r for c in range(n_col_groups)]
            g = _dist.new_group(ranks)
            if me in ranks:
                mine["cols"] = cls(g, ranks)
        for c in range(n_col_groups):
            ranks = [c * n_row_parts + r for r in range(n_row_parts)]
            g = _dist.new_group(ranks)
            if me in ranks:
                mine["rows"] = cls(g, ranks)
        how = os.environ.get("SRH_2D_EXCHANGE", "twohop").lower()
        if how not in ("twohop", "direct"):
            raise SelfrecHipError(f"SRH_2D_EXCHANGE={how!r}: twohop or direct")
        if how == "twohop" and n_col_groups > 1 and n_row_parts > 1:
            mine["rows"] = TwoHopRows(mine["rows"], n_col_groups, n_row_parts)
        return mine["cols"], mine["rows"]

    def all_gather(self, out, inp):
        # (flat views: rank r's contribution is the r-th equal piece of `out`, whatever the shapes)
        _dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group)

    def all_reduce_sum(self, t):
        _dist.all_reduce(t, op=_dist.ReduceOp.SUM, group=self.group)

    def assert_replicated(self, what, values, device):
        """Every rank must hold the same `values` (a short list of floats: checksums of state the step code
        assumes replicated -- initial tables, the epoch's sampled indices).  Raises on the ranks that differ
        from rank 0 AND on rank 0, so a mis-seeded job stops instead of training on different batches."""
        mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
        everyone = torch.empty((self.world, mine.numel()), dtype=torch.float64, device=device)
        _dist.all_gather_into_tensor(everyone.view(-1), mine, group=self.group)
        bad = [r for r in range(self.world) if not torch.equal(everyone[r], everyone[0])]
        if bad:
            raise SelfrecHipError(f"{what} differs between ranks (rank 0 vs ranks {bad}): every rank must be seeded "
                                  f"identically (torch.manual_seed for the tables, the sampler seed / python `random` "
                                  f"state for the batches); this rank is {self.rank}")


class AbiComm:
    """The same three methods over the LIBRARY's collectives -- ``srh_comm_init_rank`` / ``srh_allgather_rows`` /
    ``srh_allreduce_sum_f32`` (include/selfrec_hip.h, csrc/collectives.cpp: thin RCCL wrappers) -- i.e. what a host that
    binds the C header runs, without ``torch.distributed`` in the data path.  ``FusedTrainer(shard="rows", comm=AbiComm())``
    / ``SRH_COLLECTIVES=abi`` for the default communicators (``default_comm``).  The 128-byte unique id reaches the other
    ranks through ``bootstrap(id_bytes_or_None) -> id_bytes`` (default: ``torch.distributed.broadcast_object_list`` on an
    initialised group of any backend; one rank needs none)."""

    def __init__(self, world=None, rank=None, bootstrap=None):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib.load()
        if world is None:
            world, rank = (_dist.get_world_size(), _dist.get_rank()) if _dist.is_initialized() else (1, 0)
        self.world, self.rank, self.group = int(world), int(rank), None
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            _lib.check(self._lib.srh_comm_unique_id(uid), "srh_comm_unique_id")
        if self.world > 1:
            if bootstrap is None:
                def bootstrap(mine):
                    box = [mine]
                    _dist.broadcast_object_list(box, src=0)
                    return box[0]
            got = bootstrap(bytes(uid) if self.rank == 0 else None)
            uid = (C.c_uint8 * 128)(*got)
        h = C.c_void_p()
        _lib.check(self._lib.srh_comm_init_rank(C.byref(h), self.world, self.rank, uid), "srh_comm_init_rank")
        self._h = h
        n = C.c_int32()
        _lib.check(self._lib.srh_comm_world(self._h, C.byref(n)), "srh_comm_world")
        if n.value != self.world:
            raise SelfrecHipError(f"AbiComm: the communicator counts {n.value} ranks, expected {self.world}")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.srh_comm_destroy(h)
            except Exception:        # (interpreter shutdown: the library or RCCL may be gone already)
                pass
            self._h = None

    @staticmethod
    def _as_f32(t):
        if not t.is_contiguous():
            raise SelfrecHipError("AbiComm: contiguous tensors only")
        return t.view(-1) if t.dtype == torch.float32 else t.view(-1).view(torch.float32)     # (4- and 8-byte types: bit transport)

    def all_gather(self, out, inp):
        from . import _lib
        o, i = self._as_f32(out), self._as_f32(inp)
        if o.numel() != self.world * i.numel():
            raise SelfrecHipError(f"AbiComm.all_gather: {o.numel()} != {self.world} x {i.numel()}")
        _lib.check(self._lib.srh_allgather_rows(self._C.c_void_p(i.data_ptr()), self._C.c_void_p(o.data_ptr()), i.numel(), 1,
                                                self._h, self._C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "srh_allgather_rows")

    def reduce_scatter_sum(self, out, inp):
        from . import _lib
        if out.dtype != torch.float32 or inp.dtype != torch.float32 or inp.numel() != self.world * out.numel():
            raise SelfrecHipError("AbiComm.reduce_scatter_sum: float32, world x the slice")
        _lib.check(self._lib.srh_reducescatter_rows(self._C.c_void_p(inp.data_ptr()), self._C.c_void_p(out.data_ptr()),
                                                    out.numel(), 1, self._h,
                                                    self._C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "srh_reducescatter_rows")

    def all_reduce_sum(self, t):
        from . import _lib
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise SelfrecHipError("AbiComm.all_reduce_sum: contiguous float32")
        _lib.check(self._lib.srh_allreduce_sum_f32(self._C.c_void_p(t.data_ptr()), t.numel(), self._h,
                                                   self._C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "srh_allreduce_sum_f32")

    def assert_replicated(self, what, values, device):
        mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
        everyone = torch.empty((self.world, mine.numel()), dtype=torch.float64, device=device)
        self.all_gather(everyone, mine)
        bad = [r for r in range(self.world) if not torch.equal(everyone[r], everyone[0])]
        if bad:
            raise SelfrecHipError(f"{what} differs between ranks (rank 0 vs ranks {bad}); this rank is {self.rank}")


def default_comm():
    """The communicator of a placement nobody handed one: ``TorchComm`` over the default process group, or -- with
    ``SRH_COLLECTIVES=abi`` -- the library's own RCCL wrappers (``AbiComm``)."""
    how = os.environ.get("SRH_COLLECTIVES", "torch").lower()
    if how not in ("torch", "abi"):
        raise SelfrecHipError(f"SRH_COLLECTIVES={how!r}: torch or abi")
    return AbiComm() if how == "abi" else TorchComm()


class TwoHopRows:
    """The table-row all-gather of the 2-D layout, moved over EVERY xGMI link instead of the Gr - 1 direct ones.

    xGMI is a full mesh of point-to-point links (7 per GPU, ~75 GB/s per direction each).  A column block's all-gather
    over its Gr ranks uses Gr - 1 of a rank's links and leaves the others idle: at Gr = 2 one link carries the whole
    (N / 2, w) slab (96 MB at the 1 M x 500 k, d = 128 shape on a 4 x 2 grid: ~1.3 ms per layer).
    Two all-to-alls over ALL G ranks move the same slab through every link at once:

        hop 1   rank q cuts its slab into G pieces and sends piece k to rank k          (1/G of the slab per link)
        hop 2   rank k forwards piece k of rank q's slab to q's partner(s)              (again 1/G per link)

    so a rank's link carries 2/G of a slab per partner instead of a whole one (Gr = 2, G = 8: 4x less time on the wire).
    With Gr > 2 hop 2 runs once per partner offset j = 1 .. Gr - 1 (rank k sends to rank t the piece it holds of the
    slab of t's j-th partner).  Pieces are padded to equal size; the slab's own piece never leaves the rank.
    Same interface and same result as the direct group all-gather (tests/test_dist_cpu.py).

    LOCKSTEP.  Hops 1 and 2 are ``all_to_all_single`` on the WORLD group, although the component gathers within one
    column block: every rank of EVERY column block has to make the same sequence of ``all_gather`` calls with the same
    ``n`` and dtype -- and therefore the same direct / two-hop decision -- or the job hangs instead of raising.  The engine
    satisfies this by construction (all ranks run the same step on tables of the same padded shape); the first call of
    every (n, dtype) checks it with one small all-reduce over WORLD and raises on a mismatch."""

    def __init__(self, direct, n_col_groups, n_row_parts):
        self.direct = direct                              # (tiny messages -- checksums, the D^-1/2 vector -- go direct)
        self.world, self.rank, self.group = direct.world, direct.rank, direct.group
        self.Gc, self.Gr = int(n_col_groups), int(n_row_parts)
        self.G, self.me = self.Gc * self.Gr, _dist.get_rank()
        self._buf = {}
        self.min_bytes = int(os.environ.get("SRH_2D_TWOHOP_MIN_BYTES", 1 << 20))

    def assert_replicated(self, what, values, device):
        self.direct.assert_replicated(what, values, device)

    def _check_lockstep(self, n, inp):
        """first two-hop gather of this (n, dtype): every rank of the world is about to make the same call (else: raise,
        on every rank, before the all-to-all that would hang)"""
        key = ("lockstep", int(n), str(inp.dtype))
        if key in self._buf:
            return
        sig = float(n) * 4.0 + inp.element_size()
        t = torch.tensor([sig, -sig], dtype=torch.float64, device=inp.device)
        _dist.all_reduce(t, op=_dist.ReduceOp.MAX)           # max(sig), max(-sig) = -min(sig): equal everywhere iff both are ours
        if float(t[0]) != sig or float(t[1]) != -sig:
            raise RuntimeError(f"TwoHopRows.all_gather: ranks disagree on the gather they are making (this rank: {n} elements "
                               f"of {inp.dtype}; the world's range of n * 4 + itemsize: {-float(t[1])} .. {float(t[0])}) -- "
                               "every rank of every column block must call it in lockstep")
        self._buf[key] = True

    def _scratch(self, piece, dtype, device):
        key = (piece, dtype, str(device))
        if key not in self._buf:
            self._buf[key] = tuple(torch.empty(self.G * piece, dtype=dtype, device=device) for _ in range(3))
        return self._buf[key]

    def all_gather(self, out, inp):
        out, inp = out.view(-1), inp.reshape(-1)
        n = inp.numel()
        if n * inp.element_size() < self.min_bytes or out.numel() != self.Gr * n:
            return self.direct.all_gather(out, inp)
        self._check_lockstep(n, inp)
        G, Gr = self.G, self.Gr
        piece = (n + G - 1) // G
        send, held, fwd = self._scratch(piece, inp.dtype, inp.device)
        send[:n].copy_(inp)                               # (inp may alias out's own piece: read it before anything lands)
        _dist.all_to_all_single(held, send)               # hop 1: held[q] = piece `me` of rank q's slab
        c, r = self.me // Gr, self.me % Gr
        pieces = held.view(G, piece)
        for j in range(1, Gr):
            # hop 2, partner offset j: destination t receives the piece of t's j-th partner's slab that this rank holds
            key = (j, str(inp.device))
            if key not in self._buf:
                self._buf[key] = torch.tensor([(t // Gr) * Gr + (t % Gr + j) % Gr for t in range(G)], device=inp.device)
            src = self._buf[key]
            torch.index_select(pieces, 0, src, out=fwd.view(G, piece))
            _dist.all_to_all_single(send, fwd)            # send[k] = piece k of my j-th partner's slab
            part = (r + j) % Gr
            out[part * n:(part + 1) * n].copy_(send[:n])
        if out.data_ptr() + r * n * out.element_size() != inp.data_ptr():
            out[r * n:(r + 1) * n].copy_(inp)


def shard_adjacency(norm_adj_csr, rank, world):
    """CSR rows of the nodes owned by `rank` (nodes rank, rank + world, ...), columns rewritten to the
    all-gather layout (owner * n_pad + local row).  Returns (indptr, indices, data, n_pad)."""
    n = norm_adj_csr.shape[0]
    n_pad = (n + world - 1) // world
    own = np.arange(rank, n, world)
    sub = norm_adj_csr[own].tocsr()
    sub.sort_indices()
    cols = sub.indices.astype(np.int64)
    new_cols = (cols % world) * n_pad + cols // world
    indptr = np.zeros(n_pad + 1, dtype=np.int32)
    indptr[1:len(own) + 1] = sub.indptr[1:]
    indptr[len(own) + 1:] = sub.indptr[-1]                 # padding rows are empty
    return indptr, new_cols.astype(np.int32), sub.data.astype(np.float32), n_pad
