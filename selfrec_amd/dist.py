"""Row-sharded training across the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Partition.  The N = U + I graph nodes are dealt round-robin: node p lives on rank
p % G at local row p // G.  Interleaving spreads users, items and the power-law heavy rows
evenly, so every rank owns ~N/G table rows, ~nnz/G non-zeros, the matching slice of every layer
output, gradient buffer and Adam moment.  Each rank keeps the CSR rows of A_hat for its nodes,
with column ids rewritten to the layout an all-gather produces: node q sits at row
(q % G) * N_pad + q // G of the gathered (G * N_pad, d) buffer, N_pad = ceil(N / G).

Exchange per step.  A_hat is symmetric, so forward layers and backward layers are the same
row-block product  Y_local = A_local . X_full : one all-gather of the (N_pad, d) shards before
each of the 2L products (the two halves of the "all-reduce of layer embeddings and gradients";
no rank ever reduces rows it does not own).  The O(batch) rows the losses need are assembled
with one small all-reduce of a zero-padded (rows, d) buffer (each row is contributed by its
owner only), the losses are evaluated redundantly on every rank (B x d work), and each rank
scatters the resulting row gradients it owns into its local buffers.  Adam is purely local.
The sampler is replicated: every rank replays the same MT19937 stream, so there is no index
broadcast.  Message sizes at Yelp2018 shape (d=64): 17.8 MB per all-gather, ~3.7 MB for the
batch rows; on the fully connected xGMI mesh an all-gather is 7 concurrent point-to-point
sends per rank, which is why all-gather (not a ring all-reduce) is the collective used.

The kernels are reached through a small backend object so that the sharding logic itself --
partition, column rewrite, collectives, owner scatter -- is exercised on CPU (gloo, world
size 2) by tests/test_dist_cpu.py with a torch-CPU stand-in; the product backend is
``HipBackend`` and nothing else is ever selected implicitly.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from ._lib import SelfrecHipError


class HipBackend:
    """The product backend: every call is a libselfrec_hip.so kernel on this rank's GPU."""

    def __init__(self, device=None):
        from . import ops
        ops._lib.require_gpu()
        self.ops = ops
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device

    def csr(self, indptr, indices, vals, shape):
        return self.ops.DeviceCSR(indptr, indices, vals, shape, device=self.device)

    def spmm(self, csr, x, out, *, perturb=None, mean=None, add=None, add_scale=None, alpha=1.0):
        kw = {}
        if perturb is not None:
            eps, noise, seed, offset = perturb
            kw.update(perturb_eps=eps, noise=noise, rng_seed=seed, rng_offset=offset)
        if mean is not None:
            prev, div, mean_out = mean
            kw.update(prev=prev, mean_div=div, mean_out=mean_out)
        if add:
            kw.update(add=add, add_scale=add_scale)
        if alpha != 1.0 or add:
            kw.update(alpha=alpha)
        self.ops.spmm(csr, x, out=out, epilogue=self.ops.make_epilogue(**kw) if kw else None)

    def bpr_l2(self, u, p, n, ru, rp, rn, reg_coef, include_neg, losses):
        """Rows already gathered: (B, d) each.  Returns grads (gu, gp, gn, gru, grp, grn)."""
        B, d = u.shape
        dev = u.device
        items, ritems = torch.cat([p, n]), torch.cat([rp, rn])
        ar = torch.arange(B, dtype=torch.int32, device=dev)
        gu, gi = torch.zeros_like(u), torch.zeros_like(items)
        same = ru is u
        gru, gri = (gu, gi) if same else (torch.zeros_like(u), torch.zeros_like(items))
        self.ops.bpr_l2_fwd_bwd(u, items, u if same else ru.contiguous(), items if same else ritems, ar, ar,
                                ar + B, batch=B, reg_coef=reg_coef, reg_include_neg=include_neg, loss_scale=1.0,
                                g_user=gu, g_item=gi, greg_user=gru, greg_item=gri, losses=losses[0:2],
                                ws=self.ops.bpr_ws(B, dev))
        return gu, gi[:B], gi[B:], gru, gri[:B], gri[B:]

    def infonce(self, v1, v2, tau, scale, loss):
        n, d = v1.shape
        g1, g2 = torch.zeros_like(v1), torch.zeros_like(v2)
        self.ops.infonce_fwd_bwd(v1.contiguous(), v2.contiguous(), None, n, tau=tau, loss_scale=scale, loss=loss,
                                 g1=g1, g2=g2, ws=self.ops.infonce_ws(n, d, v1.device))
        return g1, g2

    def adam(self, p, g, m, v, step, lr):
        self.ops.adam_step(p, g, m, v, step=step, lr=lr)

    def make_sampler(self, edge_u, edge_i, n_users, n_items):
        return self.ops.Sampler(edge_u, edge_i, n_users, n_items)


def shard_adjacency(norm_adj_csr, rank, world):
    """CSR rows of the nodes owned by `rank`, columns rewritten to the all-gather layout.
    Returns (indptr, indices, data, n_pad)."""
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


class ShardedTrainer:
    """LightGCN / XSimGCL / MF training step with the tables and the graph row-sharded over the
    ranks of the default process group.  Same numerical spec as engine.FusedTrainer."""

    def __init__(self, data, emb_size, *, model, n_layers=2, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2,
                 layer_cl=1, batch_size=2048, user_emb=None, item_emb=None, noise_fn=None, rng_seed=0x5E1F0EC,
                 backend=None, use_graph=False, **_unused):
        if model not in ("MF", "LightGCN", "XSimGCL"):
            raise SelfrecHipError(f"ShardedTrainer: model {model!r} is not sharded yet (MF, LightGCN, XSimGCL are)")
        if not dist.is_initialized():
            raise SelfrecHipError("ShardedTrainer needs an initialised torch.distributed process group")
        self.k = backend if backend is not None else HipBackend()
        dev = self.k.device
        self.rank, self.G = dist.get_rank(), dist.get_world_size()
        self.model, self.d, self.L = model, int(emb_size), (0 if model == "MF" else int(n_layers))
        self.lr, self.reg, self.cl_rate, self.eps, self.tau = float(lr), float(reg), float(cl_rate), float(eps), float(tau)
        self.layer_cl, self.B = int(layer_cl), int(batch_size)
        self.noise_fn, self.rng_seed = noise_fn, int(rng_seed)
        self.U, self.I = data.user_num, data.item_num
        self.N = self.U + self.I
        G, r, N, d = self.G, self.rank, self.N, self.d
        indptr, indices, vals, self.n_pad = shard_adjacency(data.norm_adj.tocsr(), r, G)
        self.adj = self.k.csr(indptr, indices, vals, (self.n_pad, G * self.n_pad))
        self.n_own = len(range(r, N, G))
        self.graph = _GraphInfo(self.U, self.I, len(data.train_u), self.adj)

        if user_emb is None or item_emb is None:
            ue = torch.nn.init.xavier_uniform_(torch.empty(self.U, d))
            ie = torch.nn.init.xavier_uniform_(torch.empty(self.I, d))
        else:
            ue, ie = torch.as_tensor(user_emb, dtype=torch.float32), torch.as_tensor(item_emb, dtype=torch.float32)
        full = torch.cat([ue, ie])

        def buf():
            return torch.zeros((self.n_pad, d), dtype=torch.float32, device=dev)

        self.E0 = buf()
        self.E0[:self.n_own].copy_(full[r::G])
        self.m, self.v, self.gE0 = buf(), buf(), buf()
        self.F = self.E0 if model == "MF" else buf()
        self.gF = self.gE0 if model == "MF" else buf()
        self.Y = [buf() for _ in range(self.L)]
        self.Ha, self.Hb = (buf(), buf()) if self.L else (None, None)
        self.gCL = buf() if model == "XSimGCL" else None
        self.full = torch.zeros((G * self.n_pad, d), dtype=torch.float32, device=dev)   # all-gather target
        self.losses = torch.zeros(4, dtype=torch.float64, device=dev)
        self.sampler = self.k.make_sampler(data.train_u, data.train_i, self.U, self.I)
        self.epoch_batches = (len(data.train_u) + self.B - 1) // self.B
        self._host, self._batch_no, self.step_count = None, 0, 0
        self._noise_call = 0
        self.dev = dev

    # ---- sampling: replicated, identical on every rank ------------------------------------
    def sample_epoch_host(self):
        return self.sampler.epoch(self.B, 1, with_unique=True)

    def upload_epoch(self, host):
        """Node ids of the whole epoch go to the device once; a step only slices them (no host sync)."""
        self._host, self._batch_no = host, 0
        U = self.U
        dev = lambda a, off=0: torch.from_numpy(a.astype(np.int64) + off).to(self.dev)   # noqa: E731
        self._ep = {"u": dev(host["u"]), "p": dev(host["i"], U), "n": dev(host["j"], U),
                    "uu": dev(host["uniq_u"]), "ui": dev(host["uniq_i"], U)}

    def begin_epoch(self):
        self.upload_epoch(self.sample_epoch_host())
        return self.epoch_batches

    # ---- collectives ------------------------------------------------------------------------
    def _gather(self, local):
        dist.all_gather_into_tensor(self.full, local)
        return self.full

    def _owner_rows(self, ids):
        """For global node ids (device int64): (0/1 float mask of ids this rank owns, their local rows).
        Every id has a valid local row on every rank (loc < n_pad), so gathers need no compaction."""
        return ((ids % self.G) == self.rank).to(torch.float32).unsqueeze(1), ids // self.G

    def _collect_rows(self, tables_and_ids):
        """One all-reduce assembling rows of sharded tables: [(local_table, global ids), ...] ->
        list of dense (len(ids), d) tensors, identical on every rank.  Sync-free: non-owned rows are
        gathered from whatever lives at that local row and multiplied by zero."""
        parts, metas = [], []
        for table, ids in tables_and_ids:
            own, loc = self._owner_rows(ids)
            parts.append(table[loc] * own)
            metas.append((own, loc))
        flat = torch.cat(parts)
        dist.all_reduce(flat)
        outs, at = [], 0
        for p in parts:
            outs.append(flat[at:at + p.shape[0]])
            at += p.shape[0]
        return outs, metas

    # ---- encoder ------------------------------------------------------------------------------
    def _noise_shard(self):
        if self.noise_fn is None:
            return None
        t = torch.as_tensor(self.noise_fn((self.N, self.d)), dtype=torch.float32)
        out = torch.zeros((self.n_pad, self.d), dtype=torch.float32)
        out[:self.n_own] = t[self.rank::self.G]
        return out.to(self.dev)

    def _forward(self, Ys, F, perturbed, include_ego):
        x = self.E0
        for k in range(self.L):
            kw = {}
            if perturbed:
                off = (self.step_count * 16 + self._noise_call) * self.n_pad * self.G + self.rank * self.n_pad
                kw["perturb"] = (self.eps, self._noise_shard(), self.rng_seed, off)
                self._noise_call += 1
            if k == self.L - 1:
                prev = ([self.E0] if include_ego else []) + Ys[:self.L - 1]
                kw["mean"] = (prev, float(self.L + 1 if include_ego else self.L), F)
            self.k.spmm(self.adj, self._gather(x), Ys[k], **kw)
            x = Ys[k]

    def _backward(self, gF, include_ego, gCL=None, layer_cl=None):
        L = self.L
        s = 1.0 / (L + 1 if include_ego else L)
        cl_at = layer_cl if gCL is not None else None
        if cl_at == L:
            H = self.Ha
            H.copy_(gF).mul_(s).add_(gCL)
            src, alpha = H, 1.0
        else:
            src, alpha = gF, s
        bufs = [self.Hb, self.Ha] if src is self.Ha else [self.Ha, self.Hb]
        for k in range(L - 1, 0, -1):
            add, sc = [gF], [s]
            if cl_at == k:
                add.append(gCL); sc.append(1.0)
            dst = bufs[0]
            self.k.spmm(self.adj, self._gather(src), dst, add=add, add_scale=sc, alpha=alpha)
            src, alpha = dst, 1.0
            bufs.reverse()
        add, sc = [self.gE0], [1.0]
        if include_ego:
            add.append(gF); sc.append(s)
        if cl_at == 0:
            add.append(gCL); sc.append(1.0)
        while len(add) > 2:
            self.gE0.add_(add.pop(), alpha=sc.pop())
        self.k.spmm(self.adj, self._gather(src), self.gE0, add=add, add_scale=sc, alpha=alpha)

    # ---- one step -------------------------------------------------------------------------------
    def step(self):
        if self._host is None:
            raise SelfrecHipError("call begin_epoch() first")
        h, b, B, U, ep = self._host, self._batch_no, self.B, self.U, self._ep
        lo, hi = b * B, min((b + 1) * B, len(h["u"]))
        u_ids, p_ids, n_ids = ep["u"][lo:hi], ep["p"][lo:hi], ep["n"][lo:hi]
        self._batch_no += 1
        self._noise_call = 0
        m = self.model
        for t in (self.gE0, self.gF, self.gCL):
            if t is not None:
                t.zero_()
        self.losses.zero_()
        include_ego = m == "LightGCN"
        if m != "MF":
            self._forward(self.Y, self.F, perturbed=(m == "XSimGCL"), include_ego=include_ego)
        want = [(self.F, u_ids), (self.F, p_ids), (self.F, n_ids)]
        if m == "LightGCN":
            want += [(self.E0, u_ids), (self.E0, p_ids), (self.E0, n_ids)]
        if m == "XSimGCL":
            uu = ep["uu"][b * B:b * B + int(h["n_uniq_u"][b])]
            ui = ep["ui"][b * B:b * B + int(h["n_uniq_i"][b])]
            CL = self.E0 if self.layer_cl == 0 else self.Y[self.layer_cl - 1]
            want += [(self.F, uu), (CL, uu), (self.F, ui), (CL, ui)]
        rows, metas = self._collect_rows(want)
        ru, rp, rn = rows[0], rows[1], rows[2]
        if m == "LightGCN":
            eu, ep_, en = rows[3], rows[4], rows[5]
            g = self.k.bpr_l2(ru, rp, rn, eu, ep_, en, self.reg / self.B, True, self.losses)
            for (own, loc), grad in zip(metas[3:6], g[3:6]):
                self.gE0.index_add_(0, loc, grad * own)
        else:
            coef = self.reg / self.B if m == "MF" else self.reg
            g = self.k.bpr_l2(ru, rp, rn, ru, rp, rn, coef, m == "MF", self.losses)
        for (own, loc), grad in zip(metas[0:3], g[0:3]):
            self.gF.index_add_(0, loc, grad * own)
        if m == "XSimGCL":
            base = 3
            for side in range(2):
                v1, v2 = rows[base + 2 * side], rows[base + 2 * side + 1]
                g1, g2 = self.k.infonce(v1, v2, self.tau, self.cl_rate, self.losses[2:3])
                (own, loc) = metas[base + 2 * side]
                self.gF.index_add_(0, loc, g1 * own)
                self.gCL.index_add_(0, loc, g2 * own)
        if m == "XSimGCL":
            self._backward(self.gF, False, gCL=self.gCL, layer_cl=self.layer_cl)
        elif m == "LightGCN":
            self._backward(self.gF, True)
        self.step_count += 1
        self.k.adam(self.E0, self.gE0, self.m, self.v, self.step_count, self.lr)

    def read_losses(self):
        return tuple(self.losses[:3].tolist())

    # ---- whole tables, for evaluation and tests ---------------------------------------------------
    def _unshard(self, local):
        full = self._gather(local).view(self.G, self.n_pad, self.d)
        out = torch.empty((self.N, self.d), dtype=torch.float32, device=self.dev)
        for r in range(self.G):
            cnt = len(range(r, self.N, self.G))
            out[r::self.G] = full[r, :cnt]
        return out

    def parameters_full(self):
        t = self._unshard(self.E0)
        return t[:self.U], t[self.U:]

    @torch.no_grad()
    def embeddings(self):
        if self.model == "MF":
            return self.parameters_full()
        F = torch.zeros_like(self.E0)
        Ys = [torch.zeros_like(self.E0) for _ in range(self.L)]
        self._forward(Ys, F, perturbed=False, include_ego=self.model == "LightGCN")
        t = self._unshard(F)
        return t[:self.U], t[self.U:]


class _GraphInfo:
    """The attributes bench.py reads from ``trainer.graph``."""

    def __init__(self, n_users, n_items, n_edges, adj):
        self.n_users, self.n_items, self.n_edges = n_users, n_items, n_edges
        self.n_nodes = n_users + n_items
        self.adj = adj
