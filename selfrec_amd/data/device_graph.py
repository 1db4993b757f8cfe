"""HBM-resident mirrors of the interaction graph.

Layout (all int32 structure / fp32 values, sized for one MI355X's 288 GB):
  r_indptr, r_indices          R  (U x I) CSR, columns sorted       -> eval mask, edge ids
  adj.indptr / adj.indices     A  (N x N), N = U + I: rows 0..U-1 are R's rows with columns
                               shifted by U, rows U..N-1 are R^T's rows
  edge_id[p]                   position in R's row-major order of the interaction behind
                               non-zero p (both copies of an edge share it) -- this is the
                               index space of GraphAugmentor.edge_dropout's keep-set
                               (reference data/augmentor.py:33-37: sp_adj.nonzero() order)
  weight[p]                    interaction weight (None when all ones; duplicates in the
                               training file sum, as scipy does for the reference)
  adj.vals                     D^-1/2 A D^-1/2 computed on device by srh_adj_sym_normalize
Edge-dropped views share the structure and own only a value array (``dropped_view``).
"""
import numpy as np
import scipy.sparse as sp
import torch

from .. import ops


class DeviceGraph:
    def __init__(self, interaction_mat, device=None, column_classes=True, split_len=0):
        r = interaction_mat.tocsr()
        r.sum_duplicates()
        r.sort_indices()
        self.n_users, self.n_items = r.shape
        self.n_nodes = self.n_users + self.n_items
        self.n_edges = r.nnz
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.device = dev
        eid = np.arange(r.nnz, dtype=np.int64)
        # transpose with edge ids riding along as the data
        rt = sp.csr_matrix((eid + 1, r.indices, r.indptr), shape=r.shape).tocsc()
        rt.sort_indices()
        indptr = np.concatenate([r.indptr, r.indptr[-1] + rt.indptr[1:]]).astype(np.int32)
        indices = np.concatenate([r.indices.astype(np.int64) + self.n_users, rt.indices]).astype(np.int32)
        edge_id = np.concatenate([eid, rt.data - 1]).astype(np.int32)
        w_full = None if np.all(r.data == 1.0) else np.concatenate([r.data, r.data[rt.data - 1]]).astype(np.float32)
        # long rows are stored [even columns | odd columns] so the SpMM plan can give each half to its own
        # XCDs (each 4 MiB L2 then caches a quarter of the table instead of a half; ops.column_class_order)
        row_mid = None
        min_len = 64          # = the SpMM plan's short-row bound: only cooperative rows are split by column class
        # (a column slice of 8 .. 32 columns fits every XCD's L2 whole: the split would only add hand-offs --
        # measured 26.5 -> 22.1 us per 8-column launch at the Yelp2018 shape without it)
        if min_len > 0 and column_classes:
            perm, row_mid = ops.column_class_order(indptr, indices, min_len)
            indices, edge_id = indices[perm], edge_id[perm]
            w_full = None if w_full is None else w_full[perm]
        self.r_indptr = torch.from_numpy(r.indptr.astype(np.int32)).to(dev)
        self.r_indices = torch.from_numpy(r.indices.astype(np.int32)).to(dev)
        self.edge_id = torch.from_numpy(edge_id).to(dev)
        self.weight = None if w_full is None else torch.from_numpy(w_full).to(dev)
        self.h_r_indptr, self.h_r_indices = r.indptr.astype(np.int32), r.indices.astype(np.int32)
        self._deg_ws = torch.empty(self.n_nodes, dtype=torch.float32, device=dev)
        # k^-1/2 exactly as the host's numpy evaluates np.power(float32(k), -0.5) (graph.py:14)
        max_deg = int(max(np.diff(r.indptr).max(initial=0), np.diff(rt.indptr).max(initial=0)))
        with np.errstate(divide='ignore'):
            table = np.power(np.arange(max_deg + 1, dtype=np.float32), -0.5)
        table[0] = 0.0
        self._inv_sqrt = torch.from_numpy(table.astype(np.float32)).to(dev)
        self.adj = ops.DeviceCSR(indptr, indices, torch.zeros(indices.size, dtype=torch.float32, device=dev),
                                 (self.n_nodes, self.n_nodes), device=dev, xcd_split_row=self.n_users, row_mid=row_mid,
                                 split_len=split_len)
        ops.adj_sym_normalize(self.adj.indptr, self.adj.indices, self.edge_id, None, self.n_nodes,
                              weight=self.weight, out=self.adj.vals, deg_ws=self._deg_ws,
                              inv_sqrt_table=self._inv_sqrt)
        # D^-1/2 of the full graph (the workspace is reused by dropped views): the row scale of value-free products
        self.dinv = self._deg_ws.clone()

    def dropped_view(self, keep_mask: torch.Tensor, out: torch.Tensor | None = None) -> "ops.DeviceCSR":
        """Normalised adjacency of the graph restricted to interactions with keep_mask != 0
        (uint8, indexed by R's row-major edge order).  Degrees are those of the dropped
        graph, as reference ui_graph.py:58-65 recomputes them."""
        if out is None:
            out = torch.empty_like(self.adj.vals)
        # (unit weights: GraphAugmentor rebuilds the dropped matrix with np.ones_like -- augmentor.py:22-25,36-39 --
        # so an interaction duplicated in the training file weighs 2 in norm_adj but 1 in every dropped view)
        ops.adj_sym_normalize(self.adj.indptr, self.adj.indices, self.edge_id, keep_mask, self.n_nodes,
                              weight=None, out=out, deg_ws=self._deg_ws, inv_sqrt_table=self._inv_sqrt)
        return self.adj.with_values(out)


class ShardedDeviceGraph:
    """One rank's rows of the same (N x N) adjacency for the row-sharded engine (engine.FusedTrainer with
    shard=True): nodes rank, rank + G, ... as local rows 0.., columns rewritten to all-gather order
    (owner * n_pad + local row), edge ids and weights riding along, normalised on the device in two
    phases with one all-gather of the D^-1/2 vector in between (srh_adj_sym_normalize phase 1 / 2) --
    so edge-dropped views (SGL) work exactly as on one GPU: a keep mask over interactions, the degrees
    of the dropped graph, a new value array over the shared structure."""

    def __init__(self, interaction_mat, rank, world, device, all_gather):
        r = interaction_mat.tocsr()
        r.sum_duplicates()
        r.sort_indices()
        self.n_users, self.n_items = r.shape
        self.n_nodes = N = self.n_users + self.n_items
        self.n_edges = r.nnz
        self.device, self._all_gather = device, all_gather
        eid = np.arange(r.nnz, dtype=np.int64)
        rt = sp.csr_matrix((eid + 1, r.indices, r.indptr), shape=r.shape).tocsc()
        rt.sort_indices()
        indptr = np.concatenate([r.indptr, r.indptr[-1] + rt.indptr[1:]]).astype(np.int64)
        indices = np.concatenate([r.indices.astype(np.int64) + self.n_users, rt.indices])
        edge_id = np.concatenate([eid, rt.data - 1]).astype(np.int32)
        w_full = None if np.all(r.data == 1.0) else np.concatenate([r.data, r.data[rt.data - 1]]).astype(np.float32)
        self.h_r_indptr, self.h_r_indices = r.indptr.astype(np.int32), r.indices.astype(np.int32)
        self.n_pad = n_pad = (N + world - 1) // world
        self.P = world * n_pad
        own = np.arange(rank, N, world)
        lens = (indptr[1:] - indptr[:-1])[own]
        l_indptr = np.zeros(n_pad + 1, dtype=np.int32)
        l_indptr[1:own.size + 1] = np.cumsum(lens)
        l_indptr[own.size + 1:] = l_indptr[own.size]                      # padding rows are empty
        entry = np.repeat(indptr[own] - l_indptr[:own.size], lens) + np.arange(int(l_indptr[own.size]), dtype=np.int64)
        cols = indices[entry]
        l_cols = ((cols % world) * n_pad + cols // world).astype(np.int32)
        self.edge_id = torch.from_numpy(edge_id[entry]).to(device)
        self.weight = None if w_full is None else torch.from_numpy(w_full[entry]).to(device)
        max_deg = int(max(np.diff(r.indptr).max(initial=0), np.diff(rt.indptr).max(initial=0)))
        with np.errstate(divide='ignore'):
            table = np.power(np.arange(max_deg + 1, dtype=np.float32), -0.5)
        table[0] = 0.0
        self._inv_sqrt = torch.from_numpy(table.astype(np.float32)).to(device)
        self._dinv = torch.zeros(self.P, dtype=torch.float32, device=device)
        self.row_offset = rank * n_pad
        self.adj = ops.DeviceCSR(l_indptr, l_cols, torch.zeros(l_cols.size, dtype=torch.float32, device=device),
                                 (n_pad, self.P), device=device, xcd_split_row=len(range(rank, self.n_users, world)))
        self._normalize(None, self.adj.vals)

    def _normalize(self, keep, out):
        # (dropped views are rebuilt with unit weights: see DeviceGraph.dropped_view)
        kw = dict(weight=self.weight if keep is None else None, out=out, deg_ws=self._dinv, inv_sqrt_table=self._inv_sqrt, row_offset=self.row_offset)
        ops.adj_sym_normalize(self.adj.indptr, self.adj.indices, self.edge_id, keep, self.n_pad, phase=1, **kw)
        self._all_gather(self._dinv)                 # every rank needs D^-1/2 of the columns it references
        ops.adj_sym_normalize(self.adj.indptr, self.adj.indices, self.edge_id, keep, self.n_pad, phase=2, **kw)

    def dropped_view(self, keep_mask, out=None):
        if out is None:
            out = torch.empty_like(self.adj.vals)
        self._normalize(keep_mask, out)
        return self.adj.with_values(out)
