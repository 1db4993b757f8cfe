"""Graph augmentation with the call surface of reference data/augmentor.py:6-40, kept on
the device.

``GraphAugmentor.edge_dropout(sp_adj, drop_rate)`` draws the SAME keep-set as the reference
(``random.sample(range(E), int(E*(1-rate)))`` on the global ``random`` stream, replayed
bit-exactly in C++) but returns a ``DroppedInteraction`` -- a keep mask over the resident
graph -- instead of a rebuilt scipy matrix.  The next two calls a model makes
(``data.convert_to_laplacian_mat(m)`` then ``convert_sparse_mat_to_tensor(m).cuda()``,
SGL.py:89-96) recognise it and produce the re-normalised adjacency with one device kernel
pair; nothing is rebuilt on the host and nothing is re-uploaded but E mask bytes.
"""
import numpy as np
import torch

from .. import ops


class DroppedInteraction:
    """Lazy (U x I) interaction matrix minus dropped edges / nodes."""

    def __init__(self, shape, keep_mask):
        self.shape = tuple(shape)
        self.keep_mask = keep_mask            # numpy uint8 over R's row-major edges

    def get_shape(self):
        return self.shape

    def count_nonzero(self):
        return int(self.keep_mask.sum())

    def to_device_laplacian(self, data):
        from ..base.torch_interface import SparseAdjHandle
        g = data.device_graph()
        mask = torch.from_numpy(self.keep_mask).to(g.device)
        return SparseAdjHandle(g.dropped_view(mask), symmetric=True)

    def to_scipy(self, data):
        r = data.interaction_mat.tocsr()
        r.sort_indices()
        out = r.copy()
        out.data = out.data * self.keep_mask
        out.eliminate_zeros()
        return out


def _sampler_on_global_stream(n_rows, n_cols):
    smp = ops.Sampler(np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), max(n_rows, 1), max(n_cols, 1))
    smp.set_state_from_python()
    return smp


class GraphAugmentor:
    @staticmethod
    def edge_dropout(sp_adj, drop_rate):
        shape = sp_adj.get_shape()
        edge_count = sp_adj.count_nonzero()
        smp = _sampler_on_global_stream(*shape)
        keep_idx = smp.sample_range(edge_count, int(edge_count * (1 - drop_rate)))
        smp.push_state_to_python()
        mask = np.zeros(edge_count, dtype=np.uint8)
        mask[keep_idx] = 1
        return DroppedInteraction(shape, mask)

    @staticmethod
    def node_dropout(sp_adj, drop_rate):
        n_users, n_items = sp_adj.get_shape()
        smp = _sampler_on_global_stream(n_users, n_items)
        drop_u = smp.sample_range(n_users, int(n_users * drop_rate))
        drop_i = smp.sample_range(n_items, int(n_items * drop_rate))
        smp.push_state_to_python()
        csr = sp_adj.tocsr()
        csr.sort_indices()
        rows = np.repeat(np.arange(n_users), np.diff(csr.indptr))
        dead_u = np.zeros(n_users, dtype=bool)
        dead_i = np.zeros(n_items, dtype=bool)
        dead_u[drop_u] = True
        dead_i[drop_i] = True
        mask = (~dead_u[rows] & ~dead_i[csr.indices]).astype(np.uint8)
        return DroppedInteraction((n_users, n_items), mask)
