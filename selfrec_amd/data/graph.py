"""Host-side graph normalisation with the semantics of reference data/graph.py:10-24.

Used for the scipy ``norm_adj`` attribute that model files read; the device-resident
variant (and SGL's per-epoch re-normalisation) is ``srh_adj_sym_normalize``.
"""
import numpy as np
import scipy.sparse as sp


class Graph:
    @staticmethod
    def normalize_graph_mat(adj_mat):
        """Square input: D^-1/2 A D^-1/2; rectangular: D^-1 A.  Empty rows give 0, not inf.
        All arithmetic in the matrix dtype (fp32 for the interaction graphs)."""
        n_rows, n_cols = adj_mat.shape
        degree = np.asarray(adj_mat.sum(axis=1)).ravel()
        power = -0.5 if n_rows == n_cols else -1.0
        with np.errstate(divide='ignore'):
            scale = np.power(degree, power)
        scale[np.isinf(scale)] = 0.0
        left = sp.diags(scale)
        if n_rows == n_cols:
            return left.dot(adj_mat).dot(left)
        return left.dot(adj_mat)

    def convert_to_laplacian_mat(self, adj_mat):
        raise NotImplementedError
