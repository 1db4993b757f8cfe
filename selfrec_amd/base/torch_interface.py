"""Adjacency upload + SpMM hook: the drop-in for reference base/torch_interface.py:8-13.

The reference turns a scipy matrix into an un-coalesced COO tensor (int64 indices) and
model files later call raw ``torch.sparse.mm(adj, dense)`` on it (LightGCN.py:72,
XSimGCL.py:88, SimGCL.py:85, SGL.py:104-108).  Here ``convert_sparse_mat_to_tensor``
returns a ``SparseAdjHandle``: a device CSR (int32 structure, 2.5x fewer bytes than
COO/int64) that

  * answers ``.cuda()`` / ``.to(...)`` with itself (it already lives in HBM),
  * implements ``__torch_function__`` so the unmodified ``torch.sparse.mm(handle, x)`` call
    dispatches to the hand-written HIP SpMM through an ``autograd.Function`` whose backward
    is the same kernel on the transposed matrix (the normalised bipartite adjacency is
    symmetric, so the transpose is the matrix itself; a general input gets an explicit
    transposed copy).
"""
import numpy as np
import torch

from .. import ops


class _SpmmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle, dense):
        ctx.handle = handle
        x = dense.contiguous()
        if x.dtype != torch.float32 or not x.is_cuda:
            raise ops.SelfrecHipError("torch.sparse.mm(SparseAdjHandle, x): x must be an fp32 HIP tensor")
        return ops.spmm_any(handle.csr, x)        # (any embedding.size: base/recommender.py:16)

    @staticmethod
    def backward(ctx, grad_out):
        g = grad_out.contiguous()
        return None, ops.spmm_any(ctx.handle.transposed().csr, g)


class SparseAdjHandle:
    """Device-resident sparse matrix standing in for the reference's sparse COO tensor."""

    def __init__(self, csr: "ops.DeviceCSR", symmetric: bool, scipy_source=None):
        self.csr = csr
        self.shape = torch.Size(csr.shape)
        self._symmetric = symmetric
        self._scipy = scipy_source
        self._t = None

    # --- tensor-like surface used by model files -----------------------------------
    def cuda(self, *args, **kwargs):
        return self

    def to(self, *args, **kwargs):
        return self

    @property
    def is_sparse(self):
        return True

    @property
    def device(self):
        return self.csr.vals.device

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def _nnz(self):
        return self.csr.nnz

    # BUIR.py:118-127 / MixGCF.py:84-94 drop entries of the adjacency themselves: they read the COO parts of the
    # tensor and build a new torch sparse tensor, which then multiplies through torch (outside the accelerated path)
    def _coo(self):
        """(2, nnz) int64 indices in (row, column) order and the permutation that puts the CSR's entries in that order
        (the resident graph stores long rows by column class: data/device_graph.py)."""
        if getattr(self, "_coo_idx", None) is None:
            indptr = self.csr.indptr.to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device), indptr[1:] - indptr[:-1])
            cols = self.csr.indices.to(torch.int64)
            perm = torch.argsort(rows * self.shape[1] + cols)
            self._coo_idx, self._coo_perm = torch.stack([rows[perm], cols[perm]]), perm
        return self._coo_idx, self._coo_perm

    def _indices(self):
        return self._coo()[0]

    def _values(self):
        return self.csr.vals[self._coo()[1]]

    def dropout(self, keep, scale=1.0):
        """Entry-wise dropped copy as a VALUE ARRAY over this structure: entry k (in (row, column) order, the order
        of ``_indices()``) survives where ``keep[k]`` and is multiplied by ``scale`` -- what BUIR.py:118-127 /
        MixGCF.py:84-94 build as a new torch COO tensor.  The result multiplies on the HIP SpMM, backward included
        (its transpose is the same structure with mirrored values)."""
        idx, perm = self._coo()
        # the mask crosses the link as it is (one byte per entry for a bool mask) and becomes fp32 on the device: a
        # host-side bool -> float conversion of 2.5 M entries costs up to 30 ms per call on the GPU box's host
        keep = torch.as_tensor(keep).to(device=self.device).to(dtype=torch.float32)
        if keep.numel() != self.csr.nnz:
            raise ops.SelfrecHipError(f"dropout: mask has {keep.numel()} entries, the matrix {self.csr.nnz}")
        vals = torch.zeros_like(self.csr.vals)
        vals[perm] = self.csr.vals[perm] * keep * float(scale)
        out = SparseAdjHandle(self.csr.with_values(vals), symmetric=False)
        if self._symmetric or getattr(self, "_mirror", None) is not None:
            out._mirror = self._mirror_perm()
        out._coo_idx, out._coo_perm = idx, perm
        return out

    def _mirror_perm(self):
        """mirror[p] = storage position of entry (j, i) for the entry (i, j) stored at p (structurally symmetric
        matrices only): A^T has this structure with values vals[mirror]."""
        if getattr(self, "_mirror", None) is None:
            idx, perm = self._coo()
            n = self.shape[1]
            keys = idx[0] * n + idx[1]                       # ascending: idx is in (row, column) order
            where = torch.searchsorted(keys, idx[1] * n + idx[0])
            if where.numel() and (int(where.max()) >= keys.numel() or not torch.equal(keys[where], idx[1] * n + idx[0])):
                raise ops.SelfrecHipError("transpose by mirroring needs a structurally symmetric matrix")
            mirror = torch.empty_like(perm)
            mirror[perm] = perm[where]
            self._mirror = mirror
        return self._mirror

    def transposed(self):
        if self._symmetric:
            return self
        if self._t is None and getattr(self, "_mirror", None) is not None:
            self._t = SparseAdjHandle(self.csr.with_values(self.csr.vals[self._mirror]), symmetric=False)
            self._t._t, self._t._mirror = self, self._mirror
        if self._t is None:
            if self._scipy is None:
                raise ops.SelfrecHipError("transpose of a device-only non-symmetric adjacency is not available")
            self._t = SparseAdjHandle(ops.DeviceCSR.from_scipy(self._scipy.T.tocsr(), device=self.device), False)
            self._t._t = self
        return self._t

    def to_sparse_coo(self):
        """Escape hatch for code that wants the reference's COO tensor (BUIR/MixGCF style
        ``_indices()/_values()`` access is outside the accelerated path)."""
        return torch.sparse_coo_tensor(self._indices(), self._values(), tuple(self.shape))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.sparse.mm, torch.mm, torch.matmul, torch.spmm) and len(args) == 2 \
                and isinstance(args[0], SparseAdjHandle) and isinstance(args[1], torch.Tensor):
            return _SpmmFn.apply(args[0], args[1])
        raise ops.SelfrecHipError(f"SparseAdjHandle does not support {getattr(func, '__name__', func)}; "
                                  f"call .to_sparse_coo() for a plain torch sparse tensor")


def _is_symmetric(mat) -> bool:
    if mat.shape[0] != mat.shape[1]:
        return False
    diff = (mat - mat.T).tocsr()
    return diff.nnz == 0 or float(np.abs(diff.data).max()) == 0.0


class TorchGraphInterface:
    @staticmethod
    def convert_sparse_mat_to_tensor(X):
        if isinstance(X, SparseAdjHandle):     # already on device (dropped views)
            return X
        owner = getattr(X, "_srh_owner", None)
        owner = owner() if owner is not None else None
        if owner is not None and ops.gpu_available():
            # data.norm_adj itself: the Interaction's resident device graph IS this matrix (normalised on device,
            # bit-identical values -- tests/test_gpu_kernels.py) with the row / column-class SpMM schedule
            return SparseAdjHandle(owner.device_graph().adj, symmetric=True, scipy_source=X)
        csr = X.tocsr().astype(np.float32)
        csr.sort_indices()
        return SparseAdjHandle(ops.DeviceCSR.from_scipy(csr), _is_symmetric(csr), scipy_source=csr)
