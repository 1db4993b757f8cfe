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
        return ops.spmm(handle.csr, x)

    @staticmethod
    def backward(ctx, grad_out):
        g = grad_out.contiguous()
        return None, ops.spmm(ctx.handle.transposed().csr, g)


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
    def _indices(self):
        if getattr(self, "_coo_idx", None) is None:
            self._coo_idx = self.to_sparse_coo()._indices()
        return self._coo_idx

    def _values(self):
        return self.csr.vals

    def transposed(self):
        if self._symmetric:
            return self
        if self._t is None:
            if self._scipy is None:
                raise ops.SelfrecHipError("transpose of a device-only non-symmetric adjacency is not available")
            self._t = SparseAdjHandle(ops.DeviceCSR.from_scipy(self._scipy.T.tocsr(), device=self.device), False)
            self._t._t = self
        return self._t

    def to_sparse_coo(self):
        """Escape hatch for code that wants the reference's COO tensor (BUIR/MixGCF style
        ``_indices()/_values()`` access is outside the accelerated path)."""
        indptr = self.csr.indptr.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device), indptr[1:] - indptr[:-1])
        idx = torch.stack([rows, self.csr.indices.to(torch.int64)])
        return torch.sparse_coo_tensor(idx, self.csr.vals, tuple(self.shape))

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
        csr = X.tocsr().astype(np.float32)
        csr.sort_indices()
        return SparseAdjHandle(ops.DeviceCSR.from_scipy(csr), _is_symmetric(csr), scipy_source=csr)
