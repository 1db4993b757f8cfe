"""Torch-CPU stand-in for selfrec_amd.dist.HipBackend -- TEST INFRASTRUCTURE ONLY.

Lets the world_size-2 gloo tests exercise ShardedTrainer's partition / column rewrite /
collective / owner-scatter logic on a machine without GPUs.  Numerics come from the CPU oracle;
the product never constructs this class."""
import numpy as np
import scipy.sparse as sp
import torch

from oracle import selfrec_oracle as O


class CpuBackend:
    device = torch.device("cpu")

    def csr(self, indptr, indices, vals, shape):
        m = sp.csr_matrix((np.asarray(vals), np.asarray(indices), np.asarray(indptr)), shape=shape)
        return O.to_torch_sparse(m)

    def spmm(self, csr, x, out, *, perturb=None, mean=None, add=None, add_scale=None, alpha=1.0):
        y = torch.sparse.mm(csr, x)
        if add or alpha != 1.0:
            y = y * alpha
            for a, s in zip(add or [], add_scale or []):
                y = y + s * a
        if perturb is not None:
            eps, noise, _, _ = perturb
            assert noise is not None, "CPU stand-in needs injected noise"
            y = O.perturb_(y, noise, eps)
        out.copy_(y)
        if mean is not None:
            prev, div, mean_out = mean
            mean_out.copy_(torch.stack(list(prev) + [out], dim=1).sum(1) / div)

    def bpr_l2(self, u, p, n, ru, rp, rn, reg_coef, include_neg, losses):
        same = ru is u
        leaves = [t.detach().clone().requires_grad_() for t in ((u, p, n) if same else (u, p, n, ru, rp, rn))]
        a, b, c = leaves[:3]
        ra, rb, rc = (a, b, c) if same else leaves[3:]
        bpr = O.bpr_loss(a, b, c)
        reg = O.l2_reg_loss(reg_coef, *([ra, rb, rc] if include_neg else [ra, rb]))
        (bpr + reg).backward()
        losses[0] += bpr.item(); losses[1] += reg.item()
        g = [t.grad if t.grad is not None else torch.zeros_like(t) for t in leaves]
        return tuple(g) if not same else (g[0], g[1], g[2], g[0], g[1], g[2])

    def infonce(self, v1, v2, tau, scale, loss):
        a, b = v1.detach().clone().requires_grad_(), v2.detach().clone().requires_grad_()
        l = scale * O.info_nce(a, b, tau)
        l.backward()
        loss += l.item()
        return a.grad, b.grad

    def adam(self, p, g, m, v, step, lr):
        pn, gn, mn, vn = p.numpy(), g.numpy(), m.numpy(), v.numpy()
        O.adam_step(pn, gn, mn, vn, step, lr)

    def make_sampler(self, edge_u, edge_i, n_users, n_items):
        from selfrec_amd import ops
        return ops.Sampler(edge_u, edge_i, n_users, n_items)      # host-only C++ code: runs without a GPU
