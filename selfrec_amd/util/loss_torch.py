"""Differentiable losses with the signatures of reference util/loss_torch.py
(``bpr_loss`` :6-10, ``l2_reg_loss`` :18-22, ``InfoNCE`` :35-50), computed by fused HIP
forward+backward kernels.

What runs where:
  * 2-D fp32 HIP tensors -- what every model file of the reference passes once ``.cuda()`` has run -- ALWAYS take the
    HIP kernels; a missing library or device raises (``ops.SelfrecHipError``), nothing is substituted.  Any
    ``embedding.size`` (base/recommender.py:16): rows are zero-padded to the next width the kernels serve -- zero
    columns change no inner product, norm or F.normalize result and receive exactly zero gradient -- BPR up to
    256 columns, InfoNCE up to 256 (above 128: the split path only), the regulariser any size.
  * CPU tensors, other dtypes, other ranks: the reference's own torch expression (the arithmetic of loss_torch.py:6-50
    restated below), so code that calls these functions off the device -- unit tests of a model file, a CPU dry run --
    behaves as it does with the reference.  This is not a fallback for the HIP path: device fp32 input never reaches it.
  * InfoNCE or BPR on HIP rows wider than 256 columns: the same torch expression on the device (rocBLAS +
    ATen), announced once with a RuntimeWarning -- no kernel of this package serves those widths yet.
"""
import warnings

import torch
import torch.nn.functional as F

from .. import ops


_warned_wide = []


def _on_hip_path(*tensors):
    """True when every tensor is what the kernels take (2-D fp32 HIP); False -> the torch expression."""
    return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 for t in tensors)


def _padded(width_set, what, *tensors):
    d = int(tensors[0].shape[1])
    w = ops.padded_width(d, width_set)
    if w is None:
        raise ops.SelfrecHipError(f"{what}: rows of {d} columns -- the HIP kernels serve up to {width_set[-1]}")
    return d, w


class _BprFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, p, n):
        d, w = _padded(ops.ROW_WIDTHS, "bpr_loss", u)
        u, p, n = (ops.pad_cols(t, w) for t in (u, p, n))
        rows = u.shape[0]
        loss_sum = torch.zeros(1, dtype=torch.float64, device=u.device)
        coef = torch.empty(rows, dtype=torch.float32, device=u.device)
        ops.bpr_fwd(u, p, n, loss_sum, coef)
        ctx.save_for_backward(u, p, n, coef)
        ctx.d = d
        return (loss_sum / rows).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, gout):
        u, p, n, coef = ctx.saved_tensors
        gu, gp, gn = torch.empty_like(u), torch.empty_like(p), torch.empty_like(n)
        # the upstream gradient stays ON THE DEVICE (folded into the per-row coefficients): a float(gout) here is a
        # device-to-host sync in the middle of every backward pass -- the host could no longer run ahead of the device
        ops.bpr_bwd(u, p, n, coef * (gout.to(torch.float32) / u.shape[0]), 1.0, gu, gp, gn)
        d = ctx.d
        return gu[:, :d], gp[:, :d], gn[:, :d]


def _announce_wide(what, d, limit):
    if what not in _warned_wide:
        _warned_wide.append(what)
        warnings.warn(f"{what} on rows of {d} columns: the HIP kernels serve up to {limit}; evaluating the reference's "
                      f"expression (util/loss_torch.py) with ATen on the device", RuntimeWarning, stacklevel=3)


def bpr_loss(user_emb, pos_item_emb, neg_item_emb):
    wide = _on_hip_path(user_emb) and int(user_emb.shape[1]) > ops.ROW_WIDTHS[-1]
    if wide:
        _announce_wide("bpr_loss", int(user_emb.shape[1]), ops.ROW_WIDTHS[-1])
    if wide or not _on_hip_path(user_emb, pos_item_emb, neg_item_emb):
        # loss_torch.py:6-10
        pos_score = torch.mul(user_emb, pos_item_emb).sum(dim=1)
        neg_score = torch.mul(user_emb, neg_item_emb).sum(dim=1)
        loss = -torch.log(10e-6 + torch.sigmoid(pos_score - neg_score))
        return torch.mean(loss)
    return _BprFn.apply(user_emb, pos_item_emb, neg_item_emb)


class _FrobNormFn(torch.autograd.Function):
    """||x||_F of a 2-D block (torch.norm(emb, p=2) in the reference); zero gradient at 0."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        acc = torch.zeros(1, dtype=torch.float64, device=x.device)
        ops.sumsq(x, acc)
        norm = torch.sqrt(acc).to(torch.float32).reshape(())
        ctx.save_for_backward(x, norm)
        return norm

    @staticmethod
    def backward(ctx, gout):
        x, norm = ctx.saved_tensors
        # gout / ||x|| (0 at ||x|| = 0) as a DEVICE scalar: no float(...) -- every one of those is a device-to-host sync
        # that stops the host from running ahead (four per step of the reference's XSimGCL.py before round 4)
        coef = torch.where(norm > 0, gout.to(torch.float32) / norm, torch.zeros_like(norm))
        return x * coef


def l2_reg_loss(reg, *args):
    emb_loss = 0
    for emb in args:
        if _on_hip_path(emb):
            emb_loss = emb_loss + _FrobNormFn.apply(emb) / emb.shape[0]
        else:
            emb_loss = emb_loss + torch.norm(emb, p=2) / emb.shape[0]          # loss_torch.py:18-22
    return emb_loss * reg


class _InfoNceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v1, v2, temperature):
        d, w = _padded(ops.NCE_WIDTHS, "InfoNCE", v1)
        v1, v2 = ops.pad_cols(v1, w), ops.pad_cols(v2, w)
        n = v1.shape[0]
        dev = v1.device
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        g1, g2 = torch.zeros_like(v1), torch.zeros_like(v2)
        ws = ops.infonce_ws(n, w, dev)
        # forward and backward share every intermediate, so both are produced here with unit
        # upstream gradient and scaled in backward()
        ops.infonce_fwd_bwd(v1, v2, None, n, tau=temperature, loss_scale=1.0, loss=loss, g1=g1, g2=g2, ws=ws)
        ctx.save_for_backward(g1[:, :d], g2[:, :d])
        return loss.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, gout):
        g1, g2 = ctx.saved_tensors
        return g1 * gout, g2 * gout, None


def _infonce_expression(view1, view2, temperature, b_cos=True):
    """loss_torch.py:35-50"""
    if b_cos:
        view1, view2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
    pos_score = (view1 @ view2.T) / temperature
    score = torch.diag(F.log_softmax(pos_score, dim=1))
    return -score.mean()


def InfoNCE(view1, view2, temperature: float, b_cos: bool = True):
    if not _on_hip_path(view1, view2):
        return _infonce_expression(view1, view2, temperature, b_cos)
    if not b_cos:
        raise ops.SelfrecHipError("InfoNCE(b_cos=False) has no caller in the reference and no HIP kernel")
    if view1.shape != view2.shape:
        raise ops.SelfrecHipError("InfoNCE: the two views must have the same shape")
    if int(view1.shape[1]) > ops.NCE_WIDTHS[-1]:
        _announce_wide("InfoNCE", int(view1.shape[1]), ops.NCE_WIDTHS[-1])
        return _infonce_expression(view1, view2, temperature, True)
    return _InfoNceFn.apply(view1, view2, float(temperature))
