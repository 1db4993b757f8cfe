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


def _dev_scalar(gout, like):
    """the upstream gradient as the kernels read it: a 0-dim f32 tensor ON THE DEVICE (a float(gout) would be a
    device-to-host sync in the middle of every backward pass -- the host could no longer run ahead of the device)"""
    if gout.device != like.device or gout.dtype != torch.float32:
        gout = gout.to(device=like.device, dtype=torch.float32)
    return gout.contiguous()


class _BprFn(torch.autograd.Function):
    """one launch forward (the mean is finished on the device), one backward"""

    @staticmethod
    def forward(ctx, u, p, n):
        d, w = _padded(ops.ROW_WIDTHS, "bpr_loss", u)
        u, p, n = (ops.pad_cols(t, w) for t in (u, p, n))
        loss = torch.empty((), dtype=torch.float32, device=u.device)
        coef = torch.empty(u.shape[0], dtype=torch.float32, device=u.device)
        ops.bpr_fwd(u, p, n, loss, coef)
        ctx.save_for_backward(u, p, n, coef)
        ctx.d = d
        return loss

    @staticmethod
    def backward(ctx, gout):
        u, p, n, coef = ctx.saved_tensors
        g = torch.empty((3,) + tuple(u.shape), dtype=torch.float32, device=u.device)
        ops.bpr_bwd(u, p, n, coef, _dev_scalar(gout, u), g[0], g[1], g[2])
        d = ctx.d
        return g[0][:, :d], g[1][:, :d], g[2][:, :d]


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


class _L2RegFn(torch.autograd.Function):
    """reg * sum_k ||emb_k||_F / rows_k over 1..4 blocks of rows: one launch forward, one backward (zero gradient where a
    norm is zero, as torch.norm's)."""
    MAX_BLOCKS = 4

    @staticmethod
    def forward(ctx, reg, *embs):
        xs = [e.contiguous() for e in embs]
        dev = xs[0].device
        norms = torch.empty(len(xs), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        ops.l2_reg_fwd(xs, reg, norms, loss)
        ctx.save_for_backward(norms, *xs)
        ctx.reg = float(reg)
        return loss

    @staticmethod
    def backward(ctx, gout):
        norms, *xs = ctx.saved_tensors
        gxs = [torch.empty_like(x) for x in xs]
        ops.l2_reg_bwd(xs, ctx.reg, norms, _dev_scalar(gout, norms), gxs)
        return (None, *gxs)


def l2_reg_loss(reg, *args):
    if args and len(args) <= _L2RegFn.MAX_BLOCKS and all(_on_hip_path(e) and e.numel() > 0 for e in args):
        return _L2RegFn.apply(float(reg), *args)
    emb_loss = 0
    for emb in args:
        if _on_hip_path(emb) and emb.numel() > 0:
            emb_loss = emb_loss + _L2RegFn.apply(1.0, emb)                     # = ||emb|| / rows
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
        # [dL/dv1 | dL/dv2 | loss (one double)]: the kernels ADD into all three, so all three start at zero -- one fill
        buf = torch.zeros(2 * n * w + 2, dtype=torch.float32, device=dev)
        g = buf[:2 * n * w].view(2, n, w)
        loss = buf[2 * n * w:].view(torch.float64)
        ws = ops.infonce_ws(n, w, dev)
        # forward and backward share every intermediate, so both are produced here with unit
        # upstream gradient and scaled in backward()
        ops.infonce_fwd_bwd(v1, v2, None, n, tau=temperature, loss_scale=1.0, loss=loss, g1=g[0], g2=g[1], ws=ws)
        ctx.save_for_backward(g)
        ctx.d = d
        return loss.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        g = g * gout
        return g[0][:, :ctx.d], g[1][:, :ctx.d], None


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
