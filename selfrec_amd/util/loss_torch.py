"""Differentiable losses with the signatures of reference util/loss_torch.py
(``bpr_loss`` :6-10, ``l2_reg_loss`` :18-22, ``InfoNCE`` :35-50), computed by fused HIP
forward+backward kernels.  Inputs are fp32 HIP tensors of shape (rows, d); anything else
raises -- the HIP path is the only path.
"""
import torch

from .. import ops


def _check(*tensors):
    for t in tensors:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
            raise ops.SelfrecHipError("loss kernels take 2-D fp32 HIP tensors (no CPU fallback)")


class _BprFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, p, n):
        u, p, n = u.contiguous(), p.contiguous(), n.contiguous()
        rows = u.shape[0]
        loss_sum = torch.zeros(1, dtype=torch.float64, device=u.device)
        coef = torch.empty(rows, dtype=torch.float32, device=u.device)
        ops.bpr_fwd(u, p, n, loss_sum, coef)
        ctx.save_for_backward(u, p, n, coef)
        return (loss_sum / rows).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, gout):
        u, p, n, coef = ctx.saved_tensors
        gu, gp, gn = torch.empty_like(u), torch.empty_like(p), torch.empty_like(n)
        ops.bpr_bwd(u, p, n, coef, float(gout) / u.shape[0], gu, gp, gn)
        return gu, gp, gn


def bpr_loss(user_emb, pos_item_emb, neg_item_emb):
    _check(user_emb, pos_item_emb, neg_item_emb)
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
        n = float(norm)
        g = torch.empty_like(x)
        ops.axpby(float(gout) / n if n > 0.0 else 0.0, x, 0.0, g)
        return g


def l2_reg_loss(reg, *args):
    emb_loss = 0
    for emb in args:
        _check(emb)
        emb_loss = emb_loss + _FrobNormFn.apply(emb) / emb.shape[0]
    return emb_loss * reg


class _InfoNceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v1, v2, temperature):
        v1, v2 = v1.contiguous(), v2.contiguous()
        n, d = v1.shape
        dev = v1.device
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        g1, g2 = torch.zeros_like(v1), torch.zeros_like(v2)
        ws = ops.infonce_ws(n, d, dev)
        # forward and backward share every intermediate, so both are produced here with unit
        # upstream gradient and scaled in backward()
        ops.infonce_fwd_bwd(v1, v2, None, n, tau=temperature, loss_scale=1.0, loss=loss, g1=g1, g2=g2, ws=ws)
        ctx.save_for_backward(g1, g2)
        return loss.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, gout):
        g1, g2 = ctx.saved_tensors
        return g1 * gout, g2 * gout, None


def InfoNCE(view1, view2, temperature: float, b_cos: bool = True):
    _check(view1, view2)
    if not b_cos:
        raise ops.SelfrecHipError("InfoNCE(b_cos=False) has no caller in the reference and no HIP kernel")
    if view1.shape != view2.shape:
        raise ops.SelfrecHipError("InfoNCE: the two views must have the same shape")
    return _InfoNceFn.apply(view1, view2, float(temperature))
