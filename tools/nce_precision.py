#!/usr/bin/env python3
"""InfoNCE forward + backward of the library (srh_infonce_fwd_bwd, whatever arithmetic it was built / set to) against
the same expression (loss_torch.py:35-50) evaluated in float64 by ATen on the device: relative error of the loss and of
both gradients (max |diff| / max |want|), at n = 2048 rows, d = 64 / 128, tau = 0.2 and 0.05, views that are noisy
copies of each other (the training regime) and independent views (the worst case for the softmax's dynamic range)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import ops  # noqa: E402


def reference(v1, v2, tau):
    a, b = v1.double().requires_grad_(), v2.double().requires_grad_()
    x, y = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(b, dim=1)
    loss = -torch.diag(torch.nn.functional.log_softmax(x @ y.T / tau, dim=1)).mean()
    loss.backward()
    return loss.item(), a.grad, b.grad


def main():
    torch.manual_seed(0)
    dev = "cuda"
    modes = [m for m in ("split", "f32") if m in ops.NCE_PRECISIONS] or list(ops.NCE_PRECISIONS)
    for d in (64, 128):
        for tau in (0.2, 0.05):
            for kind in ("correlated", "independent"):
                base = torch.randn(2048, d, device=dev)
                v1 = base + 0.3 * torch.randn(2048, d, device=dev)
                v2 = (base if kind == "correlated" else torch.randn(2048, d, device=dev)) + 0.3 * torch.randn(2048, d, device=dev)
                want, g1w, g2w = reference(v1, v2, tau)
                line = f"d={d} tau={tau} {kind:11s}"
                for mode in modes:
                    ops.set_infonce_precision(mode)
                    loss = torch.zeros(1, dtype=torch.float64, device=dev)
                    g1, g2 = torch.zeros_like(v1), torch.zeros_like(v2)
                    ws = ops.infonce_ws(2048, d, dev)
                    ops.infonce_fwd_bwd(v1, v2, None, 2048, tau=tau, loss_scale=1.0, loss=loss, g1=g1, g2=g2, ws=ws)
                    e_l = abs(loss.item() - want) / abs(want)
                    e_1 = ((g1.double() - g1w).abs().max() / g1w.abs().max()).item()
                    e_2 = ((g2.double() - g2w).abs().max() / g2w.abs().max()).item()
                    line += f" | {mode}: loss {e_l:.1e} g1 {e_1:.1e} g2 {e_2:.1e}"
                print(line, flush=True)
    ops.set_infonce_precision(modes[0])


if __name__ == "__main__":
    main()
