#!/usr/bin/env python3
"""Host cost per call of the op-level tier's pieces (no device synchronisation inside the timed loops; tensors small enough
that the device is never the bottleneck): what a model file written the reference's way pays on the HOST per step for each
of this package's functions, beside two of torch's own ops for scale.

  python tools/oplevel_host_probe.py            -> table on stdout
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("PROBE_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (another build of the package: A/B)
from selfrec_amd import dropin, synth                                      # noqa: E402
from selfrec_amd.base.torch_interface import TorchGraphInterface            # noqa: E402
from selfrec_amd.data.ui_graph import Interaction                           # noqa: E402
from selfrec_amd.util import fastpath                                       # noqa: E402
from selfrec_amd.util.loss_torch import InfoNCE, bpr_loss, l2_reg_loss      # noqa: E402

DEV = torch.device("cuda", 0)
N = int(os.environ.get("PROBE_ITERS", 300))


def timed(label, fn, n=N):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t0) / n)
        torch.cuda.synchronize()
    print(f"{label:58s} {best * 1e6:8.1f} us / call")
    return best


def main():
    tu, ti, _, _, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    adj = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj).cuda()
    d, B = 64, 2048
    ego = torch.randn(U + I, d, device=DEV, requires_grad=True)
    u, p, q = (torch.randn(B, d, device=DEV, requires_grad=True) for _ in range(3))
    one = torch.ones((), device=DEV)
    print(f"# torch {torch.__version__}; {N} calls per loop, best of 3; B = {B}, d = {d}, graph {U} x {I}")
    timed("torch: a * b (elementwise, no grad)", lambda: torch.mul(u.detach(), p.detach()))
    timed("torch: (a * b).sum().backward()", lambda: torch.mul(u, p).sum().backward())
    timed("torch.sparse.mm(handle, x) forward", lambda: torch.sparse.mm(adj, ego.detach()))
    timed("torch.sparse.mm(handle, x) forward + backward", lambda: torch.sparse.mm(adj, ego).backward(ego.detach()))
    timed("bpr_loss forward", lambda: bpr_loss(u.detach(), p.detach(), q.detach()))
    timed("bpr_loss forward + backward", lambda: bpr_loss(u, p, q).backward(one))
    timed("l2_reg_loss(reg, u, p) forward", lambda: l2_reg_loss(1e-4, u.detach(), p.detach()))
    timed("l2_reg_loss(reg, u, p) forward + backward", lambda: l2_reg_loss(1e-4, u, p).backward(one))
    timed("l2_reg_loss(reg, u, p, q) forward + backward", lambda: l2_reg_loss(1e-4, u, p, q).backward(one))
    timed("InfoNCE forward (both gradients inside)", lambda: InfoNCE(u.detach(), p.detach(), 0.2))
    timed("InfoNCE forward + backward", lambda: InfoNCE(u, p, 0.2).backward(one))
    dropin.install(fuse=False)
    try:
        table = torch.nn.Parameter(torch.randn(U, d, device=DEV))
        idx = torch.randint(0, U, (B,), device=DEV)
        lists = ([int(x) for x in np.random.randint(0, U, B)],) * 3
        fastpath.register_batch(lists, [np.asarray(lists[0], dtype=np.int32)] * 3)
        timed("table[device int64 index of the caller's] (torch's own path)", lambda: table[idx])
        timed("table[registered python list] (fast path)", lambda: table[lists[0]])
        timed("table[idx] forward + backward", lambda: table[idx].backward(u.detach()))
        opt = torch.optim.Adam([table, torch.nn.Parameter(torch.randn(I, d, device=DEV))], lr=1e-3)
        for prm in opt.param_groups[0]["params"]:
            prm.grad = torch.randn_like(prm)
        timed("torch.optim.Adam.step() (fast path, two tables)", opt.step)
        timed("fastpath.register_batch (one pinned copy per batch)",
              lambda: fastpath.register_batch(lists, [np.asarray(lists[0], dtype=np.int32)] * 3), n=100)
    finally:
        dropin.uninstall()
    opt = torch.optim.Adam([table, torch.nn.Parameter(torch.randn(I, d, device=DEV))], lr=1e-3)
    for prm in opt.param_groups[0]["params"]:
        prm.grad = torch.randn_like(prm)
    timed("torch.optim.Adam.step() (torch's own foreach, two tables)", opt.step)


if __name__ == "__main__":
    main()
