#!/usr/bin/env python3
"""A/B the InfoNCE launch set (2 problems of n rows, d=64) over key splits and MFMA path.
Each configuration runs in its own process (the knobs are read once per process)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(HERE))
    from selfrec_amd import ops
    n, d = int(sys.argv[2]), 64
    g = torch.Generator(device="cuda").manual_seed(0)
    t1 = torch.randn((40000, d), device="cuda", generator=g) * 0.3
    t2 = t1 + torch.randn((40000, d), device="cuda", generator=g) * 0.1
    idx = torch.randperm(40000, device="cuda", generator=g)[:n].sort().values.int()
    g1, g2 = torch.zeros_like(t1), torch.zeros_like(t2)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = torch.empty(2 * ops.infonce_ws(n, d, "cuda").numel(), dtype=torch.uint8, device="cuda")
    prob = [(t1, t2, idx, n, None, g1, g2), (t2, t1, idx, n, None, g2, g1)]
    for _ in range(5):
        ops.infonce_multi(prob, d=d, tau=0.2, loss_scale=0.2, loss=loss, ws=ws)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(50):
        ops.infonce_multi(prob, d=d, tau=0.2, loss_scale=0.2, loss=loss, ws=ws)
    b.record(); torch.cuda.synchronize()
    print(f"{a.elapsed_time(b) / 50 * 1e3:8.2f} us", flush=True)
else:
    for n in (2048,):
        configs = [dict(SRH_NCE_LDS="0", SRH_NCE_QT="2", SRH_NCE_SPLITS="8")]
        for fin in (0, 1):
            for waves, qt, splits in ((8, 1, 8), (8, 1, 16), (4, 2, 8), (8, 2, 8), (8, 2, 16)):
                configs.append(dict(SRH_NCE_LDS="1", SRH_NCE_FINISH=str(fin), SRH_NCE_WAVES=str(waves), SRH_NCE_QT=str(qt),
                                    SRH_NCE_SPLITS=str(splits)))
        for cfg in configs:
            out = subprocess.run([sys.executable, __file__, "child", str(n)], env=dict(os.environ, **cfg), capture_output=True, text=True)
            tag = " ".join(f"{k[8:].lower()}={v}" for k, v in cfg.items())
            print(f"n={n} bf16x3 {tag}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
