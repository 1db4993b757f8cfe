"""The CPU baseline leg: the reference's own train() step when a staged checkout is present, else the oracle port."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3


def reference_step(args, seconds, stage=None):
    """The reference's own model/graph/XSimGCL.py train() step timed by `python -m benchlib.ref_step` in a process of its
    own, when a checkout of the reference is staged on this box (`_refstage/`: client code for the drop-in tests, never
    tracked; /root/reference itself does not exist where bench.py runs).  None when there is none, or when the run failed
    (the caller then times the port)."""
    import subprocess
    stage = stage or os.path.join(REPO, "_refstage")
    if not os.path.isfile(os.path.join(stage, "model", "graph", "XSimGCL.py")):
        return None
    cmd = [sys.executable, "-m", "benchlib.ref_step", "--ref", stage, "--shape", args.shape, "--seed", str(args.seed),
           "--layers", str(args.layers), "--emb", str(args.emb), "--batch", str(args.batch), "--tau", str(args.tau),
           "--seconds", str(seconds)]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    try:
        out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=60 + 20 * seconds)
        rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:                   # noqa: BLE001  (a baseline leg never costs the line)
        print(f"[bench] reference step not timed ({type(e).__name__}: {str(e)[:200]}); timing the port", file=sys.stderr)
        return None
    return {"value": rec["pairs_per_s"], "unit": "pairs/s", "cores": rec["cores"], "host_cpus": rec["host_cpus"],
            "kind": "reference",
            "sample": f"{rec['steps']} steps of the reference's own XSimGCL.train() (model/graph/XSimGCL.py:23-43, its python "
                      f"sampler inside) on the same graph, torch {rec['torch']} CPU: {rec['ms_per_step']} ms per step",
            "ms_per_step": rec["ms_per_step"]}


def cpu_baseline(args, raw, seconds):
    """`cpu_baseline` of the line: the reference itself when it is staged on this box (kind "reference"), else the oracle
    port of its step (kind "port"); with the reference timed, a short sample of the port rides along as `port`."""
    ref = reference_step(args, seconds) if args.model == "XSimGCL" else None
    if ref is not None:
        port = port_baseline(args, raw, min(seconds, 6.0))
        ref["port"] = {"value": port["value"], "sample": port["sample"]}
        return ref
    return port_baseline(args, raw, seconds)


def port_baseline(args, raw, seconds):
    """The CPU oracle's XSimGCL step (torch-CPU fp32, python sampler) on this host's cores."""
    import random
    from oracle import selfrec_oracle as O
    tu, ti, su, si, U, I = raw
    torch.manual_seed(args.seed)
    kw = dict(n_layers=args.layers, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=args.tau, layer_cl=1,
              batch_size=args.batch)
    tr = O.OracleTrainer(args.model, tu, ti, U, I, args.emb, **kw)
    smp = O.PairwiseSampler(tu, ti, U, I)
    random.seed(args.seed)
    t0 = time.time()
    it = smp.epoch(args.batch)
    first = next(it)                                  # includes the once-per-epoch python shuffle
    t_first = time.time() - t0
    tr.step(*first)                                   # warm-up
    n, t_steps, t_sample = 0, 0.0, 0.0
    t_begin = time.time()
    while time.time() - t_begin < seconds or n < 3:
        t1 = time.time(); batch = next(it); t2 = time.time()
        tr.step(*batch)
        t3 = time.time()
        t_sample += t2 - t1; t_steps += t3 - t2; n += 1
    per_batch_sample = t_sample / n
    shuffle_amortised = max(t_first - per_batch_sample, 0.0) / max(1, (len(tu) + args.batch - 1) // args.batch)
    step_s = t_steps / n + per_batch_sample + shuffle_amortised
    return {"value": round(args.batch / step_s, 1), "unit": "pairs/s", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{n} XSimGCL steps (B={args.batch}, L={args.layers}) of the oracle on the same graph, "
                      f"{t_steps / n * 1e3:.0f} ms compute + {per_batch_sample * 1e3:.1f} ms python sampling per step, "
                      f"shuffle {t_first:.2f} s amortised over the epoch"}
