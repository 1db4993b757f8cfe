#!/usr/bin/env python3
"""SURVEY.md 8(f-1) / VERDICT r03 next #7: time to the first training step at a shape, stage by stage -- the text files on
disk (the reference's input format, data/loader.py:26-32) -> native load + id map -> Interaction (graph products the models
read) -> device CSR + column-class order + SpMM plan -> trainer (tables, workspaces, XCD calibration) -> first sampled epoch
-> upload -> capture + first step.

    python tools/startup_probe.py [--shape 1m-500k --emb 128]        (default: yelp2018, d = 64)

The synthetic graph is generated and written to a temporary directory first (not counted: a user's files exist)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfrec_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="yelp2018")
    ap.add_argument("--emb", type=int, default=64)
    ap.add_argument("--layers", type=int, default=3)
    args = ap.parse_args()
    t0 = time.perf_counter()
    tu, ti, su, si, U, I = synth.make_dataset(args.shape, seed=2024)
    su, si = su[:200_000], si[:200_000]
    print(f"# {args.shape}: {U} users x {I} items, {len(tu)} train interactions (generated in {time.perf_counter() - t0:.1f} s, not counted)")
    stages = []

    def stage(name, t_begin):
        torch.cuda.synchronize()
        stages.append((name, time.perf_counter() - t_begin))
        print(f"  {name:<62} {stages[-1][1]:8.3f} s", flush=True)
        return time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        for name, (a, b) in (("train.txt", (tu, ti)), ("test.txt", (su, si))):
            import pandas as pd                  # (the C writer: 40 M lines in seconds; names are opaque strings to the loader)
            pd.DataFrame({"u": a, "i": b, "w": np.ones(len(a), dtype=np.int64)}).to_csv(
                os.path.join(tmp, name), sep=" ", header=False, index=False)
        print(f"# files written in {time.perf_counter() - t0:.1f} s (not counted)")
        from selfrec_amd.data.loader import FileIO
        from selfrec_amd.data.ui_graph import Interaction
        from selfrec_amd.engine import FusedTrainer
        torch.cuda.init()
        torch.zeros(1, device="cuda")
        t_all = t = time.perf_counter()
        train = FileIO.open_data_set(os.path.join(tmp, "train.txt"), "graph")       # (what selfrec_amd.SELFRec does)
        test = FileIO.open_data_set(os.path.join(tmp, "test.txt"), "graph")
        t = stage("open_data_set (lazy handles)", t)
        data = Interaction({}, train, test)
        t = stage("Interaction (native parse + first-appearance ids of both files, id arrays)", t)
        g = data.device_graph(torch.device("cuda"))
        t = stage("device graph (normalised CSR, column-class order, SpMM plans)", t)
        torch.manual_seed(0)
        tr = FusedTrainer(data, args.emb, model="XSimGCL", n_layers=args.layers, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2,
                          layer_cl=1, batch_size=2048, use_graph=True,
                          sampler_seed=None if os.environ.get("STARTUP_NO_EARLY_SAMPLING") else 1)
        t = stage("FusedTrainer (tables, workspaces, replanned CSR, XCD calibration; the first epoch is being sampled on a thread)", t)
        tr.seed_sampler(1)
        host = tr.sample_epoch_host()
        t = stage("first epoch: what of its sampling was not hidden under the construction", t)
        tr.upload_epoch(host)
        t = stage("epoch upload", t)
        tr.step()
        t = stage("first step (warm-up step + hipGraph capture + replay)", t)
        for _ in range(10):
            tr.step()
        t = stage("ten more steps", t)
        total = time.perf_counter() - t_all
    print(f"time to the first trained step: {sum(s for _, s in stages[:-1]):.3f} s; largest stage: "
          f"{max(stages[:-1], key=lambda x: x[1])[0]} ({max(s for _, s in stages[:-1]):.3f} s); with ten more steps {total:.3f} s")


if __name__ == "__main__":
    main()
