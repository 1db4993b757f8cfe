#!/usr/bin/env python3
"""is bench.py's `cpu_baseline` (kind "port": oracle/selfrec_oracle.py) the reference's speed?

Times, in ONE process on this host's cores with the same torch thread count, on bench.py's Yelp2018-shape graph:
  (1) the reference's OWN model/graph/XSimGCL.py train() loop (imported from an untracked staging copy of the reference,
      `--ref _refstage`; numba stubbed, .cuda() patched to identity: SURVEY.md 8c) -- seconds per step measured between
      consecutive batches of its own next_batch_pairwise generator, so python sampling is inside, like in its train();
  (2) the oracle's step (what bench.py times), sampling included the same way.
Prints both, the ratio and the core counts.  /root/reference does not exist on the GPU box: without the staging copy this
says so and exits 0 (nothing to compare)."""
import argparse
import importlib
import os
import random
import sys
import tempfile
import time
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=os.path.join(REPO, "_refstage"))
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--layers", type=int, default=3)
    args = ap.parse_args()
    import bench
    bargs = bench.parse([])
    from selfrec_amd import synth
    tu, ti, su, si, U, I = synth.make_dataset(bargs.shape, seed=bargs.seed)
    print(f"# host: {os.cpu_count()} logical CPUs, torch threads {torch.get_num_threads()}, torch {torch.__version__}")
    print(f"# graph: {U} users x {I} items, {len(tu)} train interactions; XSimGCL L={args.layers} d=64 B=2048 tau=0.2")

    # ---- (2) the port first (it does not touch sys.modules)
    from oracle import selfrec_oracle as O
    kw = dict(n_layers=args.layers, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1, batch_size=2048)
    torch.manual_seed(bargs.seed)
    tr = O.OracleTrainer("XSimGCL", tu, ti, U, I, 64, **kw)
    smp = O.PairwiseSampler(tu, ti, U, I)
    random.seed(bargs.seed)
    it = smp.epoch(2048)
    tr.step(*next(it))                         # warm-up (+ the epoch's shuffle)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(*next(it))
    port = (time.perf_counter() - t0) / args.steps
    print(f"port      (oracle.OracleTrainer.step + PairwiseSampler): {port * 1e3:8.1f} ms/step = {2048 / port:9.1f} pairs/s")

    def port_again():
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.step(*next(it))
        return (time.perf_counter() - t0) / args.steps

    # ---- (1) the reference's own file
    if not os.path.isfile(os.path.join(args.ref, "model", "graph", "XSimGCL.py")):
        print(f"reference: no staged checkout at {args.ref} -- nothing to compare against on this box")
        return
    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = numba
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.dont_write_bytecode = True
    for name in [m for m in sys.modules if m.split(".")[0] in ("base", "data", "util", "model")]:
        del sys.modules[name]
    sys.path.insert(0, os.path.abspath(args.ref))
    mod = importlib.import_module("model.graph.XSimGCL")
    assert os.path.abspath(mod.__file__).startswith(os.path.abspath(args.ref)), mod.__file__
    from util.conf import ModelConf
    stamps = []
    real = mod.next_batch_pairwise

    def batches(data, bs, n_negs=1):
        for k, b in enumerate(real(data, bs, n_negs)):
            stamps.append(time.perf_counter())
            if k == args.steps + 1:
                return
            yield b
    mod.next_batch_pairwise = batches
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            lines = ["training.set: ./train.txt", "test.set: ./test.txt", "model:", "  name: XSimGCL", "  type: graph",
                     "item.ranking.topN: [10,20]", "embedding.size: 64", "max.epoch: 1", "batch.size: 2048",
                     "learning.rate: 0.001", "reg.lambda: 0.0001", "output: ./results/", "XSimGCL:",
                     f"  n_layer: {args.layers}", "  l_star: 1", "  lambda: 0.2", "  eps: 0.2", "  tau: 0.2"]
            with open("XSimGCL.yaml", "w") as f:
                f.write("\n".join(lines) + "\n")
            conf = ModelConf("XSimGCL.yaml")
            torch.manual_seed(bargs.seed)
            random.seed(bargs.seed)
            model = mod.XSimGCL(conf, [list(t) for t in synth.as_triples(tu, ti)], [list(t) for t in synth.as_triples(su, si)])
            model.fast_evaluation = lambda epoch: None
            try:
                model.train()
            except AttributeError as e:          # (train() ends with self.best_user_emb, set by the evaluation we skipped)
                assert "best_user_emb" in str(e), e
        finally:
            os.chdir(cwd)
    # stamps[k] = when batch k was handed to the loop: stamps[k+1] - stamps[k] = step k's compute + batch k+1's sampling
    gaps = [b - a for a, b in zip(stamps[1:-1], stamps[2:])]          # (drop the first step: warm-up)
    ref = sum(gaps) / len(gaps)
    print(f"reference (model/graph/XSimGCL.py train(), its own sampler):  {ref * 1e3:8.1f} ms/step = {2048 / ref:9.1f} pairs/s"
          f"   [{len(gaps)} steps]")
    port2 = port_again()                  # (the port once more AFTER the reference: run order / warm allocator effects show here)
    print(f"port again, after the reference:                             {port2 * 1e3:8.1f} ms/step = {2048 / port2:9.1f} pairs/s")
    best = min(port, port2)
    print(f"port / reference step time: first run {port / ref:.3f}, second run {port2 / ref:.3f}  "
          f"({'within' if abs(best / ref - 1) <= 0.10 else 'OUTSIDE'} 10 % on the better run)")


if __name__ == "__main__":
    main()
