"""The reference's OWN training step on this host's cores: `python -m benchlib.ref_step --ref <checkout> ...` imports the
reference's model/graph/XSimGCL.py from a staged checkout (never from this repo), applies SURVEY.md 8(c)'s two shims
(numba stubbed -- it is not in this image; .cuda() -> identity) and times its train() loop on bench.py's synthetic graph:
seconds per step measured between consecutive batches of its own next_batch_pairwise generator, so the python sampling is
inside, as in the reference's loop (model/graph/XSimGCL.py:23-43).  Prints ONE JSON line.

A process of its own: the shims replace torch.Tensor.cuda and the `base / data / util / model` module names, which the
benchmark process must keep."""
import argparse
import importlib
import json
import os
import random
import sys
import tempfile
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", required=True)
    ap.add_argument("--shape", default="yelp2018")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--emb", type=int, default=64)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--tau", type=float, default=0.2)
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--max-steps", type=int, default=200)
    args = ap.parse_args()
    sys.path.insert(0, REPO)
    import torch
    from selfrec_amd import synth
    tu, ti, su, si, U, I = synth.make_dataset(args.shape, seed=args.seed)
    for name in [m for m in sys.modules if m.split(".")[0] in ("base", "data", "util", "model")]:
        del sys.modules[name]
    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = numba
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.dont_write_bytecode = True
    ref = os.path.abspath(args.ref)
    sys.path.insert(0, ref)
    mod = importlib.import_module("model.graph.XSimGCL")
    assert os.path.abspath(mod.__file__).startswith(ref), mod.__file__
    from util.conf import ModelConf
    stamps = []
    real = mod.next_batch_pairwise
    t_begin = [None]

    def batches(data, bs, n_negs=1):
        for b in real(data, bs, n_negs):
            now = time.perf_counter()
            stamps.append(now)
            if t_begin[0] is None:
                t_begin[0] = now
            if len(stamps) >= 4 and (now - stamps[1] >= args.seconds or len(stamps) >= args.max_steps + 2):
                return
            yield b
    mod.next_batch_pairwise = batches
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            lines = ["training.set: ./train.txt", "test.set: ./test.txt", "model:", "  name: XSimGCL", "  type: graph",
                     "item.ranking.topN: [10,20]", f"embedding.size: {args.emb}", "max.epoch: 1", f"batch.size: {args.batch}",
                     "learning.rate: 0.001", "reg.lambda: 0.0001", "output: ./results/", "XSimGCL:",
                     f"  n_layer: {args.layers}", "  l_star: 1", "  lambda: 0.2", "  eps: 0.2", f"  tau: {args.tau}"]
            with open("XSimGCL.yaml", "w") as f:
                f.write("\n".join(lines) + "\n")
            conf = ModelConf("XSimGCL.yaml")
            torch.manual_seed(args.seed)
            random.seed(args.seed)
            t0 = time.perf_counter()
            model = mod.XSimGCL(conf, [list(t) for t in synth.as_triples(tu, ti)], [list(t) for t in synth.as_triples(su, si)])
            build_s = time.perf_counter() - t0
            model.fast_evaluation = lambda epoch: None
            try:
                model.train()
            except AttributeError as e:          # (train() ends with self.best_user_emb, set by the evaluation we skipped)
                assert "best_user_emb" in str(e), e
        finally:
            os.chdir(cwd)
    # stamps[k] = when batch k was handed to the loop: stamps[k+1] - stamps[k] = step k's compute + batch k+1's sampling;
    # the first gap holds the warm-up step
    gaps = [b - a for a, b in zip(stamps[1:-1], stamps[2:])]
    step = sum(gaps) / len(gaps)
    print(json.dumps({"ms_per_step": round(step * 1e3, 2), "pairs_per_s": round(args.batch / step, 1), "steps": len(gaps),
                      "cores": torch.get_num_threads(), "host_cpus": os.cpu_count(), "build_seconds": round(build_s, 1),
                      "first_step_ms": round((stamps[1] - stamps[0]) * 1e3, 1), "torch": torch.__version__}), flush=True)


if __name__ == "__main__":
    main()
