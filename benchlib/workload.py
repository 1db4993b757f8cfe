"""The workload of bench.py: flags, the synthetic graph, SURVEY.md 8(d)'s byte models and the step driver."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1300)     # > 2 epochs of 616 batches
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--shape", default="yelp2018")
    ap.add_argument("--model", default="XSimGCL")
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--emb", type=int, default=64)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--tau", type=float, default=0.2)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--seed", type=int, default=2024)
    return ap.parse_args(argv)


def build_data(shape, seed):
    from selfrec_amd import synth
    from selfrec_amd.data.ui_graph import Interaction
    tu, ti, su, si, U, I = synth.make_dataset(shape, seed=seed)
    if len(tu) > 5_000_000:
        # (the 1 M x 500 k graph: 40 M python triples would be 25 GB of host objects -- what the reference needs for it;
        # the id arrays go straight in, with a bounded test set)
        data = Interaction.from_id_arrays({}, tu, ti, su[:200_000], si[:200_000], U, I)
    else:
        data = Interaction({}, synth.as_triples(tu, ti), synth.as_triples(su, si))
    return data, (tu, ti, su, si, U, I)


def spmm_alg_bytes(nnz, n_rows, n_cols, d):
    """compulsory traffic of one CSR SpMM with perfect reuse of x."""
    return nnz * 8 + (n_rows + 1) * 4 + n_cols * d * 4 + n_rows * d * 4


def step_alg_bytes(model, nnz, N, d, L, B):
    passes = {"MF": 0, "LightGCN": 1, "XSimGCL": 1, "SimGCL": 3, "SGL": 3}[model]
    bwd = {"MF": 0, "LightGCN": 1, "XSimGCL": 1, "SimGCL": 1, "SGL": 3}[model]
    spmm = (passes + bwd) * L * spmm_alg_bytes(nnz, N, N, d)
    return spmm + 7 * N * d * 4 + N * d * 4 + 2 * 3 * B * d * 4 + 4 * 2 * B * d * 4


class Runner:
    """A trainer driven the way a training run drives it: the host samples epoch e + 1 on a worker thread while the device
    works on epoch e; an epoch boundary = hand-over of the sampled arrays + a 25 MB index upload."""

    def __init__(self, trainer, seed, dist=None, watchdog=None):
        from selfrec_amd.engine import EpochPrefetcher
        self.trainer, self.dist, self.watchdog = trainer, dist, watchdog
        trainer.seed_sampler(seed)                 # (data parallel: seed + rank -- every rank its own batches)
        self.pre = EpochPrefetcher(trainer)
        self.pre.start()
        self.left, self.uploads, self.steps_done = 0, 0, 0
        self.boundary_ms = []                      # per epoch boundary: host ms (waiting for the sampled epoch, hand-over, restart)
        self.step_events = None                    # diagnostic (SRH_BENCH_STEP_EVENTS=1): a timing event behind every step

    def run(self, n_steps):
        done = 0
        while done < n_steps:
            if self.left == 0:
                self.uploads += 1
                t0 = time.perf_counter()
                epoch = self.pre.take()
                t1 = time.perf_counter()
                self.trainer.upload_epoch(epoch)
                t2 = time.perf_counter()
                self.pre.start()                       # host samples the next epoch while this one runs
                self.boundary_ms.append((round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3),
                                         round((time.perf_counter() - t2) * 1e3, 3)))
                self.left = self.trainer.epoch_batches
            take = min(self.left, n_steps - done)
            for _ in range(take):
                self.trainer.step()
                if self.step_events is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.step_events.append(ev)
            self.left -= take
            done += take
        self.steps_done += n_steps

    def align_to_epoch_boundary(self, n_steps, cap_seconds=3.0):
        """Extra UNTIMED steps so that the next n_steps hold an epoch boundary (sampler hand-over + index upload are part
        of the metric; a 20-step region placed anywhere else in a 616-batch epoch would never see one):
        run until about n_steps / 2 batches of the current epoch are left.  Returns the number of extra steps (0 when the
        region is longer than an epoch anyway, or when getting there would take more than `cap_seconds` of steps)."""
        if n_steps >= self.trainer.epoch_batches or self.left < n_steps:
            return 0                                   # (the region reaches a boundary as it is)
        extra = self.left - max(1, n_steps // 2)
        if extra <= 0:
            return 0
        probe = min(extra, 50)
        t0 = time.perf_counter()
        self.run(probe)
        self.fence()
        if (extra - probe) * (time.perf_counter() - t0) / probe > cap_seconds:
            return probe                               # (too far away: the 1 M x 500 k epoch is 19,657 steps of 25 ms)
        self.run(extra - probe)
        return extra

    def fence(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize()

    def timed(self, n_steps, phase):
        """(seconds, epoch boundaries inside, every rank's seconds) of exactly n_steps steps between two fences; the
        seconds are the MAX over the ranks."""
        self.fence()
        up0 = self.uploads
        t0 = time.perf_counter()
        self.run(n_steps)
        self.fence()
        dt = time.perf_counter() - t0
        per_rank = None
        if self.dist is not None:
            mine = torch.tensor([dt], dtype=torch.float64, device="cuda")
            everyone = torch.empty(self.dist.get_world_size(), dtype=torch.float64, device="cuda")
            self.dist.all_gather_into_tensor(everyone, mine)
            per_rank = [float(v) for v in everyone.cpu()]
            dt = max(per_rank)
        if self.watchdog is not None:
            self.watchdog.beat(phase)
        return dt, self.uploads - up0, per_rank


def steady_state(runner, step_s, pairs_per_step):
    """>= 2 epochs (>= 1 epoch boundary: sampler hand-over + 25 MB index upload inside the region) and >= 0.6 s of
    device time, whatever --steps the driver passed.  SURVEY.md 8(d): sampling and the index upload are INSIDE the metric."""
    tr = runner.trainer
    n = max(2 * tr.epoch_batches, int(0.6 / step_s))
    if n * step_s > 30.0:          # (the 1 M x 500 k shape: an epoch is 19,657 steps of 25 ms -- bounded instead)
        n = max(20, int(5.0 / step_s))
    dt, bounds, _ = runner.timed(n, "steady state")
    return {"steps": n, "seconds": round(dt, 4), "ms_per_step": round(dt / n * 1e3, 4),
            "pairs_per_s": round(n * pairs_per_step / dt, 1), "epoch_boundaries_inside": bounds}
