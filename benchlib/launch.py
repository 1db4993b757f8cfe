"""Launching bench.py over N ranks: torchrun relaunch, device-count refusal, the watchdog, guarded first steps."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3


def default_layout_is_dp(nnz):
    """bench.py --gpus N > 1 on this graph: data parallel unless SRH_SHARD_LAYOUT says otherwise or the graph is gather-bound"""
    from selfrec_amd.dist import GATHER_BOUND_NNZ
    return (os.environ.get("SRH_SHARD_LAYOUT") or "dp") == "dp" and nnz < GATHER_BOUND_NNZ


def free_port():
    """A TCP port nobody listens on right now (rendezvous of a self-launched job: never a fixed number -- two jobs on one
    node, or a stale listener of a crashed one, would collide on it)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """HIP devices this process can open (0 without a GPU: never an error)."""
    try:
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def refuse_gpu_count(n, backend):
    """`--gpus N` with N above the visible devices: say so and leave with exit code 2 instead of letting N ranks fight over
    fewer GPUs (RCCL refuses two ranks per device with an error that names neither N nor the device count).  The CPU
    launch check ("gloo") and the shared-device test mode ("gloo:device") do not need N devices."""
    have = visible_gpus()
    if backend in ("gloo", "gloo:device") or n <= have:
        return
    msg = (f"bench.py --gpus {n}: only {have} HIP device(s) visible on this node (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES "
           f"= {os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES') or 'unset'}); run with "
           f"--gpus <= {have}")
    print(f"[bench] {msg}", file=sys.stderr, flush=True)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"error": msg, "n_gpus": n, "visible_gpus": have}), flush=True)
    raise SystemExit(2)


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` (no launcher, the way the driver's single-GPU command line is spelled): start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` -- one rank per GPU over
    RCCL -- and hand its exit code back.  Rank 0 of the child job prints the JSON line on the inherited stdout."""
    import subprocess
    refuse_gpu_count(n, os.environ.get("SRH_DIST_BACKEND", "nccl"))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool's hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(REPO, "bench.py")] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


class Watchdog:
    """First-contact safety of the N > 1 run (no line of it has met more than one real GPU): a daemon thread that ends the
    PROCESS -- message on stderr, an error JSON line on rank 0, exit code 3 -- when the main thread has not reported progress
    for `seconds`.  A collective that never returns cannot be recovered from inside the process that is stuck in it; what can
    be done is to stop within a bounded time, say where, and leave the GPU free (a job that hangs until the driver's own
    limit is a strike against the box).  `beat(phase)` after every phase that contains a collective."""

    def __init__(self, seconds, rank=0, enabled=True):
        import threading
        self.seconds, self.rank = float(seconds), rank
        self.phase, self.last = "start", time.monotonic()
        self.partial = None            # rank 0: a finished headline record -- printed (with the error) instead of being lost
        self._stop = threading.Event()
        self.thread = None
        if enabled and self.seconds > 0:
            self.thread = threading.Thread(target=self._watch, daemon=True)
            self.thread.start()

    def beat(self, phase):
        self.phase, self.last = phase, time.monotonic()

    def stop(self):
        self._stop.set()

    def _watch(self):
        while not self._stop.wait(min(1.0, self.seconds / 4)):
            idle = time.monotonic() - self.last
            if idle > self.seconds:
                msg = (f"watchdog: rank {self.rank} made no progress for {idle:.0f} s in phase '{self.phase}' "
                       f"(limit SRH_BENCH_WATCHDOG_S = {self.seconds:.0f} s) -- a collective or a captured graph around one "
                       f"did not return; rerun with SRH_SHARDED_GRAPH=0 (eager launches) or SRH_SHARD_LAYOUT=rows|cols to "
                       f"narrow it down")
                print(f"[bench] {msg}", file=sys.stderr, flush=True)
                if self.rank == 0:
                    line = dict(self.partial, error=msg, phase=self.phase) if self.partial else {"error": msg, "phase": self.phase}
                    print(json.dumps(line), flush=True)
                os._exit(3)


def first_steps_guarded(make_trainer, runner_of, watchdog, what):
    """The first steps of a multi-rank trainer (capture of the two graphs around the collective, the first replays): if
    they RAISE -- an RCCL timeout, a capture error next to a live communicator, the engine's replay-vs-eager check -- fall
    back to eager launches ONCE; a second failure ends the run with exit code 4 and the message.  (A hang that raises
    nothing is the watchdog's.)  Returns (trainer, runner, note)."""
    note = None
    for attempt in (0, 1):
        trainer = make_trainer(eager=attempt == 1)
        runner = runner_of(trainer)
        try:
            runner.run(2)
            runner.fence()
            if watchdog is not None:
                watchdog.beat(f"{what}: first steps")
            return trainer, runner, note
        except (RuntimeError, ValueError) as e:
            note = f"{what}: {type(e).__name__} in the first steps ({str(e)[:300]})"
            print(f"[bench] {note}; " + ("falling back to eager launches once" if attempt == 0 else "giving up"),
                  file=sys.stderr, flush=True)
            if attempt == 1:
                if int(os.environ.get("RANK", "0")) == 0:
                    print(json.dumps({"error": note}), flush=True)
                os._exit(4)
    raise AssertionError("unreachable")
