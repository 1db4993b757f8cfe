#!/usr/bin/env python3
"""Host cost of driving the captured step: how far the enqueueing thread runs ahead of the GPU, and what an epoch boundary
costs it piece by piece (a 20-step timed region placed over a boundary sees every host stall that the queue cannot cover)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, args.emb, model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1,
                  batch_size=2048, use_graph=True, nce_precision="f32")
r = bench.Runner(tr, args.seed)
r.run(100); r.fence()
for n in (20, 200):
    r.fence()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    r.left -= n
    print(f"{n} replays after a fence: enqueued in {(t1 - t0) / n * 1e6:.1f} us each, done after {(t2 - t0) / n * 1e6:.1f} us each "
          f"(the host is ahead by {(t2 - t1) * 1e3:.2f} ms at the end)")


def us(fn, n=200):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


ev = torch.cuda.Event()
ev.record()
st = torch.cuda.current_stream()
view = tr.cursor[0:1]
val = int(tr.cursor[0].item())
src = torch.tensor([val], dtype=torch.int64, device=tr.cursor.device)
print(f"pieces (us per call, idle stream): wait_event {us(lambda: st.wait_event(ev)):.1f} | Event() + record "
      f"{us(lambda: torch.cuda.Event().record()):.1f} | cursor[0:1].fill_ {us(lambda: tr.cursor[0:1].fill_(val)):.1f} | "
      f"pre-sliced view.fill_ {us(lambda: view.fill_(val)):.1f} | view.copy_(device scalar) {us(lambda: view.copy_(src)):.1f} | "
      f"Thread(target=noop).start() + join {us(lambda: (lambda t: (t.start(), t.join()))(threading.Thread(target=lambda: None)), 50):.1f}")
torch.cuda.synchronize()
# the same pieces with the stream busy (50 replays queued in front)
for _ in range(50):
    tr.step()
r.left -= 50
busy = (us(lambda: st.wait_event(ev), 20), us(lambda: torch.cuda.Event().record(), 20), us(lambda: view.fill_(val), 20))
torch.cuda.synchronize()
print(f"with ~50 replays queued: wait_event {busy[0]:.1f} | Event() + record {busy[1]:.1f} | view.fill_ {busy[2]:.1f} us")

# ---- which piece of an epoch boundary costs GPU time: 10 replays, the piece, 10 replays, between two fences
import numpy as np  # noqa: E402


def region(piece, reps=9, manage=True):
    out = []
    for _ in range(reps):
        if manage and r.left < 25:
            r.run(r.left + 1)             # (cross the epoch boundary the runner's way, outside the timing)
        r.fence()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step()
        piece()
        for _ in range(10):
            tr.step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 20 * 1e6)
        if manage:
            r.left -= 20
    return float(np.median(out)), float(np.min(out)), float(np.max(out))


def busy_python_thread():
    def spin():
        x = 0
        for i in range(200000):
            x += i
    threading.Thread(target=spin, daemon=True).start()


pieces = {"nothing": lambda: None, "view.fill_": lambda: view.fill_(int(0)), "Event().record()": lambda: torch.cuda.Event().record(),
          "wait_event": lambda: st.wait_event(ev), "Thread(noop).start()": lambda: threading.Thread(target=lambda: None).start(),
          "thread: 10 ms of python bytecode": busy_python_thread}
for name, fn in pieces.items():
    med, lo, hi = region(fn)
    print(f"10 replays + [{name}] + 10 replays: median {med:.1f} us per step (min {lo:.1f}, max {hi:.1f})", flush=True)
# ---- the cross-stream wait on the staging copy's event, and letting go of an epoch's worth of host arrays.
# (Regions around a real upload_epoch / a real sampler-thread start were tried here too and measure the GPU's clock ramp instead: the
# sampled epoch has to be waited for first, 0.1 s of idle -- every such region came out + 12 us per step whatever was inside.)
cs = tr._copy_stream if tr._copy_stream is not None else torch.cuda.Stream()
ev_other = torch.cuda.Event()
ev_other.record(cs)
torch.cuda.synchronize()
junk = {}


def make_junk():
    junk["a"] = {k: np.empty(6_300_000, dtype=np.int32) for k in "uijxy"}      # ~ an epoch's host arrays (126 MB here)
    for v in junk["a"].values():
        v[::1024] = 1                                                           # (touched: really mapped)


def free_junk():
    junk.pop("a")


for name, before, fn in (("wait_event(an event of the COPY stream, complete)", lambda: None, lambda: st.wait_event(ev_other)),
                         ("free 126 MB of numpy arrays (munmap)", make_junk, free_junk)):
    vals = []
    for _ in range(5):
        if r.left < 25:
            r.run(r.left + 1)
        before()
        med, lo, hi = region(fn, reps=1)
        vals.append(med)
    print(f"10 replays + [{name}] + 10 replays: median {float(np.median(vals)):.1f} us per step (min {min(vals):.1f}, max {max(vals):.1f})",
          flush=True)
