#!/usr/bin/env python3
"""Headline benchmark: XSimGCL training throughput (user-item pairs/s) on a synthetic Yelp2018-shaped graph
(BASELINE.json config 3: 31,668 x 38,048, ~1.26 M train edges, d=64, B=2048, L=3, l*=1, eps=0.2, lambda=0.2,
tau=0.2), plus full-rank eval users/s.

    python bench.py --gpus N --steps K --warmup W          (N>1: one rank per GPU via torchrun)

A "step" is one pass of the whole hot path over one batch: host sampling of that batch (C++ MT19937 replay, on a
worker thread, one epoch ahead), index staging, L propagation SpMMs with fused perturbation / mean, BPR + L2 +
2 x InfoNCE forward / backward, L backward SpMMs, dense Adam.  Inputs (graph, tables, sampled epoch) are resident in
HBM when the timed region starts; the region is EXACTLY --steps steps between barrier + torch.cuda.synchronize() on
both sides, positioned (by extra untimed steps) so that it holds an epoch boundary -- sampler hand-over and index
upload are inside the metric -- and the max over ranks is reported.

The arithmetic of the headline is fp32 throughout: InfoNCE's two n x n x d products run on the f32 MFMA
(v_mfma_f32_16x16x4_f32, exact f32 multiply-adds), like the reference's fp32 matmul (util/loss_torch.py:46-47).
The library's faster mode for those two products (16-bit split operands, f32 accumulation) is timed beside it and
reported as `value_split16`; it is not the headline.

N > 1: the headline keeps the global batch at B = 2048 pairs per step and divides that batch's work over the ranks
(tables / graph sharded: "scaling": "strong", north_star's partition); the data-parallel run (N x B pairs per step,
weak) is the `dp` sub-record.

One JSON line on rank 0: the contract fields, `roofline` (dominant kernel: CSR SpMM, HBM-bound, algorithmic bytes per
launch / mean launch duration from HIP events on the launch stream), `cpu_baseline` (the reference's own train() step
when a checkout is staged on this box, else the oracle port, on this host's cores), `eval`.  The parts live in
benchlib/: workload (flags, data, step driver), launch (N ranks, watchdog), probes (roofline), evalbench, baseline,
dropin_bench."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from benchlib.baseline import cpu_baseline  # noqa: E402
from benchlib.dropin_bench import dropin_fused_throughput, dropin_throughput  # noqa: E402
from benchlib.evalbench import eval_cpu_baseline, eval_throughput, eval_throughput_sharded, ops_filtered  # noqa: E402,F401
from benchlib.launch import (Watchdog, default_layout_is_dp, first_steps_guarded, free_port,  # noqa: E402,F401
                             refuse_gpu_count, relaunch_under_torchrun)
from benchlib.probes import (git_blob_hash, pmc_traffic_cols, spmm_roofline, time_spmm_kernel)  # noqa: E402,F401
from benchlib.workload import (Runner, build_data, parse, spmm_alg_bytes, steady_state, step_alg_bytes)  # noqa: E402,F401

NCE_MODELS = ("XSimGCL", "SimGCL", "SGL")


def launch_check(args, rank, world, backend, wd_seconds, coll_timeout):
    """`SRH_DIST_BACKEND=gloo`: the launch path without GPUs (tests/test_dist_cpu.py) -- rendezvous, one collective, the
    layouts this world size would take -- then stop; everything after it needs the HIP library and a device."""
    import torch.distributed as dist
    from selfrec_amd.dist import describe_layout, pick_layout
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=coll_timeout)
    wd = Watchdog(wd_seconds, rank)
    if os.environ.get("SRH_BENCH_TEST_STALL"):          # (tests/test_dist_cpu.py: the watchdog's exit path)
        time.sleep(float(os.environ["SRH_BENCH_TEST_STALL"]))
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    ones = torch.ones(1)
    dist.all_reduce(ones)
    wd.beat("launch check")
    if rank == 0:
        head = headline_layout(args, world)
        print(json.dumps({"launch_check": True, "backend": backend, "world": world, "rank_sum": float(t.item()),
                          "ranks_in_collective": int(ones.item()), "master_port": int(os.environ["MASTER_PORT"]),
                          "headline": {"layout": pick_layout(args.emb, world, head, None), "scaling": scaling_of(head),
                                       "global_batch": args.batch * (world if head == "dp" else 1),
                                       "parallelism": describe_layout(args.emb, world, head)},
                          "baseline_partition": baseline_partition_note(pick_layout(args.emb, world, head, None)),
                          "sub_records": sub_record_layouts(args, world, head)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    wd.stop()


def headline_layout(args, world):
    """The layout `--gpus N` reports first: north_star's -- one batch of B pairs, tables / graph divided over the ranks
    (pick_layout's cols / rows / 2-D choice: "auto") -- unless SRH_SHARD_LAYOUT names another."""
    return (os.environ.get("SRH_SHARD_LAYOUT") or "auto").lower()


def scaling_of(layout):
    return "weak" if layout == "dp" else "strong"


def sub_record_layouts(args, world, head):
    """Layouts timed after the headline: data parallel (weak scaling), and -- at EVERY shape -- the plain row partition
    BASELINE.json's north_star names (tables and graph rows dealt over the ranks, an all-gather of (N, d) after every
    product: `layout_rows`), so that the first line from real multi-GPU hardware holds that configuration's number whatever
    pick_layout chose for the headline (column blocks at the Yelp2018 shape, the 2-D grid at 1 M x 500 k)."""
    if os.environ.get("SRH_SUB_RECORDS", "1") == "0" or world < 2:
        return []
    subs = [s for s in (os.environ.get("SRH_SUB_LAYOUTS") or "").split(",") if s]
    if not subs:
        subs = ["dp"] if head != "dp" else ["auto"]
        if head != "rows":
            subs.append("rows")
    return subs


def baseline_partition_note(chosen):
    """config.baseline_partition: which record of this line is the partition BASELINE.json / SURVEY.md 8(e) name."""
    if chosen == "rows":
        return "this record: rows (tables + graph rows dealt over the ranks, one all-gather of (N, d) per product)"
    return (f"the `layout_rows` sub-record (tables + graph rows dealt over the ranks, one all-gather of (N, d) per product); the "
            f"headline is pick_layout's choice for this shape ({chosen}), same fixed global batch")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU of this node)
        raise SystemExit(relaunch_under_torchrun(args.gpus))
    if world != args.gpus:
        args.gpus = world
    sharded = world > 1 or bool(os.environ.get("SRH_FORCE_SHARDED"))     # (knob: exercise the N>1 path on one GPU)
    # SRH_DIST_BACKEND: "nccl" (RCCL, the default); "gloo": the CPU test of the launch path (stops after one collective);
    # "gloo:device": the WHOLE benchmark with gloo moving device tensors and the ranks dealt over the visible GPUs modulo
    # their count -- N ranks on the one GPU of a test box (RCCL refuses two ranks per device); its numbers mean nothing,
    # the point is that every line of the N > 1 path has run with N > 1 (tools/gpu_session.sh stage benchworld2)
    backend = os.environ.get("SRH_DIST_BACKEND", "nccl")
    shared_device = backend == "gloo:device"
    wd_seconds = float(os.environ.get("SRH_BENCH_WATCHDOG_S", "240"))
    import datetime
    coll_timeout = datetime.timedelta(seconds=max(30.0, wd_seconds))
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                # (one process standing in for a job: SRH_FORCE_SHARDED)
            os.environ["MASTER_PORT"] = str(free_port())
    if sharded and backend not in ("nccl", "gloo:device"):
        return launch_check(args, rank, world, backend, wd_seconds, coll_timeout)
    refuse_gpu_count(world, backend)
    from selfrec_amd import _lib
    _lib.require_gpu()
    if shared_device:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    watchdog, dist, comm = None, None, None
    if sharded:
        import torch.distributed as dist
        if shared_device:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=coll_timeout)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                    timeout=coll_timeout)
        watchdog = Watchdog(wd_seconds, rank, enabled=world > 1)
        # the communicator's own account of itself, after a real collective on device memory: how many ranks added their 1
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        comm = {"backend": "gloo (TEST MODE: ranks share GPUs, figures are not measurements)" if shared_device
                else f"{dist.get_backend()} (RCCL)", "ranks_in_collective": int(ones.item()), "world_size": dist.get_world_size()}
        watchdog.beat("first collective")

    data, raw = build_data(args.shape, args.seed)
    if watchdog is not None:
        watchdog.beat("data built")
    nnz_adj = 2 * data.interaction_mat.nnz
    has_nce = args.model in NCE_MODELS
    kw = dict(model=args.model, n_layers=args.layers, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=args.tau,
              layer_cl=1, batch_size=args.batch, nce_precision=("f32" if has_nce else None))

    def make(layout):
        """a trainer of this layout on the SAME initial tables (torch.manual_seed before every construction)"""
        def build(eager=False):
            torch.manual_seed(args.seed)
            use_graph = not args.no_graph and not eager
            if layout is False:
                from selfrec_amd.engine import FusedTrainer
                # (the first epoch is drawn on a thread while the trainer is built: Runner seeds with the same value)
                return FusedTrainer(data, args.emb, use_graph=use_graph, sampler_seed=args.seed, **kw)
            from selfrec_amd.dist import ShardedTrainer
            return ShardedTrainer(data, args.emb, layout=layout, use_graph=use_graph, **kw)
        return build

    def parallelism_of(tr):
        from selfrec_amd.dist import describe_layout
        return describe_layout(args.emb, world, str(tr.layout) if (world > 1 or tr.dp) else ("cols" if tr.cols else "rows"),
                               2 * tr.graph.n_edges)

    def timed_record(trainer, runner, label):
        """warm-up to --warmup, then exactly --steps steps over an epoch boundary; the fields every record carries"""
        runner.run(max(0, args.warmup - runner.steps_done))
        extra = runner.align_to_epoch_boundary(args.steps)
        seen = len(runner.boundary_ms)
        # Short regions carry a timing event behind every step: `step_us` shows where inside the region the time goes (the
        # last, partial batch of an epoch; the first step of the next).  Observed beside it, not understood: with the events in
        # place the host's work at an epoch boundary (hand-over, sampler thread) no longer delays the steps behind it -- 20-step
        # regions over a boundary: 0.290 ms per step in 20 runs of 22, against 0.287 .. 0.316 without
        # (profiles/r05_o_epoch_boundary_host_cost.txt).  SRH_BENCH_STEP_EVENTS=0: without.
        trace = os.environ.get("SRH_BENCH_STEP_EVENTS", "1") not in ("", "0") and args.steps <= 200
        if trace:
            runner.fence()
            first = torch.cuda.Event(enable_timing=True)
            first.record()
            runner.step_events = [first]
        dt, bounds, per_rank = runner.timed(args.steps, f"{label}: timed region")
        step_us = None
        if trace:
            evs, runner.step_events = runner.step_events, None
            step_us = [round(a.elapsed_time(b) * 1e3, 1) for a, b in zip(evs[:-1], evs[1:])]
        dp = bool(getattr(trainer, "dp", False))
        pairs = args.batch * (world if dp else 1)
        return {"value": round(args.steps * pairs / dt, 1), "unit": "pairs/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
                "steps": args.steps, "global_batch": pairs, "scaling": "weak" if dp else "strong",
                "epoch_boundaries_inside": bounds, "untimed_steps_before": runner.steps_done - args.steps,
                # host ms per boundary inside the region: (waiting for the sampled epoch, hand-over, restarting the sampler thread)
                "epoch_boundary_host_ms": [list(t) for t in runner.boundary_ms[seen:]],
                **({"step_us": step_us} if step_us is not None else {}),
                "aligned_by_extra_steps": extra,
                "ms_per_step_by_rank": [round(t / args.steps * 1e3, 4) for t in per_rank] if per_rank else None,
                "launch": "hipGraph replay" if trainer.use_graph else "eager"}, dt

    notes = []
    head = headline_layout(args, world)
    if sharded:
        from selfrec_amd.dist import pick_layout
        chosen = pick_layout(args.emb, world, head, nnz_adj)
        trainer, runner, note = first_steps_guarded(make(chosen), lambda t: Runner(t, args.seed, dist, watchdog), watchdog,
                                                    f"layout {chosen}")
        if note:
            notes.append(note)
    else:
        trainer = make(False)()
        runner = Runner(trainer, args.seed, None, None)
        runner.run(min(2, max(args.warmup, 1)))    # epoch upload, hipGraph capture, first replays
        runner.fence()
    g = trainer.graph
    dp = bool(getattr(trainer, "dp", False))
    # The roofline probes (per-flavour launch times, the gather bound, the stream rates: ~30 ms of launches the line needs
    # anyway) run HERE, between the first warm-up steps and the rest: the timed region then starts on a chip that has been
    # busy, not on one that idled through graph capture (profiles/r04_b_step_timeline.txt: after >= 0.3 s of idle the first
    # 60 steps run 4 / 2.5 / 1.5 % slow -- DVFS ramp).
    roof = None
    if rank == 0:
        try:
            roof = spmm_roofline(args, trainer, sharded, dp, None, g)
        except RuntimeError as e:             # (never lose a multi-GPU line to its footnotes)
            if not sharded:
                raise
            roof = {"error": str(e)}
    if watchdog is not None:
        watchdog.beat("roofline")
    rec, elapsed = timed_record(trainer, runner, "headline")
    losses = trainer.read_losses()
    step_s = max(elapsed / args.steps, 1e-6)

    out = {
        "metric": f"train pairs/sec ({args.model}, {'Yelp2018' if args.shape == 'yelp2018' else args.shape}-shape)",
        "value": rec["value"], "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
        # N = 1 carries the label of its N > 1 companions, so that one SCALE series reads as one mode
        "scaling": rec["scaling"] if sharded else scaling_of(head), "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.model} L={args.layers} l*=1 eps=0.2 lambda=0.2 tau={args.tau}, synthetic {args.shape}-shape "
                               f"graph ({g.n_users} users x {g.n_items} items, {g.n_edges} train edges), d={args.emb}, "
                               f"B={args.batch}, Adam lr=1e-3",
                   "global_batch": rec["global_batch"],
                   "parallelism": parallelism_of(trainer) if sharded else "single",
                   **({"baseline_partition": baseline_partition_note(str(trainer.layout))} if sharded and world > 1 else {}),
                   # every sum of the step has one fixed order (no float atomics): two runs give the same bits
                   "bit_reproducible_step": bool(getattr(trainer, "det_scatter", False)),
                   "rccl_ranks": comm["ranks_in_collective"] if comm and not shared_device else None,
                   "comm": comm, "launch": rec["launch"],
                   "epoch_boundaries_in_region": rec["epoch_boundaries_inside"],
                   "epoch_boundary_host_ms": rec["epoch_boundary_host_ms"],
                   **({"step_us": rec["step_us"]} if "step_us" in rec else {}),
                   "nce_products": "f32 MFMA (v_mfma_f32_16x16x4_f32)" if has_nce else None,
                   "perturbation_rng": "counter-based integer hash in the SpMM epilogue (not Philox): moments / decorrelation tested",
                   # workgroups of real tasks per XCD in the dense plan after the engine's start-up calibration
                   "xcd_shares": (None if getattr(trainer, "xcd_shares", None) is None
                                  else [int(v) for v in trainer.xcd_shares])},
        "ms_per_step_by_rank": rec["ms_per_step_by_rank"],
        "final_losses": {"bpr": losses[0], "reg": losses[1], "cl": losses[2]},
    }
    if watchdog is not None:
        watchdog.partial = out                     # (from here on a hang in a later leg still leaves this line)
    out["steady_state"] = steady_state(runner, step_s, rec["global_batch"])
    out["value_steady_state"] = out["steady_state"]["pairs_per_s"]
    if has_nce and not sharded:
        # the same step with InfoNCE's two products on 16-bit split operands (the library's faster mode; re-captured graph)
        trainer.set_nce_precision("split")
        runner.run(20); runner.fence()
        n16 = max(300, args.steps)
        dt16, _, _ = runner.timed(n16, "split16 region")
        out["value_split16"] = round(n16 * args.batch / dt16, 1)
        out["ms_per_step_split16"] = round(dt16 / n16 * 1e3, 4)
        out["config"]["value_split16"] = out["value_split16"]
        trainer.set_nce_precision("f32")
        runner.run(5); runner.fence()
    if rank == 0 and roof:
        if "step_alg_bytes" in roof:
            roof["step_GBps"] = round(roof["step_alg_bytes"] / step_s / 1e9, 1)
        out["roofline"] = roof

    # ---- N > 1: the other layouts, each a record of its own next to the headline
    for sub in sub_record_layouts(args, world, head) if sharded else []:
        dist.barrier()
        try:
            from selfrec_amd.dist import pick_layout
            chosen = pick_layout(args.emb, world, sub, nnz_adj)
            s_tr, s_run, s_note = first_steps_guarded(make(chosen), lambda t: Runner(t, args.seed, dist, watchdog), watchdog,
                                                      f"sub-record {chosen}")
            s_rec, s_dt = timed_record(s_tr, s_run, f"sub-record {chosen}")
            s_rec.update(layout=chosen, n_gpus=world, parallelism=parallelism_of(s_tr))
            s_rec["steady_state"] = steady_state(s_run, max(s_dt / args.steps, 1e-6), s_rec["global_batch"])
            if s_note:
                s_rec["note"] = s_note
            out["dp" if chosen == "dp" else f"layout_{chosen.replace(':', '_')}"] = s_rec
            del s_tr, s_run
        except Exception as e:                    # (never lose the headline to a companion record)
            out[f"layout_{sub}_error"] = f"{type(e).__name__}: {e}"
        if watchdog is not None:
            watchdog.beat(f"sub-record {sub}")
        dist.barrier()

    if rank == 0:
        small = len(raw[0]) <= 5_000_000
        if not args.no_eval and not sharded and small:
            out["eval"] = eval_throughput(trainer, data)
            if out["eval"]:
                # eval time runs "up to and including the python rec_list" -- the materialised figure is the
                # metric; the lazy one is what fast_evaluation() pays per epoch
                out["eval_users_per_s"] = out["eval"]["end_to_end_materialised_users_per_s"]
                out["eval_users_per_s_lazy"] = out["eval"]["end_to_end_users_per_s"]
                out["config"]["eval_users_per_s"] = out["eval_users_per_s"]
        if not sharded and not args.no_dropin and args.model == "XSimGCL" and small:
            out["dropin"] = dropin_throughput(args, raw)
            out["dropin_pairs_per_s"] = out["dropin"]["pairs_per_s"]
            out["dropin_fused"] = dropin_fused_throughput(args, raw)
            out["dropin_fused_pairs_per_s"] = out["dropin_fused"]["pairs_per_s"]
        # the CPU baseline rides on rank 0 at every N (the other ranks wait at the barrier below: ~15 s)
        if not args.no_cpu_baseline and not small:
            out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "kind": "port", "cores": torch.get_num_threads(),
                                   "sample": "not run at this shape: the reference's step needs ~25 GB of python objects and "
                                             "minutes per step (tests/golden/make_golden_shapes.py ran it once: 216 s on 8 threads)"}
        elif not args.no_cpu_baseline:
            if watchdog is not None:
                watchdog.beat("cpu baseline (rank 0 alone)")
            out["cpu_baseline"] = cpu_baseline(args, raw, min(args.cpu_seconds, max(5.0, wd_seconds / 4)) if sharded
                                               else args.cpu_seconds)
            if out.get("eval"):
                out["eval"]["cpu_baseline"] = eval_cpu_baseline(trainer, data)
    if watchdog is not None:
        watchdog.beat("cpu baseline")
    if dist is not None:
        dist.barrier()
        if watchdog is not None:
            watchdog.beat("after baseline barrier")
        if not args.no_eval:
            try:
                ev = eval_throughput_sharded(trainer, data, dist, rank, world)      # every rank takes part
            except Exception as e:                # (never lose the training line to the evaluation leg)
                ev = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                out["eval"] = ev
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.stop()
    if notes:
        out["notes"] = notes
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)        # RCCL's banner goes through C stdio: keep the JSON line last
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
