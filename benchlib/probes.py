"""Kernel-level probes behind the `roofline` block: per-flavour SpMM launch times (HIP events), stream rates,
the committed PMC traffic records."""
import json
import os
import sys
import time

import numpy as np
import torch

from .workload import spmm_alg_bytes, step_alg_bytes

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3


def git_blob_hash(path):
    """`git hash-object path` without git: the id of the file's contents."""
    import hashlib
    with open(path, "rb") as f:
        body = f.read()
    return hashlib.sha1(b"blob %d\0" % len(body) + body).hexdigest()


def pmc_traffic(args):
    """HBM-side bytes per dense SpMM launch from the committed PMC passes (bench.py cannot run under rocprofv3 --pmc
    itself): profiles/spmm_dense_traffic[_<shape>_d<emb>].json, written by tools/pmc_to_json.py from a counter session
    over tools/spmm_pmc.py.  The record carries the git blob id of the csrc/spmm.hip it was measured on: a record taken
    from another kernel source is REFUSED (traffic = null, with the reason), not quoted."""
    here = REPO
    default = args.shape == "yelp2018" and args.emb == 64
    name = "spmm_dense_traffic.json" if default else f"spmm_dense_traffic_{args.shape}_d{args.emb}.json"
    path = os.path.join(here, "profiles", name)
    if not os.path.exists(path):
        return None, f"no PMC pass committed for this workload (profiles/{name})"
    with open(path) as f:
        rec = json.load(f)
    want, have = rec.get("spmm_hip_blob"), git_blob_hash(os.path.join(here, "selfrec_amd", "csrc", "spmm.hip"))
    if want != have:
        return None, (f"profiles/{name} was measured on csrc/spmm.hip blob {str(want)[:12]}, this tree has {have[:12]}: stale "
                      f"record refused (re-run tools/gpu_session.sh pmc)")
    return rec["traffic_bytes_per_launch"], f"{rec['summary']}: {rec['how']}"


def eval_mfma_busy():
    """MFMA-pipe utilisation of the ranking's filter kernel from the committed counter pass (profiles/eval_mfma_busy.json,
    written from tools/gpu_session.sh evalpmc; bench.py cannot run under rocprofv3 --pmc itself).  Stamped with the git blob of
    the csrc/eval.hip it was measured on: a record of another kernel source is refused (None + the reason)."""
    here = REPO
    path = os.path.join(here, "profiles", "eval_mfma_busy.json")
    if not os.path.exists(path):
        return None, "no counter pass committed (profiles/eval_mfma_busy.json)"
    with open(path) as f:
        rec = json.load(f)
    have = git_blob_hash(os.path.join(here, "selfrec_amd", "csrc", "eval.hip"))
    if rec.get("eval_hip_blob") != have:
        return None, (f"profiles/eval_mfma_busy.json was measured on csrc/eval.hip blob {str(rec.get('eval_hip_blob'))[:12]}, this tree "
                      f"has {have[:12]}: stale record refused (re-run tools/gpu_session.sh evalpmc)")
    return rec["mfma_busy_filter16"], f"{rec['summary']}: {rec['how']}"


def eval_kernel_us():
    """{kernel: {"calls", "avg_us"}} of the ranking's kernels from the committed rocprofv3 --kernel-trace --stats pass of
    tools/eval_probe.py (profiles/eval_mfma_busy.json: kernel_us), or {} when the record is of another csrc/eval.hip."""
    path = os.path.join(REPO, "profiles", "eval_mfma_busy.json")
    if not os.path.exists(path):
        return {}, None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("eval_hip_blob") != git_blob_hash(os.path.join(REPO, "selfrec_amd", "csrc", "eval.hip")):
        return {}, None
    return rec.get("kernel_us") or {}, rec.get("kernel_us_source")


def pmc_traffic_cols(args, w):
    """Same for one rank's launch on (N, w) tables in the column-sharded layout: profiles/spmm_cols_traffic.json."""
    path = os.path.join(REPO, "profiles", "spmm_cols_traffic.json")
    if not (args.shape == "yelp2018" and args.emb == 64) or not os.path.exists(path):
        return None, "no PMC pass committed for this workload"
    with open(path) as f:
        rec = json.load(f)
    t = rec["traffic_bytes_per_launch"].get(str(w))
    return t, f"{rec['summary']}: {rec['how']}"


def slice_kernel_name(w):
    """The SpMM kernel that serves (N, w) tables (csrc/spmm.hip)."""
    return {8: "spmm_pair_kernel", 16: "spmm_slice_kernel<4>", 32: "spmm_slice_kernel<8>"}.get(w, f"spmm_rows_kernel<{w // 4}>")


def time_spmm_kernel(trainer, iters=50):
    """Mean duration (s) of the propagation SpMM launch in the three flavours a step issues, HIP events on
    the launch stream: dense (forward layers / inner backward layers: all rows, perturb epilogue),
    row-masked (last forward layer: batch rows only) and column-masked (first backward layer: batch
    columns only).  An XSimGCL step with L layers issues 2L launches: 2L-2 dense + 1 + 1."""
    from selfrec_amd import ops
    adj = trainer.adj                                         # (this rank's rows when the graph is sharded)
    x, y = trainer.E0, trainer._loc(trainer.Ha)
    stamp = (trainer.cursor[1:2] - 1).contiguous()            # the marks of the batch that just ran
    sl = trainer._slice_kw()                                  # (column-sharded: where the slice sits in the row)
    flavours = {
        "dense": ops.make_epilogue(perturb_eps=trainer.eps, rng_seed=1, rng_offset=0, **sl),
        "row_masked": ops.make_epilogue(perturb_eps=trainer.eps, rng_seed=1, rng_offset=0,
                                        row_mark=trainer._loc(trainer.mark), mark_stamp=stamp, **sl),
        "col_masked": ops.make_epilogue(col_mark=trainer.mark, mark_stamp=stamp, **sl),
    }
    pattern = {}
    if getattr(trainer, "vfree", False):
        # value-free launches (layers >= 2 and every backward product but the first): pattern + row scale
        sc = dict(row_scale=trainer.dinv, scale_in=True, scale_out=True)
        flavours["dense_value_free"] = ops.make_epilogue(perturb_eps=trainer.eps, rng_seed=1, rng_offset=0, **sc)
        flavours["row_masked_value_free"] = ops.make_epilogue(perturb_eps=trainer.eps, rng_seed=1, rng_offset=0,
                                                              row_mark=trainer.mark, mark_stamp=stamp, **sc)
        pattern = {"dense_value_free": True, "row_masked_value_free": True}
    out = {}
    adj_cm = getattr(trainer, "adj_cm", None) or adj          # (the column-masked launch runs on its own plan: engine.py)
    for name, ep in flavours.items():
        kw = {"pattern": True} if pattern.get(name) else {}
        m = adj_cm if name == "col_masked" else adj
        for _ in range(5):
            ops.spmm(m, x, out=y, epilogue=ep, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(iters):
            ops.spmm(m, x, out=y, epilogue=ep, **kw)
        b.record()
        torch.cuda.synchronize()
        out[name] = a.elapsed_time(b) / iters * 1e-3
    n = 2 * max(trainer.L, 1)
    if pattern and n >= 4:
        # L layers: forward = 1 dense (values) + (L-2) dense value-free + 1 row-masked value-free;
        # backward = 1 column-masked (values) + (L-1) dense value-free
        out["step_mix"] = (out["dense"] + (n - 4) * out["dense_value_free"] + out["row_masked_value_free"] + out["col_masked"]
                           + out["dense_value_free"]) / n
    else:
        out["step_mix"] = ((n - 2) * out["dense"] + out["row_masked"] + out["col_masked"]) / n if n >= 2 else out["dense"]
    return out


def stream_bandwidth(dev):
    """Measured streaming rates of this GPU with the library's own elementwise kernel, y = a*x + b*y
    (srh_axpby: 2 reads + 1 write per element): arrays that stay in the 256 MiB Infinity Cache (the regime
    of the engine's 17.8 MB tables) and arrays far beyond it (HBM proper) -- the achievable counterparts
    of the 8 TB/s spec (tools/stream_bw.py prints the same for Adam's 7-stream pattern)."""
    from selfrec_amd import ops
    out = {}
    for label, mib in (("infinity_cache_64MiB_arrays", 64), ("hbm_1GiB_arrays", 1024)):
        n = mib * (1 << 20) // 4
        x, y = torch.ones(n, device=dev), torch.ones(n, device=dev)
        for _ in range(3):
            ops.axpby(0.5, x, 0.5, y)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10):
            ops.axpby(0.5, x, 0.5, y)
        b.record(); torch.cuda.synchronize()
        out[label] = round(3.0 * n * 4 * 10 / (a.elapsed_time(b) * 1e-3) / 1e9, 1)
        del x, y
    return out


def spmm_roofline(args, trainer, sharded, dp, step_s, g):
    """The roofline block of one trainer: its dominant propagation launch against HBM on algorithmic bytes,
    the counters' traffic where a PMC pass of this kernel source is committed, and the bare gather stream on the live
    launch's measured lower bound (ops.spmm_gather_bound) and its vector-memory roofline."""
    from selfrec_amd import ops
    if trainer.L < 1:
        return None
    stream = stream_bandwidth(trainer.dev)      # (first: it allocates and frees 2 GiB -- the chip idles through the frees)
    t_spmm = time_spmm_kernel(trainer)
    alg = spmm_alg_bytes(trainer.adj.nnz, trainer.adj.shape[0], trainer.adj.shape[1], trainer.w)
    # the dominant launch: the value-free dense product when the engine uses it (2L - 3 of the 2L launches of a
    # step), else the dense product with values.  Algorithmic bytes stay SURVEY 8(d)'s CSR figure either way.
    dom = "dense_value_free" if "dense_value_free" in t_spmm else "dense"
    t_spmm["dominant"] = t_spmm[dom]
    ach = alg / t_spmm[dom] / 1e9
    cols = bool(getattr(trainer, "cols", False))
    if not sharded or dp:              # (data parallel: every rank runs the single-GPU launch)
        traffic, traffic_note = pmc_traffic(args)
    elif cols and trainer.w != args.emb:
        traffic, traffic_note = pmc_traffic_cols(args, trainer.w)
    else:
        traffic, traffic_note = None, "PMC passes exist for the unsharded and the column-sharded launches only"
    # What the launch could at best be on this chip, measured two ways:
    #  * gather_bound_us -- srh_spmm_gather_bound: the product's own task list, XCD placement, column stream and eight-in-
    #    flight gathers, with nothing after them (no values / reduction / hand-off / epilogue / y).  A strict subset of the
    #    product's work on its schedule: launch_us >= gather_bound_us, frac_of_attainable = bound / launch <= 1.
    #  * vector_memory -- the second roofline of this kernel: nnz x d x 4 bytes of x rows have to pass the per-CU vector-
    #    memory path whatever the caches hold; peak = the same number of row fetches from a table small enough to sit in
    #    every XCD's L2 (1 MiB), measured here with the stand-alone gather stream.
    floor = None
    if trainer.w in (64, 128, 256) and not cols:
        try:
            bound_us = ops.spmm_gather_bound(trainer.adj, trainer.E0)
            gather_bytes = trainer.adj.nnz * trainer.w * 4
            small_rows = (1 << 20) // (trainer.w * 4)
            small = trainer.E0[:small_rows].contiguous()
            idx_small = (trainer.adj.indices % small_rows).contiguous()
            l2_us = ops.gather_floor_probe(idx_small, small)
            floor = {"gather_bound_us": round(bound_us, 2),
                     "gather_bound_what": "srh_spmm_gather_bound: same tasks, XCD shares and 8-in-flight gathers as the launch, "
                                          "nothing after them",
                     "frac_of_attainable": round(bound_us / (t_spmm[dom] * 1e6), 4),
                     "vector_memory": {"bound": "l2_gather", "unit": "TB/s", "bytes_per_launch": gather_bytes,
                                       "achieved": round(gather_bytes / t_spmm[dom] / 1e12, 2),
                                       "peak": round(gather_bytes / (l2_us * 1e-6) / 1e12, 2),
                                       "frac": round(l2_us / (t_spmm[dom] * 1e6), 4),
                                       "peak_what": f"{trainer.adj.nnz} fetches of {trainer.w * 4}-byte rows from a 1 MiB table "
                                                    f"(L2-resident on every XCD): {l2_us:.1f} us"}}
        except Exception as e:          # (a footnote never costs the line)
            floor = {"gather_bound_us": None, "gather_bound_error": f"{type(e).__name__}: {e}"}
    step_bytes = step_alg_bytes(args.model, 2 * g.n_edges, g.n_nodes, args.emb, args.layers, args.batch)
    return {"bound": "hbm",
            "kernel": (f"{slice_kernel_name(trainer.w)} "
                       f"(one propagation layer over the whole graph for this rank's {trainer.w} of "
                       f"{args.emb} columns, perturb epilogue)") if cols else
                      (f"spmm_rows_kernel<{args.emb // 4}> (one propagation layer over "
                       f"{'the rows of one rank of the' if sharded and not dp else 'the whole'} graph, "
                       "perturb epilogue; split rows finished in-kernel"
                       + ("; value-free form: pattern of A over a table pre-scaled by D^-1/2, row scale in "
                          "the epilogue" if "dense_value_free" in t_spmm else "") + ")"),
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
            **(floor or {}),
            "measured_stream_GBps": stream,
            "traffic_source": traffic_note,
            "traffic_GBps": round(traffic / t_spmm["dominant"] / 1e9, 1) if traffic else None,
            "alg_bytes_per_launch": alg, "launch_us": round(t_spmm["dominant"] * 1e6, 2),
            # the value-free launch streams no value array -- the same launch priced by the
            # bytes ITS formulation has to move (indices + indptr + D^-1/2 + x + y), next to SURVEY
            # 8(d)'s figure for the problem (CSR with values) that `achieved` / `frac` use
            **({"value_free_byte_model": {
                "bytes_per_launch": alg - trainer.adj.nnz * 4 + trainer.adj.shape[0] * 4,
                "achieved": round((alg - trainer.adj.nnz * 4 + trainer.adj.shape[0] * 4) / t_spmm[dom] / 1e9, 1),
                "frac": round((alg - trainer.adj.nnz * 4 + trainer.adj.shape[0] * 4) / t_spmm[dom] / 1e9 / HBM_PEAK_GBS, 4)}}
               if dom == "dense_value_free" else {}),
            "with_values": ({"launch_us": round(t_spmm["dense"] * 1e6, 2),
                             "achieved": round(alg / t_spmm["dense"] / 1e9, 1),
                             "frac": round(alg / t_spmm["dense"] / 1e9 / HBM_PEAK_GBS, 4)}
                            if "dense" in t_spmm else None),
            "launch_us_by_flavour": {k: round(v * 1e6, 2) for k, v in t_spmm.items()},
            "note": "rocprofv3's per-kernel average mixes the three flavours: compare it with "
                    "launch_us_by_flavour.step_mix (profiles/)",
            "step_alg_bytes": step_bytes,
            "step_GBps": round(step_bytes / step_s / 1e9, 1) if step_s else None}
