"""Fused training step of the LightGCN family on one MI355X.

What a model file of the reference spells as ~60 ATen launches per step --
cat, L x sparse.mm, rand_like/normalize/sign/mul/add, stack, mean, 3 fancy-index gathers,
bpr/l2/InfoNCE elementwise chains, their autograd mirror images, dense Adam -- is issued here
as ~25 hand-written HIP kernels whose sequence is *numerically specified by* those model
files:

    model      encoder passes / step               spec (reference file:lines)
    MF         none (F = E0)                       model/graph/MF.py:13-31
    LightGCN   1, mean over layers 0..L            model/graph/LightGCN.py:17-36,68-78
    XSimGCL    1 perturbed, mean over 1..L, CL     model/graph/XSimGCL.py:23-50,83-101
               view = layer l*
    SimGCL     1 clean + 2 perturbed               model/graph/SimGCL.py:21-50,81-93
    SGL        1 + 2 on edge-dropped graphs        model/graph/SGL.py:24-47,98-125

Data layout in HBM (fp32 row-major, one allocation each, never re-created):
    E0   (N, d)  the parameter table; user_emb = E0[:U], item_emb = E0[U:] are views, so the
                 reference's torch.cat is free;   m, v  Adam moments, same shape
    Y_k  (N, d)  layer outputs, F (N, d) their mean, written by the SpMM epilogue
    gF, gCL, gE0, H_a, H_b (N, d) gradient buffers
Backward uses two facts of the model family: the perturbation has identity Jacobian
(sign() has zero gradient, XSimGCL.py:91), and A_hat is symmetric, so every backward
product is the same SpMM kernel; encoder passes that share A_hat and the layer mean share
one backward chain because it is linear (SimGCL: gradients of the three passes are summed
before a single chain).

Indices: the C++ sampler produces a whole epoch (shuffle + batches + sorted unique ids) on a
host thread; it is uploaded once per epoch, and a one-block kernel (srh_batch_fetch) stages
batch b into fixed buffers and publishes its sizes in device memory.  Every kernel of the
step reads sizes from there, so the step has no host-dependent argument and can be
captured once in a hipGraph and replayed.
"""
from __future__ import annotations

import os
import threading

import numpy as np
import torch
import torch.distributed as _dist

from . import layouts, ops
from ._lib import SelfrecHipError
from .comm import AbiComm, TorchComm, TwoHopRows, default_comm, shard_adjacency  # noqa: F401  (public: dist.py and the tests import them from here)
from .layouts import SLICE_WIDTHS, parse_grid  # noqa: F401

MODELS = ("MF", "LightGCN", "XSimGCL", "SimGCL", "SGL")


class FusedTrainer:
    reuse_epoch_arrays = True     # EpochPrefetcher draws epochs into two sets of host arrays in turn (see its __init__)

    def __init__(self, data, emb_size, *, model, n_layers=2, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2,
                 tau=0.2, layer_cl=1, drop_rate=0.1, aug_type=1, batch_size=2048, user_emb=None, item_emb=None,
                 noise_fn=None, rng_seed=0x5E1F0EC, use_graph=False, device=None, shard=False, comm=None,
                 nce_precision=None, sampler_seed=None):
        if model not in MODELS:
            raise SelfrecHipError(f"FusedTrainer: unknown model {model!r}")
        ops.require_gpu()
        self.model, self.data = model, data
        # Any embedding.size (base/recommender.py:16): every table is stored with zero columns up to the next width the
        # kernels serve (d_valid -> d).  Zero columns stay exactly zero through products, means, losses (inner products,
        # norms and F.normalize ignore them), their gradients are zero and Adam's update of a zero gradient is zero; the
        # perturbation normalises its noise over the d_valid real columns (epilogue field noise_d_valid).
        self.d_valid = int(emb_size)
        widths = ops.NCE_WIDTHS if model in ("XSimGCL", "SimGCL", "SGL") else ops.ROW_WIDTHS
        self.d = ops.padded_width(self.d_valid, widths) if self.d_valid > 0 else None
        if self.d is None:
            raise SelfrecHipError(f"{model}: embedding.size = {emb_size} -- the fused kernels serve up to {widths[-1]} columns"
                                  + (" (InfoNCE; the op-level tier, selfrec_amd.dropin, takes wider rows)" if widths[-1] < 256 else ""))
        self.L = 0 if model == "MF" else int(n_layers)
        self.lr, self.reg, self.cl_rate, self.eps, self.tau = float(lr), float(reg), float(cl_rate), float(eps), float(tau)
        if model == "SimGCL":
            self.tau = 0.2                       # hard-coded in the reference, SimGCL.py:48-49
        self.layer_cl, self.drop_rate, self.aug_type = int(layer_cl), float(drop_rate), int(aug_type)
        self.B = int(batch_size)
        self.noise_fn = noise_fn                  # (N, d) -> tensor; None = in-kernel counter RNG
        # arithmetic of InfoNCE's two n x n x d products, carried by THIS trainer and passed with every loss call ('split' |
        # 'f32'; None = the process default -- f32 unless srh_infonce_set_precision / SRH_NCE_SPLIT16 says otherwise --, resolved at launch / capture time)
        if nce_precision is not None and nce_precision not in ops.NCE_PRECISIONS:
            raise SelfrecHipError(f"FusedTrainer: nce_precision {nce_precision!r}: one of {sorted(ops.NCE_PRECISIONS)} or None")
        self.nce_precision = nce_precision
        if self.d == 256 and model in ("XSimGCL", "SimGCL", "SGL") and nce_precision in (None, "f32"):
            # (the all-f32 MFMA passes serve d = 64 / 128: loss_torch.py:46-47's fp32 matmul has no exact counterpart here)
            if nce_precision == "f32":
                raise SelfrecHipError("FusedTrainer: nce_precision='f32' is served for embedding sizes up to 128; "
                                      f"embedding.size = {emb_size} pads to 256 columns -- pass nce_precision='split'")
            if ops.get_infonce_precision() == "f32":
                import logging
                logging.getLogger("selfrec_amd").warning(
                    "%s with embedding.size = %d (256 columns): InfoNCE's two products run on split 16-bit operands "
                    "(nce_precision='split'), not the library's fp32 default, which serves up to 128 columns", model, int(emb_size))
        self.rng_seed = int(rng_seed)
        if model != "MF" and self.L < 1:
            raise SelfrecHipError("n_layers must be >= 1")
        if model == "XSimGCL" and not (0 <= self.layer_cl <= self.L):
            raise SelfrecHipError("l_star must be in [0, n_layer]")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dev = dev
        self.U, self.I = data.user_num, data.item_num
        self.N = self.U + self.I
        # ---- placement (layouts.py): who owns which rows and columns of the tables, what the ranks exchange per step.
        # `shard`: False (one GPU) | "rows" | "cols" | "2d[:GCxGR]" | "dp"; every layout is the SAME step code below,
        # run against the placement's three components -- self.rows, self.colx, self.sync.
        self.place = layouts.make_placement(shard, comm, int(emb_size), default_comm, TorchComm.grid)
        self.rows, self.colx, self.sync = self.place.rows, self.place.colx, self.place.sync
        self.layout = self.place.name
        self.G, self.rank = self.place.G, self.place.rank
        self.rng_seed = self.sync.rng_seed(self.rng_seed)
        N, d, B = self.N, self.d, self.B
        self.colx.bind(d, self.d_valid)
        self.rows.bind(N, self.U, dev)
        self.w, self.col0 = self.colx.w, self.colx.col0     # width / first column of this rank's tables
        self.n_pad, self.P = self.rows.n_pad, self.rows.P   # rows this rank owns / rows of a whole table
        # Fixed-order batch gradients (srh_batch_segments_t): the sampler hands every batch's row -> slot lists over with the
        # epoch, and the loss section's last kernel writes every touched gradient row ONCE, summed in slot order -- no float
        # atomics, so a step is reproducible bit for bit (run twice, bit-compare; the reference's single-threaded
        # index_put(accumulate) backward of XSimGCL.py:30 is).  Whole rows and columns on this rank (single, dp); the
        # column-block layouts run the losses on compact slot tables and keep the atomic scatter.  SRH_DET_SCATTER=0: atomics.
        # (Decided before the first epoch is drawn: the epoch carries the lists.)
        self.det_scatter = (dev.type == "cuda" and self.place.single_gpu_step and not self.colx.split
                            and hasattr(ops, "_segments") and os.environ.get("SRH_DET_SCATTER", "1") != "0")
        # The sampler needs the interaction arrays only: with `sampler_seed` given, the FIRST epoch is drawn on a host thread
        # from here on -- under the graph upload, the SpMM plans and the XCD calibration below (all of them C++ / device work
        # that releases the GIL) -- instead of after them: 3.6 of 13.3 s to the first trained step at 1 M x 500 k.
        # (SGL draws its edge-dropped views from the same stream and needs the graph for them: sampled after construction.)
        self.sampler = ops.Sampler(data.train_u, data.train_i, self.U, self.I)
        self._first_epoch, self._first_epoch_seed = None, None
        if sampler_seed is not None and model != "SGL":
            self._first_epoch_seed = int(sampler_seed)
            self.sampler.seed(self.sync.sampler_seed(int(sampler_seed)))
            self._first_epoch = EpochPrefetcher(self, first=True)
            self._first_epoch.start()
        self.graph = self.rows.build_graph(data, dev, column_classes=not self.colx.split)
        self.adj = self.graph.adj
        g = self.graph
        P = self.P

        w = self.w

        def buf(rows=P):
            return torch.zeros((rows, w), dtype=torch.float32, device=dev)

        self.E0 = buf()
        if user_emb is None or item_emb is None:
            ue = torch.nn.init.xavier_uniform_(torch.empty(self.U, self.d_valid))      # XSimGCL.py:76-80
            ie = torch.nn.init.xavier_uniform_(torch.empty(self.I, self.d_valid))
        else:
            ue, ie = torch.as_tensor(user_emb, dtype=torch.float32), torch.as_tensor(item_emb, dtype=torch.float32)
        if ue.shape[1] != self.d_valid or ie.shape[1] != self.d_valid:
            raise SelfrecHipError(f"initial tables have {ue.shape[1]} / {ie.shape[1]} columns, embedding.size is {self.d_valid}")
        whole = torch.zeros((N, d), dtype=torch.float32)
        whole[:, :self.d_valid] = torch.cat([ue, ie])
        self.E0[self.rows.pos_dev] = whole[:, self.col0:self.col0 + w].contiguous().to(dev)
        # the step code assumes the tables (and, per epoch, the sampled batches) are replicated: check, don't trust
        self._assert_replicated("initial embedding tables",
                                [ue.double().sum().item(), ie.double().sum().item(), ue.double().abs().sum().item(),
                                 ie.double().abs().sum().item()])
        self.m, self.v = buf(self.n_pad), buf(self.n_pad)       # Adam state: owned rows only
        self.gE0 = buf()
        self.F = self.E0 if model == "MF" else buf()
        self.gF = self.gE0 if model == "MF" else buf()
        self.Y = [buf() for _ in range(self.L)]
        self.Ha, self.Hb = (buf(), buf()) if self.L >= 1 else (None, None)
        self.gCL = buf() if model == "XSimGCL" else None
        self.gReg = buf() if model == "LightGCN" else None    # ego-row regulariser gradient (sparse rows)
        # activity marks: mark[node] == optimiser step  <=>  the node is a row of the current batch
        self.mark = torch.zeros(P, dtype=torch.int32, device=dev)
        self.use_marks = True
        # models whose gradient buffers only ever hold O(batch) non-zero rows are reset row-wise
        self.sparse_reset = True
        self.views = []                           # SimGCL / SGL: extra passes [(F_v, Y_v list, gF_v)]
        if model in ("SimGCL", "SGL"):
            for _ in range(2):
                self.views.append({"F": buf(), "Y": [buf() for _ in range(self.L)], "gF": buf()})
        self.view_adj = [None, None]              # SGL: dropped adjacencies (DeviceCSR)
        self._view_vals = [torch.empty_like(self.adj.vals), torch.empty_like(self.adj.vals)] if model == "SGL" else None
        self.losses = torch.zeros(4, dtype=torch.float64, device=dev)      # bpr, reg, cl
        self.stage = {k: torch.zeros(B, dtype=torch.int32, device=dev) for k in ("u", "i", "j", "uniq_u", "uniq_i")}
        self.stage_cat = torch.zeros(2 * B, dtype=torch.int32, device=dev)  # SGL: [uniq users ; uniq items + U]
        self.meta = torch.zeros(4, dtype=torch.int32, device=dev)          # rows, n_uniq_u, n_uniq_i, batch no
        self.cursor = torch.tensor([0, 1], dtype=torch.int64, device=dev)  # batch no, optimiser step (1-based)
        self.now = self.cursor.clone()             # batch_fetch's copy of the cursor: what Adam reads its step from
        # (running batch_fetch beside the first L - 1 products and zero_rows beside Adam on a side stream was measured:
        # 0.3435 against 0.3167 ms per step -- fork / join edges inside the captured graph cost more than the two
        # launches; round 1 saw the same with BPR beside InfoNCE.  The reset is folded into Adam instead.)
        single = dev.type == "cuda" and self.place.single_gpu_step     # whole rows and columns here (single, dp)
        self.fused_reset = single
        # Value-free products: on a unit-weight graph A_hat_ij = d_i^-1/2 d_j^-1/2, so a layer that reads
        # a table stored PRE-SCALED by D^-1/2 needs no value stream -- the kernel sums x rows over the pattern and the
        # epilogue multiplies the row by d_i^-1/2 (-10 % per launch at the Yelp2018 shape).  Layer outputs Y_1 .. Y_(L-1)
        # and the backward intermediates are therefore kept as Z = D^-1/2 Y: the next product gathers them as they are,
        # the last layer's mean un-scales them on the batch rows, InfoNCE is invariant to a positive row scale (its
        # gradient w.r.t. Z re-enters the chain times d_i^-1/2), BPR / L2 read the true mean F.  The first product of a
        # chain reads true values (E0, gF) and keeps the value array.
        self.vfree = (single and self.L >= 2 and self.d >= 64
                      and model in ("LightGCN", "XSimGCL", "SimGCL") and self.graph.weight is None)
        self.dinv = self.graph.dinv if self.vfree else None
        # The column-masked launch (first backward product: 3/4 of its entries are dead, its waves run on a latency chain)
        # gains nothing from the L2 classes of the dense plan and loses to their imbalance -- the batch's item rows are the
        # heavy ones: on a class-free plan of the same arrays (256-entry segments, tasks dealt to all XCDs alike) it takes
        # 23.1 instead of 27.4 us at the Yelp2018 shape (profiles/r02_i_spmm_plans_by_flavour.txt; the dense launches lose
        # 10 us without the classes, the row-masked one is indifferent).
        self.adj_cm = None
        if single and self.L >= 1 and self.d >= 64:
            self.adj_cm = self.adj.replanned(split_len=256)
        # batch_fetch as a rider of the step's first product (see _step_front): models whose step starts with a plain
        # srh_spmm_f32 launch of a layer that is not the last
        self.ride_fetch = single and self.L >= 2 and self.d >= 64 and model in ("LightGCN", "XSimGCL", "SimGCL")
        # Adam inside the LAST backward product's row epilogue (SRH_EPI_ADAM): the finished gradient row updates its
        # parameter row where it is produced -- no 18 MB gradient written and read back, no separate 21 us pass; the same
        # launch clears the batch rows of the sparse gradient buffers and advances the cursor (what Adam's pass did).  Whole
        # rows on one GPU only: under dp the all-reduce sits between the product and the optimiser.  L >= 2: at L = 1 the
        # product's x is gF itself, which the launch would be clearing.  SRH_FUSE_ADAM=0: the separate pass (A/B).
        self.adam_coef = torch.zeros(2, dtype=torch.float32, device=dev)
        self.fuse_adam = (single and not self.sync.active and self.sparse_reset and self.L >= 2 and self.w >= 64
                          and model != "MF" and tuple(self.m.shape) == tuple(self.E0.shape)
                          and getattr(ops, "ADAM_EPILOGUE", False) and os.environ.get("SRH_FUSE_ADAM", "1") != "0")
        self.n_cat = torch.zeros(1, dtype=torch.int32, device=dev)
        self.bpr_ws = ops.bpr_ws(B, dev)
        self.nce_ws = None
        if model in ("XSimGCL", "SimGCL", "SGL"):       # user side + item side share one workspace / launch set
            one = ops.infonce_ws(2 * B if model == "SGL" else B, d, dev)
            self.nce_ws = torch.empty(one.numel() * (1 if model == "SGL" else 2), dtype=torch.uint8, device=dev)
        E = self.sampler.n_edges
        self.epoch_batches = (E + B - 1) // B
        # The device holds TWO epochs back to back (srh_batch_fetch_args_t::half_batches): the steps read one half while the
        # next epoch's arrays are copied into the other on a stream of their own (stage_epoch, run by the prefetch thread),
        # so an epoch boundary costs one cursor write instead of a 25 MB upload in front of the next step (1.5 ms at the
        # Yelp2018 shape -- five steps).  Fixed addresses: a captured hipGraph keeps pointing at them across epochs.
        nb = self.epoch_batches
        self._epoch_slot = {"u": nb * B, "i": nb * B, "j": nb * B, "uniq_u": nb * B, "uniq_i": nb * B,
                            "n_uniq_u": nb, "n_uniq_i": nb}                       # entries of ONE half
        if self.det_scatter:
            self._epoch_slot.update({"n_uniq_n": nb, "seg_rows": 3 * nb * B, "seg_end": 3 * nb * B, "seg": 3 * nb * B,
                                     "seg_a": 3 * nb * B, "seg_b": nb * B})
        self._epoch_dev = {k: torch.zeros(2 * n, dtype=torch.int32, device=dev) for k, n in self._epoch_slot.items()}
        self._live_half = None                 # the half the steps read (None: nothing uploaded yet)
        self._half_free = [None, None]         # event on the step stream: every step that read this half has been enqueued
        self._copy_stream = None
        self._pinned = [None, None]
        self._stage_pending = [None, None]     # event behind the last staged copy INTO each half (upload_epoch orders in-line copies behind it)
        # stage_epoch (prefetch thread) allocates pinned memory, creates a stream, records and waits for events; a hipGraph
        # capture in "global" error mode on the main thread is invalidated by such calls from ANY thread.  One lock: the worker
        # makes no HIP call while a capture is open, and a capture does not open in the middle of a staging.
        self._hip_lock = threading.RLock()
        self._epoch_ready = False
        self.step_count = 0
        # (a sharded step is captured only on request: RCCL collectives inside a hipGraph could not be
        # exercised beyond one rank on the development box)
        # Column-sharded steps capture as TWO graphs with the one all-gather issued between them, so RCCL stays
        # outside the captured region (one GPU, stand-in communicator, 8 ranks: 197 us captured, 211 us eager);
        # SRH_SHARDED_GRAPH=0 launches eagerly.  The row-sharded step has a collective after every product: it
        # launches eagerly unless SRH_SHARDED_GRAPH=1 asks for RCCL inside the capture.
        self.use_graph = bool(use_graph) and self.place.graph_by_default(os.environ.get("SRH_SHARDED_GRAPH"))
        self._graph = None
        self._noise_call = 0
        # counter RNG layout: (optimiser step) * rng_stride + (perturbed-layer call of the step) * P + row; SimGCL makes
        # 2L calls per step, XSimGCL L -- the stride leaves room for all of them at any depth
        self._rng_calls = max(16, 2 * self.L)
        self.colx.init_exchange(self)                  # (column blocks: the compact batch-row tables; else nothing)
        self.xcd_shares = self._calibrate_xcd_shares()

    # ------------------------------------------------------------------------------------
    # start-up calibration of the dense plan's XCD shares
    # ------------------------------------------------------------------------------------
    def _calibrate_xcd_shares(self, rounds=5):
        """The plan deals every XCD the same number of workgroups, but the XCDs do not take the same time over them (at the
        Yelp2018 shape they finish a propagation launch 3.5 us apart: the odd column classes run 15 % slower for the same
        non-zeros).  A few probe launches of the step's dominant product (srh_spmm_f32_probe: per-XCD finish times) move the
        last workgroups of the late XCDs' queues to the early ones' (srh_spmm_plan_set_xcd_shares) until they finish
        together: 45.1 -> 42.7 us per dense launch in the lab (profiles/r02_j_xcd_balance_closed_loop.txt).  Same tasks, same
        sums -- placement only.  Once per (plan, table width), before anything captures a launch of the plan;
        SRH_XCD_CALIBRATE=0 keeps the equal dealing.  Returns the shares, or None."""
        self.xcd_calibration = None                    # the record of what was decided (also logged); None: equal dealing
        if not layouts.calibration_enabled() or self.dev.type != "cuda" or not self.place.single_gpu_step:
            return None
        if self.L < 1 or self.d != 64:                  # (the entry points serve d = 128 / 256 too; measured at d = 64 only so far)
            return None
        # Cached on the PLAN OBJECT (so: per plan identity) and table width.  One calibration per plan and width, by the
        # first trainer that asks: re-calibrating for another launch flavour would rewrite a task list that an earlier
        # trainer's captured graph still launches (ADVICE r02 asked for the key to include the plan -- it is the plan's
        # own attribute -- or the flavour; the flavours of one plan share its list by construction).
        done = self.adj.__dict__.setdefault("_xcd_calibrated", {})
        key = self.d
        if key in done:
            self.xcd_calibration = self.adj.__dict__.get("_xcd_calibration_record", {}).get(key)
            return done[key]
        nb = (ops.spmm_plan_run_tasks(self.adj, self.d) + 3) // 4
        if nb < 4096:                                   # nothing to balance on a small graph
            done[key] = None
            return None
        kw = dict(perturb_eps=self.eps, rng_seed=1, rng_offset=0) if self.model == "XSimGCL" else {}
        if self.vfree:
            kw.update(row_scale=self.dinv, scale_in=True, scale_out=True)
        ep = ops.make_epilogue(**kw) if kw else None
        canon = np.array([len(range(k, nb, 8)) for k in range(8)], dtype=np.int64)
        shares, best = canon.copy(), (float("inf"), canon.copy())
        spread = []                                     # us between the first and the last XCD to finish, per round

        def finish_times():
            runs = [ops.spmm_probe(self.adj, self.E0, self.Ha, epilogue=ep, pattern=bool(self.vfree))[0] for _ in range(3)]
            return np.median(np.stack(runs), axis=0)
        finish_times()                                  # warm-up (module load, caches)
        for rnd in range(rounds):
            fin = finish_times()
            spread.append(float(fin.max() - fin.min()))
            if fin.max() < best[0]:
                best = (float(fin.max()), shares.copy())
            if fin.max() - fin.min() < 0.4 or rnd == rounds - 1:
                break
            per_block = fin.mean() / (nb / 8.0)         # us of an XCD's finish time per workgroup of its queue
            new = shares - np.round((fin - fin.mean()) / per_block).astype(np.int64)
            new = np.clip(new, (canon * 6) // 10, (canon * 14) // 10)
            new[int(np.argmin(fin))] += nb - int(new.sum())
            if new.min() < 0 or int(new.sum()) != nb:
                break
            shares = new
            ops.spmm_set_xcd_shares(self.adj, self.d, shares)
        if not np.array_equal(best[1], shares):
            ops.spmm_set_xcd_shares(self.adj, self.d, None if np.array_equal(best[1], canon) else best[1])
        done[key] = best[1]
        # what was decided, on the record (logging: logger "selfrec_amd", level INFO) and on the trainer
        self.xcd_calibration = {"shares": [int(v) for v in best[1]], "canonical": [int(v) for v in canon],
                                "xcd_finish_spread_us_by_round": [round(v, 2) for v in spread], "last_finish_us": round(best[0], 2)}
        self.adj.__dict__.setdefault("_xcd_calibration_record", {})[key] = self.xcd_calibration
        import logging
        logging.getLogger("selfrec_amd").info(
            "XCD shares of the dense SpMM plan calibrated at start-up (d = %d, %d workgroups): %s (equal dealing: %s); XCDs "
            "finished %.2f us apart before, %.2f after; SRH_XCD_CALIBRATE=0 keeps the equal dealing", self.d, nb,
            self.xcd_calibration["shares"], self.xcd_calibration["canonical"], spread[0] if spread else 0.0,
            spread[-1] if spread else 0.0)
        return best[1]

    # ------------------------------------------------------------------------------------
    # the placement's answers under the names the rest of the package (and the tests) read
    # ------------------------------------------------------------------------------------
    @property
    def ops(self):
        """the kernel module the step runs on (this module's `ops`: the CPU tests swap it for stand-ins)"""
        return ops

    sharded = property(lambda self: self.rows.dealt)            # rows of the graph / tables dealt over Gr ranks
    cols = property(lambda self: self.colx.split)               # columns of the tables split over Gc ranks
    dp = property(lambda self: self.sync.active)                # data parallel: gradient all-reduce before Adam
    Gc = property(lambda self: self.colx.blocks)
    cr = property(lambda self: self.colx.block)
    Gr = property(lambda self: self.rows.parts)
    rr = property(lambda self: self.rows.part)
    loc = property(lambda self: self.rows.own)
    _pos = property(lambda self: self.rows.pos)
    _pos_dev = property(lambda self: self.rows.pos_dev)
    comm = property(lambda self: self.sync.comm if self.sync.active else self.colx.comm if self.colx.split else self.rows.comm)
    comm_rows = property(lambda self: self.rows.comm)

    def _exchange(self):
        """the step's one collective of the column layouts (tests drive virtual ranks through it)"""
        self.colx.exchange()

    def _dp_allreduce(self):
        """the step's one collective of the data-parallel layout"""
        self.sync.reduce(self)

    def _assert_replicated(self, what, values):
        for comm in self.place.comms():
            check = getattr(comm, "assert_replicated", None)       # (test stand-in communicators may not have it)
            if check is not None and comm.world > 1:
                check(what, values, self.dev)                      # (2-D: row part + column block span the grid)

    # ------------------------------------------------------------------------------------
    # views of the tables
    # ------------------------------------------------------------------------------------
    @property
    def user_emb(self):
        t = self._valid(self.colx.full(self.E0))         # (column blocks: a collective -- every rank must ask)
        return self.rows.rows_of_users(t)

    @property
    def item_emb(self):
        t = self._valid(self.colx.full(self.E0))
        return self.rows.rows_of_items(t)

    def _valid(self, t):
        """The real columns of a whole-row table (tables are stored zero-padded to a width the kernels serve)."""
        return t if self.d_valid == self.d else t[:, :self.d_valid]

    def _loc(self, t):
        """The rows of a table this rank owns (the whole table unless the rows are dealt)."""
        return self.rows.mine(t)

    def _allgather(self, t):
        """Make a table whose owned rows were just written whole again on every rank."""
        self.rows.make_whole(t)

    # ------------------------------------------------------------------------------------
    # sampling
    # ------------------------------------------------------------------------------------
    def seed_sampler(self, seed: int):
        """Seed the batch sampler.  Replicated layouts (rows / cols / 2-D) need the SAME stream on every rank; data
        parallel needs a DIFFERENT one per rank (seed + rank): every rank trains on its own batches."""
        if self._first_epoch is not None:
            if int(seed) == self._first_epoch_seed:
                return                                     # (the epoch being drawn since construction IS this seed's first)
            self._drop_first_epoch()                       # another seed after all
        self.sampler.seed(self.sync.sampler_seed(int(seed)))

    def _drop_first_epoch(self):
        """The epoch drawn since construction is not wanted: let its thread finish, and start from a FRESH sampler -- the
        discarded shuffle has permuted the sampler's edge order, which the next seed's stream must not inherit."""
        self._first_epoch.take()
        self._first_epoch = None
        self.sampler = ops.Sampler(self.data.train_u, self.data.train_i, self.U, self.I)

    def seed_sampler_from_python(self):
        """Adopt the global ``random`` state (bit-exact mode, as the reference consumes it).  Data parallel over more than
        one rank has no reference stream to be exact to -- every rank needs ITS OWN batches -- so there one 63-bit draw of
        the (replicated) global stream seeds the sampler, offset by the rank (adopting the state itself would
        hand every rank the same batches)."""
        if self._first_epoch is not None:
            self._drop_first_epoch()
        if self.sync.active and self.sync.world > 1:
            import random
            self.sampler.seed(self.sync.sampler_seed(random.getrandbits(63)) & ((1 << 63) - 1))
            return
        self.sampler.set_state_from_python()

    def sample_epoch_host(self, slot=None):
        """Host part of an epoch: SGL's two edge-dropped views are drawn first (SGL.py:28-29),
        then shuffle + batches.  Pure host work -- safe to run on a worker thread.  (The epoch that has been in the making
        since construction -- ``sampler_seed`` -- is handed out first.)  slot: see ops.Sampler.epoch -- the arrays of a
        slot are the sampler's and are overwritten by the next epoch drawn into the same slot (EpochPrefetcher alternates
        two); None = fresh arrays."""
        if self._first_epoch is not None:
            pre, self._first_epoch = self._first_epoch, None
            return pre.take()
        return self._sample_epoch_host_now(slot)

    def _sample_epoch_host_now(self, slot=None):
        out = {}
        if self.model == "SGL":
            masks = []
            e = self.graph.n_edges
            for _ in range(2):
                if self.aug_type == 0:
                    # node dropout (augmentor.py:10-27): int(U rho) users and int(I rho) items lose all their edges;
                    # as a keep mask over interactions it is the same value-array view as edge dropout
                    dead_u = np.zeros(self.U, dtype=bool)
                    dead_i = np.zeros(self.I, dtype=bool)
                    dead_u[self.sampler.sample_range(self.U, int(self.U * self.drop_rate))] = True
                    dead_i[self.sampler.sample_range(self.I, int(self.I * self.drop_rate))] = True
                    rows = np.repeat(np.arange(self.U), np.diff(self.graph.h_r_indptr))
                    mk = (~dead_u[rows] & ~dead_i[self.graph.h_r_indices]).astype(np.uint8)
                else:                       # aug_type 1 and 2 both take SGL.py:92-93's edge_dropout branch
                    keep = self.sampler.sample_range(e, int(e * (1 - self.drop_rate)))
                    mk = np.zeros(e, dtype=np.uint8)
                    mk[keep] = 1
                masks.append(mk)
            out["masks"] = masks
        ep = self.sampler.epoch(self.B, 1, with_unique=True, slot=slot,
                                **({"with_segments": self.rows.segment_row_offsets()} if self.det_scatter else {}))
        # node ids -> table rows (items follow the users; all-gather order when the rows are dealt)
        out.update(self.rows.epoch_to_table_rows(ep))
        return out

    def epoch_node_ids(self, host=None):
        """(u, i, j) of an epoch as the reference's user / item ids (the staged arrays hold table rows)."""
        host = self._epoch_host if host is None else host
        return self.rows.epoch_node_ids(host, self.P)

    def _next_half(self):
        return 0 if self._live_half is None else 1 - self._live_half

    def stage_epoch(self, host):
        """Start copying a sampled epoch into the half of the device arrays the steps are NOT reading, on the copy stream,
        from pinned memory; `upload_epoch(host)` later only waits for it.  Called by the prefetch thread right after
        sampling (GPU only; without it upload_epoch copies in line).  The target half was last read by the epoch before the
        live one: the copy is ordered behind the event recorded when that epoch's last step had been enqueued."""
        if self.dev.type != "cuda" or "_staged" in host:
            return host
        with self._hip_lock:
            return self._stage_epoch_locked(host)

    def _stage_epoch_locked(self, host):
        torch.cuda.set_device(self.dev)                                   # (a worker thread starts on device 0)
        half = self._next_half()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.dev)
        if self._pinned[half] is None:
            self._pinned[half] = {k: torch.empty(n, dtype=torch.int32).pin_memory() for k, n in self._epoch_slot.items()}
        pin = self._pinned[half]
        if "_pin_done" in pin:
            pin["_pin_done"].synchronize()                              # (two epochs ago: long complete)
        for k, n in self._epoch_slot.items():
            src = torch.from_numpy(host[k])
            pin[k][:src.numel()].copy_(src)
        with torch.cuda.stream(self._copy_stream):
            if self._half_free[half] is not None:
                self._copy_stream.wait_event(self._half_free[half])
            for k, n in self._epoch_slot.items():
                m = int(host[k].size)
                self._epoch_dev[k][half * n:half * n + m].copy_(pin[k][:m], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        pin["_pin_done"] = done
        self._stage_pending[half] = done
        host["_staged"] = (half, done)
        return host

    def upload_epoch(self, host):
        dev = self.dev
        if "masks" in host:
            for v, mk in enumerate(host["masks"]):
                self.view_adj[v] = self.graph.dropped_view(torch.from_numpy(mk).to(dev), out=self._view_vals[v])
        half = self._next_half()
        staged = host.pop("_staged", None)
        if staged is not None and staged[0] == half:
            torch.cuda.current_stream().wait_event(staged[1])           # the copy stream's work, ordered before the next step
        else:
            # in line, on the step stream (in order behind every step that read this half): tests, begin_epoch(), the CPU
            # stand-ins -- or a staged copy that went to the other half (an epoch was uploaded in between).  A staged copy
            # of ANOTHER epoch may still be in flight into this very half (begin_epoch() with a prefetch outstanding): the
            # in-line copy goes behind it, so the late copy cannot overwrite the epoch that is about to go live.
            pending = self._stage_pending[half]
            if pending is not None and dev.type == "cuda":
                torch.cuda.current_stream().wait_event(pending)
            for k, n in self._epoch_slot.items():
                src = torch.from_numpy(host[k])
                self._epoch_dev[k][half * n:half * n + src.numel()].copy_(src, non_blocking=True)
        if dev.type == "cuda" and self._live_half is not None:
            ev = torch.cuda.Event()
            ev.record()                                                   # every step of the epoch that read the old half
            self._half_free[self._live_half] = ev
        if self.place.replicated_batches:   # same seed => same batches on every rank; one tiny collective per epoch says so
            # (data parallel: every rank samples ITS OWN batches by design)
            w3 = np.arange(1, 4, dtype=np.int64)
            self._assert_replicated("the sampled epoch (u, i, j streams)",
                                    [int(host[k].astype(np.int64).sum()) for k in ("u", "i", "j")] +
                                    [int((host[k][:3 * (host[k].size // 3)].astype(np.int64).reshape(-1, 3) * w3).sum())
                                     for k in ("u", "i", "j")])
        self._epoch_host = host
        self._epoch_ready = True
        self._live_half = half
        self.cursor[0:1].fill_(half * self.epoch_batches)

    def begin_epoch(self):
        self.upload_epoch(self.sample_epoch_host())
        return self.epoch_batches

    # ------------------------------------------------------------------------------------
    # encoder
    # ------------------------------------------------------------------------------------
    def _noise(self):
        if self.noise_fn is None:
            return None
        t = torch.as_tensor(self.noise_fn((self.N, self.d_valid)), dtype=torch.float32)     # torch.rand_like(h): XSimGCL.py:90
        if self.d_valid != self.d:
            t = torch.nn.functional.pad(t, (0, self.d - self.d_valid))
        return self.rows.place_noise(t.to(self.dev), self.d)     # (column blocks: whole rows too -- the unit vector spans the row)

    def _rng_offset(self, call):
        """Counter offset of perturbed-layer call number `call` of this step (rows of all ranks)."""
        return (call * self.P + self.rows.rng_row_offset()) & ((1 << 62) - 1)

    def _slice_kw(self):
        """PERTURB on a column slice: tell the kernel where the slice sits in the whole row; on zero-padded rows: where
        the real columns end."""
        kw = self.colx.slice_kw()
        if self.d_valid != self.d:
            kw["d_valid"] = self.d_valid
        return kw

    def _forward_pass(self, adj, Ys, F, *, perturbed, include_ego, batch_rows_only=False, need_last=False,
                      start_layer=0, noises=None, call_base=None):
        """L SpMMs; layer k's epilogue perturbs (optional) and the last one also writes the
        layer mean into F.  Returns nothing; the CL view of XSimGCL is Ys[l*-1] (or E0).
        batch_rows_only: the last layer's output (and F) feed nothing but the batch losses, so
        only the rows marked for this step are computed (the others keep stale values that are
        never read).  start_layer > 0: Ys[:start_layer] are already there (SimGCL's shared first
        product); noises / call_base: injected noise per layer and the RNG call number of layer 0."""
        L = self.L
        x = self.E0 if start_layer == 0 else Ys[start_layer - 1]
        vf = self.vfree and adj is self.adj            # (Ys[0 .. L-2] then hold D^-1/2 Y: see __init__)
        for k in range(start_layer, L):
            kw = {}
            if perturbed:
                noise = noises[k] if noises is not None else self._noise()
                call = self._noise_call if call_base is None else call_base + k
                kw.update(perturb_eps=self.eps, noise=noise, rng_seed=self.rng_seed, rng_offset=self._rng_offset(call),
                          rng_step=self.cursor[1:2] if noise is None else None,
                          rng_stride=self.P * self._rng_calls, **self._slice_kw())
                self._noise_call += 1
            if k == L - 1:
                prev = ([self.E0] if include_ego else []) + Ys[:L - 1]
                kw.update(prev=[self._loc(t) for t in prev], mean_div=float(L + 1 if include_ego else L),
                          mean_out=self._loc(F))
                if vf:
                    kw.update(prev_unscale=([False] if include_ego else []) + [True] * (L - 1))
                if batch_rows_only and self.use_marks:
                    kw.update(row_mark=self._loc(self.mark), mark_stamp=self.cursor[1:2])
            if vf:
                # layer 1 reads the true table E0 through the value array; later layers read pre-scaled tables through
                # the pattern.  Every layer but the last stores its output pre-scaled.
                kw.update(row_scale=self.dinv, scale_in=k > 0, scale_out=k < L - 1)
            rider = {}
            if getattr(self, "_rider", None) is not None:
                rider, self._rider = {"fetch": self._rider}, None
            ops.spmm(adj, x, out=self._loc(Ys[k]), epilogue=ops.make_epilogue(**kw) if kw else None,
                     **({"pattern": True} if vf and k > 0 else {}), **rider)
            if k < L - 1 or need_last:
                self._allgather(Ys[k])          # the next layer (or the contrast view) reads every row
            x = Ys[k]
        self._allgather(F)

    def _simgcl_forward(self, adj):
        """SimGCL.py:81-93 three times (clean pass for the recommendation loss, two perturbed views): the
        first layer of all three is the SAME product A.E0, so it is gathered once and leaves three
        outputs (clean, +noise_a, +noise_b); the remaining layers differ in their inputs."""
        L, a, b = self.L, self.views[0], self.views[1]
        if L < 2:                               # the only layer also carries the mean: three plain passes
            self._forward_pass(adj, self.Y, self.F, perturbed=False, include_ego=False, batch_rows_only=True)
            for v in (a, b):
                self._forward_pass(adj, v["Y"], v["F"], perturbed=True, include_ego=False, batch_rows_only=True)
            return
        # injected noise is drawn in the reference's order: view a layers 1..L, then view b
        na = [self._noise() for _ in range(L)] if self.noise_fn is not None else [None] * L
        nb = [self._noise() for _ in range(L)] if self.noise_fn is not None else [None] * L
        rider = {}
        if getattr(self, "_rider", None) is not None:
            rider, self._rider = {"fetch": self._rider}, None
        ops.spmm(adj, self.E0, out=self._loc(self.Y[0]), **rider, epilogue=ops.make_epilogue(
            perturb_eps=self.eps, noise=None, rng_seed=self.rng_seed, rng_offset=0,
            rng_step=self.cursor[1:2] if self.noise_fn is None else None, rng_stride=self.P * self._rng_calls, main_clean=True,
            **self._slice_kw(), **(dict(row_scale=self.dinv, scale_out=True) if self.vfree and adj is self.adj else {}),
            extra_out=[self._loc(a["Y"][0]), self._loc(b["Y"][0])], extra_noise=[na[0], nb[0]],
            extra_rng_offset=[self._rng_offset(0), self._rng_offset(L)]))
        for t in (self.Y[0], a["Y"][0], b["Y"][0]):
            self._allgather(t)
        self._forward_pass(adj, self.Y, self.F, perturbed=False, include_ego=False, batch_rows_only=True, start_layer=1)
        self._forward_pass(adj, a["Y"], a["F"], perturbed=True, include_ego=False, batch_rows_only=True,
                           start_layer=1, noises=na, call_base=0)
        self._forward_pass(adj, b["Y"], b["F"], perturbed=True, include_ego=False, batch_rows_only=True,
                           start_layer=1, noises=nb, call_base=L)

    def _backward_chain(self, adj, gF, *, include_ego, gCL=None, layer_cl=None, extra=None, accumulate=False, final=False):
        """gE0 (+)= d loss / d E0 through one encoder pass (accumulate=False overwrites gE0).  final: the step's last chain
        -- with fuse_adam its last product ends in the optimiser step (and gE0 is not written).

        H_L = s gF + [l*==L] gCL ;  H_k = A H_{k+1} + s gF + [l*==k] gCL ;
        gE0 += A H_1 + [ego] s gF + [l*==0] gCL (+ extra)       with s = 1/#averaged layers.
        """
        L = self.L
        s = 1.0 / (L + 1 if include_ego else L)
        cl_at = layer_cl if gCL is not None else None
        vf = self.vfree and adj is self.adj
        # (value-free: the contrast view at layer 1 .. L-1 was the pre-scaled table, so gCL is the gradient w.r.t.
        # D^-1/2 Y and enters the chain times d_i^-1/2; at layer 0 / L the view held true values)
        cl_rowscale = vf and cl_at is not None and 0 < cl_at < L
        if cl_at == L:
            H = self.Ha
            ops.axpby(s, gF, 0.0, H)
            ops.axpby(1.0, gCL, 1.0, H)
            src, alpha = H, 1.0
        else:
            src, alpha = gF, s                     # A (s gF) = s (A gF): no materialised H_L
        # the incoming gradient is non-zero only on this step's batch rows: the first product
        # skips every other column
        sparse_src = dict(col_mark=self.mark, mark_stamp=self.cursor[1:2]) if self.use_marks else {}
        loc = self._loc
        # gF, gCL and the regulariser gradient are zero outside the batch rows: read them only there
        batch_sparse = {id(gF), id(gCL), id(extra)} - {id(None)} if self.use_marks else set()

        def sparse_add(add):
            if not batch_sparse:
                return {}
            return dict(add_mark=loc(self.mark), mark_stamp=self.cursor[1:2], add_sparse=[id(a) in batch_sparse for a in add])
        bufs = [self.Hb, self.Ha] if src is self.Ha else [self.Ha, self.Hb]
        first = True                               # the chain's first product reads true values (gF / H_L)

        def plan_for(kw):
            """the matrix under the plan its launch flavour wants: column-masked launches run on the class-free plan"""
            if self.adj_cm is None or "col_mark" not in kw:
                return adj
            if adj is self.adj:
                return self.adj_cm
            return ops.DeviceCSR(None, None, adj.vals, adj.shape, structure_of=self.adj_cm)    # (SGL view: same arrays)

        def scale_kw(add, last):
            """value-free bookkeeping of one product: pattern + row scale on input unless it is the chain's first,
            pre-scaled store unless it is the chain's last."""
            if not vf:
                return {}, {}
            kw = dict(row_scale=self.dinv, scale_in=not first, scale_out=not last,
                      add_rowscale=[cl_rowscale and a is gCL for a in add])
            return kw, ({"pattern": True} if not first else {})
        for k in range(L - 1, 0, -1):              # produce H_k
            add, sc = [gF], [s]
            if cl_at == k:
                add.append(gCL)
                sc.append(1.0)
            dst = bufs[0]
            skw, pkw = scale_kw(add, last=False)
            ops.spmm(plan_for(sparse_src), src, out=loc(dst),
                     epilogue=ops.make_epilogue(add=[loc(a) for a in add], add_scale=sc, alpha=alpha,
                                                **{**sparse_add(add), **sparse_src, **skw}), **pkw)
            self._allgather(dst)
            src, alpha, sparse_src, first = dst, 1.0, {}, False
            bufs.reverse()
        add, sc = ([self.gE0], [1.0]) if accumulate else ([], [])   # (aliasing y is allowed)
        if include_ego:
            add.append(gF)
            sc.append(s)
        if cl_at == 0:
            add.append(gCL)
            sc.append(1.0)
        if extra is not None:
            add.append(extra)
            sc.append(1.0)
        while len(add) > 2:                        # epilogue takes two addends: fold the rest first
            if not accumulate:
                raise SelfrecHipError("internal: more than two addends without an accumulator")
            ops.axpby(sc.pop(), add.pop(), 1.0, self.gE0)
        skw, pkw = scale_kw(add, last=True)
        akw = {}
        if final and self.fuse_adam:
            akw = dict(adam=dict(param=self.E0, m=self.m, v=self.v, coef=self.adam_coef, clear=self._sparse_tables(),
                                 clear_mark=self.mark, cursor=self.cursor),
                       mark_stamp=self.now[1:2])          # (the launch moves the cursor: marks compare with the copy)
        ops.spmm(plan_for(sparse_src), src, out=loc(self.gE0),
                 epilogue=ops.make_epilogue(add=[loc(a) for a in add], add_scale=sc, alpha=alpha,
                                            **{**sparse_add(add), **sparse_src, **skw, **akw}), **pkw)

    def _sparse_tables(self):
        """the gradient buffers that hold non-zeros on this step's batch rows only"""
        clear = [self.gF] + [t for t in (self.gCL, self.gReg) if t is not None]
        if self.model == "SGL":
            clear += [v["gF"] for v in self.views]
        return clear

    # ------------------------------------------------------------------------------------
    # one training step on the staged batch
    # ------------------------------------------------------------------------------------
    # A step = before | collective | after.  The collective is the placement's: the batch-row exchange of column blocks
    # (then the losses come after it), the gradient all-reduce of data parallel (then the losses come before it), or
    # nothing.  Layouts with a collective run the three as separate phases (step_phases) and capture the two halves.
    def _step_before(self):
        self._step_front()
        if self.sync.active:
            self._step_grad()

    def _step_collective(self):
        self.colx.exchange()
        self.sync.reduce(self)

    def _step_after(self):
        if not self.sync.active:
            self._step_grad()
        self._step_opt()

    def _step_kernels(self):
        self._step_before()
        self._step_collective()
        self._step_after()

    def _sgl_shared_first(self):
        return self.model == "SGL" and self.L >= 2 and self.w == 64

    def _step_front(self):
        """batch_fetch + the encoder passes (+ packing the batch rows when the tables are column slices)."""
        m, st = self.model, self.stage
        adj = self.adj
        # the staged ids are table rows (items already offset / permuted): one table, one index space
        cat = dict(stage_cat=self.stage_cat, n_cat=self.n_cat) if m == "SGL" else {}
        fetch = ((self._epoch_dev, self.sampler.n_edges, self.B, self.cursor, st, self.meta),
                 dict(row_mark=self.mark, mark_item_offset=0, zero4=self.losses, now=self.now,
                      half_batches=self.epoch_batches, **cat,
                      **(dict(adam_coef=self.adam_coef, adam_lr=self.lr) if self.fuse_adam else {})))
        # The first product of the step does not depend on the batch (marks and staged ids enter at the last forward
        # layer and at the losses), so the fetch rides on its launch as eight extra workgroups instead of being a 5 us
        # launch of its own at the head of the step (srh_spmm_f32_with_fetch).
        self._rider = None
        if self.ride_fetch:
            self._rider = ops.batch_fetch_args(*fetch[0], **fetch[1])
        else:
            ops.batch_fetch(*fetch[0], **fetch[1])
        self._noise_call = 0      # RNG counter = (adam step, perturbed-layer call no, row)

        include_ego = m in ("LightGCN", "SGL")
        sgl_shared_first = self._sgl_shared_first()
        if m == "SimGCL":
            self._simgcl_forward(adj)
        elif sgl_shared_first:
            # SGL.py:104-108 on the full graph and on both dropped views: the first layer of all three
            # multiplies the same ego table -- one traversal of the shared structure, three value arrays
            a, b = self.views
            firsts = [self.Y[0], a["Y"][0], b["Y"][0]]
            ops.spmm3([adj, self.view_adj[0], self.view_adj[1]], self.E0, [self._loc(t) for t in firsts])
            for t in firsts:
                self._allgather(t)
            self._forward_pass(adj, self.Y, self.F, perturbed=False, include_ego=True, batch_rows_only=True, start_layer=1)
            for vi, v in enumerate(self.views):
                self._forward_pass(self.view_adj[vi], v["Y"], v["F"], perturbed=False, include_ego=True,
                                   batch_rows_only=True, start_layer=1)
        elif m != "MF":
            self._forward_pass(adj, self.Y, self.F, perturbed=(m == "XSimGCL"), include_ego=include_ego,
                               batch_rows_only=True, need_last=(m == "XSimGCL" and self.layer_cl == self.L))
        if m == "SGL" and not sgl_shared_first:
            for vi, v in enumerate(self.views):
                self._forward_pass(self.view_adj[vi], v["Y"], v["F"], perturbed=False, include_ego=include_ego,
                                   batch_rows_only=True)
        if self._rider is not None:
            raise SelfrecHipError("internal: the batch fetch found no product to ride on")
        self.colx.pack(self)                      # (column blocks: this rank's slices of the batch rows; else nothing)

    def _step_back(self):
        """losses on the batch rows, backward through the encoder, (gradient sync,) optimiser, row-wise resets."""
        self._step_grad()
        self.sync.reduce(self)
        self._step_opt()

    def _step_grad(self):
        """losses on the batch rows and the backward chain: leaves d loss / d E0 in gE0."""
        m, st = self.model, self.stage
        adj = self.adj
        rows_dev, nuu_dev, nui_dev = self.meta[0:1], self.meta[1:2], self.meta[2:3]
        F = self.F
        # the tables the losses read and the index lists into them: the rank's own (whole rows here), or -- column blocks --
        # the compact (5B, d) tables the batch-row exchange just filled, with slot -> slot lists
        T, GT, ix, cat_idx = self.colx.loss_views(self)
        # ---- recommendation loss + regulariser (a-5..a-7)
        if m == "LightGCN":
            # regulariser on the EGO rows (LightGCN.py:25); its gradient joins gE0 in the last product
            reg_t, greg_t = self.E0, self.gReg
            reg_coef, inc_neg = self.reg / self.B, True
        elif m == "MF":
            reg_t, greg_t = F, self.gF
            reg_coef, inc_neg = self.reg / self.B, True                     # MF.py:21
        else:
            reg_t, greg_t = F, self.gF
            reg_coef, inc_neg = self.reg, (m == "SGL")                      # XSimGCL.py:33, SGL.py:36
        # (forking BPR onto a second stream beside InfoNCE was measured: 0.367 vs 0.353 ms/step -- the
        # extra graph edges cost more than the overlap buys; instead the two losses' O(batch) kernels
        # share launches inside srh_bpr_infonce_fwd_bwd)
        bpr = dict(batch=self.B, n_rows_dev=rows_dev, reg_coef=reg_coef, reg_include_neg=inc_neg, loss_scale=1.0,
                   g_user=GT(self.gF), g_item=GT(self.gF), greg_user=GT(greg_t), greg_item=GT(greg_t),
                   losses=self.losses[0:2])
        bpr_in = (T(F), T(F), T(reg_t), T(reg_t), ix["u"], ix["i"], ix["j"])
        seg = {}
        if self.det_scatter and not getattr(self, "_seg_off", False):     # (_seg_off: tools/det_scatter_ab.py's A/B switch)
            ed = self._epoch_dev
            # (rows_are_zero: the batch rows of every batch-sparse gradient table were cleared by the previous step's last launch)
            seg = dict(seg=dict(n_uniq_u=nuu_dev, n_uniq_i=nui_dev, n_uniq_n=ed["n_uniq_n"], seg_rows=ed["seg_rows"],
                                seg_end=ed["seg_end"], seg=ed["seg"], seg_a=ed["seg_a"], seg_b=ed["seg_b"],
                                batch_no=self.meta[3:4], rows_are_zero=self.sparse_reset))
        nce = dict(tau=self.tau, cl_scale=self.cl_rate, cl_loss=self.losses[2:3], nce_ws=self.nce_ws,
                   precision=self.nce_precision)
        # ---- recommendation + contrastive loss (a-5..a-8)
        if m == "XSimGCL":
            CL = self.E0 if self.layer_cl == 0 else self.Y[self.layer_cl - 1]
            ops.bpr_infonce(*bpr_in, **bpr, bpr_ws=self.bpr_ws, **nce, **seg, **({"nce_rows": 1} if seg else {}), problems=[
                # (gCL's rows: the user-side problem names user rows, the item-side one item rows, nobody else writes them)
                (T(F), T(CL), ix["uniq_u"], self.B, nuu_dev, GT(self.gF), GT(self.gCL), True),
                (T(F), T(CL), ix["uniq_i"], self.B, nui_dev, GT(self.gF), GT(self.gCL), True)])
        elif m in ("SimGCL", "SGL"):
            a, b = self.views
            if m == "SimGCL":
                # the three passes share one backward chain (same linear operator), so the views'
                # gradients go straight into gF
                problems = [(T(a["F"]), T(b["F"]), ix["uniq_u"], self.B, nuu_dev, GT(self.gF), GT(self.gF)),
                            (T(a["F"]), T(b["F"]), ix["uniq_i"], self.B, nui_dev, GT(self.gF), GT(self.gF))]
            else:
                problems = [(T(a["F"]), T(b["F"]), cat_idx, 2 * self.B, self.n_cat, GT(a["gF"]), GT(b["gF"]))]
            ops.bpr_infonce(*bpr_in, **bpr, bpr_ws=self.bpr_ws, **nce, problems=problems, **seg,
                            **({"nce_rows": 1 if m == "SimGCL" else 2} if seg else {}))
        else:
            ops.bpr_l2_fwd_bwd(*bpr_in, **bpr, ws=self.bpr_ws, **seg)
        self.colx.scatter(self)                   # (column blocks: this rank's columns of the batch-row gradients go home)
        # ---- backward through the encoder (a-4) and optimiser (a-9)
        if m == "MF":
            pass                                     # gF is gE0
        elif m == "XSimGCL":
            self._backward_chain(adj, self.gF, include_ego=False, gCL=self.gCL, layer_cl=self.layer_cl, final=True)
        elif m == "LightGCN":
            self._backward_chain(adj, self.gF, include_ego=True, extra=self.gReg, final=True)
        elif m == "SimGCL":
            self._backward_chain(adj, self.gF, include_ego=False, final=True)
        else:                                        # SGL: three operators, three chains
            self._backward_chain(adj, self.gF, include_ego=True)
            for vi, v in enumerate(self.views):
                self._backward_chain(self.view_adj[vi], v["gF"], include_ego=True, accumulate=True,
                                     final=vi == len(self.views) - 1)

    def _step_opt(self):
        """Adam on the (owned rows of the) table, row-wise resets of the batch-sparse gradient buffers, cursor advance."""
        m, st = self.model, self.stage
        rows_dev, nuu_dev, nui_dev = self.meta[0:1], self.meta[1:2], self.meta[2:3]
        if self.fuse_adam:
            return                      # done by the last backward product (_backward_chain, final=True)
        if self.fused_reset and self.sparse_reset:
            # Adam's pass over the table also clears this batch's rows (the marked ones) of the batch-sparse gradient
            # buffers and advances the cursor: the separate zero_rows launch (4.5 us) is gone.  Adam reads its step from
            # `now` (batch_fetch's copy), so the advance cannot race with it.
            ops.adam_step(self.E0, self.gE0, self.m, self.v, step_dev=self.now[1:2], lr=self.lr, clear=self._sparse_tables(),
                          row_mark=self.mark, advance_cursor=self.cursor)
            if self.sync.active and self.gF is self.gE0:
                self.gE0.zero_()        # MF: gE0 IS the accumulation buffer, and after the all-reduce it holds the OTHER
            return                      # ranks' batch rows too, which this rank's marks do not name
        ops.adam_step(self._loc(self.E0), self._loc(self.gE0), self.m, self.v, step_dev=self.now[1:2], lr=self.lr)
        if self.sync.active and self.gF is self.gE0:
            self.gE0.zero_()
        self._allgather(self.E0)                     # every rank's next forward pass reads the whole table
        if self.sparse_reset:
            # the gradient buffers hold non-zeros only on this batch's rows: clear just those
            B = self.B
            lists = [(self.gF, st["u"], rows_dev, B, 0), (self.gF, st["i"], rows_dev, B, 0),
                     (self.gF, st["j"], rows_dev, B, 0)]
            if self.gCL is not None:
                lists += [(self.gCL, st["uniq_u"], nuu_dev, B, 0), (self.gCL, st["uniq_i"], nui_dev, B, 0)]
            if self.gReg is not None:
                lists += [(self.gReg, st["u"], rows_dev, B, 0), (self.gReg, st["i"], rows_dev, B, 0),
                          (self.gReg, st["j"], rows_dev, B, 0)]
            if m == "SGL":                       # the views' gradients live on the contrast rows
                lists += [(v["gF"], self.stage_cat, self.n_cat, 2 * B, 0) for v in self.views]
            ops.zero_rows(lists, self.w, cursor_advance=self.cursor)      # last kernel of the step
        else:
            ops.cursor_advance(self.cursor)

    def step(self):
        """Run one training step on the next batch of the current epoch."""
        if not self._epoch_ready:
            raise SelfrecHipError("call begin_epoch() first")
        for phase in self.step_phases():
            phase()

    def step_phases(self):
        """The step as a sequence of calls: (whole step, done), or -- placements with a collective in the step --
        (before, the collective, after, done), so that a test can drive several ranks in lock-step on one GPU and the
        collective stays outside the two captured graphs."""
        if not self._epoch_ready:
            raise SelfrecHipError("call begin_epoch() first")
        graphed = self.use_graph and self.noise_fn is None
        if graphed and self._graph is None:
            try:
                self._capture()
            except RuntimeError as e:
                if self.G == 1 and not self.place.collective_in_step and not self.rows.dealt:
                    raise
                # (a capture next to a live process group is the one thing that could not be exercised beyond one
                # rank here: fall back to eager launches rather than lose the run -- the state is the snapshot's)
                import sys
                print(f"[selfrec_amd] hipGraph capture failed on rank {self.rank} ({e}); launching eagerly", file=sys.stderr)
                self.use_graph, self._graph, graphed = False, None, False

        def done():
            self.step_count += 1
        if self.place.collective_in_step:
            return (self._graph[0].replay if graphed else self._step_before, self._step_collective,
                    self._graph[1].replay if graphed else self._step_after, done)
        return (self._graph.replay if graphed else self._step_kernels, done)

    def reset_graph(self):
        """Forget the captured hipGraph (it is re-captured on the next step): needed after anything that changes which
        kernels a step launches."""
        self._graph = None

    def set_nce_precision(self, mode):
        """'split' | 'f32' | None (the process default) for this trainer's InfoNCE products from the next step on; a captured
        step is re-captured (it keeps the kernels it was captured with)."""
        if mode is not None and mode not in ops.NCE_PRECISIONS:
            raise SelfrecHipError(f"nce_precision {mode!r}: one of {sorted(ops.NCE_PRECISIONS)} or None")
        if mode != self.nce_precision:
            self.nce_precision = mode
            self.reset_graph()

    def _capture(self):
        """Capture the step as hipGraph(s).  No destructor with a HIP call in it may run while a stream of this thread is
        capturing: an older trainer's device plans (hipFree), CUDAGraphs, streams, collected by python's cyclic GC in that
        window, left a graph whose first replayed step was wrong (half of the world-2 data-parallel runs of
        tests/test_gpu_multiproc.py; never with the collection below -- tools/dp_graph_probe.py, DESIGN.md 6.4;
        torch >= 2.9 no longer collects on entering torch.cuda.graph).  So: collect first, keep the collector off inside."""
        import gc
        guard = not os.environ.get("SRH_NO_CAPTURE_GC")          # (diagnostic knob: reproduces the failure)
        was_enabled = gc.isenabled()
        if guard:
            gc.collect()
            gc.disable()
        try:
            with self._hip_lock:                                 # (the prefetch thread's staging stays out of the window)
                self._capture_guarded()
        finally:
            if guard and was_enabled:
                gc.enable()

    def _capture_guarded(self):
        # warm up once eagerly on a side stream (allocator + lazy module loads), then capture.  (The warm-up skips the
        # placement's collective -- its numbers are thrown away with the snapshot -- so capturing is a purely local act.)
        two = self.place.collective_in_step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        snapshot = (self.E0.clone(), self.m.clone(), self.v.clone(), self.cursor.clone())

        def restore():
            self.E0.copy_(snapshot[0]); self.m.copy_(snapshot[1]); self.v.copy_(snapshot[2]); self.cursor.copy_(snapshot[3])
        with torch.cuda.stream(side):
            self._step_before()
            self._step_after()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        restore()
        if two:
            # two graphs with the collective between them: RCCL stays outside the captured region
            self._graph = (torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph())
            pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(self._graph[0], pool=pool, capture_error_mode="thread_local"):
                self._step_before()
            with torch.cuda.graph(self._graph[1], pool=pool, capture_error_mode="thread_local"):
                self._step_after()
        else:
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._step_kernels()
        restore()
        if two and self.G > 1 and all(isinstance(c, (TorchComm, TwoHopRows)) for c in self.place.comms()) \
                and os.environ.get("SRH_CAPTURE_CHECK", "1") != "0":
            self._check_replay_against_eager(restore, snapshot[0])

    def _check_replay_against_eager(self, restore, e0_before):
        """Two graphs around a live collective have never met more than one real GPU (the one failure seen so far,
        a collector-run destructor inside the capture window, left a graph whose FIRST replayed step was wrong): before the
        graphs are trusted, one step is run eagerly and once more as a replay from the same state -- collective included,
        same batch, same noise counters -- and the parameters must agree on every rank (the verdict is all-reduced over
        the whole job, so the ranks fall back together).  A mismatch raises; step_phases() then launches eagerly and says so."""
        results = []
        for phases in ((self._step_before, self._step_collective, self._step_after),
                       (self._graph[0].replay, self._step_collective, self._graph[1].replay)):
            for ph in phases:
                ph()
            torch.cuda.synchronize()
            results.append(self.E0.clone())
            restore()
        moved = (results[0] - e0_before).abs().max().item()
        diff = (results[0] - results[1]).abs().max().item()
        ok = torch.tensor([1.0 if (diff <= 0.1 * self.lr and moved > 0.0) else 0.0], device=self.dev)
        _dist.all_reduce(ok, op=_dist.ReduceOp.MIN)
        if ok.item() < 1.0:
            self._graph = None
            raise RuntimeError(f"captured step differs from the eager step on some rank (this rank {self.rank}: max |E0 eager - "
                               f"E0 replay| = {diff:.3e}, one step moves E0 by {moved:.3e}, lr = {self.lr:g})")

    def read_losses(self):
        """(bpr, reg, cl) of the last step -- a device-to-host sync, call sparingly."""
        bpr, reg, cl = self.losses[:3].tolist()
        return bpr, reg, cl

    # ------------------------------------------------------------------------------------
    # inference-time embeddings (model() with no perturbation, XSimGCL.py:40-41)
    # ------------------------------------------------------------------------------------
    @torch.no_grad()
    def embeddings(self):
        if self.model == "MF":
            return self.user_emb, self.item_emb
        out = torch.zeros_like(self.E0)
        Ys = [torch.zeros_like(self.E0) for _ in range(self.L)]
        self._forward_pass(self.adj, Ys, out, perturbed=False, include_ego=self.model in ("LightGCN", "SGL"))
        out = self._valid(self.colx.full(out))    # (column blocks: a collective -- every rank must ask)
        return self.rows.rows_of_users(out), self.rows.rows_of_items(out)


class EpochPrefetcher:
    """Runs ``trainer.sample_epoch_host()`` for epoch e+1 on a host thread while the device
    works on epoch e.  The sampler is sequential by construction (one MT19937 stream with
    data-dependent rejection), so a single producer is all there is to overlap."""

    def __init__(self, trainer: FusedTrainer, first: bool = False):
        self.trainer = trainer
        self._thread = None
        self._result = None
        self._error = None
        self._first = first           # the trainer's own early start: draws directly (sample_epoch_host hands THIS one out)
        # Epochs are drawn into two sets of host arrays in turn (ops.Sampler.epoch(slot=...)): by the time a set is drawn into
        # again, the epoch it held has been replaced as the trainer's live one.  Fresh arrays per epoch are 25 MB unmapped and
        # 25 MB page-faulted in at every epoch boundary, beside the thread that enqueues the steps -- and unmapping memory there
        # stalls the GPU's supply of work (126 MB freed between two replays: + 2.2 ms, tools/host_cost_probe.py).
        self._slots = bool(getattr(trainer, "reuse_epoch_arrays", False)) and not first
        self._turn = 0

    def _work(self):
        try:
            if self._first:
                self._result = self.trainer._sample_epoch_host_now()
            elif self._slots:
                self._result = self.trainer.sample_epoch_host(slot=self._turn)
                self._turn ^= 1
            else:
                self._result = self.trainer.sample_epoch_host()
            stage = getattr(self.trainer, "stage_epoch", None)
            if stage is not None and not self._first:      # (the early first epoch is drawn while the trainer is still being built)
                self._result = stage(self._result)
        except BaseException as e:  # surfaced on the consumer side
            self._error = e

    def start(self):
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def take(self):
        if self._thread is None:
            self.start()
        self._thread.join()
        self._thread = None
        if self._error is not None:
            raise self._error
        res, self._result = self._result, None
        return res
