"""The column-sharded multi-GPU layout on ONE GPU: the thin SpMM kernel against the wide kernel's column
slices and scipy, the batch-row exchange kernels against torch indexing, and G "virtual ranks" -- G
trainers with layout "cols" driven in lock-step, their all-gather done by the test -- against the
unsharded trainer and the reference's own 3-step runs (tests/golden/models.npz)."""
import random

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from selfrec_amd import ops
from selfrec_amd.dist import ShardedTrainer
from selfrec_amd.engine import FusedTrainer

pytestmark = pytest.mark.gpu
WIDTHS = (8, 16, 32)


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


def power_law_csr(n=3000, seed=0):
    """Square CSR with empty rows, short rows, rows of 65..512 non-zeros (coop) and split rows (> 512)."""
    rng = np.random.default_rng(seed)
    lens = np.minimum((rng.pareto(1.1, n) * 6).astype(np.int64), n - 1)
    lens[:5] = [0, 1, 700, 1500, 64]
    lens[n // 2] = 0
    rows = np.repeat(np.arange(n), lens)
    cols = np.concatenate([rng.choice(n, size=k, replace=False) for k in lens]) if lens.sum() else np.zeros(0, np.int64)
    vals = rng.standard_normal(rows.size).astype(np.float32)
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    m.sort_indices()
    return m


@pytest.fixture(scope="module")
def csr_case():
    m = power_law_csr()
    csr = ops.DeviceCSR.from_scipy(m)
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((m.shape[0], 64), device="cuda", generator=gen)
    return m, csr, x


def slices(t, w):
    return [t[:, c:c + w].contiguous() for c in range(0, t.shape[1], w)]


@pytest.mark.parametrize("w", WIDTHS)
def test_thin_spmm_plain_matches_scipy_and_wide_kernel(csr_case, w):
    m, csr, x = csr_case
    want64 = m.astype(np.float64) @ x.cpu().numpy().astype(np.float64)
    wide = ops.spmm(csr, x)
    for r, xs in enumerate(slices(x, w)):
        ys = ops.spmm(csr, xs, epilogue=ops.make_epilogue(d_full=64, col0=r * w))       # (d = 32 too goes thin)
        assert rel_err(ys.cpu().numpy(), want64[:, r * w:(r + 1) * w]) < 2e-6
        assert rel_err(ys.cpu().numpy(), wide[:, r * w:(r + 1) * w].cpu().numpy()) < 2e-6
    # bitwise repeatable (fixed-order reductions, slot-ordered split rows)
    xs = slices(x, w)[0]
    a = ops.spmm(csr, xs, epilogue=ops.make_epilogue(d_full=64, col0=0))
    for _ in range(5):
        assert torch.equal(a, ops.spmm(csr, xs, epilogue=ops.make_epilogue(d_full=64, col0=0)))


@pytest.mark.parametrize("w", WIDTHS)
def test_thin_spmm_epilogues_match_wide_kernel_slices(csr_case, w):
    """PERTURB (injected noise and counter RNG: the unit vector spans the WHOLE row), MEAN, AXPY with sparse
    addends, row / column marks, FANOUT -- every slice equals the same columns of the d = 64 launch."""
    m, csr, x = csr_case
    n = m.shape[0]
    gen = torch.Generator(device="cuda").manual_seed(2)
    noise = torch.rand((n, 64), device="cuda", generator=gen)
    noise2 = torch.rand((n, 64), device="cuda", generator=gen)
    prev = [torch.randn((n, 64), device="cuda", generator=gen) for _ in range(2)]
    add = [torch.randn((n, 64), device="cuda", generator=gen) for _ in range(2)]
    stamp = torch.tensor([7], dtype=torch.int64, device="cuda")
    mark = torch.where(torch.rand(n, device="cuda", generator=gen) < 0.3, 7, 3).to(torch.int32)
    add[1][mark != 7] = 0.0                      # a batch-sparse addend is zero off the marked rows
    xm = x * (mark == 7).unsqueeze(1)             # col_mark contract: x is zero on dead columns
    step = torch.tensor([5], dtype=torch.int64, device="cuda")

    def run(xin, width, col0, sl):
        """All epilogue flavours on tables of `width` columns starting at col0; sl() slices a 64-wide tensor."""
        kw = {} if width == 64 else dict(d_full=64, col0=col0)
        out = {}
        out["noise"] = ops.spmm(csr, xin, epilogue=ops.make_epilogue(perturb_eps=0.2, noise=noise, **kw))
        out["rng"] = ops.spmm(csr, xin, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=11, rng_offset=3 * n,
                                                                 rng_step=step, rng_stride=16 * n, **kw))
        mean_out = torch.zeros((n, width), device="cuda")
        out["mean_y"] = ops.spmm(csr, xin, epilogue=ops.make_epilogue(perturb_eps=0.1, noise=noise, prev=[sl(p) for p in prev],
                                                                    mean_div=3.0, mean_out=mean_out, **kw))
        out["mean"] = mean_out
        out["axpy"] = ops.spmm(csr, sl(xm) if width != 64 else xm, epilogue=ops.make_epilogue(
            add=[sl(a) for a in add], add_scale=[0.5, 2.0], alpha=0.25, add_mark=mark, add_sparse=[False, True],
            col_mark=mark, mark_stamp=stamp, **kw))
        rm = torch.full((n, width), -5.0, device="cuda")
        ops.spmm(csr, xin, out=rm, epilogue=ops.make_epilogue(row_mark=mark, mark_stamp=stamp, **kw))
        out["rowmark"] = rm
        e1, e2 = torch.zeros((n, width), device="cuda"), torch.zeros((n, width), device="cuda")
        out["fan_main"] = ops.spmm(csr, xin, epilogue=ops.make_epilogue(
            perturb_eps=0.2, main_clean=True, extra_out=[e1, e2], extra_noise=[noise, noise2], **kw))
        out["fan1"], out["fan2"] = e1, e2
        return out

    wide = run(x, 64, 0, lambda t: t)
    for r in range(64 // w):
        c0 = r * w
        sl = lambda t, c0=c0: t[:, c0:c0 + w].contiguous()      # noqa: E731
        thin = run(sl(x), w, c0, sl)
        for k, v in thin.items():
            assert rel_err(v.cpu().numpy(), wide[k][:, c0:c0 + w].cpu().numpy()) < 3e-6, (k, w, r)
    assert (wide["rowmark"][mark != 7] == -5.0).all()


def test_exchange_kernels_match_torch_indexing():
    B, n, G, w = 96, 500, 4, 16
    d = G * w
    gen = torch.Generator().manual_seed(3)
    stage = {k: torch.randint(0, n, (B,), generator=gen, dtype=torch.int32).cuda() for k in ("u", "i", "j", "uniq_u", "uniq_i")}
    meta = torch.tensor([70, 33, 96, 0], dtype=torch.int32).cuda()
    counts = {"u": 70, "i": 70, "j": 70, "uniq_u": 33, "uniq_i": 96}
    lists = ops.batch_lists(stage, meta, B)
    full = [torch.randn((n, d), generator=gen).cuda() for _ in range(2)]
    sends = []
    cat = torch.full((2 * B,), -1, dtype=torch.int32).cuda()
    for r in range(G):
        send = torch.full((2, 5 * B, w), 9.0).cuda()
        ops.batch_pack(lists, [t[:, r * w:(r + 1) * w].contiguous() for t in full], send, cat_idx=cat)
        sends.append(send)
    recv = torch.stack(sends)
    compact = [torch.full((5 * B, d), 7.0).cuda() for _ in range(2)]
    cg = [torch.full((5 * B, d), 7.0).cuda() for _ in range(2)]
    ops.batch_unpack(lists, recv, G, w, compact, cg)
    assert cat[:33].tolist() == list(range(3 * B, 3 * B + 33)) and cat[33:33 + 96].tolist() == list(range(4 * B, 5 * B))
    for s, k in enumerate(("u", "i", "j", "uniq_u", "uniq_i")):
        live = slice(s * B, s * B + counts[k])
        dead = slice(s * B + counts[k], (s + 1) * B)
        for t in range(2):
            assert torch.equal(compact[t][live], full[t][stage[k][:counts[k]].long()])
            assert (compact[t][dead] == 7.0).all() and (cg[t][live] == 0.0).all() and (cg[t][dead] == 7.0).all()
    # scatter: column slice of every live compact row, added to the node's row (duplicates sum)
    grads = [torch.randn((5 * B, d), generator=gen).cuda() for _ in range(2)]
    grads[1][: 3 * B] = 0.0
    for r in range(G):
        local = [torch.zeros((n, w)).cuda() for _ in range(2)]
        ops.batch_scatter(lists, list(zip(grads, local)), d, r * w, w)
        for t in range(2):
            want = torch.zeros((n, w), dtype=torch.float64)
            for s, k in enumerate(("u", "i", "j", "uniq_u", "uniq_i")):
                rows = grads[t][s * B:s * B + counts[k], r * w:(r + 1) * w].double().cpu()
                want.index_add_(0, stage[k][:counts[k]].long().cpu(), rows)
            assert rel_err(local[t].cpu().numpy(), want.numpy()) < 1e-6


# ---------------------------------------------------------------------------------------------------
# virtual ranks: G column-sharded trainers on one GPU, their collective done by the driver
# ---------------------------------------------------------------------------------------------------
class LockstepGroup:
    def __init__(self, world):
        self.world, self.calls = world, []

    def comm(self, rank):
        group = self

        class Comm:
            world, rank_ = group.world, rank

            def all_gather(self, out, inp):
                group.calls.append((rank, out, inp))
        c = Comm()
        c.rank = rank
        return c

    def flush(self):
        assert sorted(r for r, _, _ in self.calls) == list(range(self.world))
        ins = [inp for _, _, inp in sorted(self.calls, key=lambda t: t[0])]
        for _, out, _ in self.calls:
            for r, inp in enumerate(ins):
                out.view(self.world, -1)[r].copy_(inp.reshape(-1))
        self.calls = []


def lockstep_step(group, trainers):
    phases = [tr.step_phases() for tr in trainers]
    for k in range(len(phases[0])):
        for ph in phases:
            ph[k]()
        if group.calls:
            group.flush()


def gathered(trainers, name):
    return torch.cat([getattr(tr, name) for tr in trainers], dim=1)


def make_kw(name, gm, meta, noise=True):
    m = meta[name]; c = m["conf"]
    kw = dict(model=name, n_layers=int(c.get("n_layer", 0)), lr=m["lr"], reg=m["reg"],
              cl_rate=float(c.get("lambda", 0.0)), eps=float(c.get("eps", 0.0)),
              tau=float(c.get("tau", c.get("temp", 0.2))), layer_cl=int(c.get("l_star", 1)),
              drop_rate=float(c.get("drop_rate", 0.1)), batch_size=m["batch"],
              user_emb=gm[f"{name}_init_user"], item_emb=gm[f"{name}_init_item"])
    if noise:
        kw["noise_seed"] = m["noise_seed"]
    return kw


@pytest.mark.parametrize("name,world", [("MF", 2), ("LightGCN", 4), ("XSimGCL", 2), ("XSimGCL", 4), ("XSimGCL", 8),
                                        ("SimGCL", 4), ("SGL", 8)])
def test_virtual_ranks_match_reference_run(golden_models, golden_meta, tiny_data, name, world):
    """Column-sharded over `world` virtual ranks == the reference's own 3 training steps (noise injected)."""
    gm, meta = golden_models, golden_meta
    d = meta[name]["emb"]
    if d % world or d // world not in WIDTHS:
        pytest.skip(f"d / world = {d}/{world} is not a thin width")
    group = LockstepGroup(world)
    trainers = []
    for r in range(world):
        kw = make_kw(name, gm, meta)
        gen = torch.Generator().manual_seed(kw.pop("noise_seed"))      # every rank draws the same noise stream
        trainers.append(ShardedTrainer(tiny_data, d, layout="cols", comm=group.comm(r),
                                       noise_fn=lambda shape, gen=gen: torch.rand(shape, generator=gen), **kw))
    for tr in trainers:
        assert tr.cols and tr.E0.shape[1] == d // world
        random.seed(meta[name]["sampler_seed"])
        tr.seed_sampler_from_python()
        nb = tr.begin_epoch()
    bpr, cl = [], []
    for _ in range(nb):
        lockstep_step(group, trainers)
        per_rank = [tr.read_losses() for tr in trainers]
        for other in per_rank[1:]:
            np.testing.assert_allclose(other, per_rank[0], rtol=1e-6)     # replicated loss section
        bpr.append(per_rank[0][0]); cl.append(per_rank[0][2])
    np.testing.assert_allclose(bpr, gm[f"{name}_loss_bpr"], rtol=1e-5)
    if name in ("XSimGCL", "SimGCL"):
        np.testing.assert_allclose(cl, gm[f"{name}_loss_nce"].reshape(nb, 2).sum(1) * trainers[0].cl_rate, rtol=2e-5)
    elif name == "SGL":
        np.testing.assert_allclose(cl, gm[f"{name}_loss_nce"] * trainers[0].cl_rate, rtol=2e-5)
    U = trainers[0].U
    E0 = gathered(trainers, "E0").cpu().numpy()
    assert rel_err(E0[:U], gm[f"{name}_param_user"]) < 1e-4 and rel_err(E0[U:], gm[f"{name}_param_item"]) < 1e-4
    du = E0[:U] - gm[f"{name}_init_user"]
    assert np.abs(du - (gm[f"{name}_param_user"] - gm[f"{name}_init_user"])).max() < 1e-5


@pytest.mark.selfcheck
@pytest.mark.parametrize("use_graph", [False, True])
def test_virtual_ranks_counter_rng_equal_unsharded(golden_models, golden_meta, tiny_data, use_graph, monkeypatch):
    """In-kernel noise: the sharded run regenerates exactly the unsharded run's perturbation, eager and as two
    captured graphs per step with the all-gather between them."""
    gm, meta = golden_models, golden_meta
    name, world = "XSimGCL", 4
    d = meta[name]["emb"]
    if d // world not in WIDTHS:
        pytest.skip("thin width")
    kw = make_kw(name, gm, meta, noise=False)
    monkeypatch.setenv("SRH_SHARDED_GRAPH", "1" if use_graph else "0")
    single = FusedTrainer(tiny_data, d, noise_fn=None, use_graph=False, **kw)
    group = LockstepGroup(world)
    trainers = [ShardedTrainer(tiny_data, d, layout="cols", comm=group.comm(r), noise_fn=None, use_graph=use_graph, **kw)
                for r in range(world)]
    for tr in [single] + trainers:
        tr.sampler.seed(5)
    for _ in range(2):
        nb = single.begin_epoch()
        for tr in trainers:
            assert tr.begin_epoch() == nb
        for _ in range(nb):
            single.step()
            lockstep_step(group, trainers)
    torch.cuda.synchronize()
    assert trainers[0].step_count == single.step_count == 2 * nb and trainers[0].use_graph == use_graph
    want = single.E0.cpu().numpy()
    assert np.isfinite(want).all()
    # (different kernels, different summation orders, 2 epochs of Adam in between: the parity budget, not bitwise)
    assert rel_err(gathered(trainers, "E0").cpu().numpy(), want) < 1e-4
    np.testing.assert_allclose(trainers[0].read_losses(), single.read_losses(), rtol=1e-4)


@pytest.mark.parametrize("name", ["XSimGCL", "SGL"])
def test_cols_layout_through_rccl_single_rank(golden_models, golden_meta, tiny_data, name):
    """The column-sharded step with its batch-row all-gather going through RCCL ("nccl"), world size 1 (the
    one rank keeps whole rows): pack -> all_gather_into_tensor -> unpack -> compact-table losses -> scatter
    reproduce the reference's own 3-step run."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29579")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        gm, meta = golden_models, golden_meta
        kw = make_kw(name, gm, meta)
        gen = torch.Generator().manual_seed(kw.pop("noise_seed"))
        tr = ShardedTrainer(tiny_data, meta[name]["emb"], layout="cols", noise_fn=lambda shape: torch.rand(shape, generator=gen), **kw)
        assert tr.cols and tr.G == 1 and tr.w == tr.d
        random.seed(meta[name]["sampler_seed"])
        tr.seed_sampler_from_python()
        bpr = []
        for _ in range(tr.begin_epoch()):
            tr.step()
            bpr.append(tr.read_losses()[0])
        np.testing.assert_allclose(bpr, gm[f"{name}_loss_bpr"], rtol=1e-5)
        pu, pi = tr.parameters_full()
        assert rel_err(pu.cpu().numpy(), gm[f"{name}_param_user"]) < 1e-4
        assert rel_err(pi.cpu().numpy(), gm[f"{name}_param_item"]) < 1e-4
        fu, _ = tr.embeddings()
        assert rel_err(fu.cpu().numpy(), gm[f"{name}_final_user"]) < 1e-4
    finally:
        if created:
            dist.destroy_process_group()


def test_wide_slices_of_wider_rows(csr_case):
    """64 of 128 columns (d = 128 over 2 ranks) runs the 16-lane row kernel on a slice: PERTURB still normalises over
    the whole 128-wide row -- every slice equals the same columns of the d = 128 launch."""
    m, csr, _ = csr_case
    n = m.shape[0]
    gen = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn((n, 128), device="cuda", generator=gen)
    noise = torch.rand((n, 128), device="cuda", generator=gen)
    step = torch.tensor([2], dtype=torch.int64, device="cuda")
    wide_n = ops.spmm(csr, x, epilogue=ops.make_epilogue(perturb_eps=0.2, noise=noise))
    wide_r = ops.spmm(csr, x, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=5, rng_offset=n, rng_step=step, rng_stride=16 * n))
    for w in (64, 16):
        for c0 in range(0, 128, w):
            xs = x[:, c0:c0 + w].contiguous()
            got_n = ops.spmm(csr, xs, epilogue=ops.make_epilogue(perturb_eps=0.2, noise=noise, d_full=128, col0=c0))
            got_r = ops.spmm(csr, xs, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=5, rng_offset=n, rng_step=step,
                                                                 rng_stride=16 * n, d_full=128, col0=c0))
            assert rel_err(got_n.cpu().numpy(), wide_n[:, c0:c0 + w].cpu().numpy()) < 3e-6, (w, c0)
            assert rel_err(got_r.cpu().numpy(), wide_r[:, c0:c0 + w].cpu().numpy()) < 3e-6, (w, c0)


def test_bench_helpers_accept_a_column_sharded_trainer(golden_models, golden_meta, tiny_data):
    """What bench.py's rank 0 does after an N > 1 run -- time the rank's SpMM launch in its three flavours, look up
    the committed PMC traffic -- works on a column-sharded trainer (here: rank 1 of 4 virtual ranks)."""
    import bench
    gm, meta = golden_models, golden_meta
    group = LockstepGroup(4)
    trainers = [ShardedTrainer(tiny_data, 64, layout="cols", comm=group.comm(r), noise_fn=None, **make_kw("XSimGCL", gm, meta, noise=False))
                for r in range(4)]
    for tr in trainers:
        tr.sampler.seed(3)
        tr.begin_epoch()
    lockstep_step(group, trainers)
    t = bench.time_spmm_kernel(trainers[1], iters=3)
    assert set(t) == {"dense", "row_masked", "col_masked", "step_mix"} and all(v > 0 for v in t.values())
    args = bench.parse([])
    traffic, note = bench.pmc_traffic_cols(args, trainers[1].w)
    assert trainers[1].w == 16 and traffic > 0 and "pmc" in note
    alg = bench.spmm_alg_bytes(trainers[1].adj.nnz, trainers[1].adj.shape[0], trainers[1].adj.shape[1], trainers[1].w)
    assert alg > 0


# ---------------------------------------------------------------------------------------------------
# the 2-D grid (column blocks x row parts) on ONE GPU: Gc * Gr virtual ranks, one python THREAD each
# (the row all-gathers sit inside the step's halves, so the ranks cannot be driven phase by phase), their two
# communicators stood in by barriers + device copies on the shared stream
# ---------------------------------------------------------------------------------------------------
class ThreadGrid:
    class _Shared:
        def __init__(self, n):
            import threading
            self.n, self.slots, self.barrier = n, [None] * n, threading.Barrier(n, timeout=120)

    class _Comm:
        def __init__(self, shared, rank):
            self.shared, self.rank, self.world = shared, rank, shared.n

        def all_gather(self, out, inp):
            sh = self.shared
            sh.slots[self.rank] = inp
            sh.barrier.wait()
            flat, n = out.view(-1), inp.numel()
            for r, t in enumerate(sh.slots):
                piece = flat[r * n:(r + 1) * n]
                if piece.data_ptr() != t.data_ptr():
                    piece.copy_(t.reshape(-1))
            sh.barrier.wait()

    def __init__(self, gc, gr):
        self.gc, self.gr = gc, gr
        self.col_groups = [self._Shared(gc) for _ in range(gr)]      # one per row part: its Gc ranks trade batch rows
        self.row_groups = [self._Shared(gr) for _ in range(gc)]      # one per column block: its Gr ranks trade table rows

    def comms(self, q):
        c, r = q // self.gr, q % self.gr
        return self._Comm(self.col_groups[r], c), self._Comm(self.row_groups[c], r)

    def run(self, body):
        """body(q, comms) on one thread per virtual rank; returns their results in rank order, re-raises a failure."""
        import threading
        res, err = [None] * (self.gc * self.gr), []

        def work(q):
            try:
                torch.cuda.set_device(0)
                res[q] = body(q, self.comms(q))
            except BaseException as e:       # noqa: BLE001  (a dead rank would leave the others at a barrier)
                err.append(e)
                for g in self.col_groups + self.row_groups:
                    g.barrier.abort()
        threads = [threading.Thread(target=work, args=(q,)) for q in range(self.gc * self.gr)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if err:
            raise next((e for e in err if not isinstance(e, __import__("threading").BrokenBarrierError)), err[0])
        return res


def whole_table_2d(results, gc, gr):
    """[(E0 of rank q in all-gather order, pos)] -> the (N, d) table in node order (row part 0 of every column block)."""
    blocks = [results[c * gr][0][results[c * gr][1]] for c in range(gc)]
    return torch.cat(blocks, dim=1)


@pytest.mark.parametrize("name,grid", [("XSimGCL", (2, 2)), ("XSimGCL", (4, 2)), ("LightGCN", (2, 2)), ("SGL", (2, 4)),
                                       ("SimGCL", (2, 2)), ("MF", (2, 2))])
def test_virtual_ranks_2d_match_reference_run(golden_models, golden_meta, tiny_data, name, grid):
    """The 2-D layout over Gc x Gr virtual ranks == the reference's own 3 training steps (noise injected): slice
    kernels on row parts of the graph, per-layer row all-gathers inside a column block, the batch-row exchange inside a
    row part, Adam on (N / Gr, d / Gc) slices."""
    gm, meta = golden_models, golden_meta
    gc, gr = grid
    d = meta[name]["emb"]
    if d % gc or d // gc not in WIDTHS + (64,):
        pytest.skip(f"d / Gc = {d}/{gc} is not a slice width")
    tiny_data.interaction_mat                                  # (lazy attribute: build it before the threads race for it)

    def body(q, comms):
        kw = make_kw(name, gm, meta)
        gen = torch.Generator().manual_seed(kw.pop("noise_seed"))      # every rank draws the same noise stream
        tr = ShardedTrainer(tiny_data, d, layout=f"2d:{gc}x{gr}", comm=comms,
                            noise_fn=lambda shape: torch.rand(shape, generator=gen), **kw)
        assert tr.cols and tr.sharded and (tr.Gc, tr.Gr, tr.cr, tr.rr) == (gc, gr, q // gr, q % gr)
        assert tr.E0.shape == (gr * tr.n_pad, d // gc) and tr.adj.shape == (tr.n_pad, gr * tr.n_pad)
        rnd = random.Random(meta[name]["sampler_seed"])                # (threads: no shared global `random` state)
        tr.sampler.set_state_from_python(rnd.getstate())
        nb = tr.begin_epoch()
        losses = []
        for _ in range(nb):
            tr.step()
            losses.append(tr.read_losses())
        return tr.E0.clone(), tr._pos_dev, losses, tr.cl_rate

    res = ThreadGrid(gc, gr).run(body)
    for other in res[1:]:
        np.testing.assert_allclose(other[2], res[0][2], rtol=1e-6)        # replicated loss section
    bpr, cl = [l[0] for l in res[0][2]], [l[2] for l in res[0][2]]
    np.testing.assert_allclose(bpr, gm[f"{name}_loss_bpr"], rtol=1e-5)
    nb = len(bpr)
    if name in ("XSimGCL", "SimGCL"):
        np.testing.assert_allclose(cl, gm[f"{name}_loss_nce"].reshape(nb, 2).sum(1) * res[0][3], rtol=2e-5)
    elif name == "SGL":
        np.testing.assert_allclose(cl, gm[f"{name}_loss_nce"] * res[0][3], rtol=2e-5)
    E0 = whole_table_2d(res, gc, gr).cpu().numpy()
    U = gm[f"{name}_param_user"].shape[0]
    assert rel_err(E0[:U], gm[f"{name}_param_user"]) < 1e-4 and rel_err(E0[U:], gm[f"{name}_param_item"]) < 1e-4
    # every row part of a column block ends with the same table (the all-gather after Adam)
    for c in range(gc):
        for r in range(1, gr):
            assert torch.equal(res[c * gr + r][0], res[c * gr][0])


@pytest.mark.parametrize("use_graph", [False, True])
def test_dp_layout_through_rccl_single_rank(golden_models, golden_meta, tiny_data, use_graph):
    """shard="dp" (data parallel) with its all-reduce of the dense gradient going through RCCL ("nccl"), world size 1: the
    single-GPU step with one collective between the backward chain and Adam -- eager, and as the two captured graphs around
    the all-reduce -- reproduces the reference's own 3-step run (injected noise keeps the step eager either way) and, with
    in-kernel noise, the unsharded trainer."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        gm, meta = golden_models, golden_meta
        name = "XSimGCL"
        kw = make_kw(name, gm, meta)
        gen = torch.Generator().manual_seed(kw.pop("noise_seed"))
        tr = ShardedTrainer(tiny_data, meta[name]["emb"], layout="dp", noise_fn=lambda shape: torch.rand(shape, generator=gen), **kw)
        assert tr.dp and tr.G == 1 and tr.w == tr.d and tr.vfree == FusedTrainer(tiny_data, meta[name]["emb"], **make_kw(name, gm, meta, noise=False)).vfree
        random.seed(meta[name]["sampler_seed"])
        tr.seed_sampler_from_python()
        bpr = []
        for _ in range(tr.begin_epoch()):
            tr.step()
            bpr.append(tr.read_losses()[0])
        np.testing.assert_allclose(bpr, gm[f"{name}_loss_bpr"], rtol=1e-5)
        pu, pi = tr.parameters_full()
        assert rel_err(pu.cpu().numpy(), gm[f"{name}_param_user"]) < 1e-4
        assert rel_err(pi.cpu().numpy(), gm[f"{name}_param_item"]) < 1e-4
        # in-kernel noise, graph capture on / off: the same steps as the unsharded trainer
        kw = make_kw(name, gm, meta, noise=False)
        single = FusedTrainer(tiny_data, meta[name]["emb"], noise_fn=None, use_graph=use_graph, **kw)
        dp = ShardedTrainer(tiny_data, meta[name]["emb"], layout="dp", noise_fn=None, use_graph=use_graph, **kw)
        assert dp.use_graph == use_graph
        for t in (single, dp):
            t.seed_sampler(5)
        for _ in range(2):
            nb = single.begin_epoch()
            assert dp.begin_epoch() == nb
            for _ in range(nb):
                single.step(); dp.step()
        torch.cuda.synchronize()
        assert dp.step_count == single.step_count == 2 * nb
        assert rel_err(dp.E0.cpu().numpy(), single.E0.cpu().numpy()) < 1e-4
        np.testing.assert_allclose(dp.read_losses(), single.read_losses(), rtol=1e-4)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["rows", "dp"])
def test_layouts_through_the_c_abi_collectives_single_rank(golden_models, golden_meta, tiny_data, layout):
    """The collectives behind the C ABI (include/selfrec_hip.h: srh_comm_init_rank, srh_allgather_rows,
    srh_reducescatter_rows, srh_allreduce_sum_f32 -- thin RCCL wrappers; comm.AbiComm binds them): a real RCCL communicator of
    ONE rank made through the header, the row-sharded step's all-gather after every product (rows) and the data-parallel
    step's gradient all-reduce (dp) issued through it on the step's stream -- no torch.distributed in the data path -- and the
    reference's own 3-step run reproduced.  (More ranks than one need more GPUs than this box has: the call shapes are the
    ones tests/test_dist_cpu.py drives at world 2-8 over gloo.)"""
    from selfrec_amd.comm import AbiComm
    comm = AbiComm(world=1, rank=0)
    x = torch.arange(24, dtype=torch.float32, device="cuda").reshape(6, 4)
    out = torch.zeros_like(x)
    comm.all_gather(out, x)
    rs = torch.zeros_like(x)
    comm.reduce_scatter_sum(rs, x)
    ar = x.clone()
    comm.all_reduce_sum(ar)
    torch.cuda.synchronize()
    assert torch.equal(out, x) and torch.equal(rs, x) and torch.equal(ar, x)
    comm.assert_replicated("a checksum", [1.5, 2.5], torch.device("cuda"))
    gm, meta = golden_models, golden_meta
    name = "XSimGCL"
    kw = make_kw(name, gm, meta)
    gen = torch.Generator().manual_seed(kw.pop("noise_seed"))
    tr = ShardedTrainer(tiny_data, meta[name]["emb"], layout=layout, comm=comm,
                        noise_fn=lambda shape: torch.rand(shape, generator=gen), **kw)
    assert tr.layout == layout and tr.comm is comm
    random.seed(meta[name]["sampler_seed"])
    tr.seed_sampler_from_python()
    bpr = []
    for _ in range(tr.begin_epoch()):
        tr.step()
        bpr.append(tr.read_losses()[0])
    np.testing.assert_allclose(bpr, gm[f"{name}_loss_bpr"], rtol=1e-5)
    pu, pi = tr.parameters_full()
    assert rel_err(pu.cpu().numpy(), gm[f"{name}_param_user"]) < 1e-4
    assert rel_err(pi.cpu().numpy(), gm[f"{name}_param_item"]) < 1e-4
