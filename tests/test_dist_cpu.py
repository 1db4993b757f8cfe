"""world_size-2 (3, 4, 8) gloo runs of the sharded trainer on CPU, both layouts -- column-sharded tables
(one all-gather of the batch rows per step) and row-sharded graph + tables (an all-gather per layer): the
product's step code (engine.FusedTrainer via dist.ShardedTrainer) over CPU stand-ins of the kernels
(tests/cpu_ops.py) gives the single-process oracle's result on the same batches and injected noise."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import selfrec_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_moves_device_tensors(out_dir, rank):
    """tests/test_gpu_multiproc.py: ranks of ONE GPU talk over gloo with device tensors (RCCL refuses two ranks on a
    device).  A torch build whose gloo cannot do that leaves a flag and the test skips."""
    try:
        t = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        return True
    except Exception as e:                                   # noqa: BLE001
        with open(os.path.join(out_dir, "no_device_gloo"), "w") as f:
            f.write(repr(e))
        return False


def _worker(rank, world, port, model, layout, out_path, d=64, device="cpu"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      SRH_2D_TWOHOP_MIN_BYTES="0")           # (2-D: even these small tables take the two-hop exchange)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from selfrec_amd import engine, synth
    from selfrec_amd.data import device_graph
    from selfrec_amd.data.ui_graph import Interaction
    from selfrec_amd.dist import ShardedTrainer
    if device == "cpu":
        from tests import cpu_ops
        engine.ops = device_graph.ops = cpu_ops   # the product's step code over CPU stand-ins of the kernels
    elif not _gloo_moves_device_tensors(os.path.dirname(out_path), rank):
        dist.destroy_process_group()
        return
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    torch.manual_seed(0)
    ue = torch.nn.init.xavier_uniform_(torch.empty(U, d)); ie = torch.nn.init.xavier_uniform_(torch.empty(I, d))
    gen = torch.Generator().manual_seed(7)
    tr = ShardedTrainer(data, d, model=model, n_layers=3, batch_size=1000, layer_cl=2, tau=0.2, eps=0.2, cl_rate=0.2,
                        drop_rate=0.1, user_emb=ue, item_emb=ie, noise_fn=lambda s: torch.rand(s, generator=gen), device=device,
                        layout=layout)
    assert tr.G == world
    if layout == "rows":
        assert tr.sharded and tr.P == world * tr.n_pad
    elif layout.startswith("2d"):
        # column blocks x row parts: (N / Gr, d / Gc) slices, tables kept whole in all-gather order of the row parts
        assert tr.sharded and tr.cols and tr.Gc * tr.Gr == world and tr.Gr >= 2
        assert tr.w == d // tr.Gc and tr.E0.shape == (tr.Gr * tr.n_pad, tr.w) and tr.m.shape == (tr.n_pad, tr.w)
        assert (tr.cr, tr.rr) == (rank // tr.Gr, rank % tr.Gr)
    else:
        assert tr.cols and tr.w == d // world and tr.E0.shape == (U + I, d // world)
    import random
    random.seed(11)
    tr.seed_sampler_from_python()             # (SGL: the two dropped views come out of this stream first)
    tr.begin_epoch()
    losses = []
    for _ in range(3):
        tr.step()
        losses.append(tr.read_losses())
    pu, pi = (t.cpu() for t in tr.parameters_full())
    fu, fi = (t.cpu() for t in tr.embeddings())
    if rank == 0:
        eu, ei, ej = tr.epoch_node_ids()      # staged as table rows: back to node ids for the oracle
        np.savez(out_path, pu=pu.numpy(), pi=pi.numpy(), fu=fu.numpy(), fi=fi.numpy(), losses=np.asarray(losses),
                 u=eu, i=ei, j=ej, train_u=data.train_u, train_i=data.train_i, ue=ue.numpy(), ie=ie.numpy())
    dist.barrier()
    dist.destroy_process_group()


CASES = [("XSimGCL", 2, "rows"), ("LightGCN", 2, "rows"), ("MF", 2, "rows"), ("XSimGCL", 3, "rows"), ("SimGCL", 2, "rows"),
         ("SGL", 2, "rows"),
         ("XSimGCL", 2, "cols"), ("XSimGCL", 4, "cols"), ("XSimGCL", 8, "cols"), ("LightGCN", 2, "cols"), ("MF", 4, "cols"),
         ("SimGCL", 2, "cols"), ("SGL", 2, "cols"),
         # the 2-D grid: 4 x 2 at d = 128 is the layout BASELINE configs[3] takes on 8 GPUs
         ("XSimGCL", 8, "2d@128"), ("XSimGCL", 4, "2d"), ("LightGCN", 4, "2d"), ("SGL", 4, "2d"), ("SimGCL", 4, "2d:1x4"),
         ("MF", 4, "2d")]


@pytest.mark.parametrize("model,world,layout", CASES + [("XSimGCL", 2, "cols@128")])
def test_sharded_equals_single_process_oracle(tmp_path, model, world, layout):
    out = str(tmp_path / "res.npz")
    layout, _, dd = layout.partition("@")               # (d = 128 over 2 ranks: 64-column slices)
    d = int(dd or 64)
    mp.spawn(_worker, args=(world, _free_port(), model, layout, out, d), nprocs=world, join=True)
    check_against_oracle(out, model, d)


def check_against_oracle(out, model, d, atol=2e-6):
    """(atol: the device kernels sum in another order than the oracle, and three Adam steps of lr 1e-3 amplify that where
    a gradient is nearly zero -- tests/test_gpu_multiproc.py passes 2e-5, the bound of the single-process GPU engine tests)"""
    r = np.load(out)
    gen = torch.Generator().manual_seed(7)
    ref = O.OracleTrainer(model, r["train_u"], r["train_i"], 300, 500, d, n_layers=3, batch_size=1000, layer_cl=2,
                          tau=0.2, eps=0.2, cl_rate=0.2, drop_rate=0.1, user_emb=r["ue"], item_emb=r["ie"],
                          noise_fn=lambda s: torch.rand(s, generator=gen))
    if model == "SGL":                       # the oracle draws its two dropped views from the same stream
        import random
        random.seed(11)
        ref.resample_views()
    want = []
    for b in range(3):
        lo, hi = b * 1000, (b + 1) * 1000
        want.append(ref.step(r["u"][lo:hi].tolist(), r["i"][lo:hi].tolist(), r["j"][lo:hi].tolist()))
    np.testing.assert_allclose(r["losses"], np.asarray(want), rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(r["pu"], ref.user_emb.detach().numpy(), rtol=1e-4, atol=atol)
    np.testing.assert_allclose(r["pi"], ref.item_emb.detach().numpy(), rtol=1e-4, atol=atol)
    fu, fi = ref.embeddings()
    np.testing.assert_allclose(r["fu"], fu, rtol=1e-4, atol=atol)
    np.testing.assert_allclose(r["fi"], fi, rtol=1e-4, atol=atol)


def _twohop_worker(rank, world, port, gc, gr, flag_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      SRH_2D_TWOHOP_MIN_BYTES="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from selfrec_amd.engine import TorchComm, TwoHopRows
    cols, rows = TorchComm.grid(gc, gr)
    assert isinstance(rows, TwoHopRows) and (cols.world, rows.world) == (gc, gr)
    assert (cols.rank, rows.rank) == (rank // gr, rank % gr)
    for n in (world * 5, 1003, 7):                           # divisible by G, ragged (padded pieces), smaller than G
        table = torch.full((gr * n,), -1.0)
        mine = table[rows.rank * n:(rows.rank + 1) * n]      # in place, as the engine calls it
        mine.copy_(torch.arange(n, dtype=torch.float32) + 1000.0 * rank)
        rows.all_gather(table, mine)
        want = torch.cat([torch.arange(n, dtype=torch.float32) + 1000.0 * ((rank // gr) * gr + r) for r in range(gr)])
        assert torch.equal(table, want), (rank, n)
        direct = torch.empty_like(table)
        rows.direct.all_gather(direct, mine.clone())
        assert torch.equal(direct, want)
    # the batch-row communicator: ranks of the same row part, one per column block
    got = torch.empty(gc)
    cols.all_gather(got, torch.tensor([float(rank)]))
    assert got.tolist() == [float(c * gr + rank % gr) for c in range(gc)]
    open(os.path.join(flag_dir, f"ok{rank}"), "w").close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gc,gr", [(4, 2), (2, 4), (2, 3)])
def test_two_hop_row_exchange_equals_direct_all_gather(tmp_path, gc, gr):
    """the per-layer all-gather of a column block moved as two all-to-alls over all G ranks (every xGMI
    link carries 1/G of the slab per hop) leaves exactly what the direct group all-gather leaves."""
    world = gc * gr
    mp.spawn(_twohop_worker, args=(world, _free_port(), gc, gr, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_shard_adjacency_layout():
    import scipy.sparse as sp
    from selfrec_amd.dist import shard_adjacency
    rng = np.random.default_rng(0)
    a = sp.random(23, 23, density=0.2, random_state=1, dtype=np.float32).tocsr()
    x = rng.standard_normal((23, 4)).astype(np.float32)
    world = 4
    n_pad = 6
    gathered = np.zeros((world * n_pad, 4), dtype=np.float32)
    for q in range(23):
        gathered[(q % world) * n_pad + q // world] = x[q]
    full = a @ x
    for r in range(world):
        indptr, indices, vals, npad = shard_adjacency(a, r, world)
        assert npad == n_pad and len(indptr) == n_pad + 1
        loc = sp.csr_matrix((vals, indices, indptr), shape=(n_pad, world * n_pad)) @ gathered
        own = np.arange(r, 23, world)
        np.testing.assert_allclose(loc[:len(own)], full[own], rtol=1e-6)
        assert not loc[len(own):].any()


def test_layout_choice():
    from selfrec_amd.dist import pick_layout
    assert [pick_layout(64, w) for w in (1, 2, 3, 4, 8, 16)] == ["rows", "cols", "rows", "cols", "cols", "rows"]
    assert pick_layout(128, 8) == "cols" and pick_layout(128, 2) == "cols" and pick_layout(64, 2, "rows") == "rows"
    assert pick_layout(256, 8) == "cols" and pick_layout(128, 3) == "rows"
    # the 2-D grid: on request, and by itself for gather-bound graphs once the column blocks would fall below 32
    assert pick_layout(128, 8, "2d") == "2d:4x2" and pick_layout(64, 4, "2d") == "2d:2x2" and pick_layout(64, 8, "2d:2x4") == "2d:2x4"
    assert pick_layout(128, 8, nnz=80_600_000) == "2d:4x2" and pick_layout(128, 4, nnz=80_600_000) == "cols"
    assert pick_layout(64, 8, nnz=2_521_586) == "cols" and pick_layout(64, 8, nnz=80_600_000) == "2d:2x4"
    with pytest.raises(Exception):
        pick_layout(128, 8, "2d:3x2")


def _eval_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from selfrec_amd.dist import deal_users, gather_ranked
    n_users, k = 37, 5                                       # (not a multiple of the world size: padded shares)
    uid = list(range(100, 100 + n_users))
    mine, n_max = deal_users(uid, rank, world)
    assert n_max == (n_users + world - 1) // world and len(mine) in (n_max, n_max - 1)
    ids = np.asarray([[u * 10 + c for c in range(k)] for u in mine], dtype=np.int32).reshape(len(mine), k)   # stand-in ranking
    table = gather_ranked(ids, n_users, rank, world, "cpu")
    want = np.asarray([[u * 10 + c for c in range(k)] for u in uid], dtype=np.int32)
    assert np.array_equal(table.numpy(), want)               # every rank holds every user's list, in test-set order
    if rank == 0:
        np.save(out_path, table.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_evaluation_deals_and_gathers(tmp_path, world):
    """bench.py's eval_throughput_sharded: users dealt round-robin, ranked shares all-gathered and re-ordered."""
    out = str(tmp_path / "ids.npy")
    mp.spawn(_eval_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.load(out).shape == (37, 5)


def _seed_worker(rank, world, port, flag_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from selfrec_amd import engine, synth
    from selfrec_amd._lib import SelfrecHipError
    from selfrec_amd.data import device_graph
    from selfrec_amd.data.ui_graph import Interaction
    from selfrec_amd.dist import ShardedTrainer
    from tests import cpu_ops
    engine.ops = device_graph.ops = cpu_ops
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    kw = dict(model="LightGCN", n_layers=2, batch_size=1000, device="cpu", layout="cols")
    # (1) different torch seeds -> different initial tables: construction must fail on every rank
    torch.manual_seed(rank)
    try:
        ShardedTrainer(data, 64, **kw)
        caught = False
    except SelfrecHipError as e:
        caught = "initial embedding tables" in str(e)
    assert caught
    # (2) same tables, different sampler seeds -> different batches: the first epoch upload must fail
    torch.manual_seed(0)
    tr = ShardedTrainer(data, 64, **kw)
    tr.sampler.seed(100 + rank)
    try:
        tr.begin_epoch()
        caught = False
    except SelfrecHipError as e:
        caught = "sampled epoch" in str(e)
    assert caught
    # (3) identical seeds: fine (a fresh trainer: the sampler keeps its shuffled edge order from epoch to epoch)
    torch.manual_seed(0)
    tr = ShardedTrainer(data, 64, **kw)
    tr.sampler.seed(5)
    tr.begin_epoch()
    tr.step()
    open(os.path.join(flag_dir, f"ok{rank}"), "w").close()
    dist.barrier()
    dist.destroy_process_group()


def test_unequal_seeds_across_ranks_are_detected(tmp_path):
    """the step code assumes replicated tables and batches -- a rank seeded differently now
    stops the job with an error instead of silently training on different data."""
    mp.spawn(_seed_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_bench_gpus_n_launches_itself_under_torchrun():
    """`python bench.py --gpus 2` -- no launcher, the way the driver spells the single-GPU command --
    must start one rank per GPU itself (torch.distributed.run) instead of exiting.  Here: gloo instead of RCCL, and the
    ranks stop after the rendezvous + one collective (everything later needs a GPU)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SRH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["launch_check"] and rec["world"] == 2 and rec["rank_sum"] == 3.0 and rec["ranks_in_collective"] == 2
    # the HEADLINE of `--gpus N` is north_star's configuration: ONE batch of 2048 pairs divided over the ranks (d / 2 = 32
    # columns per rank), strong scaling; data parallel (N x 2048 pairs per step) is the sub-record
    assert rec["headline"]["global_batch"] == 2048 and rec["headline"]["scaling"] == "strong"
    assert rec["headline"]["layout"] == "cols" and rec["headline"]["parallelism"].startswith("column-sharded tables x2")
    # ... and so is the plain row partition BASELINE.json names (the headline at this shape is column blocks)
    assert rec["sub_records"] == ["dp", "rows"] and "layout_rows" in rec["baseline_partition"]
    assert "torch.distributed.run" in p.stderr
    # the rendezvous port is picked free per launch, never a fixed number
    assert rec["master_port"] != 29511 and f"--master-port {rec['master_port']}" in p.stderr


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_headline_keeps_the_global_batch_at_2048(world):
    """whatever N, the first record of `bench.py --gpus N` is the fixed-batch partition (B = 2048, strong);
    the plain row partition BASELINE.json names is timed beside pick_layout's choice at every shape."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from selfrec_amd.dist import pick_layout
    for shape, emb in (("yelp2018", 64), ("1m-500k", 128)):
        args = bench.parse(["--gpus", str(world), "--shape", shape, "--emb", str(emb)])
        head = bench.headline_layout(args, world)
        chosen = pick_layout(args.emb, world, head, 100_000_000 if shape == "1m-500k" else 2_500_000)
        assert chosen != "dp" and bench.scaling_of(chosen) == "strong"
        assert args.batch * (world if chosen == "dp" else 1) == 2048
        subs = bench.sub_record_layouts(args, world, head)
        assert subs[0] == "dp" and "rows" in subs          # north_star's row partition is timed at every shape


def _bench(args, **env_extra):
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SRH_DIST_BACKEND")}
    env.update(OMP_NUM_THREADS="1", **env_extra)
    return subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=600)


def test_bench_refuses_more_ranks_than_visible_devices():
    """`--gpus N` above the visible HIP devices (none in this container): a clear message and exit code 2 BEFORE any rank is
    started, not N ranks fighting over fewer GPUs."""
    import json
    p = _bench(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 2, (p.returncode, p.stderr[-1500:])
    assert "--gpus 4" in p.stderr and "HIP device(s) visible" in p.stderr
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 4 and rec["visible_gpus"] < 4 and "error" in rec
    assert "torch.distributed.run" not in p.stderr


def test_bench_watchdog_ends_a_stalled_job():
    """A rank that makes no progress for SRH_BENCH_WATCHDOG_S seconds (here: a stall injected before the first collective)
    ends the job with exit code 3 and a message that names the phase -- it never hangs until the driver's limit."""
    import time
    t0 = time.time()
    p = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], SRH_DIST_BACKEND="gloo", SRH_BENCH_WATCHDOG_S="3",
               SRH_BENCH_TEST_STALL="600")
    assert p.returncode != 0
    assert "watchdog: rank" in p.stderr and "made no progress" in p.stderr and "phase 'start'" in p.stderr
    assert time.time() - t0 < 120


def test_bench_runner_counts_epoch_boundaries_and_takes_the_max_over_ranks():
    """bench.Runner / first_steps_guarded on a stub trainer: exactly n steps per timed region, epoch boundaries counted,
    and a trainer whose first steps raise is rebuilt ONCE with eager launches."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    class Stub:
        epoch_batches = 5

        def __init__(self, fail=False):
            self.steps, self.uploads, self.fail = 0, 0, fail

        def seed_sampler(self, seed):
            self.seed = seed

        def sample_epoch_host(self):
            return {"e": 1}

        def upload_epoch(self, host):
            self.uploads += 1

        def step(self):
            if self.fail:
                raise RuntimeError("captured replay differs from the eager step")
            self.steps += 1

    class NoFence(bench.Runner):
        def fence(self):
            pass

    r = NoFence(Stub(), 7)
    r.run(3)
    dt, bounds, per_rank = r.timed(9, "x")
    assert r.trainer.steps == 12 and r.trainer.uploads == 3 and bounds == 2 and dt >= 0 and per_rank is None
    assert r.steps_done == 12
    # a short region is moved (untimed steps) to where it straddles an epoch boundary: 5-batch epochs, 3 left now
    assert r.left == 3 and r.align_to_epoch_boundary(2) == 2 and r.left == 1
    _, bounds, _ = r.timed(2, "y")
    assert bounds == 1
    assert r.align_to_epoch_boundary(9) == 0                     # (longer than an epoch: nothing to align)
    built = []

    def make(eager=False):
        built.append(eager)
        return Stub(fail=not eager)
    tr, runner, note = bench.first_steps_guarded(make, lambda t: NoFence(t, 7), None, "layout dp")
    assert built == [False, True] and tr.steps == 2 and "RuntimeError" in note


@pytest.mark.parametrize("model,d", [("XSimGCL", 50), ("SGL", 96), ("LightGCN", 20), ("SimGCL", 100)])
def test_any_embedding_size_is_stored_padded_and_trains_like_the_oracle(monkeypatch, model, d):
    """base/recommender.py:16 takes any `embedding.size`.  The engine stores its tables zero-padded to the next width
    the kernels serve (50 -> 64, 96 / 100 -> 128, 20 -> 32): here its step code runs over the CPU stand-ins of the
    kernels and must give the single-process oracle's result AT THE REAL WIDTH, with the padding columns still exactly
    zero after the Adam steps (the GPU kernels are held to the same in tests/test_gpu_engine.py)."""
    from selfrec_amd import engine, synth
    from selfrec_amd.data import device_graph
    from selfrec_amd.data.ui_graph import Interaction
    from tests import cpu_ops
    monkeypatch.setattr(engine, "ops", cpu_ops)
    monkeypatch.setattr(device_graph, "ops", cpu_ops)
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    torch.manual_seed(0)
    ue = torch.nn.init.xavier_uniform_(torch.empty(U, d)); ie = torch.nn.init.xavier_uniform_(torch.empty(I, d))
    kw = dict(n_layers=2, batch_size=1000, layer_cl=1, tau=0.2, eps=0.2, cl_rate=0.2, drop_rate=0.1)
    gen = torch.Generator().manual_seed(7)
    tr = engine.FusedTrainer(data, d, model=model, user_emb=ue, item_emb=ie, device="cpu",
                             noise_fn=lambda s: torch.rand(s, generator=gen), **kw)
    assert tr.d_valid == d and tr.d in (32, 64, 128) and tr.d > d and tr.E0.shape[1] == tr.d
    import random
    random.seed(11)
    tr.seed_sampler_from_python()
    tr.begin_epoch()
    losses = []
    for _ in range(3):
        tr.step()
        losses.append(tr.read_losses())
    assert not tr.E0[:, d:].any() and tr.user_emb.shape == (U, d)
    gen = torch.Generator().manual_seed(7)
    ref = O.OracleTrainer(model, data.train_u, data.train_i, U, I, d, user_emb=ue.numpy(), item_emb=ie.numpy(),
                          noise_fn=lambda s: torch.rand(s, generator=gen), **kw)
    if model == "SGL":
        random.seed(11)
        ref.resample_views()
    eu, ei, ej = tr.epoch_node_ids()
    want = [ref.step(eu[b * 1000:(b + 1) * 1000].tolist(), ei[b * 1000:(b + 1) * 1000].tolist(),
                     ej[b * 1000:(b + 1) * 1000].tolist()) for b in range(3)]
    np.testing.assert_allclose(np.asarray(losses), np.asarray(want), rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(tr.user_emb.numpy(), ref.user_emb.detach().numpy(), rtol=1e-4, atol=2e-6)
    fu, fi = tr.embeddings()
    wu, wi = ref.embeddings()
    np.testing.assert_allclose(fi.numpy(), wi, rtol=1e-4, atol=2e-6)


def _dp_worker(rank, world, port, model, out_dir, device="cpu"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from selfrec_amd import engine, synth
    from selfrec_amd.data import device_graph
    from selfrec_amd.data.ui_graph import Interaction
    from selfrec_amd.dist import ShardedTrainer
    if device == "cpu":
        from tests import cpu_ops
        engine.ops = device_graph.ops = cpu_ops
    elif not _gloo_moves_device_tensors(out_dir, rank):
        dist.destroy_process_group()
        return
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    torch.manual_seed(0)
    ue = torch.nn.init.xavier_uniform_(torch.empty(U, 64)); ie = torch.nn.init.xavier_uniform_(torch.empty(I, 64))
    gen = torch.Generator().manual_seed(100 + rank)           # every rank draws its OWN perturbation noise
    tr = ShardedTrainer(data, 64, model=model, n_layers=2, batch_size=500, layer_cl=1, tau=0.2, eps=0.2, cl_rate=0.2,
                        user_emb=ue, item_emb=ie, noise_fn=lambda s: torch.rand(s, generator=gen), device=device, layout="dp")
    assert tr.dp and not tr.cols and not tr.sharded and tr.G == world and tr.E0.shape == (U + I, 64)
    assert (tr.rng_seed == 0x5E1F0EC) == (rank == 0)           # in-kernel noise: every rank its own stream, rank 0 the single-GPU one
    tr.seed_sampler(40)                                        # seed + rank: every rank its own batches
    tr.begin_epoch()
    losses = []
    for _ in range(3):
        tr.step()
        losses.append(tr.read_losses())
    eu, ei, ej = tr.epoch_node_ids()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), E0=tr.E0.cpu().numpy(), u=eu, i=ei, j=ej, losses=np.asarray(losses),
             train_u=data.train_u, train_i=data.train_i, ue=ue.numpy(), ie=ie.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("model,world", [("XSimGCL", 2), ("LightGCN", 4), ("MF", 2)])
def test_data_parallel_equals_the_mean_gradient_oracle(tmp_path, model, world):
    """shard="dp": every rank its own batches, one all-reduce of the dense gradient per step, Adam on the MEAN -- against the
    single-process oracle that evaluates each rank's loss on that rank's batch (with that rank's noise), averages the
    gradients and takes one Adam step; every rank ends with the same table."""
    mp.spawn(_dp_worker, args=(world, _free_port(), model, str(tmp_path)), nprocs=world, join=True)
    check_dp_against_oracle(tmp_path, model, world)


def check_dp_against_oracle(tmp_path, model, world):
    r = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    for k in range(1, world):
        assert np.array_equal(r[k]["E0"], r[0]["E0"])                          # replicas stay replicas
        assert not np.array_equal(r[k]["u"][:500], r[0]["u"][:500])            # ... on different batches
    gens = [torch.Generator().manual_seed(100 + k) for k in range(world)]
    cur = [0]
    ref = O.OracleTrainer(model, r[0]["train_u"], r[0]["train_i"], 300, 500, 64, n_layers=2, batch_size=500, layer_cl=1,
                          tau=0.2, eps=0.2, cl_rate=0.2, user_emb=r[0]["ue"], item_emb=r[0]["ie"],
                          noise_fn=lambda s: torch.rand(s, generator=gens[cur[0]]))
    for b in range(3):
        ref.opt.zero_grad()
        for k in range(world):
            cur[0] = k
            sl = slice(b * 500, (b + 1) * 500)
            rec, regl, cl = ref.losses(r[k]["u"][sl].tolist(), r[k]["i"][sl].tolist(), r[k]["j"][sl].tolist())
            ((rec + regl + cl) / world).backward()
            np.testing.assert_allclose(r[k]["losses"][b], [float(rec), float(regl), float(cl)], rtol=2e-5, atol=1e-9)
        ref.opt.step()
    want = torch.cat([ref.user_emb, ref.item_emb]).detach().numpy()
    np.testing.assert_allclose(r[0]["E0"], want, rtol=1e-4, atol=2e-6)


def test_abi_communicator_has_the_call_shape_the_gloo_worlds_exercise():
    """comm.AbiComm (the placement exchanges through include/selfrec_hip.h's RCCL wrappers) needs a GPU per rank; what the
    world-2 .. 8 tests above drive over gloo is comm.TorchComm.  The two are interchangeable for a placement only if they
    expose the same calls with the same parameters: all_gather(out, inp), all_reduce_sum(t), assert_replicated(what, values,
    device), world / rank / group -- checked here, so that a change to one side cannot leave the other behind."""
    import inspect
    from selfrec_amd.comm import AbiComm, TorchComm
    for name in ("all_gather", "all_reduce_sum", "assert_replicated"):
        a, t = inspect.signature(getattr(AbiComm, name)), inspect.signature(getattr(TorchComm, name))
        assert list(a.parameters) == list(t.parameters), name
    src = inspect.getsource(AbiComm)
    for sym in ("srh_comm_unique_id", "srh_comm_init_rank", "srh_comm_world", "srh_allgather_rows", "srh_reducescatter_rows",
                "srh_allreduce_sum_f32", "srh_comm_destroy"):
        assert sym in src, sym
    for attr in ("self.world", "self.rank", "self.group"):
        assert attr in src
