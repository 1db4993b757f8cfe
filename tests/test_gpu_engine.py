"""End-to-end parity of the fused engine against the reference's own training runs
(tests/golden/models.npz: 3 Adam steps of each model on the 200 x 300 graph, noise injected)
and full-size property checks at the BASELINE.json shapes."""
import random

import numpy as np
import pytest
import torch

from oracle import selfrec_oracle as O
from selfrec_amd import ops, synth
from selfrec_amd.data.ui_graph import Interaction
from selfrec_amd.engine import FusedTrainer

from .conftest import EDGE_TAGS

pytestmark = pytest.mark.gpu
MODELS = ["MF", "LightGCN", "XSimGCL", "SimGCL", "SGL"]


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


def make_trainer(name, gm, meta, data, **over):
    m = meta[name]; c = m["conf"]
    tag, name = name, m.get("model", name)
    gen = torch.Generator().manual_seed(m["noise_seed"])
    kw = dict(model=name, n_layers=int(c.get("n_layer", 0)), lr=m["lr"], reg=m["reg"],
              cl_rate=float(c.get("lambda", 0.0)), eps=float(c.get("eps", 0.0)),
              tau=float(c.get("tau", c.get("temp", 0.2))), layer_cl=int(c.get("l_star", 1)),
              drop_rate=float(c.get("drop_rate", 0.1)), batch_size=m["batch"],
              user_emb=gm[f"{tag}_init_user"], item_emb=gm[f"{tag}_init_item"],
              noise_fn=lambda shape: torch.rand(shape, generator=gen))
    kw.update(over)
    return FusedTrainer(data, m["emb"], **kw)


@pytest.mark.parametrize("name", MODELS)
def test_three_steps_match_reference_run(golden_models, golden_meta, tiny_data, name):
    gm, meta = golden_models, golden_meta
    tr = make_trainer(name, gm, meta, tiny_data)
    random.seed(meta[name]["sampler_seed"])
    tr.seed_sampler_from_python()
    nb = tr.begin_epoch()
    assert nb == meta[name]["n_batches"]
    eu, _, ej = tr.epoch_node_ids()
    assert np.array_equal(eu, gm[f"{name}_batch_u"]) and np.array_equal(ej, gm[f"{name}_batch_j"])
    bpr, cl = [], []
    for _ in range(nb):
        tr.step()
        b, _, c = tr.read_losses()
        bpr.append(b); cl.append(c)
    np.testing.assert_allclose(bpr, gm[f"{name}_loss_bpr"], rtol=1e-5)
    if name in ("XSimGCL", "SimGCL"):          # reference logs user and item InfoNCE separately, unscaled
        ref = gm[f"{name}_loss_nce"].reshape(nb, 2).sum(1) * tr.cl_rate
        np.testing.assert_allclose(cl, ref, rtol=2e-5)
    elif name == "SGL":
        np.testing.assert_allclose(cl, gm[f"{name}_loss_nce"] * tr.cl_rate, rtol=2e-5)
    assert rel_err(tr.user_emb.cpu().numpy(), gm[f"{name}_param_user"]) < 1e-4
    assert rel_err(tr.item_emb.cpu().numpy(), gm[f"{name}_param_item"]) < 1e-4
    # element-wise too: Adam normalises tiny gradients, so check the update, not only the scale
    du = tr.user_emb.cpu().numpy() - gm[f"{name}_init_user"]
    du_ref = gm[f"{name}_param_user"] - gm[f"{name}_init_user"]
    assert np.abs(du - du_ref).max() < 1e-5          # << one Adam step (lr = 1e-3)
    fu, fi = tr.embeddings()
    assert rel_err(fu.cpu().numpy(), gm[f"{name}_final_user"]) < 1e-4
    assert rel_err(fi.cpu().numpy(), gm[f"{name}_final_item"]) < 1e-4


@pytest.mark.parametrize("tag", EDGE_TAGS)
def test_edge_configurations_match_reference_run(golden_edges, tiny_data, tag):
    """The reference's own runs at d = 256 (64 lanes per row), with one layer (the only product also carries the mean),
    with four, and with the contrast view at the ego table (l* = 0) or at the last layer -- tests/golden/make_golden.py
    edges; the same bounds as the five main runs above."""
    ge, meta = golden_edges
    m = meta[tag]
    tr = make_trainer(tag, ge, meta, tiny_data)
    random.seed(m["sampler_seed"])
    tr.seed_sampler_from_python()
    nb = tr.begin_epoch()
    assert nb == m["n_batches"]
    eu, ei, ej = tr.epoch_node_ids()
    assert np.array_equal(eu, ge[f"{tag}_batch_u"]) and np.array_equal(ei, ge[f"{tag}_batch_i"]) \
        and np.array_equal(ej, ge[f"{tag}_batch_j"])
    bpr, cl = [], []
    for _ in range(nb):
        tr.step()
        b, _, c = tr.read_losses()
        bpr.append(b); cl.append(c)
    np.testing.assert_allclose(bpr, ge[f"{tag}_loss_bpr"], rtol=1e-5)
    nce, lam = ge[f"{tag}_loss_nce"], tr.cl_rate
    if m["model"] in ("XSimGCL", "SimGCL"):
        np.testing.assert_allclose(cl, nce.reshape(nb, 2).sum(1) * lam, rtol=2e-5)
    elif m["model"] == "SGL":
        np.testing.assert_allclose(cl, nce * lam, rtol=2e-5)
    for side, got in (("user", tr.user_emb), ("item", tr.item_emb)):
        got, want, init = got.cpu().numpy(), ge[f"{tag}_param_{side}"], ge[f"{tag}_init_{side}"]
        assert rel_err(got, want) < 1e-4
        assert np.abs((got - init) - (want - init)).max() < 1e-5       # << one Adam step (lr = 1e-3)


@pytest.mark.parametrize("name", ["LightGCN", "XSimGCL"])
def test_engine_backed_model_class_ranks_like_reference(golden_models, golden_meta, tiny_data, name, tmp_path, monkeypatch):
    """GraphRecommender.test() through srh_score_mask_topk + ranking_evaluation strings."""
    import importlib
    from selfrec_amd.util.conf import ModelConf
    monkeypatch.chdir(tmp_path)
    gm, meta = golden_models, golden_meta
    m = meta[name]
    conf = ModelConf({"model": {"name": name, "type": "graph"}, "item.ranking.topN": [10, 20],
                      "embedding.size": m["emb"], "max.epoch": 1, "batch.size": m["batch"], "learning.rate": m["lr"],
                      "reg.lambda": m["reg"], "output": "./results/", "training.set": "x", "test.set": "y",
                      name: m["conf"], "engine.hipgraph": False})
    cls = getattr(importlib.import_module(f"selfrec_amd.model.graph.{name}"), name)
    model = cls(conf, tiny_data.training_data, tiny_data.test_data)
    model.user_emb = torch.from_numpy(gm[f"{name}_final_user"]).cuda()
    model.item_emb = torch.from_numpy(gm[f"{name}_final_item"]).cuda()
    rec = model.test()
    from selfrec_amd.util.evaluation import ranking_evaluation
    assert ranking_evaluation(model.data.test_set, rec, [10, 20]) == m["measure"]
    u = next(iter(rec))
    assert len(rec[u]) == 20 and isinstance(rec[u][0][0], str) and isinstance(rec[u][0][1], float)
    assert np.allclose(model.predict(u)[model.data.item[rec[u][0][0]]], rec[u][0][1], rtol=1e-5)


@pytest.mark.selfcheck
def test_first_epoch_sampled_during_construction_is_the_seeds_first_epoch(golden_models, golden_meta, tiny_data):
    """FusedTrainer(sampler_seed=s) draws its first epoch on a host thread while the graph, plans and calibration are built:
    same epochs as seeding after construction; another seed afterwards drops that epoch AND the permuted edge order."""
    def epochs(tr, n=2):
        out = []
        for _ in range(n):
            tr.begin_epoch()
            out.append(tuple(a.copy() for a in tr.epoch_node_ids()))
        return out
    plain = make_trainer("XSimGCL", golden_models, golden_meta, tiny_data, noise_fn=None)
    plain.seed_sampler(5)
    want5 = epochs(plain)
    early = make_trainer("XSimGCL", golden_models, golden_meta, tiny_data, noise_fn=None, sampler_seed=5)
    assert early._first_epoch is not None
    early.seed_sampler(5)                                   # (what bench.Runner does: the same seed keeps the epoch)
    got5 = epochs(early)
    assert early._first_epoch is None and all(np.array_equal(a, b) for e, w in zip(got5, want5) for a, b in zip(e, w))
    other = make_trainer("XSimGCL", golden_models, golden_meta, tiny_data, noise_fn=None, sampler_seed=5)
    other.seed_sampler(7)
    fresh = make_trainer("XSimGCL", golden_models, golden_meta, tiny_data, noise_fn=None)
    fresh.seed_sampler(7)
    assert all(np.array_equal(a, b) for e, w in zip(epochs(other), epochs(fresh)) for a, b in zip(e, w))


def _bare_recommender(data, user_emb, item_emb, max_n):
    from selfrec_amd.base import graph_recommender as gr
    rec = gr.GraphRecommender.__new__(gr.GraphRecommender)
    rec.data, rec.max_N, rec.topN = data, max_n, [max_n]
    rec.user_emb, rec.item_emb = user_emb.cuda(), item_emb.cuda()
    return rec


def test_device_ranking_orders_ties_like_the_reference_heap(golden_ops):
    """reference util/algorithm.py:144-156 keeps a size-K min-heap: WHICH of several equal scores stay, and in what order,
    is a property of that walk (tests/golden: candidates with three 0.9s, three 0.5s and a masked entry -> ids
    [2, 10, 5, 8, 7], not the lowest-id order [2, 5, 10, 8, 0]).  rank_on_device ranks K + 1 on the device, flags rows with
    equal neighbours and redoes exactly those the reference's way."""
    g = golden_ops
    cand = g["topk_ties_in"]
    n = len(cand)
    masked = int(np.argmin(cand))                                       # the entry the golden holds at -10e8
    # items named by their column, appearing in column order; user "q" rated only the masked item
    train = [["p" if j != masked else "q", str(j), 1.0] for j in range(n)]
    data = Interaction({}, train, [["q", "0", 1.0]])
    assert [data.item[str(j)] for j in range(n)] == list(range(n))
    d = 64
    ue = torch.zeros(data.user_num, d); ue[:, 0] = 1.0
    ie = torch.zeros(n, d); ie[:, 0] = torch.from_numpy(np.where(cand < -1e8, 2.0, cand).astype(np.float32))
    rec = _bare_recommender(data, ue, ie, 5)
    ids, sc = rec.rank_on_device(np.asarray([data.user["q"]], dtype=np.int32))
    assert ids[0].tolist() == g["topk_ties_ids"].tolist()
    assert np.array_equal(sc[0], g["topk_ties_scores"])
    # the path test() takes (hit flags and metric rows from the ids): the ties are settled on the device first
    ids_h, sc_h, flags, cuts = rec.rank_on_device(np.asarray([data.user["q"]], dtype=np.int32), with_hits=True, metric_cuts=[5])
    assert ids_h[0].tolist() == g["topk_ties_ids"].tolist() and np.array_equal(sc_h[0], g["topk_ties_scores"])
    assert flags[0].tolist() == [int(i == data.item["0"]) for i in ids_h[0]] and int(cuts[5][0][0]) == int(flags[0].sum())
    # the other user rated everything but the masked item: only that one is left, then masked entries at -10e8 in heap order
    from selfrec_amd.util.algorithm import find_k_largest
    ids_p, sc_p = rec.rank_on_device(np.asarray([data.user["p"]], dtype=np.int32))
    c2 = np.full(n, -10e8, dtype=np.float32); c2[masked] = 2.0
    want_ids, want_sc = find_k_largest(5, c2)
    assert ids_p[0].tolist() == want_ids and np.array_equal(sc_p[0], np.asarray(want_sc, dtype=np.float32))


def test_filtered_ranking_orders_planted_ties_like_the_reference_heap():
    """The same through the filtered pipeline (catalogue >= 4 x the bound slice): duplicate item rows planted among some
    users' best -- some straddling the K-th place -- against util.algorithm.find_k_largest (pinned to the reference's
    outputs by the goldens) on the ranking's own scores; rows without ties are untouched."""
    from selfrec_amd.util.algorithm import find_k_largest
    rng = np.random.default_rng(12)
    U, I, K, d = 300, 20000, 20, 64
    tu, ti = synth.generate_edges(U, I, 60000, 5)
    data = Interaction({}, synth.as_triples(tu, ti), [])
    U, I = data.user_num, data.item_num
    assert I >= 4 * 4096
    ue = (rng.standard_normal((U, d)) * 0.3).astype(np.float32)
    ie = (rng.standard_normal((I, d)) * 0.3).astype(np.float32)
    tied_users = [3, 77, 150]
    for n_dup, u in zip((2, 5, 30), tied_users):                        # 30 > K: the tie straddles the K-th place
        dup = rng.choice(I, size=n_dup, replace=False)
        ie[dup] = ue[u] * 3.0                                            # equal rows, far above everything else for user u
    rec = _bare_recommender(data, torch.from_numpy(ue), torch.from_numpy(ie), K)
    users = np.arange(U, dtype=np.int32)
    ids, sc = rec.rank_on_device(users)
    g = data.device_graph(torch.device("cuda"))
    scores = ops.gemm_nt(rec.user_emb, rec.item_emb).cpu().numpy()
    indptr, indices = g.r_indptr.cpu().numpy(), g.r_indices.cpu().numpy()
    n_heap = 0
    for u in range(U):
        c = scores[u].copy()
        c[indices[indptr[u]:indptr[u + 1]]] = -10e8
        want_ids, want_sc = find_k_largest(K, c)
        assert ids[u].tolist() == want_ids, u
        assert np.array_equal(sc[u], np.asarray(want_sc, dtype=np.float32))
        top = np.sort(c)[::-1][:K + 1]
        n_heap += bool((top[1:] == top[:-1]).any())
    assert n_heap >= len(tied_users)


@pytest.mark.parametrize("I", [20000, 3000])
def test_top128_without_a_spare_column_still_orders_ties_like_the_heap(I):
    """K = 128 leaves the device kernels no (K + 1)-th column to see a tie across the K-th place with
    (_mark_ties_without_spare: equal neighbours among the K, or more catalogue scores equal to the K-th than the K places
    hold).  Planted duplicates inside the best 128 and straddling the 128-th place, both pipelines (filtered / exact slab),
    against find_k_largest on the ranking's own scores."""
    from selfrec_amd.util.algorithm import find_k_largest
    rng = np.random.default_rng(128)
    U, K, d = 60, 128, 64
    tu, ti = synth.generate_edges(U, I, 6000, 9)
    data = Interaction({}, synth.as_triples(tu, ti), [])
    U, I = data.user_num, data.item_num
    ue = (rng.standard_normal((U, d)) * 0.3).astype(np.float32)
    ie = (rng.standard_normal((I, d)) * 0.3).astype(np.float32)
    ie[rng.choice(I, size=3, replace=False)] = ue[5] * 3.0               # a tie at the top of user 5's list
    base = ue[9] / np.linalg.norm(ue[9])
    rows = rng.choice(I, size=140, replace=False)
    ie[rows[:120]] = base * np.linspace(9.0, 5.0, 120, dtype=np.float32)[:, None]     # 120 distinct leaders for user 9 ...
    ie[rows[120:]] = base * 4.0                                                         # ... then 20 equal: 8 places for them
    rec = _bare_recommender(data, torch.from_numpy(ue), torch.from_numpy(ie), K)
    users = np.arange(U, dtype=np.int32)
    ids, sc = rec.rank_on_device(users)
    assert rec._last_tie_rows >= 2
    g = data.device_graph(torch.device("cuda"))
    scores = ops.gemm_nt(rec.user_emb, rec.item_emb).cpu().numpy()
    indptr, indices = g.r_indptr.cpu().numpy(), g.r_indices.cpu().numpy()
    for u in range(U):
        c = scores[u].copy()
        c[indices[indptr[u]:indptr[u + 1]]] = -10e8
        want_ids, want_sc = find_k_largest(K, c)
        assert ids[u].tolist() == want_ids, u
        assert np.array_equal(sc[u], np.asarray(want_sc, dtype=np.float32))


@pytest.mark.selfcheck
def test_epochs_staged_by_the_prefetch_thread_train_like_epochs_uploaded_in_line(golden_models, golden_meta, tiny_data):
    """The device holds two epochs back to back: the prefetch thread copies epoch e + 1 into the half epoch e is not reading
    (pinned memory, its own stream) and the boundary is a cursor write.  Four epochs driven that way -- hipGraph replay, the
    host never waiting for the device -- against the same four uploaded in line (begin_epoch): same batches, same
    parameters, bit for bit."""
    from selfrec_amd.engine import EpochPrefetcher
    outs = []
    for staged in (False, True):
        tr = make_trainer("XSimGCL", golden_models, golden_meta, tiny_data, noise_fn=None, use_graph=True)
        tr.sampler.seed(5)
        halves = []
        if staged:
            pre = EpochPrefetcher(tr)
            pre.start()
            for _ in range(4):
                host = pre.take()
                assert "_staged" in host                                     # the copy is already on its way
                tr.upload_epoch(host)
                pre.start()                                                  # epoch e + 1 is drawn and staged under epoch e
                halves.append(tr._live_half)
                for _ in range(tr.epoch_batches):
                    tr.step()
        else:
            for _ in range(4):
                for _ in range(tr.begin_epoch()):
                    tr.step()
                halves.append(tr._live_half)
        assert halves == [0, 1, 0, 1]
        torch.cuda.synchronize()
        outs.append((tr.E0.cpu().numpy(), tr.read_losses()))
    assert np.isfinite(outs[0][0]).all()
    # (no float atomics in the step -- engine.det_scatter -- so "the same" means the same bits)
    assert np.array_equal(outs[1][0], outs[0][0]) and outs[1][1] == outs[0][1]


@pytest.mark.selfcheck
@pytest.mark.parametrize("name", ["XSimGCL", "SGL", "LightGCN"])
def test_hipgraph_replay_equals_eager(golden_models, golden_meta, tiny_data, name):
    """Same RNG stream, same batches: a captured step replayed == the eager launch sequence."""
    outs = []
    for use_graph in (False, True):
        tr = make_trainer(name, golden_models, golden_meta, tiny_data, noise_fn=None, use_graph=use_graph)
        tr.sampler.seed(5)
        for _ in range(2):
            for _ in range(tr.begin_epoch()):
                tr.step()
        torch.cuda.synchronize()
        outs.append((tr.E0.cpu().numpy(), tr.read_losses()))
    assert np.isfinite(outs[0][0]).all()
    # every sum of the step has a fixed order (the batch gradients too: engine.det_scatter), so the two launch forms agree bit for bit
    assert np.array_equal(outs[1][0], outs[0][0]) and outs[1][1] == outs[0][1]


@pytest.mark.selfcheck
@pytest.mark.parametrize("name", MODELS)
def test_step_is_bit_reproducible(golden_models, golden_meta, tiny_data, name):
    """SURVEY.md 5 (run twice, bit-compare): two trainers, the same seeds, two epochs of captured steps with the in-kernel
    perturbation -- parameters, both Adam moments and the last step's losses come out with the SAME BITS.  The reference's
    step on one CPU thread is reproducible (XSimGCL.py:27-37); here that takes a loss section without float atomics: the
    sampler's row -> slot lists (srh_sampler_epoch_segments) and one writer per gradient row (rows_finish, csrc/losses.hip).
    SRH_DET_SCATTER=0 (the atomic scatter) is the control: it must NOT be what a default trainer runs."""
    outs = []
    for _ in range(2):
        tr = make_trainer(name, golden_models, golden_meta, tiny_data, noise_fn=None, use_graph=True)
        assert tr.det_scatter
        tr.sampler.seed(5)
        for _ in range(2):
            for _ in range(tr.begin_epoch()):
                tr.step()
        torch.cuda.synchronize()
        outs.append((tr.E0.clone(), tr.m.clone(), tr.v.clone(), tr.losses.clone()))
    assert torch.isfinite(outs[0][0]).all() and float(outs[0][1].abs().max()) > 0
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.selfcheck
@pytest.mark.parametrize("name", ["XSimGCL", "LightGCN", "SimGCL", "SGL"])
def test_adam_in_the_last_backward_product_trains_like_the_separate_pass(golden_models, golden_meta, tiny_data, name, monkeypatch):
    """engine.fuse_adam: the optimiser step inside the last backward product's row epilogue (SRH_EPI_ADAM) against the
    srh_adam_step_reset pass after it (SRH_FUSE_ADAM=0) -- same batches, same noise counters, two epochs of captured steps:
    the same parameters, moments, cursor and cleared gradient buffers, bit for bit (as the kernel-level statement in
    test_gpu_kernels.py)."""
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("SRH_FUSE_ADAM", fuse)
        tr = make_trainer(name, golden_models, golden_meta, tiny_data, noise_fn=None, use_graph=True)
        assert tr.fuse_adam == (fuse == "1")
        tr.sampler.seed(5)
        for _ in range(2):
            for _ in range(tr.begin_epoch()):
                tr.step()
        torch.cuda.synchronize()
        sparse = [float(t.abs().max()) for t in tr._sparse_tables()]
        outs.append((tr.E0.cpu().numpy(), tr.m.cpu().numpy(), tr.v.cpu().numpy(), tr.cursor.tolist(), tr.read_losses(), sparse))
    assert np.isfinite(outs[0][0]).all() and outs[0][3] == outs[1][3]
    # (one adam_element for both forms, fixed-order sums everywhere: the same bits)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert outs[0][4] == outs[1][4]
    assert outs[0][5] == outs[1][5] == [0.0] * len(outs[0][5])          # every batch row of the sparse buffers was cleared
    # where the epilogue form does not apply, the pass stays: no propagation (MF), one layer (the product's x is gF itself)
    monkeypatch.setenv("SRH_FUSE_ADAM", "1")
    assert not make_trainer("MF", golden_models, golden_meta, tiny_data).fuse_adam
    assert not make_trainer(name, golden_models, golden_meta, tiny_data, n_layers=1, layer_cl=1).fuse_adam


def test_full_size_properties_yelp_shape():
    """Size-independent checks at BASELINE.json config 2/3 shape (31,668 x 38,048, ~1.26 M train edges)."""
    tu, ti, su, si, U, I = synth.make_dataset("yelp2018")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    g = data.device_graph()
    N, d = U + I, 64
    assert g.adj.nnz == 2 * len(tu)
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((N, d), device="cuda", generator=gen)
    y = torch.randn((N, d), device="cuda", generator=gen)
    ax, ay = ops.spmm(g.adj, x), ops.spmm(g.adj, y)
    # against scipy on the host at full size (741 split rows reduced by their last-arriving segment,
    # all 256 CUs busy: the load pattern under which a hand-off bug would show), twice for determinism
    import scipy.sparse as sp
    host = sp.csr_matrix((g.adj.vals.cpu().numpy(), g.adj.indices.cpu().numpy(), g.adj.indptr.cpu().numpy()), shape=(N, N))
    want = host.astype(np.float64) @ x.cpu().numpy().astype(np.float64)
    assert rel_err(ax.cpu().numpy(), want) < 2e-6
    for _ in range(20):
        assert torch.equal(ops.spmm(g.adj, x), ax)
    # activity marks: only marked rows are written; unmarked columns are treated as zero
    stamp = torch.tensor([7], dtype=torch.int64, device="cuda")
    mark = torch.zeros(N, dtype=torch.int32, device="cuda")
    live = torch.randperm(N, device="cuda")[:6000]
    mark[live] = 7
    out = torch.full_like(x, -5.0)
    ops.spmm(g.adj, x, out=out, epilogue=ops.make_epilogue(row_mark=mark, mark_stamp=stamp))
    assert torch.equal(out[live], ax[live]) and bool((out[mark != 7] == -5.0).all())
    xz = torch.zeros_like(x); xz[live] = x[live]
    got = ops.spmm(g.adj, x, epilogue=ops.make_epilogue(col_mark=mark, mark_stamp=stamp))
    assert rel_err(got.cpu().numpy(), ops.spmm(g.adj, xz).cpu().numpy()) < 2e-6
    # symmetry of A_hat:  <y, A x> == <x, A y>
    l, r = (y.double() * ax.double()).sum().item(), (x.double() * ay.double()).sum().item()
    assert abs(l - r) / abs(l) < 1e-6
    # linearity
    axy = ops.spmm(g.adj, 2.0 * x - 0.5 * y)
    assert rel_err(axy.cpu().numpy(), (2.0 * ax - 0.5 * ay).cpu().numpy()) < 1e-5
    # A_hat (D^1/2 1) = D^1/2 1  for every non-isolated node
    # (node ids are the first-appearance ids Interaction assigned, not the generator's raw ids)
    deg = torch.from_numpy(np.bincount(np.concatenate([data.train_u, data.train_i.astype(np.int64) + U]),
                                       minlength=N).astype(np.float32)).cuda()
    v = deg.sqrt().unsqueeze(1).repeat(1, d).contiguous()
    assert rel_err(ops.spmm(g.adj, v).cpu().numpy(), v.cpu().numpy()) < 1e-5
    # top-K: sorted, unmasked, consistent with the scores it came from
    ue, ie = x[:U].contiguous(), x[U:].contiguous()
    q = torch.arange(0, 2048, dtype=torch.int32, device="cuda")
    ids, sc = ops.score_mask_topk(ue, q, ie, g.r_indptr, g.r_indices, 20)
    assert bool((sc[:, :-1] >= sc[:, 1:]).all()) and bool((sc > -1e8).all())
    chk = (ue[:2048].double().unsqueeze(1) * ie[ids.long()].double()).sum(-1)
    assert rel_err(sc.cpu().numpy(), chk.cpu().numpy()) < 1e-5
    rid, rsc = O.full_rank_topk_fast(ue.cpu().numpy(), ie.cpu().numpy(), np.arange(64), data.interaction_mat.tocsr(), 20)
    assert (ids[:64].cpu().numpy() == rid).mean() > 0.999
    # one full training step at full size keeps everything finite and moves every touched row
    tr = FusedTrainer(data, d, model="XSimGCL", n_layers=3, batch_size=2048, tau=0.2, use_graph=True)
    tr.sampler.seed(1)
    tr.begin_epoch()
    before = tr.E0.clone()
    for _ in range(3):
        tr.step()
    bpr, reg, cl = tr.read_losses()
    assert np.isfinite([bpr, reg, cl]).all() and 0.3 < bpr < 1.0 and cl > 0
    assert torch.isfinite(tr.E0).all() and (tr.E0 != before).float().mean() > 0.99


@pytest.mark.parametrize("name", ["XSimGCL", "LightGCN", "SimGCL", "SGL"])
def test_sharded_trainer_on_hip_backend_single_rank(golden_models, golden_meta, tiny_data, name):
    """The row-sharded trainer through RCCL ("nccl") with the real HIP kernels, world size 1 (one GPU box):
    same reference run as the fused engine.  World sizes 2 and 3 are covered on CPU (tests/test_dist_cpu.py)."""
    import os
    import torch.distributed as dist
    from selfrec_amd.dist import ShardedTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        gm, meta = golden_models, golden_meta
        m = meta[name]; c = m["conf"]
        gen = torch.Generator().manual_seed(m["noise_seed"])
        tr = ShardedTrainer(tiny_data, m["emb"], model=name, n_layers=int(c["n_layer"]), lr=m["lr"], reg=m["reg"],
                            cl_rate=float(c.get("lambda", 0.0)), eps=float(c.get("eps", 0.0)),
                            tau=float(c.get("tau", c.get("temp", 0.2))), layer_cl=int(c.get("l_star", 1)),
                            drop_rate=float(c.get("drop_rate", 0.1)), batch_size=m["batch"],
                            user_emb=gm[f"{name}_init_user"], item_emb=gm[f"{name}_init_item"],
                            noise_fn=lambda shape: torch.rand(shape, generator=gen))
        random.seed(m["sampler_seed"])
        tr.sampler.set_state_from_python()
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


@pytest.mark.selfcheck
def test_sharded_step_in_a_hipgraph_equals_the_single_gpu_step(golden_models, golden_meta, tiny_data, monkeypatch):
    """The sharded layout with its RCCL all-gathers captured in a hipGraph (opt-in, SRH_SHARDED_GRAPH=1),
    world size 1: same in-kernel RNG stream and batches as the unsharded eager step => same parameters."""
    import os
    import torch.distributed as dist
    from selfrec_amd.dist import ShardedTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        outs = []
        for sharded in (False, True):
            monkeypatch.setenv("SRH_SHARDED_GRAPH", "1")
            m = golden_meta["XSimGCL"]; c = m["conf"]
            kw = dict(model="XSimGCL", n_layers=int(c["n_layer"]), lr=m["lr"], reg=m["reg"], cl_rate=float(c["lambda"]),
                      eps=float(c["eps"]), tau=float(c["tau"]), layer_cl=int(c["l_star"]), batch_size=m["batch"],
                      user_emb=golden_models["XSimGCL_init_user"], item_emb=golden_models["XSimGCL_init_item"])
            tr = ShardedTrainer(tiny_data, m["emb"], use_graph=True, **kw) if sharded else \
                FusedTrainer(tiny_data, m["emb"], use_graph=False, **kw)
            assert tr.use_graph == sharded
            tr.sampler.seed(5)
            for _ in range(2):
                for _ in range(tr.begin_epoch()):
                    tr.step()
            torch.cuda.synchronize()
            outs.append((torch.cat([tr.user_emb, tr.item_emb]).cpu().numpy(), tr.read_losses()))
        assert np.isfinite(outs[0][0]).all()
        # (the sharded graph is stored without column classes: long rows are summed in a different order, and six
        # Adam steps amplify that where a gradient element is ~1e-8 itself -- see tests/test_gpu_shapes.py's docstring.
        # Observed: max |diff| 0.8e-6 .. 3.0e-6 on single elements, i.e. 0.3 % of ONE Adam step of lr = 1e-3; so the
        # bulk is held tight and the outliers to 2 % of a step)
        diff = np.abs(outs[1][0] - outs[0][0])
        assert (diff > 1e-6).mean() < 1e-3 and diff.max() < 2e-5
        np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=1e-5)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("name,d", [("XSimGCL", 128), ("SGL", 128), ("LightGCN", 32), ("LightGCN", 256), ("MF", 128),
                                    ("XSimGCL", 256), ("SimGCL", 200)])
def test_other_embedding_sizes_match_oracle(name, d):
    """d = 128 is BASELINE.json config 4's size (two rows per wave, D=128 InfoNCE tiles); d = 32 / 256 use the
    8- and 64-lane row shapes.  Three steps against the CPU oracle on the same batches and injected noise."""
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    torch.manual_seed(3)
    ue = torch.nn.init.xavier_uniform_(torch.empty(U, d)); ie = torch.nn.init.xavier_uniform_(torch.empty(I, d))
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    kw = dict(n_layers=2, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=2, drop_rate=0.1, batch_size=1500)
    tr = FusedTrainer(data, d, model=name, user_emb=ue, item_emb=ie, noise_fn=lambda s: torch.rand(s, generator=g1), **kw)
    ref = O.OracleTrainer(name, data.train_u, data.train_i, U, I, d, user_emb=ue, item_emb=ie,
                          noise_fn=lambda s: torch.rand(s, generator=g2), **kw)
    random.seed(17)
    tr.seed_sampler_from_python()
    nb = tr.begin_epoch()
    eu, ei, ej = tr.epoch_node_ids()
    if name == "SGL":                       # the oracle draws its two dropped views from the same stream
        random.seed(17)
        ref.resample_views()
    for b in range(min(nb, 3)):
        tr.step()
        got = tr.read_losses()
        lo, hi = b * 1500, min((b + 1) * 1500, len(eu))
        want = ref.step(eu[lo:hi].tolist(), ei[lo:hi].tolist(), ej[lo:hi].tolist())
        np.testing.assert_allclose(got, want, rtol=3e-5, atol=1e-9)
    # Gradients agree to ~3e-6 relative (|diff| ~ 1e-9).  Adam turns that into lr * g / (|g| + 1e-8): on the
    # handful of elements whose gradient is itself ~1e-9..1e-8 the first update differs by up to ~2e-5
    # (measured: 5 of 102,400 elements at d = 128), everywhere else by < 3e-6.
    got = np.concatenate([tr.user_emb.cpu().numpy(), tr.item_emb.cpu().numpy()])
    want = np.concatenate([ref.user_emb.detach().numpy(), ref.item_emb.detach().numpy()])
    diff = np.abs(got - want)
    assert diff.max() < 5e-5 and (diff > 3e-6).mean() < 1e-3
    assert rel_err(got, want) < 5e-4


@pytest.mark.parametrize("name,L,l_star", [("XSimGCL", 1, 0), ("XSimGCL", 1, 1), ("XSimGCL", 4, 0), ("XSimGCL", 4, 4),
                                           ("LightGCN", 1, 0), ("LightGCN", 4, 0), ("SimGCL", 1, 0), ("SimGCL", 4, 0),
                                           ("SGL", 1, 0), ("SGL", 4, 0)])
def test_layer_count_and_contrast_layer_edge_cases(name, L, l_star):
    """One layer (the only product also carries the mean; SimGCL / SGL cannot share a first layer), four layers,
    the contrast view at the ego table (l* = 0) and at the last layer (l* = L): two steps against the CPU oracle."""
    d = 64
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    torch.manual_seed(5)
    ue = torch.nn.init.xavier_uniform_(torch.empty(U, d)); ie = torch.nn.init.xavier_uniform_(torch.empty(I, d))
    g1, g2 = torch.Generator().manual_seed(4), torch.Generator().manual_seed(4)
    kw = dict(n_layers=L, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=l_star, drop_rate=0.1, batch_size=1500)
    tr = FusedTrainer(data, d, model=name, user_emb=ue, item_emb=ie, noise_fn=lambda s: torch.rand(s, generator=g1), **kw)
    ref = O.OracleTrainer(name, data.train_u, data.train_i, U, I, d, user_emb=ue, item_emb=ie,
                          noise_fn=lambda s: torch.rand(s, generator=g2), **kw)
    random.seed(23)
    tr.seed_sampler_from_python()
    tr.begin_epoch()
    eu, ei, ej = tr.epoch_node_ids()
    if name == "SGL":
        random.seed(23)
        ref.resample_views()
    for b in range(2):
        tr.step()
        lo, hi = b * 1500, (b + 1) * 1500
        want = ref.step(eu[lo:hi].tolist(), ei[lo:hi].tolist(), ej[lo:hi].tolist())
        np.testing.assert_allclose(tr.read_losses(), want, rtol=3e-5, atol=1e-9)
    got = np.concatenate([tr.user_emb.cpu().numpy(), tr.item_emb.cpu().numpy()])
    want = np.concatenate([ref.user_emb.detach().numpy(), ref.item_emb.detach().numpy()])
    diff = np.abs(got - want)
    assert diff.max() < 5e-5 and (diff > 3e-6).mean() < 1e-3
    fu, fi = tr.embeddings()
    ru, ri = ref.embeddings()
    assert rel_err(fu.cpu().numpy(), ru) < 1e-4 and rel_err(fi.cpu().numpy(), ri) < 1e-4
