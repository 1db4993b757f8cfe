"""Pin the CPU oracle (oracle/selfrec_oracle.py) to the outputs of the reference's own
Python, committed under tests/golden/ by tests/golden/make_golden.py."""
import random

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from .conftest import EDGE_TAGS

from oracle import selfrec_oracle as O


def _sampler(g):
    return O.PairwiseSampler(g["graph_train_u_ids"], g["graph_train_i_ids"], 200, 300)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_sampler_stream_bit_exact(golden_ops, tag):
    g = golden_ops
    bs, negs, seed = (int(x) for x in g[f"sampler_{tag}_meta"])
    s = _sampler(g)
    random.seed(seed)
    us, is_, js, sizes = [], [], [], []
    for _ in range(2):
        for u, i, j in s.epoch(bs, negs):
            us += u; is_ += i; js += j; sizes.append(len(u))
    assert np.array_equal(us, g[f"sampler_{tag}_u"])
    assert np.array_equal(is_, g[f"sampler_{tag}_i"])
    assert np.array_equal(js, g[f"sampler_{tag}_j"])
    assert np.array_equal(sizes, g[f"sampler_{tag}_sizes"])
    assert random.getrandbits(32) == int(g[f"sampler_{tag}_next_u32"][0])
    assert np.array_equal(s.edges_u[s.order], g[f"sampler_{tag}_final_order_u"])
    assert np.array_equal(s.edges_i[s.order], g[f"sampler_{tag}_final_order_i"])


def test_first_appearance_ids(golden_ops):
    g = golden_ops
    u, _ = O.first_appearance_ids(g["graph_train_u_raw"].tolist())
    i, _ = O.first_appearance_ids(g["graph_train_i_raw"].tolist())
    assert np.array_equal(u, g["graph_train_u_ids"]) and np.array_equal(i, g["graph_train_i_ids"])


def test_norm_adj(golden_ops):
    g = golden_ops
    adj = O.normalize_graph_mat(O.bipartite_adjacency(g["graph_train_u_ids"], g["graph_train_i_ids"], 200, 300)).tocsr()
    adj.sort_indices()
    assert np.array_equal(adj.indptr, g["norm_adj_indptr"])
    assert np.array_equal(adj.indices, g["norm_adj_indices"])
    assert np.array_equal(adj.data.astype(np.float32), g["norm_adj_data"])
    sq = sp.csr_matrix(np.array([[0, 1, 0, 2], [1, 0, 0, 0], [0, 0, 0, 0], [2, 0, 0, 0]], dtype=np.float32))
    with np.errstate(divide="ignore"):
        assert np.array_equal(O.normalize_graph_mat(sq).toarray(), g["norm_sq_dense"])
        rect = sp.csr_matrix(np.array([[1, 1, 0], [0, 0, 0]], dtype=np.float32))
        assert np.array_equal(O.normalize_graph_mat(rect).toarray(), g["norm_rect_dense"])


def test_edge_dropout(golden_ops):
    g = golden_ops
    r = O.interaction_matrix(g["graph_train_u_ids"], g["graph_train_i_ids"], 200, 300)
    random.seed(99)
    keep = O.edge_dropout_keep_idx(r.count_nonzero(), 0.1)
    assert np.array_equal(keep, g["edge_dropout_keep"])
    random.seed(99)
    lap = O.laplacian_of(O.edge_dropout(r, 0.1)).tocsr()
    lap.sort_indices()
    assert np.array_equal(lap.indptr, g["edge_dropout_lap_indptr"])
    assert np.array_equal(lap.indices, g["edge_dropout_lap_indices"])
    assert np.array_equal(lap.data.astype(np.float32), g["edge_dropout_lap_data"])
    assert random.getrandbits(32) == int(g["edge_dropout_next_u32"][0])


@pytest.mark.parametrize("n", [1, 2, 130, 515])
def test_losses_and_grads(golden_ops, n):
    g = golden_ops
    u, p, q = (torch.tensor(x, requires_grad=True) for x in g[f"ops_{n}_in"])
    bpr, reg, nce = O.bpr_loss(u, p, q), O.l2_reg_loss(1e-4, u, p, q), O.info_nce(u, p, 0.2)
    np.testing.assert_allclose([bpr.item(), reg.item(), nce.item()], g[f"ops_{n}_loss"], rtol=1e-6)
    gb = torch.stack(torch.autograd.grad(bpr, (u, p, q))).numpy()
    gr = torch.stack(torch.autograd.grad(reg, (u, p, q))).numpy()
    gn = torch.stack(torch.autograd.grad(nce, (u, p))).numpy()
    np.testing.assert_allclose(gb, g[f"ops_{n}_g_bpr"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(gr, g[f"ops_{n}_g_reg"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(gn, g[f"ops_{n}_g_nce"], rtol=1e-5, atol=1e-9)


def test_find_k_largest_ties(golden_ops):
    g = golden_ops
    ids, sc = O.find_k_largest(5, g["topk_ties_in"])
    assert np.array_equal(ids, g["topk_ties_ids"])
    assert np.array_equal(np.asarray(sc, dtype=np.float32), g["topk_ties_scores"])


MODELS = ["MF", "LightGCN", "XSimGCL", "SimGCL", "SGL"]


def edge_cl_reference(ge, m, tag):
    """the contrastive term per step as the trainer reports it: the reference logs the user and the item InfoNCE
    separately and unscaled for XSimGCL / SimGCL, their sum for SGL"""
    nce, lam = ge[f"{tag}_loss_nce"], float(m["conf"].get("lambda", 0.0))
    if m["model"] in ("XSimGCL", "SimGCL"):
        return nce.reshape(m["n_batches"], 2).sum(1) * lam
    return nce * lam if m["model"] == "SGL" else np.zeros(m["n_batches"])


def make_oracle_trainer(name, gm, meta, golden_ops):
    m = meta[name]
    c = m["conf"]
    gen = torch.Generator().manual_seed(m["noise_seed"])
    tr = O.OracleTrainer(
        m.get("model", name), golden_ops["graph_train_u_ids"], golden_ops["graph_train_i_ids"], 200, 300, m["emb"],
        n_layers=int(c.get("n_layer", 0)), lr=m["lr"], reg=m["reg"],
        cl_rate=float(c.get("lambda", 0.0)), eps=float(c.get("eps", 0.0)),
        tau=float(c.get("tau", c.get("temp", 0.2))), layer_cl=int(c.get("l_star", 1)),
        drop_rate=float(c.get("drop_rate", 0.0)), batch_size=m["batch"],
        user_emb=gm[f"{name}_init_user"], item_emb=gm[f"{name}_init_item"],
        noise_fn=lambda shape: torch.rand(shape, generator=gen))
    return tr


@pytest.mark.parametrize("name", MODELS)
def test_training_steps_match_reference(golden_models, golden_meta, golden_ops, name):
    gm, meta = golden_models, golden_meta
    tr = make_oracle_trainer(name, gm, meta, golden_ops)
    random.seed(meta[name]["sampler_seed"])
    if name == "SGL":
        tr.resample_views()
    sampler = O.PairwiseSampler(golden_ops["graph_train_u_ids"], golden_ops["graph_train_i_ids"], 200, 300)
    sizes = gm[f"{name}_batch_sizes"]
    off = 0
    bpr, cl = [], []
    for k, (u, i, j) in enumerate(sampler.epoch(meta[name]["batch"])):
        n = int(sizes[k])
        assert np.array_equal(u, gm[f"{name}_batch_u"][off:off + n])
        assert np.array_equal(i, gm[f"{name}_batch_i"][off:off + n])
        assert np.array_equal(j, gm[f"{name}_batch_j"][off:off + n])
        off += n
        r, _, c = tr.step(u, i, j)
        bpr.append(r); cl.append(c)
    np.testing.assert_allclose(bpr, gm[f"{name}_loss_bpr"], rtol=1e-5)
    np.testing.assert_allclose(tr.user_emb.detach().numpy(), gm[f"{name}_param_user"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(tr.item_emb.detach().numpy(), gm[f"{name}_param_item"], rtol=1e-4, atol=1e-7)
    fu, fi = tr.embeddings()
    np.testing.assert_allclose(fu, gm[f"{name}_final_user"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(fi, gm[f"{name}_final_item"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("tag", EDGE_TAGS)
def test_edge_configurations_match_reference(golden_edges, golden_ops, tag):
    """d = 256, a single layer, four layers, the contrast view at the ego table (l* = 0) and at the last layer:
    the reference's own runs of those configurations (tests/golden/make_golden.py edges)."""
    ge, meta = golden_edges
    m = meta[tag]
    tr = make_oracle_trainer(tag, ge, meta, golden_ops)
    random.seed(m["sampler_seed"])
    if m["model"] == "SGL":
        tr.resample_views()
    sampler = O.PairwiseSampler(golden_ops["graph_train_u_ids"], golden_ops["graph_train_i_ids"], 200, 300)
    off, bpr, cl = 0, [], []
    for k, (u, i, j) in enumerate(sampler.epoch(m["batch"])):
        n = int(ge[f"{tag}_batch_sizes"][k])
        assert np.array_equal(u, ge[f"{tag}_batch_u"][off:off + n]) and np.array_equal(j, ge[f"{tag}_batch_j"][off:off + n])
        off += n
        r, _, c = tr.step(u, i, j)
        bpr.append(r); cl.append(c)
    assert len(bpr) == m["n_batches"]
    np.testing.assert_allclose(bpr, ge[f"{tag}_loss_bpr"], rtol=1e-5)
    np.testing.assert_allclose(cl, edge_cl_reference(ge, m, tag), rtol=2e-5)
    np.testing.assert_allclose(tr.user_emb.detach().numpy(), ge[f"{tag}_param_user"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(tr.item_emb.detach().numpy(), ge[f"{tag}_param_item"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("name", MODELS)
def test_full_rank_eval_matches_reference(golden_models, golden_meta, golden_ops, name):
    gm = golden_models
    ue, ie = gm[f"{name}_final_user"], gm[f"{name}_final_item"]
    r = O.interaction_matrix(golden_ops["graph_train_u_ids"], golden_ops["graph_train_i_ids"], 200, 300)
    users = gm[f"{name}_test_users"]
    ids, sc = O.full_rank_topk(ue, ie, users, lambda u: r.indices[r.indptr[u]:r.indptr[u + 1]], 20)
    assert np.array_equal(ids, gm[f"{name}_rec_ids"])
    np.testing.assert_allclose(sc, gm[f"{name}_rec_scores"], rtol=1e-5)
    ids2, _ = O.full_rank_topk_fast(ue, ie, users, r, 20)
    assert np.array_equal(ids2, ids)
    # metrics strings: rebuild the dict form with int ids as names
    te_u, _ = O.first_appearance_ids(golden_ops["graph_train_u_raw"].tolist())
    umap = dict(zip(golden_ops["graph_train_u_raw"].tolist(), golden_ops["graph_train_u_ids"].tolist()))
    imap = dict(zip(golden_ops["graph_train_i_raw"].tolist(), golden_ops["graph_train_i_ids"].tolist()))
    origin = {}
    for a, b in zip(gm["test_u_ids_raw"].tolist(), gm["test_i_ids_raw"].tolist()):
        if a in umap and b in imap:
            origin.setdefault(umap[a], {})[imap[b]] = 1
    res = {int(u): [(int(i), float(s)) for i, s in zip(ids[k], sc[k])] for k, u in enumerate(users)}
    assert O.ranking_evaluation(origin, res, [10, 20]) == golden_meta[name]["measure"]
