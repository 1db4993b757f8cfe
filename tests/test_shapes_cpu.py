"""Host logic and the oracle, held to reference runs at the BASELINE.json config shapes
(tests/golden/shapes.npz, made by tests/golden/make_golden_shapes.py from /root/reference):

  D  the real douban-book interactions (configs[0]) through OUR loader / id mapper (python and native) and
     the C++ sampler: id maps, one epoch of (u, i, j) -- SHA-256 against the reference's own; the oracle's
     MF + BPR steps against the reference's;
  Y  Yelp2018 shape (configs[1], [2]): two epochs of the sampler stream; the oracle's XSimGCL / LightGCN steps;
  F  iFashion shape (configs[4]): both edge_dropout keep-sets of the epoch;
  N  node_dropout (augmentor.py:10-27): drop sets, dropped Laplacian, the oracle's SGL aug_type 0 steps;
  W  duplicated interaction lines: weight 2 in norm_adj, unit weights in dropped views.
No GPU: sampler and loader are host C++; the oracle is the checker being checked."""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import selfrec_oracle as O
from selfrec_amd import ops, synth
from selfrec_amd.data.loader import FileIO
from selfrec_amd.data.ui_graph import Interaction

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a, dtype):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=dtype)).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def shapes():
    return np.load(os.path.join(GOLDEN, "shapes.npz"))


@pytest.fixture(scope="module")
def smeta():
    with open(os.path.join(GOLDEN, "shapes_meta.json")) as f:
        return json.load(f)


def write_douban(tmp):
    z = np.load(os.path.join(GOLDEN, "douban_book.npz"))
    paths = []
    for name, sel in (("train.txt", ~z["is_test"]), ("test.txt", z["is_test"])):
        p = os.path.join(tmp, name)
        with open(p, "w") as f:
            f.writelines(f"{u} {i} {r}\n" for u, i, r in zip(z["user"][sel].tolist(), z["item"][sel].tolist(), z["rating"][sel].tolist()))
        paths.append(p)
    return paths


def sampler_stream_sha(data, seed, batch, epochs):
    smp = ops.Sampler(data.train_u, data.train_i, data.user_num, data.item_num)
    random.seed(seed)
    smp.set_state_from_python()
    hs = [hashlib.sha256() for _ in range(3)]
    n_batches = last = 0
    for _ in range(epochs):
        ep = smp.epoch(batch, 1)
        for h, k in zip(hs, "uij"):
            h.update(ep[k].astype(np.int32).tobytes())      # (concatenated batches == the epoch arrays)
        n_batches += ep["n_batches"]
        last = smp.n_edges - (ep["n_batches"] - 1) * batch
    return [h.hexdigest() for h in hs], n_batches, last, smp.next_u32()


@pytest.mark.parametrize("native", [False, True])
def test_douban_book_loader_ids_and_sampler_stream(tmp_path, smeta, native):
    """BASELINE.json configs[0] plumbing: FileIO -> Interaction -> next_batch_pairwise on the shipped file
    (loader.py:22-33, ui_graph.py:29-45, sampler.py:5-28), real string ids and ratings 4 / 5."""
    train_p, test_p = write_douban(str(tmp_path))
    load = FileIO.open_data_set if native else FileIO.load_data_set
    data = Interaction({}, load(train_p, "graph"), load(test_p, "graph"))
    m, g = smeta["D_sampler"], smeta["D_MF"]
    assert (data.user_num, data.item_num, len(data.training_data)) == (g["n_users"], g["n_items"], g["n_train"])
    assert sha([int(k) for k in data.user], np.int64) == m["user_ids_sha"]          # names in first-appearance id order
    assert sha([int(k) for k in data.item], np.int64) == m["item_ids_sha"]
    got, n_batches, last, nxt = sampler_stream_sha(data, m["seed"], m["batch"], m["epochs"])
    assert got == [m["sha_u"], m["sha_i"], m["sha_j"]]
    assert (n_batches, last, nxt) == (m["n_batches"], m["last_batch"], m["next_u32"])
    if not native:
        assert data.training_data[0][2] in (4.0, 5.0)          # ratings are parsed, and ignored by the graph (ui_graph.py:50)
        assert float(data.interaction_mat.max()) == 1.0


@pytest.fixture(scope="module")
def yelp_data():
    tu, ti, su, si, U, I = synth.make_dataset("yelp2018", seed=2024)
    return Interaction({}, synth.as_triples(tu, ti), synth.as_triples(su, si))


def test_yelp_shape_sampler_two_epochs(yelp_data, smeta):
    m = smeta["Y_sampler"]
    assert sha([int(k) for k in yelp_data.user], np.int64) == m["user_ids_sha"]
    got, n_batches, last, nxt = sampler_stream_sha(yelp_data, m["seed"], m["batch"], m["epochs"])
    assert got == [m["sha_u"], m["sha_i"], m["sha_j"]]
    assert (n_batches, last, nxt) == (m["n_batches"], m["last_batch"], m["next_u32"])


def seeded_init(info):
    """The reference's parameter init (e.g. XSimGCL.py:76-80) under torch.manual_seed(init_seed); the full tables
    are pinned by SHA-256."""
    torch.manual_seed(info["init_seed"])
    ue = torch.nn.init.xavier_uniform_(torch.empty(info["n_users"], info["emb"]))
    ie = torch.nn.init.xavier_uniform_(torch.empty(info["n_items"], info["emb"]))
    assert sha(ue.numpy(), np.float32) == info["init_sha_user"] and sha(ie.numpy(), np.float32) == info["init_sha_item"]
    return ue, ie


def oracle_for(info, data, **over):
    c = info["conf"]
    ue, ie = seeded_init(info)
    gen = torch.Generator().manual_seed(info["noise_seed"])
    kw = dict(n_layers=int(c.get("n_layer", 0)), lr=info["lr"], reg=info["reg"], cl_rate=float(c.get("lambda", 0.0)),
              eps=float(c.get("eps", 0.0)), tau=float(c.get("tau", c.get("temp", 0.2))), layer_cl=int(c.get("l_star", 1)),
              drop_rate=float(c.get("drop_rate", 0.1)), aug_type=int(c.get("aug_type", 1)), batch_size=info["batch"],
              user_emb=ue, item_emb=ie, noise_fn=lambda s: torch.rand(s, generator=gen))
    kw.update(over)
    return O.OracleTrainer(info["model"], data.train_u, data.train_i, data.user_num, data.item_num, info["emb"], **kw)


def oracle_for_tables(info, data, ue, ie):
    """oracle_for with explicit initial tables (goldens that store them whole)"""
    c = info["conf"]
    gen = torch.Generator().manual_seed(info["noise_seed"])
    return O.OracleTrainer(info["model"], data.train_u, data.train_i, data.user_num, data.item_num, info["emb"],
                           n_layers=int(c.get("n_layer", 0)), lr=info["lr"], reg=info["reg"], cl_rate=float(c.get("lambda", 0.0)),
                           eps=float(c.get("eps", 0.0)), tau=float(c.get("tau", c.get("temp", 0.2))), layer_cl=int(c.get("l_star", 1)),
                           drop_rate=float(c.get("drop_rate", 0.1)), aug_type=int(c.get("aug_type", 1)), batch_size=info["batch"],
                           user_emb=ue, item_emb=ie, noise_fn=lambda s: torch.rand(s, generator=gen))


def check_oracle_run(tag, shapes, info, ref, rows=True):
    sizes = shapes[f"{tag}_batch_sizes"]
    off = np.concatenate([[0], np.cumsum(sizes)])
    got = []
    for b in range(len(sizes)):
        sl = slice(off[b], off[b + 1])
        got.append(ref.step(shapes[f"{tag}_batch_u"][sl].tolist(), shapes[f"{tag}_batch_i"][sl].tolist(),
                            shapes[f"{tag}_batch_j"][sl].tolist()))
        if b == 0 and f"{tag}_pre_grad_user" in shapes:
            # round 3: the gradient that enters Adam in the first step (embedding_dict[*].grad of the reference run) --
            # the oracle restates the reference's torch expressions, so it is held to 1e-5 of the block's largest element
            for key, param in (("user", ref.user_emb), ("item", ref.item_emb)):
                want = shapes[f"{tag}_pre_grad_{key}"].astype(np.float64)
                g = param.grad.numpy()[shapes[f"{tag}_pre_rows_{key}"].astype(np.int64)].astype(np.float64)
                assert np.abs(g - want).max() <= 1e-5 * np.abs(want).max(), (tag, key)
    got = np.asarray(got)
    np.testing.assert_allclose(got[:, 0], shapes[f"{tag}_loss_bpr"], rtol=1e-5)
    np.testing.assert_allclose(got[:, 1], shapes[f"{tag}_loss_reg"] / (info["batch"] if info["model"] in ("MF", "LightGCN") else 1.0),
                               rtol=1e-5)
    nce = shapes[f"{tag}_loss_nce"]
    if nce.size:
        per_step = nce.reshape(len(sizes), -1).sum(1) * ref.cl_rate
        np.testing.assert_allclose(got[:, 2], per_step, rtol=1e-5)
    ru = shapes[f"{tag}_rows_user"] if rows else slice(None)
    ri = shapes[f"{tag}_rows_item"] if rows else slice(None)
    np.testing.assert_allclose(ref.user_emb.detach().numpy()[ru], shapes[f"{tag}_param_user"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(ref.item_emb.detach().numpy()[ri], shapes[f"{tag}_param_item"], rtol=1e-4, atol=2e-6)
    fu, fi = ref.embeddings()
    np.testing.assert_allclose(fu[ru], shapes[f"{tag}_final_user"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(fi[ri], shapes[f"{tag}_final_item"], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("tag", ["Y_XSimGCL", "Y_LightGCN", "Y_SimGCL"])
def test_oracle_matches_reference_at_yelp_shape(yelp_data, shapes, smeta, tag):
    info = smeta[tag]
    check_oracle_run(tag, shapes, info, oracle_for(info, yelp_data))


@pytest.mark.parametrize("tag", ["E_XSimGCL50", "E_XSimGCL96", "E_SGL96", "E_LightGCN20", "E_MF50"])
def test_oracle_matches_reference_at_odd_embedding_sizes(tiny_data, shapes, smeta, tag):
    """embedding.size = 50 / 96 / 20 (goldens section E): the oracle at the real width -- what the zero-padded engine is
    compared with on the GPU."""
    info = smeta[tag]
    torch.manual_seed(info["init_seed"])
    ref = oracle_for_tables(info, tiny_data, shapes[f"{tag}_init_user"], shapes[f"{tag}_init_item"])
    if info["model"] == "SGL":
        random.seed(info["sampler_seed"])
        ref.resample_views()
    check_oracle_run(tag, shapes, info, ref, rows=False)


def test_oracle_matches_reference_mf_on_douban_book(tmp_path, shapes, smeta):
    train_p, test_p = write_douban(str(tmp_path))
    data = Interaction({}, FileIO.load_data_set(train_p, "graph"), FileIO.load_data_set(test_p, "graph"))
    info = smeta["D_MF"]
    ref = oracle_for(info, data)
    check_oracle_run("D_MF", shapes, info, ref)
    # test() + ranking_evaluation of the reference after those 3 steps (graph_recommender.py:38-58, evaluation.py:135-162)
    users = shapes["D_MF_rec_users"]
    ids, _ = O.full_rank_topk_fast(ref.user_emb.detach().numpy(), ref.item_emb.detach().numpy(), users,
                                   data.interaction_mat.tocsr(), 20)
    assert (ids == shapes["D_MF_rec_ids"]).mean() > 0.999


def test_ifashion_shape_edge_dropout_keep_sets(smeta):
    """SGL.py:28-29 at BASELINE.json configs[4]'s shape: the epoch's two keep-sets from the C++ replay of
    random.sample (pool path, k > n/3) against the reference's, order included."""
    info = smeta["F_SGL"]
    smp = ops.Sampler(np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), 1, 1)
    random.seed(info["sampler_seed"])
    smp.set_state_from_python()
    for want, want_sorted in zip(info["keep_sha"], info["keep_sorted_sha"]):
        keep = smp.sample_range(info["n_edges"], info["n_keep"])
        assert sha(keep, np.int64) == want and sha(np.sort(keep), np.int64) == want_sorted


def test_node_dropout_matches_reference(golden_ops, shapes, fresh_tiny_data):
    from selfrec_amd.data.augmentor import GraphAugmentor
    data = fresh_tiny_data
    random.seed(77)
    dropped = GraphAugmentor.node_dropout(data.interaction_mat, 0.1)
    assert random.getrandbits(32) == int(shapes["node_dropout_next_u32"][0])
    r = data.interaction_mat.tocsr(); r.sort_indices()
    rows = np.repeat(np.arange(r.shape[0]), np.diff(r.indptr))
    want_keep = ~np.isin(rows, shapes["node_dropout_users"]) & ~np.isin(r.indices, shapes["node_dropout_items"])
    assert np.array_equal(dropped.keep_mask.astype(bool), want_keep)
    lap = data.convert_to_laplacian_mat(dropped.to_scipy(data)).tocsr()      # host (scipy) path of ui_graph.py:58-65
    lap.sort_indices(); lap.eliminate_zeros()
    assert np.array_equal(lap.indptr, shapes["node_dropout_lap_indptr"])
    assert np.array_equal(lap.indices, shapes["node_dropout_lap_indices"])
    assert np.array_equal(lap.data.astype(np.float32), shapes["node_dropout_lap_data"])
    # the oracle's restatement, same stream
    random.seed(77)
    olap = O.laplacian_of(O.node_dropout(r, 0.1)).tocsr(); olap.sort_indices()
    assert np.array_equal(olap.data.astype(np.float32), shapes["node_dropout_lap_data"])


def test_oracle_sgl_node_dropout_steps(shapes, smeta, tiny_data):
    info = smeta["N_SGL"]
    gen_unused = None  # noqa: F841  (SGL draws no perturbation noise)
    c = info["conf"]
    ref = O.OracleTrainer("SGL", tiny_data.train_u, tiny_data.train_i, tiny_data.user_num, tiny_data.item_num, info["emb"],
                          n_layers=int(c["n_layer"]), lr=info["lr"], reg=info["reg"], cl_rate=float(c["lambda"]),
                          tau=float(c["temp"]), drop_rate=float(c["drop_rate"]), aug_type=0, batch_size=info["batch"],
                          user_emb=shapes["N_SGL_init_user"], item_emb=shapes["N_SGL_init_item"])
    random.seed(info["sampler_seed"])
    ref.resample_views()
    check_oracle_run("N_SGL", shapes, info, ref, rows=False)


def test_duplicate_interactions_weights(shapes):
    """a duplicated line weighs 2 in norm_adj (scipy sums it, ui_graph.py:47-56) and 1 in a dropped view
    (augmentor.py:36-39 rebuilds with ones) -- the oracle's restatement against the reference's matrices."""
    tu, ti = shapes["dup_train_u"], shapes["dup_train_i"]
    data = Interaction({}, synth.as_triples(tu, ti), [])
    na = data.norm_adj.tocsr(); na.sort_indices()
    assert np.array_equal(na.indptr, shapes["dup_norm_adj_indptr"]) and np.array_equal(na.indices, shapes["dup_norm_adj_indices"])
    assert np.array_equal(na.data.astype(np.float32), shapes["dup_norm_adj_data"])
    random.seed(5)
    r = data.interaction_mat.tocsr(); r.sort_indices()
    lap = O.laplacian_of(O.edge_dropout(r, 0.1)).tocsr(); lap.sort_indices()
    assert np.array_equal(lap.indices, shapes["dup_drop_lap_indices"])
    assert np.array_equal(lap.data.astype(np.float32), shapes["dup_drop_lap_data"])


def test_sampler_with_64_negatives_per_pair(yelp_data, smeta):
    """MixGCF's regime (MixGCF.py:24,96-114): next_batch_pairwise(data, 2048, n_negs=64) at the Yelp2018 shape, first 20
    batches bit-exact against the reference's generator; the C++ replay's rate is printed (131 k draws per batch)."""
    import time
    m = smeta["M_sampler_negs64"]
    smp = ops.Sampler(yelp_data.train_u, yelp_data.train_i, yelp_data.user_num, yelp_data.item_num)
    random.seed(m["seed"])
    smp.set_state_from_python()
    smp.shuffle()
    hs = [hashlib.sha256() for _ in range(3)]
    t0 = time.perf_counter()
    for k in range(m["batches"]):
        u, i, j = smp.next_batch(k * m["batch"], m["batch"], m["n_negs"])
        assert j.size == m["batch"] * m["n_negs"]
        for h, a in zip(hs, (u, i, j)):
            h.update(a.astype(np.int32).tobytes())
    dt = time.perf_counter() - t0
    assert [h.hexdigest() for h in hs] == [m["sha_u"], m["sha_i"], m["sha_j"]]
    print(f"n_negs=64: {m['batches'] * m['batch'] * m['n_negs'] / dt / 1e6:.1f} M negatives/s on one host core")
