#!/usr/bin/env python3
"""Reference-run goldens at the BASELINE.json config shapes ("What's weak" #1).

make_golden.py pins every op and every model on a 200 x 300 graph; this script runs the
REFERENCE'S OWN Python (/root/reference, same shims) at the shapes the configs name and
keeps what is small enough to commit -- losses, a few thousand sampled rows, SHA-256 of
index streams and keep-sets:

  Y  yelp2018 shape (31,668 x 38,048, seed 2024 = bench.py's graph): two-epoch sampler stream
     hashes; 2 training steps of XSimGCL L=3 (injected noise) and LightGCN L=3 -> per-step
     losses, 1,024 sampled rows (512 users + 512 items) of the parameters and of the final embeddings, and the
     reference's test() ranking (predict -> mask -> find_k_largest) for 64 test users.
  F  iFashion shape (300,000 x 81,614): SGL (conf/SGL.yaml: L=2, rho=0.1, aug_type=1, temp=0.2,
     lambda=0.1), the epoch's two edge_dropout keep-set hashes and 1 training step.
  D  the REAL dataset/douban-book/test.txt (10,882 x 19,075, ratings 4/5), split 80/20 by a seeded
     rule (BASELINE.json configs[0]; the train file is missing from the snapshot): id-map hashes,
     epoch-1 sampler stream hash, 3 MF + BPR steps, test() + ranking_evaluation strings.  The
     interactions themselves are committed as tests/golden/douban_book.npz (golden INPUT vectors:
     int32 ids + int8 rating + split flag) because /root/reference does not exist on the GPU box.
  N  GraphAugmentor.node_dropout (augmentor.py:10-27) on the 200 x 300 graph: dropped users /
     items, dropped Laplacian, next raw MT word; 3 SGL steps with aug_type = 0.
  W  duplicated interaction lines: weight-2 entries in norm_adj, unit weights in the dropped views.
  B  configs[3]: XSimGCL L=3, d = 128 on the synthetic 1 M x 500 k graph (40.3 M train interactions): one step.
  E  embedding.size = 50 / 96 / 20 (widths the kernels serve only as zero-padded rows): XSimGCL, SGL, LightGCN, MF on the
     200 x 300 graph, 2 steps each.
  M  the other torch graph models of SURVEY 8(f-4) -- DirectAU, MixGCF, BUIR, SelfCF -- on the 200 x 300 graph: 2 steps
     each (losses, parameters, get_embedding / model() outputs, test() ranking), MixGCF's n_negs = 64 sampler stream
     (SHA-256; also 20 batches at the Yelp2018 shape), BUIR's sparse-dropout keep masks.  Randomness the models draw on
     the DEVICE in the reference (nn.Dropout, rand_like) is drawn from seeded CPU generators by patching
     torch.nn.functional.dropout / torch.rand_like in this process; what they draw on the host (torch.rand(..).cuda(),
     torch.randn(..).cuda(), np.random.random()) needs only the seeds.

Run:  python tests/golden/make_golden_shapes.py [Y] [F] [D] [N] [W]      (default: all)
Writes shapes.npz / shapes_meta.json / douban_book.npz next to this file.
"""
import hashlib
import importlib
import json
import os
import random
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the numba stub / .cuda() shims, puts /root/reference on sys.path)

import torch  # noqa: E402
from data.augmentor import GraphAugmentor  # noqa: E402
from data.loader import FileIO  # noqa: E402
from data.ui_graph import Interaction  # noqa: E402
from util import loss_torch as ref_loss  # noqa: E402
from util import sampler as ref_sampler  # noqa: E402
from util.algorithm import find_k_largest  # noqa: E402
from util.evaluation import ranking_evaluation  # noqa: E402

from selfrec_amd import synth  # noqa: E402

SEED_GRAPH = 2024
N_PRE_RANDOM, N_PRE_BATCH = 256, 128     # rows per side whose first-step layer outputs / pre-Adam gradients are kept
N_ROWS = 512             # sampled users and sampled items whose rows are kept (the full initial tables are pinned by SHA-256)


def sha(a, dtype):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=dtype)).tobytes()).hexdigest()


EMB = {"default": 64}          # (section B overrides it: BASELINE.json configs[3] is d = 128)


def make_conf(tmp, model, extra, emb=None, batch=2048, topn="[10,20]"):
    emb = EMB["default"] if emb is None else emb
    lines = ["training.set: ./train.txt", "test.set: ./test.txt", "model:", f"  name: {model}", "  type: graph",
             f"item.ranking.topN: {topn}", f"embedding.size: {emb}", "max.epoch: 1", f"batch.size: {batch}",
             "learning.rate: 0.001", "reg.lambda: 0.0001", "output: ./results/"]
    if extra:
        lines.append(f"{model}:")
        lines += [f"  {k}: {v}" for k, v in extra.items()]
    path = os.path.join(tmp, f"{model}.yaml")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return MG.ModelConf(path)


def run_model(name, extra, train, test, n_steps, tag, out, meta, *, seeds=(31, 2718, 4242), eval_users=0,
              full_rank=False, sample_rows=True, capture=0):
    """n_steps reference training steps of `name`; everything recorded under the prefix `tag`.

    capture = K > 0 (VERDICT r02 "next" #2): also keep, from the FIRST step, sampled rows of the first K sparse-product
    outputs as the model appends them to its layer list (XSimGCL.py:88-92 / SimGCL.py:86-90 perturb the product's output
    in place, so the kept tensor IS the layer output; LightGCN.py:72-73 / SGL.py:106-107 append it as it is) and of
    embedding_dict[*].grad after the first backward(), BEFORE Adam -- quantities that summation order moves by 1e-7,
    not by the fraction of an Adam step that 1 / (sqrt(v) + 1e-8) makes of it."""
    init_seed, sampler_seed, noise_seed = seeds
    mod = importlib.import_module(f"model.graph.{name}")
    rec = {"batches": [], "bpr": [], "reg": [], "nce": []}

    def wrap(fn, key):
        def inner(*a, **k):
            r = fn(*a, **k)
            rec[key].append(float(r))
            return r
        return inner

    def batches(data, bs, n_negs=1):
        for k, b in enumerate(ref_sampler.next_batch_pairwise(data, bs, n_negs)):
            if k == n_steps:
                return
            rec["batches"].append([list(x) for x in b])
            yield b

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            conf = make_conf(tmp, name, dict(extra))
            mod.next_batch_pairwise = batches
            mod.bpr_loss = wrap(ref_loss.bpr_loss, "bpr")
            mod.l2_reg_loss = wrap(ref_loss.l2_reg_loss, "reg")
            if hasattr(mod, "InfoNCE"):
                mod.InfoNCE = wrap(ref_loss.InfoNCE, "nce")
            noise_gen = torch.Generator().manual_seed(noise_seed)
            torch.rand_like = lambda t, **k: torch.rand(t.shape, generator=noise_gen)
            kept, grads = [], {}
            real_mm, real_adam_step = torch.sparse.mm, torch.optim.Adam.step
            if capture:
                def mm(a, b):
                    y = real_mm(a, b)
                    if len(kept) < capture:
                        kept.append(y)
                    return y

                def adam_step(self, *a, **k):
                    if not grads:
                        for group in self.param_groups:
                            for q in group["params"]:
                                grads[tuple(q.shape)] = q.grad.detach().clone()
                    return real_adam_step(self, *a, **k)
                torch.sparse.mm, torch.optim.Adam.step = mm, adam_step
            torch.manual_seed(init_seed)
            random.seed(sampler_seed)
            t0 = time.time()
            model = getattr(mod, name)(conf, train, test)
            model.fast_evaluation = lambda epoch: None          # (the per-epoch evaluation is pinned separately)
            params = model.model.embedding_dict
            init_u = params["user_emb"].detach().numpy().copy()
            init_i = params["item_emb"].detach().numpy().copy()
            state0 = random.getstate()
            t1 = time.time()
            try:
                model.train()
            except AttributeError as e:                          # best_user_emb is only set by fast_evaluation
                assert "best_user_emb" in str(e), e
            finally:
                torch.sparse.mm, torch.optim.Adam.step = real_mm, real_adam_step
            with torch.no_grad():
                model.user_emb, model.item_emb = model.model()
            t2 = time.time()
            d = model.data
            U, I = d.user_num, d.item_num
            rng = np.random.default_rng(99)
            ru = np.sort(rng.choice(U, size=min(N_ROWS, U), replace=False)).astype(np.int32)
            ri = np.sort(rng.choice(I, size=min(N_ROWS, I), replace=False)).astype(np.int32)
            pu, pi = params["user_emb"].detach().numpy(), params["item_emb"].detach().numpy()
            fu, fi = model.user_emb.detach().numpy(), model.item_emb.detach().numpy()
            if sample_rows:
                out[f"{tag}_rows_user"], out[f"{tag}_rows_item"] = ru, ri
                sel_u, sel_i = ru, ri
            else:
                sel_u, sel_i = slice(None), slice(None)
            if not sample_rows:      # (sampled runs re-create the initial tables from init_seed and check init_sha_*)
                out[f"{tag}_init_user"], out[f"{tag}_init_item"] = init_u, init_i
            out[f"{tag}_param_user"], out[f"{tag}_param_item"] = pu[sel_u].copy(), pi[sel_i].copy()
            out[f"{tag}_final_user"], out[f"{tag}_final_item"] = fu[sel_u].copy(), fi[sel_i].copy()
            out[f"{tag}_batch_sizes"] = np.asarray([len(b[0]) for b in rec["batches"]], dtype=np.int32)
            for k, col in enumerate("uij"):
                out[f"{tag}_batch_{col}"] = np.concatenate([b[k] for b in rec["batches"]]).astype(np.int32)
            out[f"{tag}_loss_bpr"] = np.asarray(rec["bpr"], dtype=np.float64)
            out[f"{tag}_loss_reg"] = np.asarray(rec["reg"], dtype=np.float64)
            out[f"{tag}_loss_nce"] = np.asarray(rec["nce"], dtype=np.float64)
            info = {"model": name, "conf": extra, "n_steps": n_steps, "emb": EMB["default"], "batch": 2048, "lr": 0.001, "reg": 0.0001,
                    "init_seed": init_seed, "sampler_seed": sampler_seed, "noise_seed": noise_seed,
                    "n_users": U, "n_items": I, "n_train": len(d.training_data),
                    "init_sha_user": sha(init_u, np.float32), "init_sha_item": sha(init_i, np.float32),
                    "build_s": round(t1 - t0, 1), "train_s": round(t2 - t1, 1)}
            if capture:
                assert len(kept) == capture and len(grads) == 2, (len(kept), list(grads))
                # rows: the random sample + the first 256 distinct users / items of batch 1 (where the gradient lives)
                # (256 random + 128 batch rows per side keep shapes.npz small; the sampled tables above use all 512)
                bu = np.unique(np.asarray(rec["batches"][0][0], dtype=np.int64))[:N_PRE_BATCH]
                bi = np.unique(np.asarray(rec["batches"][0][1] + rec["batches"][0][2], dtype=np.int64))[:N_PRE_BATCH]
                g_rows_u = np.concatenate([ru[:N_PRE_RANDOM], bu]).astype(np.int32)
                g_rows_i = np.concatenate([ri[:N_PRE_RANDOM], bi]).astype(np.int32)
                out[f"{tag}_pre_rows_user"], out[f"{tag}_pre_rows_item"] = g_rows_u, g_rows_i
                out[f"{tag}_pre_n_random"] = np.asarray([len(ru[:N_PRE_RANDOM]), len(ri[:N_PRE_RANDOM])], dtype=np.int32)
                for k, y in enumerate(kept):
                    y = y.detach().numpy()
                    out[f"{tag}_pre_layer{k}_user"] = y[:U][g_rows_u].copy()
                    out[f"{tag}_pre_layer{k}_item"] = y[U:][g_rows_i].copy()
                gu, gi = grads[(U, pu.shape[1])].numpy(), grads[(I, pi.shape[1])].numpy()
                out[f"{tag}_pre_grad_user"], out[f"{tag}_pre_grad_item"] = gu[g_rows_u].copy(), gi[g_rows_i].copy()
                info["pre_adam"] = {"captured_products": capture, "grad_absmax_user": float(np.abs(gu).max()),
                                    "grad_absmax_item": float(np.abs(gi).max()),
                                    "grad_l2_user": float(np.sqrt((gu.astype(np.float64) ** 2).sum())),
                                    "grad_l2_item": float(np.sqrt((gi.astype(np.float64) ** 2).sum()))}
                del kept[:]
                grads.clear()
            if eval_users:
                # graph_recommender.py:46-53 for a sample of the test users: predict, mask, heap top-K
                users = list(d.test_set.keys())
                pick = [users[k] for k in np.sort(rng.choice(len(users), size=min(eval_users, len(users)), replace=False))]
                ids, scs = [], []
                for u in pick:
                    cand = model.predict(u)
                    rated, _ = d.user_rated(u)
                    for it in rated:
                        cand[d.item[it]] = -10e8
                    i_, s_ = find_k_largest(20, cand)
                    ids.append(i_); scs.append(s_)
                out[f"{tag}_eval_users"] = np.asarray([d.user[u] for u in pick], dtype=np.int32)
                out[f"{tag}_eval_ids"] = np.asarray(ids, dtype=np.int32)
                out[f"{tag}_eval_scores"] = np.asarray(scs, dtype=np.float32)
            if full_rank:
                rec_list = model.test()
                info["measure"] = ranking_evaluation(d.test_set, rec_list, [10, 20])
                users = list(d.test_set.keys())
                out[f"{tag}_test_users"] = np.asarray([d.user[u] for u in users], dtype=np.int32)
                pick = np.sort(rng.choice(len(users), size=min(256, len(users)), replace=False))
                out[f"{tag}_rec_users"] = out[f"{tag}_test_users"][pick]
                out[f"{tag}_rec_ids"] = np.asarray([[d.item[it] for it, _ in rec_list[users[k]]] for k in pick], dtype=np.int32)
                out[f"{tag}_rec_scores"] = np.asarray([[s for _, s in rec_list[users[k]]] for k in pick], dtype=np.float32)
            meta[tag] = info
            print(tag, info["build_s"], "s build,", info["train_s"], "s train; bpr", rec["bpr"], "nce", rec["nce"][:4], flush=True)
            return model, state0
        finally:
            os.chdir(cwd)


def stream_hashes(train, batch, epochs, seed, out_meta, tag):
    data = Interaction({"dummy": 1}, [list(t) for t in train], [])
    random.seed(seed)
    hu, hi, hj = hashlib.sha256(), hashlib.sha256(), hashlib.sha256()
    sizes = []
    for _ in range(epochs):
        for u, i, j in ref_sampler.next_batch_pairwise(data, batch, 1):
            hu.update(np.asarray(u, dtype=np.int32).tobytes())
            hi.update(np.asarray(i, dtype=np.int32).tobytes())
            hj.update(np.asarray(j, dtype=np.int32).tobytes())
            sizes.append(len(u))
    out_meta[tag] = {"seed": seed, "batch": batch, "epochs": epochs, "n_batches": len(sizes), "last_batch": sizes[-1],
                     "sha_u": hu.hexdigest(), "sha_i": hi.hexdigest(), "sha_j": hj.hexdigest(),
                     "next_u32": random.getrandbits(32),
                     "user_ids_sha": sha([int(k) for k in data.user], np.int64),      # names in id order
                     "item_ids_sha": sha([int(k) for k in data.item], np.int64)}
    print(tag, out_meta[tag], flush=True)


def section_Y(out, meta):
    tu, ti, su, si, U, I = synth.make_dataset("yelp2018", seed=SEED_GRAPH)
    train, test = synth.as_triples(tu, ti), synth.as_triples(su, si)
    meta["Y_graph"] = {"shape": "yelp2018", "seed": SEED_GRAPH, "n_users": U, "n_items": I, "n_train": len(tu), "n_test": len(su)}
    stream_hashes(train, 2048, 2, 20240924, meta, "Y_sampler")
    run_model("XSimGCL", {"n_layer": 3, "l_star": 1, "lambda": 0.2, "eps": 0.2, "tau": 0.2},
              [list(t) for t in train], [list(t) for t in test], 2, "Y_XSimGCL", out, meta, eval_users=64, capture=3)
    run_model("LightGCN", {"n_layer": 3}, [list(t) for t in train], [list(t) for t in test], 2, "Y_LightGCN", out, meta,
              eval_users=64, capture=3)
    # SimGCL.py:21-50: one clean pass + two perturbed views per step (tau = 0.2 is hard-coded there)
    run_model("SimGCL", {"n_layer": 3, "lambda": 0.5, "eps": 0.1}, [list(t) for t in train], [list(t) for t in test], 2,
              "Y_SimGCL", out, meta, capture=9)


def section_F(out, meta):
    tu, ti, su, si, U, I = synth.make_dataset("ifashion", seed=SEED_GRAPH)
    train, test = synth.as_triples(tu, ti), synth.as_triples(su, si)
    meta["F_graph"] = {"shape": "ifashion", "seed": SEED_GRAPH, "n_users": U, "n_items": I, "n_train": len(tu), "n_test": len(su)}
    extra = {"n_layer": 2, "lambda": 0.1, "drop_rate": 0.1, "aug_type": 1, "temp": 0.2}
    model, state0 = run_model("SGL", extra, [list(t) for t in train], [list(t) for t in test], 1, "F_SGL", out, meta, capture=6)
    # the epoch's two keep-sets, replayed from the state train() started with (SGL.py:28-29, augmentor.py:34)
    random.setstate(state0)
    e = model.data.interaction_mat.count_nonzero()
    k = int(e * (1 - 0.1))
    keeps = [random.sample(range(e), k) for _ in range(2)]
    meta["F_SGL"].update(n_edges=int(e), n_keep=k, keep_sha=[sha(x, np.int64) for x in keeps],
                         keep_sorted_sha=[sha(np.sort(np.asarray(x, dtype=np.int64)), np.int64) for x in keeps])
    # BASELINE.json configs[4] and the README's model matrix run SGL with THREE layers (the shipped yaml has 2)
    extra3 = dict(extra, n_layer=3)
    run_model("SGL", extra3, [list(t) for t in train], [list(t) for t in test], 1, "F_SGL3", out, meta, capture=9)
    meta["F_SGL3"].update(n_edges=int(e), n_keep=k)


def section_B(out, meta):
    """BASELINE.json configs[3]: XSimGCL on the synthetic 1 M x 500 k graph, d = 128 -- ONE reference step (the reference
    needs ~25 GB and a quarter of an hour for it: python triples, dict-of-dict sets, a 40 M element shuffle)."""
    torch.set_num_threads(8)        # (summation order inside torch's CPU kernels then differs from a 1-thread run at the
    EMB["default"] = 128            #  1e-7 level; tests/test_gpu_shapes.py's tolerances for post-Adam parameters apply)
    try:
        t0 = time.time()
        tu, ti, su, si, U, I = synth.make_dataset("1m-500k", seed=SEED_GRAPH)
        train, test = synth.as_triples(tu, ti), synth.as_triples(su[:200000], si[:200000])
        print(f"B: graph + triples in {time.time() - t0:.0f} s", flush=True)
        meta["B_graph"] = {"shape": "1m-500k", "seed": SEED_GRAPH, "n_users": U, "n_items": I, "n_train": len(tu),
                           "n_test_used": 200000}
        run_model("XSimGCL", {"n_layer": 3, "l_star": 1, "lambda": 0.2, "eps": 0.2, "tau": 0.2}, train, test, 1, "B_XSimGCL",
                  out, meta, eval_users=16, capture=3)
    finally:
        torch.set_num_threads(1)
        EMB["default"] = 64


def section_D(out, meta):
    path = os.path.join(MG.REF, "dataset", "douban-book", "test.txt")
    rows = FileIO.load_data_set(path, "graph")
    u = np.asarray([int(r[0]) for r in rows], dtype=np.int32)
    i = np.asarray([int(r[1]) for r in rows], dtype=np.int32)
    w = np.asarray([r[2] for r in rows])
    assert all(str(a) == r[0] and str(b) == r[1] for a, b, r in zip(u.tolist(), i.tolist(), rows)), "non-canonical ids"
    assert np.array_equal(w, w.astype(np.int8)) and w.min() >= 0
    is_test = np.random.default_rng(7).random(len(rows)) < 0.2
    np.savez_compressed(os.path.join(HERE, "douban_book.npz"), user=u, item=i, rating=w.astype(np.int8), is_test=is_test)
    train = [rows[k] for k in np.flatnonzero(~is_test)]
    test = [rows[k] for k in np.flatnonzero(is_test)]
    meta["D_graph"] = {"source": "dataset/douban-book/test.txt", "lines": len(rows), "split": "default_rng(7).random(n) < 0.2 -> test",
                       "n_train": len(train), "n_test": len(test), "ratings": sorted(set(w.astype(int).tolist()))}
    stream_hashes(train, 2048, 1, 424242, meta, "D_sampler")
    run_model("MF", {}, [list(t) for t in train], [list(t) for t in test], 3, "D_MF", out, meta, full_rank=True)


def section_N(out, meta):
    tu, ti, su, si = MG.tiny_graph()
    train, test = synth.as_triples(tu, ti), synth.as_triples(su, si)
    data = Interaction({"dummy": 1}, [list(t) for t in train], [])
    random.seed(77)
    st = random.getstate()
    U, I = data.interaction_mat.shape
    out["node_dropout_users"] = np.asarray(random.sample(range(U), int(U * 0.1)), dtype=np.int32)
    out["node_dropout_items"] = np.asarray(random.sample(range(I), int(I * 0.1)), dtype=np.int32)
    random.setstate(st)
    dropped = GraphAugmentor.node_dropout(data.interaction_mat, 0.1)
    out["node_dropout_next_u32"] = np.asarray([random.getrandbits(32)], dtype=np.uint32)
    lap = data.convert_to_laplacian_mat(dropped).tocsr()
    lap.sort_indices()
    out["node_dropout_lap_indptr"] = lap.indptr.astype(np.int32)
    out["node_dropout_lap_indices"] = lap.indices.astype(np.int32)
    out["node_dropout_lap_data"] = lap.data.astype(np.float32)
    meta["N_node_dropout"] = {"seed": 77, "rate": 0.1, "explicit_zeros_in_product": int((dropped.data == 0).sum()),
                              "nnz_product": int(dropped.nnz), "nnz_laplacian": int(lap.nnz)}
    extra = {"n_layer": 2, "lambda": 0.1, "drop_rate": 0.1, "aug_type": 0, "temp": 0.2}
    run_model("SGL", extra, [list(t) for t in train], [list(t) for t in test], 99, "N_SGL", out, meta, sample_rows=False)
    meta["N_SGL"]["batch"] = 2048


def section_W(out, meta):
    """Duplicated interaction lines: scipy sums them to weight 2 in norm_adj (ui_graph.py:47-56);
    edge_dropout rebuilds its view with ones (augmentor.py:36-39)."""
    tu, ti, _, _ = MG.tiny_graph()
    dup = np.arange(0, len(tu), 7)
    tu2, ti2 = np.concatenate([tu, tu[dup]]), np.concatenate([ti, ti[dup]])
    train = synth.as_triples(tu2, ti2)
    data = Interaction({"dummy": 1}, [list(t) for t in train], [])
    out["dup_train_u"], out["dup_train_i"] = tu2.astype(np.int32), ti2.astype(np.int32)
    na = data.norm_adj.tocsr(); na.sort_indices()
    out["dup_norm_adj_indptr"], out["dup_norm_adj_indices"] = na.indptr.astype(np.int32), na.indices.astype(np.int32)
    out["dup_norm_adj_data"] = na.data.astype(np.float32)
    random.seed(5)
    dropped = GraphAugmentor.edge_dropout(data.interaction_mat, 0.1)
    lap = data.convert_to_laplacian_mat(dropped).tocsr(); lap.sort_indices()
    out["dup_drop_lap_indptr"], out["dup_drop_lap_indices"] = lap.indptr.astype(np.int32), lap.indices.astype(np.int32)
    out["dup_drop_lap_data"] = lap.data.astype(np.float32)
    meta["W_dup"] = {"n_lines": len(train), "n_unique": int(data.interaction_mat.nnz), "max_weight": float(data.ui_adj.max()),
                     "drop_seed": 5}


def section_E(out, meta):
    """Any `embedding.size` (base/recommender.py:16; VERDICT r02 missing #4): the reference's XSimGCL.py / SGL.py / MF.py
    on the 200 x 300 graph with embedding.size = 50 and 96 (LightGCN: 20) -- widths no kernel of selfrec_amd serves
    natively; it stores them zero-padded.  Two steps each, full tables, first-step layer outputs and pre-Adam gradients."""
    tu, ti, su, si = MG.tiny_graph()
    train, test = synth.as_triples(tu, ti), synth.as_triples(su, si)
    cases = [("XSimGCL", 50, {"n_layer": 2, "l_star": 1, "lambda": 0.2, "eps": 0.2, "tau": 0.2}, 2),
             ("XSimGCL", 96, {"n_layer": 2, "l_star": 1, "lambda": 0.2, "eps": 0.2, "tau": 0.2}, 2),
             ("SGL", 96, {"n_layer": 2, "lambda": 0.1, "drop_rate": 0.1, "aug_type": 1, "temp": 0.2}, 6),
             ("LightGCN", 20, {"n_layer": 2}, 2),
             ("MF", 50, {}, 0)]
    try:
        for name, emb, extra, cap in cases:
            EMB["default"] = emb
            tag = f"E_{name}{emb}"
            run_model(name, extra, [list(t) for t in train], [list(t) for t in test], 2, tag, out, meta, sample_rows=False,
                      capture=cap)
            meta[tag]["batch"] = 2048
    finally:
        EMB["default"] = 64


F4_MODELS = {
    "DirectAU": {"gamma": 2, "n_layers": 3},
    "MixGCF": {"n_layer": 3, "n_negs": 64},
    "BUIR": {"n_layer": 2, "tau": 0.995, "drop_rate": 0.2},
    "SelfCF": {"n_layer": 2, "tau": 0.05},
}


def section_M(out, meta):
    import torch.nn.functional as F
    tu, ti, su, si = MG.tiny_graph()
    train, test = synth.as_triples(tu, ti), synth.as_triples(su, si)
    real_dropout, real_rand_like = F.dropout, torch.rand_like
    cwd = os.getcwd()
    for name, extra in F4_MODELS.items():
        mod = importlib.import_module(f"model.graph.{name}")
        n_steps = 2
        rec = {"batches": [], "loss": []}

        def batches(data, bs, n_negs=1):
            for k, b in enumerate(ref_sampler.next_batch_pairwise(data, bs, n_negs)):
                if k == n_steps:
                    return
                rec["batches"].append([list(x) for x in b])
                yield b
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                conf = make_conf(tmp, name, dict(extra), batch=1024)
                mod.next_batch_pairwise = batches
                gen = torch.Generator().manual_seed(777)

                def cpu_dropout(x, p=0.5, training=True, inplace=False):
                    if not training or p == 0.0:
                        return x
                    keep = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype)
                    return x * keep / (1.0 - p)
                F.dropout = cpu_dropout
                torch.rand_like = lambda t, **k: torch.rand(t.shape, generator=gen)
                torch.manual_seed(31); np.random.seed(32); random.seed(2718)
                model = getattr(mod, name)(conf, [list(t) for t in train], [list(t) for t in test])
                model.fast_evaluation = lambda epoch: None
                enc = model.model
                named0 = {k: v.detach().numpy().copy() for k, v in enc.named_parameters()}
                if name == "SelfCF":
                    named0["u_target_his"], named0["i_target_his"] = enc.u_target_his.numpy().copy(), enc.i_target_his.numpy().copy()
                # per-step losses: optimizer.zero_grad() follows each loss in every train() loop
                losses = []
                real_backward = torch.Tensor.backward

                def spy_backward(t, *a, **k):
                    losses.append(float(t.detach()))
                    return real_backward(t, *a, **k)
                torch.Tensor.backward = spy_backward
                try:
                    model.train()
                except AttributeError as e:
                    assert "best_" in str(e), e
                finally:
                    torch.Tensor.backward = real_backward
                F.dropout, torch.rand_like = real_dropout, real_rand_like
                for k, v in named0.items():
                    out[f"M_{name}_init_{k}"] = v
                for k, v in enc.named_parameters():
                    out[f"M_{name}_param_{k}"] = v.detach().numpy().copy()
                if name in ("BUIR", "SelfCF"):
                    embs = enc.get_embedding()
                    for k, v in zip(("p_u", "u", "p_i", "i"), embs):
                        out[f"M_{name}_emb_{k}"] = v.detach().numpy().copy()
                    model.p_u_online, model.u_online, model.p_i_online, model.i_online = embs
                elif name == "MixGCF":
                    with torch.no_grad():
                        model.user_emb, model.item_emb = enc.get_embeddings()
                    out[f"M_{name}_emb_u"], out[f"M_{name}_emb_i"] = model.user_emb.numpy().copy(), model.item_emb.numpy().copy()
                else:
                    with torch.no_grad():
                        model.user_emb, model.item_emb = enc()
                    out[f"M_{name}_emb_u"], out[f"M_{name}_emb_i"] = model.user_emb.numpy().copy(), model.item_emb.numpy().copy()
                rec_list = model.test()
                d = model.data
                users = list(d.test_set.keys())
                out[f"M_{name}_test_users"] = np.asarray([d.user[u] for u in users], dtype=np.int32)
                out[f"M_{name}_rec_ids"] = np.asarray([[d.item[it] for it, _ in rec_list[u]] for u in users], dtype=np.int32)
                out[f"M_{name}_loss"] = np.asarray(losses, dtype=np.float64)
                out[f"M_{name}_batch_sizes"] = np.asarray([len(b[0]) for b in rec["batches"]], dtype=np.int32)
                for k, col in enumerate("uij"):
                    out[f"M_{name}_batch_{col}"] = np.concatenate([b[k] for b in rec["batches"]]).astype(np.int32)
                meta[f"M_{name}"] = {"conf": extra, "n_steps": n_steps, "emb": 64, "batch": 1024, "lr": 0.001, "reg": 0.0001,
                                     "torch_seed": 31, "numpy_seed": 32, "sampler_seed": 2718, "device_rng_seed": 777,
                                     "measure": ranking_evaluation(d.test_set, rec_list, [10, 20])}
                print(f"M_{name}", losses, flush=True)
            finally:
                os.chdir(cwd)
                F.dropout, torch.rand_like = real_dropout, real_rand_like
    # MixGCF's sampler regime at the Yelp2018 shape: 64 negatives per pair (MixGCF.py:24,96-114), first 20 batches
    tu, ti, su, si, U, I = synth.make_dataset("yelp2018", seed=SEED_GRAPH)
    data = Interaction({"dummy": 1}, [list(t) for t in synth.as_triples(tu, ti)], [])
    random.seed(64064)
    hs = [hashlib.sha256() for _ in range(3)]
    for k, (u, i, j) in enumerate(ref_sampler.next_batch_pairwise(data, 2048, 64)):
        if k == 20:
            break
        for h, a in zip(hs, (u, i, j)):
            h.update(np.asarray(a, dtype=np.int32).tobytes())
    meta["M_sampler_negs64"] = {"seed": 64064, "batch": 2048, "n_negs": 64, "batches": 20, "sha_u": hs[0].hexdigest(),
                                "sha_i": hs[1].hexdigest(), "sha_j": hs[2].hexdigest()}
    print("M_sampler_negs64", meta["M_sampler_negs64"], flush=True)


def main():
    want = [a for a in sys.argv[1:]] or ["N", "W", "D", "Y", "F", "M", "E"]
    npz_path, meta_path = os.path.join(HERE, "shapes.npz"), os.path.join(HERE, "shapes_meta.json")
    out = dict(np.load(npz_path)) if os.path.exists(npz_path) else {}
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    meta.update(torch=torch.__version__, numpy=np.__version__)
    for s in want:
        t0 = time.time()
        {"Y": section_Y, "F": section_F, "D": section_D, "N": section_N, "W": section_W, "M": section_M, "B": section_B,
         "E": section_E}[s](out, meta)
        print(f"section {s}: {time.time() - t0:.0f} s", flush=True)
        np.savez_compressed(npz_path, **out)
        with open(meta_path, "w") as f:
            json.dump(meta, f, indent=1)
    print("written", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
