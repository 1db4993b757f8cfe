#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE'S OWN Python (from /root/reference).

The reference has no tests or golden vectors, so parity is pinned
on outputs of the reference itself, produced here in the build container (CPU) and
committed.  /root/reference does not exist on the GPU box: tests only read the .npz /
.json files this script writes, never the reference.

Shims: numba is not installed and util/algorithm.py:3 imports it
-> a stub module whose ``jit`` is the identity decorator; models hard-code ``.cuda()``
-> ``torch.Tensor.cuda`` / ``nn.Module.cuda`` patched to return self.

Run:  python tests/golden/make_golden.py        (writes next to this file)
      python tests/golden/make_golden.py edges  (only edges.npz: the configurations EDGE_CASES lists)
"""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

numba_stub = types.ModuleType("numba")
numba_stub.jit = lambda *a, **k: (lambda f: f)
sys.modules["numba"] = numba_stub

import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
torch.set_num_threads(1)

sys.path.insert(0, REF)
from util.conf import ModelConf  # noqa: E402
from util import sampler as ref_sampler  # noqa: E402
from util import loss_torch as ref_loss  # noqa: E402
from util.evaluation import ranking_evaluation  # noqa: E402
from util.algorithm import find_k_largest  # noqa: E402
from data.ui_graph import Interaction  # noqa: E402
from data.graph import Graph  # noqa: E402
from data.augmentor import GraphAugmentor  # noqa: E402
from base.torch_interface import TorchGraphInterface  # noqa: E402

from selfrec_amd import synth  # noqa: E402

GRAPH = dict(n_users=200, n_items=300, n_edges=3600, seed=11)
EMB = 64
BATCH = 1024


def tiny_graph():
    u, i = synth.generate_edges(GRAPH["n_users"], GRAPH["n_items"], GRAPH["n_edges"], GRAPH["seed"])
    (tu, ti), (su, si) = synth.split_train_test(u, i, GRAPH["n_users"], GRAPH["n_items"], 0.2, GRAPH["seed"])
    return tu, ti, su, si


def make_conf(tmp, model, extra, emb=EMB):
    lines = [
        "training.set: ./train.txt", "test.set: ./test.txt",
        "model:", f"  name: {model}", "  type: graph",
        "item.ranking.topN: [10,20]", f"embedding.size: {emb}", f"max.epoch: {extra.pop('max_epoch', 1)}",
        f"batch.size: {BATCH}", "learning.rate: 0.001", "reg.lambda: 0.0001", "output: ./results/",
    ]
    if extra:
        lines.append(f"{model}:")
        lines += [f"  {k}: {v}" for k, v in extra.items()]
    path = os.path.join(tmp, f"{model}.yaml")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return ModelConf(path)


def golden_sampler(out):
    tu, ti, _, _ = tiny_graph()
    train = synth.as_triples(tu, ti)
    conf = {"dummy": 1}
    for tag, bs, negs, seed in (("a", 1024, 1, 7), ("b", 500, 3, 123456789)):
        data = Interaction(conf, [list(t) for t in train], [])
        random.seed(seed)
        us, is_, js, sizes = [], [], [], []
        for _epoch in range(2):  # second epoch re-shuffles the already shuffled list
            for u, i, j in ref_sampler.next_batch_pairwise(data, bs, negs):
                us += u
                is_ += i
                js += j
                sizes.append(len(u))
        out[f"sampler_{tag}_u"] = np.asarray(us, dtype=np.int32)
        out[f"sampler_{tag}_i"] = np.asarray(is_, dtype=np.int32)
        out[f"sampler_{tag}_j"] = np.asarray(js, dtype=np.int32)
        out[f"sampler_{tag}_sizes"] = np.asarray(sizes, dtype=np.int32)
        out[f"sampler_{tag}_meta"] = np.asarray([bs, negs, seed], dtype=np.int64)
        # state after: the next raw draw pins how much of the global stream was consumed
        out[f"sampler_{tag}_next_u32"] = np.asarray([random.getrandbits(32)], dtype=np.uint32)
        # final order of training_data, as ids (the list is shuffled in place)
        out[f"sampler_{tag}_final_order_u"] = np.asarray([data.user[t[0]] for t in data.training_data], dtype=np.int32)
        out[f"sampler_{tag}_final_order_i"] = np.asarray([data.item[t[1]] for t in data.training_data], dtype=np.int32)
    # id maps (first-appearance order) of the un-shuffled file order
    data = Interaction(conf, [list(t) for t in train], [])
    out["graph_train_u_ids"] = np.asarray([data.user[t[0]] for t in train], dtype=np.int32)
    out["graph_train_i_ids"] = np.asarray([data.item[t[1]] for t in train], dtype=np.int32)
    out["graph_train_u_raw"] = tu.astype(np.int32)
    out["graph_train_i_raw"] = ti.astype(np.int32)
    na = data.norm_adj.tocsr()
    na.sort_indices()
    out["norm_adj_indptr"] = na.indptr.astype(np.int32)
    out["norm_adj_indices"] = na.indices.astype(np.int32)
    out["norm_adj_data"] = na.data.astype(np.float32)
    coo = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj)
    out["norm_adj_coo_shape"] = np.asarray(list(coo.shape), dtype=np.int64)
    # isolated node + rectangular case of Graph.normalize_graph_mat (graph.py:15,19-23)
    import scipy.sparse as sp
    sq = sp.csr_matrix(np.array([[0, 1, 0, 2], [1, 0, 0, 0], [0, 0, 0, 0], [2, 0, 0, 0]], dtype=np.float32))
    out["norm_sq_dense"] = Graph.normalize_graph_mat(sq).toarray().astype(np.float32)
    rect = sp.csr_matrix(np.array([[1, 1, 0], [0, 0, 0]], dtype=np.float32))
    out["norm_rect_dense"] = Graph.normalize_graph_mat(rect).toarray().astype(np.float32)
    # edge dropout keep-set + dropped laplacian (augmentor.py:29-40, ui_graph.py:58-65)
    random.seed(99)
    n_e = data.interaction_mat.count_nonzero()
    st = random.getstate()
    keep = random.sample(range(n_e), int(n_e * (1 - 0.1)))
    out["edge_dropout_keep"] = np.asarray(keep, dtype=np.int32)
    random.setstate(st)
    dropped = GraphAugmentor.edge_dropout(data.interaction_mat, 0.1)
    lap = data.convert_to_laplacian_mat(dropped).tocsr()
    lap.sort_indices()
    out["edge_dropout_lap_indptr"] = lap.indptr.astype(np.int32)
    out["edge_dropout_lap_indices"] = lap.indices.astype(np.int32)
    out["edge_dropout_lap_data"] = lap.data.astype(np.float32)
    out["edge_dropout_next_u32"] = np.asarray([random.getrandbits(32)], dtype=np.uint32)


def golden_ops(out):
    g = torch.Generator().manual_seed(5)
    for n in (1, 2, 130, 515):
        u = (torch.randn(n, EMB, generator=g) * 0.3).requires_grad_()
        p = (torch.randn(n, EMB, generator=g) * 0.3).requires_grad_()
        q = (torch.randn(n, EMB, generator=g) * 0.3).requires_grad_()
        if n > 2:  # duplicate rows, as a batch with repeated users produces
            with torch.no_grad():
                u[1] = u[0]
                p[2] = p[0]
        bpr = ref_loss.bpr_loss(u, p, q)
        reg = ref_loss.l2_reg_loss(1e-4, u, p, q)
        nce = ref_loss.InfoNCE(u, p, 0.2)
        gb = torch.autograd.grad(bpr, (u, p, q), retain_graph=True)
        gr = torch.autograd.grad(reg, (u, p, q), retain_graph=True)
        gn = torch.autograd.grad(nce, (u, p))
        out[f"ops_{n}_in"] = torch.stack([u, p, q]).detach().numpy()
        out[f"ops_{n}_loss"] = np.asarray([bpr.item(), reg.item(), nce.item()], dtype=np.float64)
        out[f"ops_{n}_g_bpr"] = torch.stack(gb).numpy()
        out[f"ops_{n}_g_reg"] = torch.stack(gr).numpy()
        out[f"ops_{n}_g_nce"] = torch.stack(gn).numpy()
    # find_k_largest with ties and a masked block (algorithm.py:144-156)
    cand = np.asarray([0.5, 0.5, 0.9, -10e8, 0.1, 0.9, 0.3, 0.5, 0.7, 0.2, 0.9, 0.0], dtype=np.float32)
    ids, sc = find_k_largest(5, cand)
    out["topk_ties_in"] = cand
    out["topk_ties_ids"] = np.asarray(ids, dtype=np.int64)
    out["topk_ties_scores"] = np.asarray(sc, dtype=np.float32)


MODELS = {
    "MF": {},
    "LightGCN": {"n_layer": 3},
    "XSimGCL": {"n_layer": 3, "l_star": 1, "lambda": 0.2, "eps": 0.2, "tau": 0.2},
    "SimGCL": {"n_layer": 2, "lambda": 0.5, "eps": 0.1},
    "SGL": {"n_layer": 2, "lambda": 0.1, "drop_rate": 0.1, "aug_type": 1, "temp": 0.2},
}


# Configurations the five runs above do not reach: the 64-lane row shape (d = 256), a single layer (the only product
# also carries the mean), four layers, and the contrast view taken at the ego table (l* = 0: XSimGCL.py's
# ``k == self.layer_cl - 1`` never fires) or at the last layer.  tag -> (model, its section, embedding.size)
EDGE_CASES = {
    "XSimGCL_d256": ("XSimGCL", {"n_layer": 2, "l_star": 2, "lambda": 0.2, "eps": 0.2, "tau": 0.2}, 256),
    "LightGCN_d256": ("LightGCN", {"n_layer": 2}, 256),
    "XSimGCL_L1_s0": ("XSimGCL", {"n_layer": 1, "l_star": 0, "lambda": 0.2, "eps": 0.2, "tau": 0.2}, 64),
    "XSimGCL_L1_s1": ("XSimGCL", {"n_layer": 1, "l_star": 1, "lambda": 0.2, "eps": 0.2, "tau": 0.2}, 64),
    "XSimGCL_L4_s0": ("XSimGCL", {"n_layer": 4, "l_star": 0, "lambda": 0.2, "eps": 0.2, "tau": 0.2}, 64),
    "XSimGCL_L4_s4": ("XSimGCL", {"n_layer": 4, "l_star": 4, "lambda": 0.2, "eps": 0.2, "tau": 0.2}, 64),
    "LightGCN_L1": ("LightGCN", {"n_layer": 1}, 64),
    "LightGCN_L4": ("LightGCN", {"n_layer": 4}, 64),
    "SimGCL_L1": ("SimGCL", {"n_layer": 1, "lambda": 0.5, "eps": 0.1}, 64),
    "SimGCL_L4": ("SimGCL", {"n_layer": 4, "lambda": 0.5, "eps": 0.1}, 64),
    "SGL_L1": ("SGL", {"n_layer": 1, "lambda": 0.1, "drop_rate": 0.1, "aug_type": 1, "temp": 0.2}, 64),
    "SGL_L4": ("SGL", {"n_layer": 4, "lambda": 0.1, "drop_rate": 0.1, "aug_type": 1, "temp": 0.2}, 64),
}


def golden_models(out, meta, cases=None, with_ranking=True):
    import importlib
    cases = cases or {name: (name, extra, EMB) for name, extra in MODELS.items()}
    tu, ti, su, si = tiny_graph()
    train, test = synth.as_triples(tu, ti), synth.as_triples(su, si)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for tag, (name, extra, emb) in cases.items():
                mod = importlib.import_module(f"model.graph.{name}")
                conf = make_conf(tmp, name, dict(extra), emb)
                rec = {"batches": [], "bpr": [], "reg": [], "nce": []}

                def wrap(fn, key):
                    def inner(*a, **k):
                        r = fn(*a, **k)
                        rec[key].append(float(r))
                        return r
                    return inner

                def batches(data, bs, n_negs=1):
                    for b in ref_sampler.next_batch_pairwise(data, bs, n_negs):
                        rec["batches"].append([list(x) for x in b])
                        yield b

                mod.next_batch_pairwise = batches
                mod.bpr_loss = wrap(ref_loss.bpr_loss, "bpr")
                mod.l2_reg_loss = wrap(ref_loss.l2_reg_loss, "reg")
                if hasattr(mod, "InfoNCE"):
                    mod.InfoNCE = wrap(ref_loss.InfoNCE, "nce")
                noise_gen = torch.Generator().manual_seed(4242)
                torch.rand_like = lambda t, **k: torch.rand(t.shape, generator=noise_gen)
                torch.manual_seed(31)
                random.seed(2718)
                model = getattr(mod, name)(conf, [list(t) for t in train], [list(t) for t in test])
                params = model.model.embedding_dict
                init_u = params["user_emb"].detach().numpy().copy()
                init_i = params["item_emb"].detach().numpy().copy()
                try:
                    model.train()
                except AttributeError as e:  # SGL only evaluates from epoch 5 on (SGL.py:44-46)
                    assert name == "SGL" and "best_user_emb" in str(e), e
                    with torch.no_grad():
                        model.user_emb, model.item_emb = model.model()
                out[f"{tag}_init_user"] = init_u
                out[f"{tag}_init_item"] = init_i
                out[f"{tag}_param_user"] = params["user_emb"].detach().numpy().copy()
                out[f"{tag}_param_item"] = params["item_emb"].detach().numpy().copy()
                nb = len(rec["batches"])
                out[f"{tag}_batch_sizes"] = np.asarray([len(b[0]) for b in rec["batches"]], dtype=np.int32)
                out[f"{tag}_batch_u"] = np.concatenate([b[0] for b in rec["batches"]]).astype(np.int32)
                out[f"{tag}_batch_i"] = np.concatenate([b[1] for b in rec["batches"]]).astype(np.int32)
                out[f"{tag}_batch_j"] = np.concatenate([b[2] for b in rec["batches"]]).astype(np.int32)
                out[f"{tag}_loss_bpr"] = np.asarray(rec["bpr"], dtype=np.float64)
                out[f"{tag}_loss_reg"] = np.asarray(rec["reg"], dtype=np.float64)
                out[f"{tag}_loss_nce"] = np.asarray(rec["nce"], dtype=np.float64)
                meta[tag] = {"model": name, "conf": extra, "n_batches": nb,
                             "emb": emb, "batch": BATCH, "lr": 0.001, "reg": 0.0001,
                             "noise_seed": 4242, "init_seed": 31, "sampler_seed": 2718}
                if with_ranking:
                    rec_list = model.test()
                    measure = ranking_evaluation(model.data.test_set, rec_list, [10, 20])
                    d = model.data
                    users = list(d.test_set.keys())
                    out[f"{tag}_final_user"] = model.user_emb.detach().numpy().copy()
                    out[f"{tag}_final_item"] = model.item_emb.detach().numpy().copy()
                    out[f"{tag}_test_users"] = np.asarray([d.user[u] for u in users], dtype=np.int32)
                    out[f"{tag}_rec_ids"] = np.asarray([[d.item[it] for it, _ in rec_list[u]] for u in users], dtype=np.int32)
                    out[f"{tag}_rec_scores"] = np.asarray([[s for _, s in rec_list[u]] for u in users], dtype=np.float32)
                    meta[tag]["measure"] = measure
                print(tag, "steps", nb, "bpr", rec["bpr"][:3], "nce", rec["nce"][:2])
        finally:
            os.chdir(cwd)
    out["test_u_ids_raw"] = su.astype(np.int32)
    out["test_i_ids_raw"] = si.astype(np.int32)


def edges_main():
    """python tests/golden/make_golden.py edges  ->  edges.npz + edges_meta.json (models.npz etc. untouched)"""
    out, meta = {}, {"graph": GRAPH, "torch": torch.__version__, "numpy": np.__version__}
    golden_models(out, meta, EDGE_CASES, with_ranking=False)
    for k in ("test_u_ids_raw", "test_i_ids_raw"):
        out.pop(k)
    np.savez_compressed(os.path.join(HERE, "edges.npz"), **out)
    with open(os.path.join(HERE, "edges_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("written edges.npz", os.path.getsize(os.path.join(HERE, "edges.npz")))


def main():
    out, meta = {}, {"graph": GRAPH, "torch": torch.__version__, "numpy": np.__version__}
    golden_sampler(out)
    golden_ops(out)
    np.savez_compressed(os.path.join(HERE, "ops_sampler_graph.npz"), **out)
    out = {}
    golden_models(out, meta)
    np.savez_compressed(os.path.join(HERE, "models.npz"), **out)
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("written", os.listdir(HERE))


if __name__ == "__main__":
    edges_main() if sys.argv[1:] == ["edges"] else main()
