"""the other torch graph models through the boundary -- DirectAU, MixGCF, BUIR, SelfCF
(selfrec_amd/model/graph/*.py, op-level tier: their own torch code over this package's sampler, SpMM handle, loss
kernels and device ranking) against 2-step runs of the REFERENCE'S files on the 200 x 300 graph
(tests/golden/shapes.npz section M).  Randomness: what the reference draws on the device (nn.Dropout, rand_like) is
drawn from the golden's seeded CPU generator by patching torch.nn.functional.dropout / torch.rand_like here exactly
as make_golden_shapes.py does; host-side draws (torch.rand(..).cuda(), torch.randn, np.random.random) follow from
the seeds because the models draw them in the reference's order."""
import importlib
import json
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_shapes_cpu import GOLDEN

pytestmark = pytest.mark.gpu
MODELS = ["DirectAU", "MixGCF", "BUIR", "SelfCF"]


@pytest.fixture(scope="module")
def shapes():
    return np.load(os.path.join(GOLDEN, "shapes.npz"))


@pytest.fixture(scope="module")
def smeta():
    with open(os.path.join(GOLDEN, "shapes_meta.json")) as f:
        return json.load(f)


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


@pytest.mark.parametrize("name", MODELS)
def test_two_steps_match_the_reference_model_file(shapes, smeta, fresh_tiny_data, name, monkeypatch, tmp_path):
    from selfrec_amd.util import sampler as sampler_mod
    from selfrec_amd.util.conf import ModelConf
    from selfrec_amd.util.evaluation import ranking_evaluation
    monkeypatch.chdir(tmp_path)
    m = smeta[f"M_{name}"]
    gen = torch.Generator().manual_seed(m["device_rng_seed"])

    def cpu_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        keep = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype).to(x.device)
        return x * keep / (1.0 - p)
    monkeypatch.setattr(F, "dropout", cpu_dropout)
    monkeypatch.setattr(torch, "rand_like", lambda t, **k: torch.rand(t.shape, generator=gen).to(t.device))
    real_batches = sampler_mod.next_batch_pairwise
    seen = []

    def batches(data, bs, n_negs=1, **kw):
        for k, b in enumerate(real_batches(data, bs, n_negs, **kw)):
            if k == m["n_steps"]:
                return
            seen.append(b)
            yield b
    monkeypatch.setattr(sampler_mod, "next_batch_pairwise", batches)
    losses = []
    real_backward = torch.Tensor.backward
    monkeypatch.setattr(torch.Tensor, "backward", lambda t, *a, **k: (losses.append(float(t.detach())), real_backward(t, *a, **k))[1])

    conf = ModelConf({"model": {"name": name, "type": "graph"}, "item.ranking.topN": [10, 20], "embedding.size": m["emb"],
                      "max.epoch": 1, "batch.size": m["batch"], "learning.rate": m["lr"], "reg.lambda": m["reg"],
                      "output": "./results/", "training.set": "x", "test.set": "y", name: m["conf"]})
    torch.manual_seed(m["torch_seed"]); np.random.seed(m["numpy_seed"]); random.seed(m["sampler_seed"])
    data = fresh_tiny_data
    cls = getattr(importlib.import_module(f"selfrec_amd.model.graph.{name}"), name)
    model = cls(conf, data.training_data, data.test_data)
    model.fast_evaluation = lambda epoch: None
    enc = model.model
    for k, v in enc.named_parameters():                       # same creation order => same initial tables
        assert np.array_equal(v.detach().cpu().numpy(), shapes[f"M_{name}_init_{k}"]), k
    if name == "SelfCF":
        assert np.array_equal(enc.u_target_his.cpu().numpy(), shapes["M_SelfCF_init_u_target_his"])
    model.train()
    # batches (bit-exact, 64 negatives per pair for MixGCF), per-step losses, parameters, embeddings, ranking
    for k, col in enumerate("uij"):
        assert np.array_equal(np.concatenate([b[k] for b in seen]), shapes[f"M_{name}_batch_{col}"])
    np.testing.assert_allclose(losses, shapes[f"M_{name}_loss"], rtol=2e-5)
    for k, v in enc.named_parameters():
        want = shapes[f"M_{name}_param_{k}"]
        assert np.abs(v.detach().cpu().numpy() - want).max() < 1e-5, k           # << one Adam step
        assert rel_err(v.detach().cpu().numpy(), want) < 2e-4, k
    if name in ("BUIR", "SelfCF"):
        embs = dict(zip(("p_u", "u", "p_i", "i"), enc.get_embedding()))
    else:
        model.model.eval()
        with torch.no_grad():
            model.snapshot()
        embs = {"u": model.user_emb, "i": model.item_emb}
    for k, v in embs.items():
        assert rel_err(v.cpu().numpy(), shapes[f"M_{name}_emb_{k}"]) < 1e-4, k
    with torch.no_grad():
        model.snapshot()
    rec = model.test()
    users = [data.id2user[int(u)] for u in shapes[f"M_{name}_test_users"]]
    ids = np.asarray([[data.item[it] for it, _ in rec[u]] for u in users])
    assert (ids == shapes[f"M_{name}_rec_ids"]).mean() > 0.99
    got = ranking_evaluation(data.test_set, rec, [10, 20])
    gv = [float(x.split(":")[1]) for x in got if ":" in x]
    wv = [float(x.split(":")[1]) for x in m["measure"] if ":" in x]
    np.testing.assert_allclose(gv, wv, atol=5e-4)


def test_handle_dropout_is_a_value_array_with_a_mirrored_transpose(fresh_tiny_data):
    """SparseAdjHandle.dropout (BUIR.py:118-127 as a value array): forward and backward products equal the torch
    COO tensor the reference would build from the same mask."""
    from selfrec_amd.base.torch_interface import TorchGraphInterface
    h = TorchGraphInterface.convert_sparse_mat_to_tensor(fresh_tiny_data.norm_adj).cuda()
    g = torch.Generator().manual_seed(3)
    keep = torch.floor(0.85 + torch.rand(h._nnz(), generator=g)).bool()
    dropped = h.dropout(keep, 1.0 / 0.85)
    want = torch.sparse_coo_tensor(h._indices()[:, keep.cuda()], h._values()[keep.cuda()], tuple(h.shape)) * (1.0 / 0.85)
    x = torch.randn((h.shape[1], 64), device="cuda", requires_grad=True)
    x2 = x.detach().clone().requires_grad_()
    w = torch.randn((h.shape[0], 64), device="cuda")
    (torch.sparse.mm(dropped, x) * w).sum().backward()
    (torch.sparse.mm(want, x2) * w).sum().backward()
    assert rel_err(torch.sparse.mm(dropped, x).detach().cpu().numpy(), torch.sparse.mm(want, x2).detach().cpu().numpy()) < 2e-6
    assert rel_err(x.grad.cpu().numpy(), x2.grad.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("name", ["BUIR", "SelfCF"])
def test_final_test_ranks_the_best_epoch_not_the_last(smeta, fresh_tiny_data, name, monkeypatch, tmp_path):
    """after train() the two-tower models report the BEST epoch (BUIR.py:36-37, SelfCF.py:33-34
    copy best_* back) -- test()'s device path ranks the concatenated tables, so restore_best() must rebuild those too.
    Three epochs whose best is the first: the final ranking equals predict() on the best tables, row by row."""
    from selfrec_amd.util.conf import ModelConf
    monkeypatch.chdir(tmp_path)
    m = smeta[f"M_{name}"]
    conf = ModelConf({"model": {"name": name, "type": "graph"}, "item.ranking.topN": [10, 20], "embedding.size": m["emb"],
                      "max.epoch": 3, "batch.size": m["batch"], "learning.rate": 0.05, "reg.lambda": m["reg"],
                      "output": "./results/", "training.set": "x", "test.set": "y", name: m["conf"]})
    torch.manual_seed(1); np.random.seed(2); random.seed(3)
    data = fresh_tiny_data
    cls = getattr(importlib.import_module(f"selfrec_amd.model.graph.{name}"), name)
    model = cls(conf, data.training_data, data.test_data)
    kept = {}

    def fast_evaluation(epoch):                     # the first epoch is "the best": save() there only
        if epoch == 0:
            model.bestPerformance = [1, {"Recall": 1.0, "NDCG": 1.0}]
            model.save()
            kept["best"] = [t.clone() for t in (model.best_p_u, model.best_u, model.best_p_i, model.best_i)]
        kept["last"] = [t.clone() for t in model.model.get_embedding()]
    model.fast_evaluation = fast_evaluation
    model.train()
    best, last = kept["best"], kept["last"]
    assert not torch.equal(best[1], last[1])        # (lr = 0.05: the tables did move after the first epoch)
    for got, want in zip((model.p_u_online, model.u_online, model.p_i_online, model.i_online), best):
        assert torch.equal(got, want)
    assert torch.equal(model.user_emb, torch.cat([best[0], best[1]], 1))
    assert torch.equal(model.item_emb, torch.cat([best[3], best[2]], 1))
    rec = model.test()                              # the final 'Testing...' pass: device ranking of the concatenation
    from selfrec_amd.util.algorithm import find_k_largest
    same = total = 0
    for user in list(data.test_set)[:40]:
        cand = model.predict(user)                  # the reference's path on the best tables
        for item in data.user_rated(user)[0]:
            cand[data.item[item]] = -10e8
        ids, _ = find_k_largest(20, cand)
        got = [data.item[it] for it, _ in rec[user]]
        same += sum(int(a == b) for a, b in zip(got, ids)); total += 20
    assert same / total > 0.99
