"""CPU tests of the host-side mirror of the reference interface (no GPU, no HIP calls)."""
import os
import random

import numpy as np
import pytest

from oracle import selfrec_oracle as O
from selfrec_amd import synth
from selfrec_amd.data.loader import FileIO
from selfrec_amd.data.ui_graph import Interaction
from selfrec_amd.util import algorithm, evaluation
from selfrec_amd.util.conf import ModelConf
from selfrec_amd.util.sampler import next_batch_pairwise


def test_interaction_matches_reference_products(golden_ops, tiny_data):
    g, d = golden_ops, tiny_data
    assert np.array_equal(d.train_u, g["graph_train_u_ids"]) and np.array_equal(d.train_i, g["graph_train_i_ids"])
    na = d.norm_adj.tocsr(); na.sort_indices()
    assert np.array_equal(na.indptr, g["norm_adj_indptr"]) and np.array_equal(na.indices, g["norm_adj_indices"])
    assert np.array_equal(na.data, g["norm_adj_data"])
    assert d.user_num == 200 and d.item_num == 300
    u0 = d.id2user[0]
    names, ones = d.user_rated(u0)
    assert set(d.item[n] for n in names) == set(d.train_i[d.train_u == 0].tolist()) and set(ones) == {1}
    assert all(u in d.user for u in d.test_set)


def test_drop_in_generator_is_bit_exact(golden_ops):
    g = golden_ops
    train = synth.as_triples(g["graph_train_u_raw"], g["graph_train_i_raw"])
    bs, negs, seed = (int(x) for x in g["sampler_a_meta"])
    data = Interaction({}, [list(t) for t in train], [])
    random.seed(seed)
    us, js = [], []
    for _ in range(2):
        for u, i, j in next_batch_pairwise(data, bs, negs):
            assert isinstance(u, list) and isinstance(j, list)
            us += u; js += j
    assert np.array_equal(us, g["sampler_a_u"]) and np.array_equal(js, g["sampler_a_j"])
    assert random.getrandbits(32) == int(g["sampler_a_next_u32"][0])
    assert np.array_equal([data.user[t[0]] for t in data.training_data], g["sampler_a_final_order_u"])


def test_sampler_construction_on_several_threads_gives_the_same_streams(monkeypatch):
    """srh_sampler_create sorts the users' item lists and fills their membership bitmaps on host threads (user ranges of
    equal edge counts; large graphs only -- SRH_SAMPLER_THREADS forces a count): the epochs do not depend on it."""
    from selfrec_amd import ops
    rng = np.random.default_rng(3)
    U, I, E = 5000, 700, 120000
    u = (rng.random(E) ** 2 * U).astype(np.int32); i = (rng.random(E) ** 2 * I).astype(np.int32)
    key = np.unique(u.astype(np.int64) * I + i)
    u, i = (key // I).astype(np.int32), (key % I).astype(np.int32)
    perm = rng.permutation(len(u)); u, i = u[perm], i[perm]
    runs = []
    for threads in ("1", "2", "7"):
        monkeypatch.setenv("SRH_SAMPLER_THREADS", threads)
        smp = ops.Sampler(u, i, U, I)
        smp.seed(11)
        eps = [smp.epoch(512, 2, with_unique=True) for _ in range(2)]
        runs.append([e[k].copy() for e in eps for k in ("u", "i", "j", "uniq_u", "n_uniq_i")])
    for other in runs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(runs[0], other))


def test_native_heap_walk_is_pythons_heapq(golden_ops):
    """srh_find_k_largest_host restates CPython's heapq (heapify, heapreplace, the stable descending sort) step for step:
    the reference's golden, and tie-heavy random vectors against the literal python walk -- ids in the same ORDER."""
    from selfrec_amd import ops
    g = golden_ops
    ids, sc = ops.find_k_largest_host(5, g["topk_ties_in"])
    assert ids.tolist() == g["topk_ties_ids"].tolist() and np.array_equal(sc, g["topk_ties_scores"])
    rng = np.random.default_rng(1)
    for trial in range(300):
        n, k = int(rng.integers(1, 400)), int(rng.integers(1, 50))
        c = (rng.integers(0, 6, n) * 0.25).astype(np.float32)            # six distinct values: ties everywhere
        if trial % 4 == 0:
            c[rng.integers(0, n, 3)] = -10e8                                # masked entries
        got, want = ops.find_k_largest_host(k, c), algorithm._heap_walk(k, c)
        assert got[0].tolist() == want[0] and np.array_equal(got[1], np.asarray(want[1], dtype=np.float32)), (trial, n, k)


def test_find_k_largest_equals_reference_heap(golden_ops):
    g = golden_ops
    ids, sc = algorithm.find_k_largest(5, g["topk_ties_in"])
    assert ids == g["topk_ties_ids"].tolist()
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = rng.standard_normal(500).astype(np.float32)
        if rng.random() < 0.5:
            x[rng.integers(0, 500, 40)] = x[0]      # inject ties
        a, b = algorithm.find_k_largest(20, x), O.find_k_largest(20, x)
        assert a[0] == b[0] and np.allclose(a[1], b[1])


def test_ranking_evaluation_strings(golden_models, golden_meta, golden_ops, tiny_data):
    gm, d = golden_models, tiny_data
    for name in ("MF", "XSimGCL"):
        users = gm[f"{name}_test_users"]
        res = {d.id2user[int(u)]: [(d.id2item[int(i)], float(s)) for i, s in zip(gm[f"{name}_rec_ids"][k], gm[f"{name}_rec_scores"][k])]
               for k, u in enumerate(users)}
        assert evaluation.ranking_evaluation(d.test_set, res, [10, 20]) == golden_meta[name]["measure"]
    with pytest.raises(SystemExit):
        evaluation.ranking_evaluation(d.test_set, {}, [10])


def test_conf_and_loader(tmp_path):
    p = tmp_path / "m.yaml"
    p.write_text("training.set: a\nmodel:\n  name: X\nX:\n  tau: 0.2\n")
    c = ModelConf(str(p))
    assert c["model"]["name"] == "X" and c.contain("X") and not c.contain("nope") and c["X"]["tau"] == 0.2
    with pytest.raises(SystemExit):
        c["nope"]
    with pytest.raises(IOError):
        ModelConf(str(tmp_path / "missing.yaml"))
    f = tmp_path / "train.txt"
    f.write_text("u1 i1 1\nu2 i1 5\n")
    assert FileIO.load_data_set(str(f), "graph") == [["u1", "i1", 1.0], ["u2", "i1", 5.0]]


def test_synthetic_generator_invariants():
    tu, ti, su, si, U, I = synth.make_dataset("small")
    assert len(np.unique(tu)) == U and len(np.unique(ti)) == I
    keys = np.concatenate([tu * I + ti, su * I + si])
    assert len(np.unique(keys)) == len(keys)
    tu2, *_ = synth.make_dataset("small")
    assert np.array_equal(tu, tu2)


@pytest.mark.skipif(not os.path.isdir("/root/reference/model/graph"), reason="reference checkout not present")
def test_reference_model_files_import_against_the_mirrors():
    """dropin.install(): the reference's unmodified model files resolve base.* / util.* / data.* to
    this package (construction needs a GPU; resolution does not)."""
    import importlib
    import sys
    from selfrec_amd import dropin
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("base", "data", "util", "model")}
    sys.dont_write_bytecode = True
    dropin.install()
    sys.path.insert(0, "/root/reference")
    try:
        for name in ("MF", "LightGCN", "XSimGCL", "SimGCL", "SGL"):
            mod = importlib.import_module(f"model.graph.{name}")
            cls = getattr(mod, name)
            assert cls.__mro__[1].__module__ == "selfrec_amd.base.graph_recommender"
            assert mod.next_batch_pairwise.__module__ == "selfrec_amd.util.sampler"
            assert mod.bpr_loss.__module__ == "selfrec_amd.util.loss_torch"
    finally:
        sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k.split(".")[0] in ("base", "data", "util", "model")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_native_loader_equals_python_path(tmp_path):
    """srh_dataset_load + Interaction's native constructor == the reference-style python loops."""
    import time
    tu, ti, su, si, U, I = synth.make_dataset("small")
    tr, te = tmp_path / "train.txt", tmp_path / "test.txt"
    synth.write_text(str(tr), tu, ti)
    with open(te, "w") as f:                                   # unseen user / item, CRLF, float weight
        f.writelines(f"{a} {b} 1\n" for a, b in zip(su.tolist(), si.tolist()))
        f.write("ghost 17 1\n17 phantom 4.5\r\n")
    slow = Interaction({}, FileIO.load_data_set(str(tr), "graph"), FileIO.load_data_set(str(te), "graph"))
    t0 = time.time()
    fast = Interaction({}, FileIO.open_data_set(str(tr), "graph"), FileIO.open_data_set(str(te), "graph"))
    assert fast.training_data._rows is None                     # no python triples were built
    assert fast.user == slow.user and fast.item == slow.item and fast.id2item == slow.id2item
    assert np.array_equal(fast.train_u, slow.train_u) and np.array_equal(fast.train_i, slow.train_i)
    assert dict(fast.test_set) == dict(slow.test_set) and fast.test_set_item == slow.test_set_item
    assert fast.test_size()[:2] == slow.test_size()[:2] and fast.training_size() == slow.training_size()
    assert (fast.norm_adj != slow.norm_adj).nnz == 0
    assert fast.training_data[0] == slow.training_data[0]       # materialises like the list on demand
    with pytest.raises(Exception):
        bad = tmp_path / "bad.txt"; bad.write_text("only_two tokens\n")
        Interaction({}, FileIO.open_data_set(str(bad), "graph"), [])


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_native_loader_in_parallel_keeps_first_appearance_ids(tmp_path, monkeypatch, threads):
    """The loader parses pieces of the file on worker threads and merges their name lists in file order: ids, name order,
    weights' acceptance and the kept test pairs must be the single pass's at ANY thread count -- names repeat across
    pieces, some appear late, lines have odd lengths, the last line has no newline, and a malformed line is reported with
    its number in the whole file."""
    import ctypes as C
    from selfrec_amd import _lib
    rng = np.random.default_rng(5)
    n = 30000
    users = [f"u{int(v)}" if v % 7 else f"user-with-a-long-name-{int(v)}" for v in rng.zipf(1.3, n) % 4000]
    items = [f"{int(v)}" for v in rng.integers(0, 2500, n)]
    weights = [("1", "1.0", "4.5", "0.25", "3")[int(v)] for v in rng.integers(0, 5, n)]
    tr, te = tmp_path / "train.txt", tmp_path / "test.txt"
    tr.write_text("\n".join(f"{a} {b} {w}" for a, b, w in zip(users, items, weights)))          # no trailing newline
    te.write_text("".join(f"{a} {b} 1\n" for a, b in zip(users[::3] + ["nobody"], items[1::3] + ["7"])) + "u1 nothing 1\n")
    monkeypatch.setenv("SRH_LOADER_THREADS", str(threads))
    slow = Interaction({}, FileIO.load_data_set(str(tr), "graph"), FileIO.load_data_set(str(te), "graph"))
    fast = Interaction({}, FileIO.open_data_set(str(tr), "graph"), FileIO.open_data_set(str(te), "graph"))
    assert list(fast.user) == list(slow.user) and list(fast.item) == list(slow.item)       # same ORDER of first appearance
    assert fast.user == slow.user and fast.item == slow.item
    assert np.array_equal(fast.train_u, slow.train_u) and np.array_equal(fast.train_i, slow.train_i)
    assert dict(fast.test_set) == dict(slow.test_set) and fast.test_set_item == slow.test_set_item
    # the weights column, straight from the library
    lib, h = _lib.load(), C.c_void_p()
    _lib.check(lib.srh_dataset_load(C.byref(h), str(tr).encode(), None))
    w = np.empty(n, dtype=np.float32)
    _lib.check(lib.srh_dataset_copy_ids(h, None, None, w.ctypes.data_as(C.c_void_p), None, None))
    lib.srh_dataset_destroy(h)
    assert np.array_equal(w, np.asarray([float(x) for x in weights], dtype=np.float32))
    bad = tmp_path / "bad.txt"
    lines = [f"{a} {b} 1" for a, b in zip(users, items)]
    lines[20011] = "two tokens"
    bad.write_text("\n".join(lines) + "\n")
    assert lib.srh_dataset_load(C.byref(h), str(bad).encode(), None) != 0
    assert b"line 20012 " in lib.srh_last_error_string()


def test_ranked_lists_reads_like_the_dict_and_reports_the_same_strings():
    """RankedLists (the array form of test()'s result) against the reference-shaped dict: same keys in the
    same order, same rows, and ranking_evaluation's vectorised branch prints exactly what the per-user
    loops print (util/evaluation.py:135-162), top-N cut shorter than K included."""
    rng = np.random.default_rng(12)
    n_users, n_items, K = 400, 300, 20
    users = [f"u{k}" for k in rng.permutation(n_users)]
    names = np.array([f"i{k}" for k in range(n_items)], dtype=object)
    origin = {}
    for u in users:
        truth = rng.choice(n_items, size=int(rng.integers(1, 40)), replace=False)
        origin[u] = {names[t]: 1 for t in truth}
    ids = np.stack([rng.choice(n_items, size=K, replace=False) for _ in users]).astype(np.int32)
    scores = np.sort(rng.standard_normal((n_users, K)).astype(np.float32))[:, ::-1].copy()
    flags = np.array([[names[i] in origin[u] for i in ids[r]] for r, u in enumerate(users)], dtype=np.uint8)
    sizes = np.array([len(origin[u]) for u in users])
    ranked = evaluation.RankedLists(users, names, ids, scores, hit_flags=flags, truth_sizes=sizes, origin=origin)
    as_dict = {u: [(names[i], float(s)) for i, s in zip(ids[r], scores[r])] for r, u in enumerate(users)}
    make = lambda: evaluation.RankedLists(users, names, ids, scores, hit_flags=flags, truth_sizes=sizes, origin=origin)
    # untouched: the vectorised branch, no row built
    assert len(ranked) == n_users and users[3] in ranked and "nobody" not in ranked
    for N in ([20], [10, 20], [1, 5, 20]):
        assert evaluation.ranking_evaluation(origin, ranked, N) == evaluation.ranking_evaluation(origin, as_dict, N)
    assert ranked.arrays_valid and dict.get(ranked, users[0]) is None
    # looked into: it is the dict, and the report then reads the rows (same strings)
    assert list(ranked) == users and len(ranked) == n_users
    assert dict(ranked) == as_dict and ranked[users[7]] == as_dict[users[7]] and ranked == as_dict
    assert not ranked.arrays_valid
    assert evaluation.ranking_evaluation(origin, ranked, [10, 20]) == evaluation.ranking_evaluation(origin, as_dict, [10, 20])
    # without flags (or against another test set) it goes through the loops like any mapping
    plain = evaluation.RankedLists(users, names, ids, scores)
    assert evaluation.ranking_evaluation(origin, plain, [10, 20]) == evaluation.ranking_evaluation(origin, as_dict, [10, 20])
    # the reference returns a dict (base/graph_recommender.py:44-58): whatever takes a dict takes this
    import copy
    import json
    import pickle
    for how in (lambda r: json.loads(json.dumps(r)), lambda r: pickle.loads(pickle.dumps(r)), copy.deepcopy, copy.copy,
                lambda r: {**r}, lambda r: dict(r.items()), lambda r: r | {}, lambda r: {} | r, lambda r: r.copy(), dict,
                lambda r: json.loads(json.JSONEncoder().encode(r)), lambda r: json.loads("".join(json.JSONEncoder().iterencode(r)))):
        got = how(make())
        assert type(got) is dict
        assert {u: [tuple(c) for c in row] for u, row in got.items()} == as_dict
    fresh = make()
    assert isinstance(fresh, dict) and list(fresh.keys()) == users and list(fresh.values())[5] == as_dict[users[5]]
    assert fresh.get("nobody") is None and fresh.get(users[2]) == as_dict[users[2]] and list(reversed(fresh)) == users[::-1]
    assert repr(make()) == repr(as_dict)
    # in-place edits behave as on the reference's dict, and the report follows them
    edited, want = make(), {u: list(r) for u, r in as_dict.items()}
    edited[users[0]] = want[users[0]] = as_dict[users[1]]
    edited[users[4]].reverse(); want[users[4]].reverse()
    assert edited.pop(users[9]) == want.pop(users[9]) and edited == want and len(edited) == n_users - 1
    cut = {u: t for u, t in origin.items() if u != users[9]}
    assert evaluation.ranking_evaluation(cut, edited, [10, 20]) == evaluation.ranking_evaluation(cut, want, [10, 20])
    other = make()
    del other[users[1]]
    other.update({"x": []}); other.setdefault("y", [1]); other |= {"z": []}
    assert "x" in other and other["y"] == [1] and "z" in other and users[1] not in other and len(other) == n_users + 2
    other.clear()
    assert not other and len(other) == 0


def test_reclist_builder_checks_its_arguments():
    """selfrec_amd/_reclist.build (csrc/reclist.c): the C construction of test()'s rec_list refuses ids outside the name
    table and arrays of the wrong size instead of reading past them; rows come back in `users` order as plain lists of
    (name, python float) tuples."""
    from selfrec_amd import _reclist
    users, names = ["a", "b", "c"], ["i0", "i1", "i2", "i3"]
    ids = np.array([[3, 0], [1, 1], [2, 3]], dtype=np.int32)
    sc = np.array([[0.5, 0.25], [2.0, -1.0], [0.0, 1e-9]], dtype=np.float32)
    got = _reclist.build(users, names, ids, sc, 2)
    assert type(got) is dict and list(got) == users
    assert got["a"] == [("i3", 0.5), ("i0", 0.25)] and got["c"][1] == ("i3", float(np.float32(1e-9)))
    assert all(type(row) is list and type(row[0]) is tuple and type(row[0][1]) is float for row in got.values())
    bad = ids.copy(); bad[1, 0] = 4
    with pytest.raises(IndexError):
        _reclist.build(users, names, bad, sc, 2)
    bad[1, 0] = -1
    with pytest.raises(IndexError):
        _reclist.build(users, names, bad, sc, 2)
    with pytest.raises(ValueError):
        _reclist.build(users, names, ids, sc, 3)
    with pytest.raises(ValueError):
        _reclist.build(users, names, ids[:2], sc, 2)
    with pytest.raises(TypeError):
        _reclist.build(tuple(users), names, ids, sc, 2)


def test_xcd_share_calibration_controller(monkeypatch):
    """engine.FusedTrainer._calibrate_xcd_shares against a simulated chip: XCD k takes speed[k] us per workgroup of its
    queue.  The controller must move workgroups from the late XCDs to the early ones until they finish together, keep the
    total, stay inside its clip range, remember the result on the plan, and do nothing when told not to."""
    import types

    import numpy as np
    from selfrec_amd import engine, ops

    nb = 7140
    canon = np.array([len(range(k, nb, 8)) for k in range(8)])
    speed = 43.0 / 893 * np.array([0.97, 0.95, 1.06, 1.03, 0.99, 0.97, 1.04, 1.00])      # (the shape measured on an MI355X)
    state = {"shares": canon.copy(), "probes": 0, "sets": []}

    def fake_probe(csr, x, out, epilogue=None, pattern=False):
        state["probes"] += 1
        rng = np.random.default_rng(state["probes"])
        return state["shares"] * speed + rng.normal(0, 0.05, 8), None, None, None

    def fake_set(csr, d, blocks=None):
        state["shares"] = canon.copy() if blocks is None else np.asarray(blocks).copy()
        state["sets"].append(None if blocks is None else list(map(int, blocks)))
    monkeypatch.setattr(ops, "spmm_probe", fake_probe)
    monkeypatch.setattr(ops, "spmm_set_xcd_shares", fake_set)
    monkeypatch.setattr(ops, "spmm_plan_run_tasks", lambda csr, d: nb * 4 - 1)
    monkeypatch.delenv("SRH_XCD_CALIBRATE", raising=False)

    def trainer(single=True, **over):
        t = types.SimpleNamespace(dev=types.SimpleNamespace(type="cuda"), place=types.SimpleNamespace(single_gpu_step=single),
                                  L=3, d=64, model="LightGCN", vfree=False, eps=0.2, dinv=None, E0=None, Ha=None,
                                  adj=types.SimpleNamespace())
        t.__dict__.update(over)
        return t
    before = (canon * speed).max() - (canon * speed).min()
    t = trainer()
    got = engine.FusedTrainer._calibrate_xcd_shares(t)
    fin = state["shares"] * speed
    assert got is not None and np.array_equal(got, state["shares"]) and int(got.sum()) == nb
    assert fin.max() - fin.min() < 0.8 < before and fin.max() < (canon * speed).max() - 1.0
    assert (got >= canon * 6 // 10).all() and (got <= canon * 14 // 10 + 8).all()
    assert got[2] < canon[2] and got[1] > canon[1]                        # the slow XCD gave, the fast one took
    # the plan remembers: a second trainer on the same matrix does not probe again
    n = state["probes"]
    t2 = trainer(adj=t.adj)
    again = engine.FusedTrainer._calibrate_xcd_shares(t2)
    assert state["probes"] == n and np.array_equal(again, got)
    # ... and what was decided is on the record of both trainers
    rec = t.xcd_calibration
    assert rec["shares"] == [int(v) for v in got] and rec["canonical"] == [int(v) for v in canon] and t2.xcd_calibration == rec
    assert rec["xcd_finish_spread_us_by_round"][0] > rec["xcd_finish_spread_us_by_round"][-1]
    # switched off, sharded layouts, unsupported widths, small graphs: nothing happens
    monkeypatch.setenv("SRH_XCD_CALIBRATE", "0")
    assert engine.FusedTrainer._calibrate_xcd_shares(trainer()) is None and state["probes"] == n
    monkeypatch.delenv("SRH_XCD_CALIBRATE")
    for over in (dict(single=False), dict(d=32), dict(L=0), dict(dev=types.SimpleNamespace(type="cpu"))):
        assert engine.FusedTrainer._calibrate_xcd_shares(trainer(**over)) is None and state["probes"] == n
    monkeypatch.setattr(ops, "spmm_plan_run_tasks", lambda csr, d: 4000)
    assert engine.FusedTrainer._calibrate_xcd_shares(trainer()) is None and state["probes"] == n



def test_loss_mirrors_evaluate_the_reference_expression_off_the_device(golden_ops):
    """SURVEY.md 8b / VERDICT r02 missing #5: util.loss_torch.{bpr_loss, l2_reg_loss, InfoNCE} called with CPU tensors (or
    float64) behave as the reference's functions do -- value and autograd gradient, against the golden values the
    REFERENCE produced for the same inputs (tests/golden/make_golden.py) and against the oracle's restatement.  HIP fp32
    tensors never take this branch (tests/test_gpu_dropin.py drives that one)."""
    import torch
    from selfrec_amd.util import loss_torch as L
    g = torch.Generator().manual_seed(3)
    for n, d in ((130, 64), (17, 50)):
        u, p, q = (torch.randn((n, d), generator=g, requires_grad=True) for _ in range(3))
        ou, op, oq = (t.detach().clone().requires_grad_() for t in (u, p, q))
        got = L.bpr_loss(u, p, q) + L.l2_reg_loss(1e-4, u, p) + 0.2 * L.InfoNCE(u, p, 0.2)
        want = O.bpr_loss(ou, op, oq) + O.l2_reg_loss(1e-4, ou, op) + 0.2 * O.info_nce(ou, op, 0.2)
        got.backward(); want.backward()
        assert abs(got.item() - want.item()) <= 1e-6 * abs(want.item())
        for a, b in ((u, ou), (p, op), (q, oq)):
            np.testing.assert_allclose(a.grad.numpy(), b.grad.numpy(), rtol=1e-5, atol=1e-8)
    # float64 input stays float64 (the kernels are fp32-only: this is the expression, not a cast)
    x = torch.randn((8, 64), generator=g, dtype=torch.float64)
    assert L.InfoNCE(x, x.flip(0), 0.2).dtype == torch.float64 and L.bpr_loss(x, x, x.flip(0)).dtype == torch.float64
    # the reference's own values and gradients for its own inputs (tests/golden/make_golden.py: duplicate rows, n = 1)
    for n in (1, 2, 130, 515):
        u, p, q = (torch.tensor(x, requires_grad=True) for x in golden_ops[f"ops_{n}_in"])
        bpr, reg, nce = L.bpr_loss(u, p, q), L.l2_reg_loss(1e-4, u, p, q), L.InfoNCE(u, p, 0.2)
        np.testing.assert_allclose([bpr.item(), reg.item(), nce.item()], golden_ops[f"ops_{n}_loss"], rtol=1e-6)
        gn = torch.stack(torch.autograd.grad(nce, (u, p))).numpy()
        np.testing.assert_allclose(gn, golden_ops[f"ops_{n}_g_nce"], rtol=1e-5, atol=1e-9)
        gb = torch.stack(torch.autograd.grad(bpr, (u, p, q))).numpy()
        np.testing.assert_allclose(gb, golden_ops[f"ops_{n}_g_bpr"], rtol=1e-5, atol=1e-9)


def test_fused_dropin_recognises_only_the_unmodified_reference_files(tmp_path, monkeypatch, capsys):
    """dropin.install(fuse=True): a model class gets the fused engine's train() iff its module is model.graph.<Name> and
    the file is the reference's -- by the SHA-256 of its bytes or of its syntax tree (comments and formatting aside); one
    edited STATEMENT keeps the file's own train(), and stderr says so.  The
    reference checkout exists in the build container only -- skipped elsewhere (the GPU-side run of the fused route is
    tools/run_reference_models.py --fuse on a staged copy)."""
    import importlib
    import shutil
    import sys
    ref = "/root/reference/model/graph"
    if not os.path.isfile(os.path.join(ref, "XSimGCL.py")):
        pytest.skip("no reference checkout here")
    from selfrec_amd import dropin
    from selfrec_amd.model.graph._fused import fused_train_of_reference_class
    stage = tmp_path / "stage"
    (stage / "model" / "graph").mkdir(parents=True)
    for name in dropin.FUSABLE:
        shutil.copy(os.path.join(ref, f"{name}.py"), stage / "model" / "graph" / f"{name}.py")
    with open(stage / "model" / "graph" / "SimGCL.py", "a") as f:          # an edited file: one more STATEMENT
        f.write("\nLOCAL_CHANGE = 1\n")
    with open(stage / "model" / "graph" / "SGL.py", "a") as f:             # a comment and blank lines: still the reference's code
        f.write("\n\n# re-formatted upstream\n")
    for d in (stage / "model", stage / "model" / "graph"):
        (d / "__init__.py").write_text("")
    monkeypatch.syspath_prepend(str(stage))
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("base", "data", "util", "model")}
    for k in saved:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.setitem(dropin._state, "fuse", False)
    monkeypatch.setitem(dropin._state, "fused", [])
    monkeypatch.setitem(dropin._state, "matched", {})
    try:
        dropin.install(fuse=True)
        for name in ("XSimGCL", "LightGCN", "SGL", "MF"):
            cls = getattr(importlib.import_module(f"model.graph.{name}"), name)
            assert cls.train is fused_train_of_reference_class and callable(cls._reference_train), name
        cls = importlib.import_module("model.graph.SimGCL").SimGCL
        assert cls.train is not fused_train_of_reference_class and not hasattr(cls, "_reference_train")
        assert sorted(dropin._state["fused"]) == ["LightGCN", "MF", "SGL", "XSimGCL"]
        assert dropin._state["matched"]["SGL"] == "syntax" and dropin._state["matched"]["XSimGCL"] == "bytes"
        assert "SimGCL differs from the reference's" in capsys.readouterr().err          # (said, not silent)
        # without fuse: nothing is rerouted
        for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[k]
        dropin.install(fuse=False)
        cls = importlib.import_module("model.graph.XSimGCL").XSimGCL
        assert cls.train is not fused_train_of_reference_class
        # this package's own classes are never rerouted (their module is selfrec_amd.model.graph.*)
        from selfrec_amd.model.graph.XSimGCL import XSimGCL as Own
        assert Own.train is not fused_train_of_reference_class
    finally:
        dropin.uninstall()
        for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[k]


def test_reordered_training_list_rebuilds_the_sampler(golden_ops):
    """`data.training_data` edited or re-ordered by the caller between epochs -- anywhere, not only at the
    64 probed positions -- must be noticed (full identity fingerprint of the records), and the next epoch must read
    the list as the reference's generator would."""
    from tests.conftest import _tiny_interaction
    data = _tiny_interaction(golden_ops)
    random.seed(5)
    first = [b for b in next_batch_pairwise(data, 512)]
    smp0 = data._srh_sampler[0]
    random.seed(6)
    list(next_batch_pairwise(data, 512))
    assert data._srh_sampler[0] is smp0                       # untouched list: the sampler is reused
    td = data.training_data
    n = len(td)
    probed = set(np.unique(np.linspace(0, n - 1, num=64).astype(np.int64)).tolist())
    a, b = [k for k in range(n) if k not in probed and k + 1 not in probed][:2]
    b += 7
    assert a not in probed and b not in probed and td[a][:2] != td[b][:2]
    td[a], td[b] = td[b], td[a]                               # a swap the value probes cannot see
    want_u = [data.user[r[0]] for r in td]
    random.seed(7)
    gen = next_batch_pairwise(data, 512)
    u0, i0, j0 = next(gen)
    assert data._srh_sampler[0] is not smp0                   # noticed: rebuilt from the list
    # ... and the epoch is what the reference's generator yields for this list: shuffle(list) then slices of it
    import copy
    ref = copy.deepcopy([r for r in data.training_data])      # (already shuffled in place by the generator)
    assert u0 == [data.user[r[0]] for r in ref[:512]] and i0 == [data.item[r[1]] for r in ref[:512]]
    assert sorted(want_u) == sorted(data.user[r[0]] for r in data.training_data)
    assert len(first) == (n + 511) // 512


def test_torch_cpu_generator_replay_is_bit_exact():
    """util/torch_rng (srh_mt19937_uniform_f32): the uniforms, the BUIR.py:118-121 keep mask and the generator state
    left behind equal torch.rand's -- fresh seed, mid-block, across block boundaries, interleaved with torch's own draws."""
    import time
    import torch
    from selfrec_amd.util import torch_rng
    for seed, sizes in ((0, (1, 623, 1, 624, 5, 100_000)), (123, (700, 3, 1248)), (7, (2_470_000,))):
        torch.manual_seed(seed)
        want = []
        for k, n in enumerate(sizes):
            want.append(torch.rand(n))
            if k % 2:
                want.append(torch.randn(3))                  # other consumers of the generator in between
        end_state = torch.get_rng_state()
        torch.manual_seed(seed)
        got = []
        for k, n in enumerate(sizes):
            got.append(torch_rng.rand(n))
            if k % 2:
                got.append(torch.randn(3))
        assert all(torch.equal(a, b) for a, b in zip(want, got))
        assert torch.equal(torch.get_rng_state(), end_state)
    rs = np.random.RandomState(5)
    for rate in [0.0, 0.05, 0.1, 0.5, 1.0 - 2.0 ** -25] + (rs.random_sample(20) * 0.3).tolist():
        torch.manual_seed(11)
        want = torch.floor(1 - rate + torch.rand(50_001)).type(torch.bool)
        state = torch.get_rng_state()
        torch.manual_seed(11)
        got = torch_rng.keep_mask(50_001, 1 - rate)
        assert got.dtype == torch.bool and torch.equal(want, got), rate
        assert torch.equal(torch.get_rng_state(), state)
    # a patched torch.rand (tests injecting their own noise) is what gets called
    real = torch.rand
    try:
        torch.rand = lambda n: torch.full((n,), 0.25)
        assert torch_rng.keep_mask(8, 0.8).tolist() == [True] * 8 and torch_rng.keep_mask(8, 0.7).tolist() == [False] * 8
    finally:
        torch.rand = real
    torch.manual_seed(1)
    t0 = time.perf_counter(); torch_rng.keep_mask(2_470_000, 0.95); fast = time.perf_counter() - t0
    t0 = time.perf_counter(); torch.floor(0.95 + torch.rand(2_470_000)).type(torch.bool); slow = time.perf_counter() - t0
    print(f"keep mask of 2.47 M entries: replay {1e3 * fast:.1f} ms, ATen {1e3 * slow:.1f} ms")


def test_dropin_fast_paths_make_the_entry_class_load_lazily(tmp_path):
    """dropin.install(): ``data.loader.FileIO.load_data_set(path, 'graph')`` -- what the reference's SELFRec.py:12-13 calls --
    returns the lazy TripleFile (equal to the list as soon as anybody reads it); without the fast paths, and after
    uninstall(), the plain list of the reference."""
    import importlib
    from selfrec_amd import dropin
    from selfrec_amd.data.loader import TripleFile
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    tr = tmp_path / "train.txt"
    synth.write_text(str(tr), tu, ti)
    want = FileIO.load_data_set(str(tr), "graph")
    assert type(want) is list
    for fast in (True, False):
        dropin.install(fuse=False, fast=fast)
        try:
            got = importlib.import_module("data.loader").FileIO.load_data_set(str(tr), "graph")
            assert isinstance(got, TripleFile) == fast
            if fast:
                assert got.unread() and Interaction({}, got, []).training_size()[2] == len(want) and got.unread()
            assert list(got) == want
            # (a sequential file, a missing file: the reference's behaviour)
            with pytest.raises(FileNotFoundError):
                importlib.import_module("data.loader").FileIO.load_data_set(str(tmp_path / "none.txt"), "graph")
        finally:
            dropin.uninstall()
    assert type(FileIO.load_data_set(str(tr), "graph")) is list


def test_lazy_training_file_keeps_the_shuffles_as_a_pending_permutation(tmp_path):
    """data/loader.TripleFile + util/sampler: on a dataset opened lazily the in-place shuffles of sampler.py:7 are carried
    as one permutation until somebody reads ``data.training_data``; batches, the list a reader then finds, and every later
    epoch equal the plain-list run's (which the goldens pin against the reference's generator)."""
    tu, ti, su, si, U, I = synth.make_dataset("small")
    tr, te = tmp_path / "train.txt", tmp_path / "test.txt"
    synth.write_text(str(tr), tu, ti)
    synth.write_text(str(te), su, si)
    plain = Interaction({}, FileIO.load_data_set(str(tr), "graph"), FileIO.load_data_set(str(te), "graph"))
    lazy = Interaction({}, FileIO.open_data_set(str(tr), "graph"), FileIO.open_data_set(str(te), "graph"))
    assert lazy.training_size() == plain.training_size() and lazy.test_size() == plain.test_size()
    assert lazy.training_data.unread() and lazy.test_data.unread()          # len() came from the loader's line counts
    runs = {}
    for name, data in (("plain", plain), ("lazy", lazy)):
        random.seed(31)
        out = []
        for epoch in range(4):
            out.append([b for b in next_batch_pairwise(data, 300, n_negs=2 if epoch == 1 else 1)])
            if name == "lazy" and epoch < 2:
                assert data.training_data.unread()                           # two epochs without a python triple
            if epoch == 1:
                snapshot = [list(r) for r in data.training_data]             # a reader: materialises the lazy list
                assert not getattr(data.training_data, "unread", lambda: False)()
                out.append(snapshot)
            if epoch == 0:
                smp = data._srh_sampler[0]
        assert data._srh_sampler[0] is smp                                   # (no rebuild after the materialisation)
        out.append([list(r) for r in data.training_data])
        out.append(random.getstate())
        runs[name] = out
    assert runs["plain"] == runs["lazy"]
    # array form: same values
    random.seed(31)
    a = [tuple(x.tolist() for x in b) for b in next_batch_pairwise(plain, 300, as_arrays=True)]
    random.seed(31)
    l = [b for b in next_batch_pairwise(lazy, 300)]
    assert a == l


def test_fused_dropin_recognises_the_reference_files_by_their_code_not_their_bytes():
    """dropin.maybe_fuse matches a model file by the SHA-256 of its bytes OR of its canonical syntax dump: comments, blank
    lines, line breaks, quote style and docstrings do not enter the second one; any changed statement does."""
    from selfrec_amd import dropin
    a = 'import torch\n\nclass M(Base):\n    """doc"""\n    def train(self):\n        x = self.f(1, 2)  # comment\n        return x + 1\n'
    b = ("# licence header\nimport torch\nclass M(Base):\n\n    def train( self ):\n        '''another doc'''\n"
         "        x = self.f(1,\n                   2)\n        return x + 1\n")
    c = a.replace("x + 1", "x + 2")
    d = a.replace("def train", "def train2")
    assert dropin.syntax_digest(a) == dropin.syntax_digest(b)
    assert len({dropin.syntax_digest(t) for t in (a, c, d)}) == 3
    # the table holds both digests for the five files the engine serves
    assert set(dropin.FUSABLE) == {"XSimGCL", "LightGCN", "SimGCL", "SGL", "MF"}
    assert all(len(v) == 2 and all(len(h) == 64 for h in v) for v in dropin.FUSABLE.values())
    ref = "/root/reference/model/graph"
    if os.path.isdir(ref):                      # (the build container: the digests ARE the reference's)
        import hashlib
        for name, (raw_sha, syn_sha) in dropin.FUSABLE.items():
            src = open(os.path.join(ref, name + ".py"), "rb").read()
            assert hashlib.sha256(src).hexdigest() == raw_sha and dropin.syntax_digest(src.decode()) == syn_sha
            assert dropin.syntax_digest("# re-formatted upstream\n" + src.decode().replace("\n\n", "\n\n\n")) == syn_sha
