"""The fused engine (HIP kernels through the C ABI) against REFERENCE runs at the BASELINE.json config shapes
(tests/golden/shapes.npz; make_golden_shapes.py):

  configs[1]/[2]  Yelp2018 shape, LightGCN L=3 / XSimGCL L=3 (tau 0.2, eps 0.2, l*=1, injected noise): 2 steps --
                  sampled batches, per-step losses, 1,024 sampled parameter / final-embedding rows, test() ranking
                  of 64 users;
  configs[4]      iFashion shape, SGL edge-drop views: keep-sets (host) + 1 step;
  configs[0]      the real douban-book file, MF + BPR: 3 steps + test() + ranking_evaluation;
  a-13            SGL with node dropout (aug_type 0): 3 steps; node-dropped Laplacian on the device;
  ADVICE r01      duplicate interactions: weight 2 in norm_adj, unit weights in the dropped views.
Tolerances: indices bit-exact; losses 1e-5 (InfoNCE 2e-5, split-bf16 products; 2e-6 on the exact-f32 path); final
embeddings 1e-4 relative (north_star) at the Yelp2018 / douban shapes.
Parameters AFTER Adam steps are a different matter: Adam divides by sqrt(v) + 1e-8, and most rows of these graphs are
far from the batch with |g| ~ 1e-9, where fp32 summation-ORDER noise in g moves the update by a fraction of a per cent of
lr.  The reference's own arithmetic shows it: the same torch-CPU step with 8 threads instead of 1 differs from itself by
8.5e-7 absolute = 1.6e-4 of the parameter scale on the iFashion-shape SGL run (tools/precision_probe.py,
profiles/r02_a_precision_probe.txt; exact-f32 InfoNCE does not change ours).  So parameters are held to an ABSOLUTE
1e-5 (1 % of one Adam step, lr = 1e-3) everywhere, plus 2e-4 relative at the Yelp2018 shape; the iFashion-shape
embeddings (one step from a +-0.0045 xavier init, i.e. a step is 22 % of the value range) to 1e-3 relative."""
import json
import os
import random

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from selfrec_amd import ops, synth
from selfrec_amd.data.loader import FileIO
from selfrec_amd.data.ui_graph import Interaction
from selfrec_amd.engine import FusedTrainer
from tests.test_shapes_cpu import GOLDEN, seeded_init, sha, write_douban

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def shapes():
    return np.load(os.path.join(GOLDEN, "shapes.npz"))


@pytest.fixture(scope="module")
def smeta():
    with open(os.path.join(GOLDEN, "shapes_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def yelp_data():
    tu, ti, su, si, U, I = synth.make_dataset("yelp2018", seed=2024)
    return Interaction({}, synth.as_triples(tu, ti), synth.as_triples(su, si))


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


def trainer_for(info, data, ue, ie, **over):
    c = info["conf"]
    gen = torch.Generator().manual_seed(info["noise_seed"])
    kw = dict(model=info["model"], n_layers=int(c.get("n_layer", 0)), lr=info["lr"], reg=info["reg"],
              cl_rate=float(c.get("lambda", 0.0)), eps=float(c.get("eps", 0.0)), tau=float(c.get("tau", c.get("temp", 0.2))),
              layer_cl=int(c.get("l_star", 1)), drop_rate=float(c.get("drop_rate", 0.1)), aug_type=int(c.get("aug_type", 1)),
              batch_size=info["batch"], user_emb=ue, item_emb=ie, noise_fn=lambda shape: torch.rand(shape, generator=gen))
    kw.update(over)
    return FusedTrainer(data, info["emb"], **kw)


class PreAdamProbe:
    """north_star's 1e-4 where 1e-4 is meaningful (VERDICT r02 "next" #2): the quantities of the FIRST step that sit
    before Adam's 1 / (sqrt(v) + 1e-8) amplifier -- every layer output of every encoder pass, as the reference's model
    appends it to its layer list, and d loss / d embedding_dict[*] after backward() -- on the golden's sampled rows
    (512 random users / items + the first 256 distinct users / items of batch 1).  The production step is probed, not a
    test-only path: hipGraph off (noise is injected), activity marks ON (the last layer of a pass exists on the batch's
    rows only, so it is compared on the sampled rows the batch marked), value-free products ON (tables kept pre-scaled by
    D^-1/2 are un-scaled here).  An element of a PERTURBED layer whose product is within 1e-6 of zero takes sign(h) from
    the summation order (XSimGCL.py:90-91: h + sign(h) * eps * unit) -- the reference run at another thread count flips
    the same elements; they are excluded and counted (|h| < 1e-8, i.e. ~100x the rounding noise of these sums; must stay
    under 0.01 % of the sample)."""

    def __init__(self, tag, shapes, info, tr, outlier_frac=0.0):
        from selfrec_amd import engine
        self.tag, self.shapes, self.info, self.tr, self.engine = tag, shapes, info, tr, engine
        # A sign-ambiguous element of layer k (see above) is one element of one row -- but the NEXT product spreads its
        # 2 eps |unit| jump over that row's neighbours (one column each), and the backward products further.  On a graph
        # with 1.9e8 elements per layer (the 1 M x 500 k shape) a handful of them exist and every later layer multiplies
        # the rows they reach (measured there: 0 of 98 k sampled elements off at layer 1, 1 at layer 2, 9 of 33 k at layer
        # 3, none above 6e-4): the elementwise bound is asserted for all but `outlier_frac` of the compared elements, and
        # the outliers are bounded by the jump itself.  0 elsewhere (Yelp2018 / iFashion shapes: no such element).
        self.outlier_frac = float(outlier_frac)
        U = tr.U
        self.ru = shapes[f"{tag}_pre_rows_user"].astype(np.int64)
        self.ri = shapes[f"{tag}_pre_rows_item"].astype(np.int64)
        self.rows = torch.from_numpy(np.concatenate([self.ru, U + self.ri])).to(DEV)
        self.noise, self.grad = [], None
        inner = tr.noise_fn
        if inner is not None:
            def noise_fn(shape):
                t = inner(shape)
                self.noise.append(torch.as_tensor(t)[self.rows.cpu()].clone())
                return t
            tr.noise_fn = noise_fn
        # (the gradient exists in memory only when the optimiser is a pass of its own: the first step runs that way, the
        # rest -- from a re-captured graph -- with Adam in the last backward product's epilogue, engine.py: fuse_adam)
        self._fuse_adam, tr.fuse_adam = tr.fuse_adam, False
        self._real_adam = engine.ops.adam_step

        def adam_step(param, grad, *a, **k):
            if self.grad is None:
                self.grad = grad.clone()
            return self._real_adam(param, grad, *a, **k)
        engine.ops.adam_step = adam_step

    def _assert_close(self, got, want, ok, what):
        err = np.abs(got - want)[ok] / np.abs(want).max()
        bad = err >= 1e-4
        assert bad.sum() <= self.outlier_frac * ok.sum(), (what, float(err.max()), int(bad.sum()), int(ok.sum()))
        # (an outlier is a neighbour of a sign-ambiguous element: its error is that element's jump times one edge weight)
        assert err.max() < 2e-2 and np.median(err) < 2e-6, (what, float(err.max()), float(np.median(err)))

    def passes(self):
        """[(layer tables, noise call of layer 0 or None)] in the order the reference's step runs its encoder passes"""
        tr, m = self.tr, self.tr.model
        if m in ("XSimGCL", "LightGCN"):
            return [(tr.Y, 0 if m == "XSimGCL" else None)]
        if m == "SimGCL":
            return [(tr.Y, None), (tr.views[0]["Y"], 0), (tr.views[1]["Y"], tr.L)]
        return [(tr.Y, None), (tr.views[0]["Y"], None), (tr.views[1]["Y"], None)]       # SGL

    def after_first_step(self):
        tr, sh, tag = self.tr, self.shapes, self.tag
        self.engine.ops.adam_step = self._real_adam
        tr.fuse_adam, tr._graph = self._fuse_adam, None
        assert tr.step_count == 1 and self.grad is not None
        rows, nu = self.rows, len(self.ru)
        live = (tr.mark[rows] == 1).cpu().numpy()                   # rows of batch 1 (stamp = optimiser step = 1)
        if f"{tag}_pre_n_random" in sh:
            n_rand_u, n_rand_i = (int(v) for v in sh[f"{tag}_pre_n_random"])
        else:
            n_rand_u, n_rand_i = len(sh[f"{tag}_rows_user"]), len(sh[f"{tag}_rows_item"])
        assert live[n_rand_u:nu].all() and live[nu + n_rand_i:].all()      # (the batch-row part of the sample is marked)
        dinv = tr.dinv[rows].cpu().numpy()[:, None] if tr.vfree else None
        k_ref, flipped, compared = 0, 0, 0
        for tables, call0 in self.passes():
            for k, t in enumerate(tables):
                got = t[rows].cpu().numpy().astype(np.float64)
                assert not got[:, tr.d_valid:].any()                # (zero-padded rows: the padding stays zero)
                got = got[:, :tr.d_valid]
                if dinv is not None and k < tr.L - 1:
                    got = got / dinv                                # stored as D^-1/2 Y (engine.py: value-free products)
                want = np.concatenate([sh[f"{tag}_pre_layer{k_ref}_user"], sh[f"{tag}_pre_layer{k_ref}_item"]]).astype(np.float64)
                k_ref += 1
                ok = np.ones(got.shape, dtype=bool)
                if k == tr.L - 1:
                    ok &= live[:, None]                             # last layer: computed on the batch's rows only
                if call0 is not None:
                    unit = torch.nn.functional.normalize(self.noise[call0 + k], dim=-1).numpy().astype(np.float64)
                    ambiguous = np.abs(np.abs(want) - tr.eps * unit) < 1e-8
                    flipped += int((ambiguous & ok).sum())
                    ok &= ~ambiguous
                compared += int(ok.sum())
                self._assert_close(got, want, ok, (tag, "layer output", k_ref - 1))
        assert flipped <= 1e-4 * compared, (flipped, compared)
        g = self.grad[rows].cpu().numpy().astype(np.float64)
        assert not g[:, tr.d_valid:].any()
        g = g[:, :tr.d_valid]
        for got, key in ((g[:nu], "user"), (g[nu:], "item")):
            want = sh[f"{tag}_pre_grad_{key}"].astype(np.float64)
            self._assert_close(got, want, np.ones(got.shape, dtype=bool), (tag, "gradient before Adam", key))
        self.grad = None
        return {"elements": compared, "sign_ambiguous": flipped}


def run_and_check(tag, shapes, info, tr, *, rows=True, nce_rtol=2e-5, param_rtol=2e-4, emb_rtol=1e-4, outliers=0.0,
                  pre_adam=True, pre_adam_outliers=0.0, param_atol=1e-5, param_outliers=0):
    """Seed the sampler like the reference run, train its steps, compare everything the golden holds.
    outliers > 0 (the 1.5 M-node shape): the element-wise bounds hold for all but that fraction of the sampled
    elements, and every element stays within 5 % of one Adam step -- see the test that uses it.
    pre_adam: goldens that hold first-step layer outputs and pre-Adam gradients are held to 1e-4 on them (PreAdamProbe)."""
    probe = (PreAdamProbe(tag, shapes, info, tr, outlier_frac=pre_adam_outliers)
             if pre_adam and f"{tag}_pre_rows_user" in shapes else None)
    random.seed(info["sampler_seed"])
    tr.seed_sampler_from_python()
    tr.begin_epoch()
    sizes = shapes[f"{tag}_batch_sizes"]
    n = int(sizes.sum())
    eu, ei, ej = tr.epoch_node_ids()
    for got, col in ((eu, "u"), (ei, "i"), (ej, "j")):                  # bit-exact index streams
        assert np.array_equal(got[:n], shapes[f"{tag}_batch_{col}"])
    losses = []
    for k in range(len(sizes)):
        tr.step()
        losses.append(tr.read_losses())
        if k == 0 and probe is not None:
            probe.after_first_step()
    losses = np.asarray(losses)
    np.testing.assert_allclose(losses[:, 0], shapes[f"{tag}_loss_bpr"], rtol=1e-5)
    reg_div = info["batch"] if info["model"] in ("MF", "LightGCN") else 1.0
    np.testing.assert_allclose(losses[:, 1], shapes[f"{tag}_loss_reg"] / reg_div, rtol=1e-5)
    nce = shapes[f"{tag}_loss_nce"]
    if nce.size:
        np.testing.assert_allclose(losses[:, 2], nce.reshape(len(sizes), -1).sum(1) * tr.cl_rate, rtol=nce_rtol)
    ru = torch.from_numpy(shapes[f"{tag}_rows_user"].astype(np.int64)).to(DEV) if rows else slice(None)
    ri = torch.from_numpy(shapes[f"{tag}_rows_item"].astype(np.int64)).to(DEV) if rows else slice(None)
    pu, pi = tr.user_emb[ru].cpu().numpy(), tr.item_emb[ri].cpu().numpy()
    fu, fi = tr.embeddings()
    if outliers:
        for got, key in ((pu, "param_user"), (pi, "param_item")):
            diff = np.abs(got - shapes[f"{tag}_{key}"])
            assert (diff > 1e-6).mean() < outliers and diff.max() < 0.05 * info["lr"], (key, (diff > 1e-6).mean(), diff.max())
        for got, key in ((fu[ru].cpu().numpy(), "final_user"), (fi[ri].cpu().numpy(), "final_item")):
            want = shapes[f"{tag}_{key}"]
            diff = np.abs(got - want) / np.abs(want).max()
            assert (diff > emb_rtol).mean() < outliers and np.median(diff) < 1e-6, (key, (diff > emb_rtol).mean(), np.median(diff), diff.max())
        return fu, fi
    if param_outliers:
        # (the caller names how many of the 65,536 sampled parameters may sit where Adam's division by sqrt(v) + 1e-8 turns a
        # 1e-10 difference of a ~1e-9 gradient into per cents of a step: those stay within 5 % of one step, all others in
        # the bounds below)
        for got, key in ((pu, "param_user"), (pi, "param_item")):
            want = shapes[f"{tag}_{key}"]
            diff = np.abs(got - want)
            bad = (diff >= param_atol) | (diff >= param_rtol * np.abs(want).max())
            assert int(bad.sum()) <= param_outliers and diff.max() < 0.05 * info["lr"], (key, int(bad.sum()), diff.max())
    else:
        assert rel_err(pu, shapes[f"{tag}_param_user"]) < param_rtol and rel_err(pi, shapes[f"{tag}_param_item"]) < param_rtol
        # element-wise: far inside one Adam step (lr = 1e-3)
        assert np.abs(pu - shapes[f"{tag}_param_user"]).max() < param_atol and np.abs(pi - shapes[f"{tag}_param_item"]).max() < param_atol
    assert rel_err(fu[ru].cpu().numpy(), shapes[f"{tag}_final_user"]) < emb_rtol
    assert rel_err(fi[ri].cpu().numpy(), shapes[f"{tag}_final_item"]) < emb_rtol
    return fu, fi


@pytest.mark.parametrize("tag", ["Y_XSimGCL", "Y_LightGCN", "Y_SimGCL"])
def test_yelp_shape_two_steps_match_reference_run(yelp_data, shapes, smeta, tag):
    info = smeta[tag]
    ue, ie = seeded_init(info)
    tr = trainer_for(info, yelp_data, ue, ie)
    assert f"{tag}_pre_grad_user" in shapes          # (first-step layer outputs + pre-Adam gradients: held to 1e-4)
    # the DEFAULT InfoNCE arithmetic (round 5: all-f32 MFMA products) holds the contrastive loss to 2e-6.
    # SimGCL's second step: ONE of the 65,536 sampled parameters -- item row 254, column 55 -- comes out 2.2e-5 from the
    # reference's (2 % of an Adam step; every other one within 1e-6, median 9e-10): a coordinate whose gradient is ~1e-9, where
    # Adam's g / (|g| + 1e-8) multiplies a 1e-10 difference in g by 1e5 x lr.  The split mode happens to land the other way
    # (test_yelp_shape_simgcl_split16 below holds it to the strict bounds); neither is closer to the reference elsewhere.
    fu, fi = run_and_check(tag, shapes, info, tr, nce_rtol=2e-6, param_outliers=2 if tag == "Y_SimGCL" else 0)
    if f"{tag}_eval_users" not in shapes:
        return
    # graph_recommender.py:46-53 for the golden's 64 test users: same ranked ids, same scores
    users = torch.from_numpy(shapes[f"{tag}_eval_users"]).to(DEV)
    ids, sc = ops.score_mask_topk(fu.contiguous(), users, fi.contiguous(), tr.graph.r_indptr, tr.graph.r_indices, 20)
    want_ids, want_sc = shapes[f"{tag}_eval_ids"], shapes[f"{tag}_eval_scores"]
    assert (ids.cpu().numpy() == want_ids).mean() > 0.995          # (exact ties / 1-ulp neighbours may swap)
    np.testing.assert_allclose(sc.cpu().numpy(), want_sc, rtol=1e-4, atol=1e-7)
    # ... and through the path production (and bench.py) ranks with: GraphRecommender.rank_on_device -- the FILTERED pipeline
    # with the shipped constants (bound from a 4096-item sample, split-bf16 filter pass, 1024 candidate slots, 16384-user
    # chunks, exact re-score) on these TRAINED embeddings (VERDICT r03 #4a): same ids, same scores as the reference's loop
    from selfrec_amd.base import graph_recommender as gr
    rec = gr.GraphRecommender.__new__(gr.GraphRecommender)
    rec.data, rec.max_N, rec.topN = yelp_data, 20, [20]
    rec.user_emb, rec.item_emb = fu.contiguous(), fi.contiguous()
    assert fi.shape[0] >= gr.FILTER_MIN_ITEMS                # (this catalogue takes the filtered pipeline)
    ids_f, sc_f = rec.rank_on_device(shapes[f"{tag}_eval_users"].astype(np.int32))
    assert getattr(rec, "_filter_ws", None) is not None            # (... and it did)
    assert (ids_f == want_ids).mean() > 0.995
    np.testing.assert_allclose(sc_f, want_sc, rtol=1e-4, atol=1e-7)
    # the two pipelines agree with each other bit for bit (ids and scores)
    assert np.array_equal(ids_f, ids.cpu().numpy()) and np.array_equal(sc_f, sc.cpu().numpy())
    # all test users at once, as test() asks: the golden users' rows are the same rows
    every = np.arange(yelp_data.user_num, dtype=np.int32)
    ids_all, sc_all = rec.rank_on_device(every)
    pick = shapes[f"{tag}_eval_users"].astype(np.int64)
    assert np.array_equal(ids_all[pick], ids_f) and np.array_equal(sc_all[pick], sc_f)


def test_yelp_shape_xsimgcl_exact_f32_infonce(yelp_data, shapes, smeta):
    """The same run with InfoNCE's products on the exact-f32 MFMA path (the trainer's own nce_precision: no process-wide
    state): the contrastive loss agrees with the reference to 2e-6, as the default split path does."""
    info = smeta["Y_XSimGCL"]
    ue, ie = seeded_init(info)
    tr = trainer_for(info, yelp_data, ue, ie)
    before = ops.get_infonce_precision()
    tr.set_nce_precision("f32")
    assert ops.get_infonce_precision() == before == "f32"  # (the trainer's own mode: the process default -- f32 -- is untouched)
    run_and_check("Y_XSimGCL", shapes, info, tr, nce_rtol=2e-6)


def test_yelp_shape_simgcl_split16(yelp_data, shapes, smeta):
    """SimGCL's two steps with the split 16-bit InfoNCE products: every sampled parameter inside the strict bounds."""
    info = smeta["Y_SimGCL"]
    ue, ie = seeded_init(info)
    tr = trainer_for(info, yelp_data, ue, ie)
    tr.set_nce_precision("split")
    run_and_check("Y_SimGCL", shapes, info, tr, nce_rtol=2e-6)


def test_yelp_shape_xsimgcl_split16_infonce(yelp_data, shapes, smeta):
    """... and with the opt-in split 16-bit operands (the faster mode bench.py reports as value_split16): same tolerances."""
    info = smeta["Y_XSimGCL"]
    ue, ie = seeded_init(info)
    tr = trainer_for(info, yelp_data, ue, ie)
    tr.set_nce_precision("split")
    assert ops.get_infonce_precision() == "f32"
    run_and_check("Y_XSimGCL", shapes, info, tr, nce_rtol=2e-6)


@pytest.mark.parametrize("tag", ["F_SGL", "F_SGL3"])
def test_ifashion_shape_sgl_step_matches_reference_run(shapes, smeta, tag):
    """BASELINE.json configs[4]: SGL with edge-dropped views at the iFashion shape -- the shipped yaml's L = 2 and the
    config's / model matrix's L = 3.  Layer outputs of all three passes and the pre-Adam gradient: 1e-4 (PreAdamProbe)."""
    info = smeta[tag]
    tu, ti, su, si, U, I = synth.make_dataset("ifashion", seed=2024)
    data = Interaction({}, synth.as_triples(tu, ti), [])
    assert (data.user_num, data.item_num) == (info["n_users"], info["n_items"])
    ue, ie = seeded_init(info)
    tr = trainer_for(info, data, ue, ie)
    assert tr.graph.n_edges == info["n_edges"] and tr.L == int(info["conf"]["n_layer"])
    assert f"{tag}_pre_grad_user" in shapes
    run_and_check(tag, shapes, info, tr, param_rtol=2e-3, emb_rtol=1e-3)          # (see the module docstring)
    for mk, want in zip(tr._epoch_host["masks"], smeta["F_SGL"]["keep_sorted_sha"]):   # the two views of the epoch
        assert sha(np.flatnonzero(mk), np.int64) == want


def first_appearance_ids(raw):
    """ids in first-appearance order of a list (ui_graph.py:29-38), vectorised"""
    uniq, first = np.unique(raw, return_index=True)
    rank = np.empty(int(uniq.max()) + 1, dtype=np.int64)
    rank[uniq[np.argsort(first)]] = np.arange(uniq.size)
    return rank[raw]


@pytest.mark.selfcheck
@pytest.mark.parametrize("model", ["LightGCN", "XSimGCL"])
def test_xcd_share_calibration_changes_placement_not_results(yelp_data, model):
    """engine.FusedTrainer calibrates the dense plan's XCD shares at start-up (probe launches -> unequal numbers of
    workgroups per XCD).  At the Yelp2018 shape it does move workgroups, and a trainer on the calibrated list takes the
    same steps as one on the canonical list: same batches, same losses, same embeddings, bit for bit.  The decision is on the
    record (trainer.xcd_calibration, logger "selfrec_amd")."""
    kw = dict(model=model, n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1, batch_size=2048)

    def run(tr):
        tr.sampler.seed(11)
        tr.begin_epoch()
        out = []
        for _ in range(3):
            tr.step()
            out.append(tr.read_losses())
        return np.asarray(out), [t.cpu().numpy() for t in tr.embeddings()]
    torch.manual_seed(5)
    a = FusedTrainer(yelp_data, 64, **kw)
    assert a.xcd_shares is not None
    shares = np.asarray(a.xcd_shares)
    assert shares.shape == (8,) and shares.min() > 0
    la, ea = run(a)
    ops.spmm_set_xcd_shares(a.adj, 64, None)                    # back to the equal dealing (a is not used again)
    torch.manual_seed(5)
    b = FusedTrainer(yelp_data, 64, **kw)                       # (the plan is marked calibrated: b does not probe again)
    lb, eb = run(b)
    # same tasks, same sums in the same order wherever they run -- and no float atomics anywhere in the step (engine.det_scatter):
    # the same bits (rounds 1-5 held this to 2e-6 / 2e-3: the atomic scatter's last bits, amplified by XSimGCL.py:90's sign())
    assert np.array_equal(la, lb)
    for x, y in zip(ea, eb):
        assert np.array_equal(x, y)
    ops.spmm_set_xcd_shares(a.adj, 64, shares)                  # leave the module's shared graph as the engine set it


def test_1m_500k_xsimgcl_step_matches_reference_run(shapes, smeta):
    """BASELINE.json configs[3]: XSimGCL L=3, d = 128 on the synthetic 1 M x 500 k graph (40.3 M train interactions) --
    one step of the reference (8 torch threads, ~25 GB of python objects) against the fused engine on one MI355X.

    Index streams bit-exact, the three losses to 1e-5 / 2e-5.  Parameters after the Adam step: the median implied
    gradient differs by 3e-6 relative, but NOT every element can agree at this size and the test says so instead of
    hiding it in a loose max-norm: XSimGCL.py:90 adds sign(h) * noise with |noise| ~ 10x |h| at d = 128, and of the
    5.8e8 layer-output elements a few hundred sit within rounding of zero, where the summation order decides the
    sign; each such element moves one column of a few batch rows' gradients, which the backward products spread
    over every node, and where the total is ~1e-9 Adam's g / (|g| + 1e-8) turns that into 1-2 % of a step.
    tools/b_probe.py shows the signature (the 12 largest differences sit in one or two COLUMNS, across unrelated
    rows) and the control: the engine against ITSELF with the products' rounding order changed (pattern products +
    row scaling vs. value products) differs the same way -- 8 of 65,536 sampled elements by > 1e-6, one by 1.1e-5.
    So: all but 0.2 % of the sampled elements within 1e-6 absolute, every element within 5 % of one Adam step."""
    if "B_XSimGCL" not in smeta:
        pytest.skip("golden section B not generated")
    info = smeta["B_XSimGCL"]
    tu, ti, su, si, U, I = synth.make_dataset("1m-500k", seed=2024)
    data = Interaction.from_id_arrays({}, first_appearance_ids(tu), first_appearance_ids(ti), np.zeros(0, np.int64),
                                      np.zeros(0, np.int64), U, I)
    del tu, ti, su, si
    assert (data.user_num, data.item_num, data.train_u.size) == (info["n_users"], info["n_items"], info["n_train"])
    ue, ie = seeded_init(info)
    tr = trainer_for(info, data, ue, ie)
    assert tr.d == 128 and tr.vfree
    assert "B_XSimGCL_pre_grad_user" in shapes       # a-4 / a-8 parity at this shape is on north_star's 1e-4 (PreAdamProbe)
    run_and_check("B_XSimGCL", shapes, info, tr, emb_rtol=1e-4, outliers=2e-3, pre_adam_outliers=1e-3)


def test_douban_book_mf_three_steps_and_ranking(tmp_path, shapes, smeta):
    """configs[0] end to end on the shipped file: native loader -> Interaction -> MF + BPR (MF.py:13-31) ->
    test() -> ranking_evaluation."""
    from selfrec_amd.base.graph_recommender import GraphRecommender
    from selfrec_amd.util.evaluation import ranking_evaluation
    train_p, test_p = write_douban(str(tmp_path))
    data = Interaction({}, FileIO.open_data_set(train_p, "graph"), FileIO.open_data_set(test_p, "graph"))
    info = smeta["D_MF"]
    ue, ie = seeded_init(info)
    tr = trainer_for(info, data, ue, ie)
    fu, fi = run_and_check("D_MF", shapes, info, tr)
    rec = GraphRecommender.__new__(GraphRecommender)
    rec.data, rec.max_N, rec.topN = data, 20, [10, 20]
    rec.user_emb, rec.item_emb = fu.contiguous(), fi.contiguous()
    out = rec.test()
    measure = ranking_evaluation(data.test_set, out, [10, 20])
    want = info["measure"]
    assert [m.split(":")[0] for m in measure] == [m.split(":")[0] for m in want]
    got_v = [float(m.split(":")[1]) for m in measure if ":" in m]
    want_v = [float(m.split(":")[1]) for m in want if ":" in m]
    np.testing.assert_allclose(got_v, want_v, atol=2e-5)            # round(.., 5) strings; a 1-ulp rank swap moves one digit
    uid = {u: k for k, u in enumerate(out.users)} if hasattr(out, "users") else None
    users = shapes["D_MF_rec_users"]
    ids, _ = rec.rank_on_device(users)
    assert (ids == shapes["D_MF_rec_ids"]).mean() > 0.995
    del uid


def test_sgl_node_dropout_steps_match_reference_run(shapes, smeta, tiny_data):
    """a-13: SGL with aug_type = 0 (augmentor.py:10-27 via SGL.py:89-96) in the fused engine, 3 reference steps."""
    info = smeta["N_SGL"]
    tr = trainer_for(info, tiny_data, shapes["N_SGL_init_user"], shapes["N_SGL_init_item"])
    assert tr.aug_type == 0
    run_and_check("N_SGL", shapes, info, tr, rows=False)


@pytest.mark.parametrize("tag", ["E_XSimGCL50", "E_XSimGCL96", "E_SGL96", "E_LightGCN20", "E_MF50"])
def test_any_embedding_size_matches_reference_run(shapes, smeta, tiny_data, tag):
    """base/recommender.py:16 takes any `embedding.size`: the fused engine stores 50 / 96 / 20 columns zero-padded to
    64 / 128 / 32 and reproduces the reference's run at the REAL width -- batches, losses, first-step layer outputs and
    pre-Adam gradients (1e-4), parameters, final embeddings; the padding columns stay exactly zero."""
    if tag not in smeta:
        pytest.skip("golden section E not generated")
    info = smeta[tag]
    tr = trainer_for(info, tiny_data, shapes[f"{tag}_init_user"], shapes[f"{tag}_init_item"])
    assert tr.d_valid == info["emb"] and tr.d > tr.d_valid and tr.E0.shape[1] == tr.d
    # parameters after the two Adam steps: on this 500-node graph most rows are far from the batch, their gradients are
    # ~1e-9 and Adam turns the last bits of them into percents of a step (module docstring) -- within 5 % of one step,
    # 5e-4 of the value range; what precedes Adam is held to 1e-4 by the probe inside run_and_check
    fu, fi = run_and_check(tag, shapes, info, tr, rows=False, param_atol=5e-5, param_rtol=5e-4)
    assert fu.shape[1] == info["emb"] and not tr.E0[:, tr.d_valid:].any() and not tr.m[:, tr.d_valid:].any()


def test_in_kernel_noise_on_padded_rows_is_normalised_over_the_real_columns(tiny_data):
    """XSimGCL.py:90 draws rand_like(h) with d_valid columns and normalises it: on zero-padded tables the counter RNG's
    unit vector must ignore the padding (epilogue field noise_d_valid) -- every perturbed row moves by exactly eps."""
    torch.manual_seed(3)
    tr = FusedTrainer(tiny_data, 50, model="XSimGCL", n_layers=2, eps=0.2, layer_cl=1, batch_size=1024)
    assert (tr.d_valid, tr.d) == (50, 64)
    x = tr.E0
    clean = ops.spmm(tr.adj, x)
    noisy = ops.spmm(tr.adj, x, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=7, rng_offset=0, d_valid=50))
    delta = (noisy - clean).cpu().numpy()
    assert not delta[:, 50:].any() and not noisy[:, 50:].any()
    norms = np.linalg.norm(delta[:, :50].astype(np.float64), axis=1)
    live = np.abs(clean.cpu().numpy()[:, :50]).min(axis=1) > 0            # (sign(0) = 0 would shorten the step)
    np.testing.assert_allclose(norms[live], 0.2, rtol=2e-6)
    # ... and the same launch without the field spreads the unit vector over all 64 columns: shorter on the real ones
    spread = ops.spmm(tr.adj, x, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=7, rng_offset=0))
    assert np.linalg.norm((spread - clean).cpu().numpy()[:, :50], axis=1)[live].max() < 0.2 * (1 - 1e-3)
    # a whole step with in-kernel noise keeps the padding at zero
    tr.sampler.seed(1)
    tr.begin_epoch()
    for _ in range(2):
        tr.step()
    assert not tr.E0[:, 50:].any() and torch.isfinite(tr.E0).all()


def test_node_dropout_dropin_laplacian_on_device(shapes, fresh_tiny_data):
    """GraphAugmentor.node_dropout -> convert_to_laplacian_mat -> convert_sparse_mat_to_tensor, as SGL.py:89-96 chains
    them with aug_type 0: the dropped, re-normalised adjacency never leaves the GPU and is bit-identical."""
    from selfrec_amd.base.torch_interface import SparseAdjHandle, TorchGraphInterface
    from selfrec_amd.data.augmentor import GraphAugmentor
    data = fresh_tiny_data
    random.seed(77)
    dropped = GraphAugmentor.node_dropout(data.interaction_mat, 0.1)
    assert random.getrandbits(32) == int(shapes["node_dropout_next_u32"][0])
    h = TorchGraphInterface.convert_sparse_mat_to_tensor(data.convert_to_laplacian_mat(dropped)).cuda()
    assert isinstance(h, SparseAdjHandle)
    g = data.device_graph()
    view = g.dropped_view(torch.from_numpy(dropped.keep_mask).to(DEV))
    m = sp.csr_matrix((view.vals.cpu().numpy(), g.adj.indices.cpu().numpy(), g.adj.indptr.cpu().numpy()),
                      shape=(g.n_nodes, g.n_nodes))
    m.eliminate_zeros(); m.sort_indices()
    assert np.array_equal(m.indptr, shapes["node_dropout_lap_indptr"])
    assert np.array_equal(m.indices, shapes["node_dropout_lap_indices"])
    assert np.array_equal(m.data, shapes["node_dropout_lap_data"])             # bit-exact (host pow table)
    x = np.random.default_rng(3).standard_normal((g.n_nodes, 64)).astype(np.float32)
    got = torch.sparse.mm(h, torch.from_numpy(x).cuda()).cpu().numpy()
    assert rel_err(got, m.astype(np.float64) @ x.astype(np.float64)) < 2e-6


def test_duplicate_interactions_weigh_two_in_adj_and_one_in_views(shapes):
    tu, ti = shapes["dup_train_u"], shapes["dup_train_i"]
    data = Interaction({}, synth.as_triples(tu, ti), [])
    g = data.device_graph()
    assert g.weight is not None
    mine = sp.csr_matrix((g.adj.vals.cpu().numpy(), g.adj.indices.cpu().numpy(), g.adj.indptr.cpu().numpy()),
                         shape=(g.n_nodes, g.n_nodes))
    mine.sort_indices()
    assert np.array_equal(mine.indices, shapes["dup_norm_adj_indices"])
    assert np.array_equal(mine.data, shapes["dup_norm_adj_data"])
    random.seed(5)
    from selfrec_amd.data.augmentor import GraphAugmentor
    dropped = GraphAugmentor.edge_dropout(data.interaction_mat, 0.1)
    view = g.dropped_view(torch.from_numpy(dropped.keep_mask).to(DEV))
    m = sp.csr_matrix((view.vals.cpu().numpy(), g.adj.indices.cpu().numpy(), g.adj.indptr.cpu().numpy()),
                      shape=(g.n_nodes, g.n_nodes))
    m.eliminate_zeros(); m.sort_indices()
    assert np.array_equal(m.indices, shapes["dup_drop_lap_indices"])
    assert np.array_equal(m.data, shapes["dup_drop_lap_data"])
