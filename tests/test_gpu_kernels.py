"""Parity of every HIP kernel against the CPU oracle / the reference-generated golden vectors.
All calls go through the C ABI (selfrec_amd.ops -> ctypes -> libselfrec_hip.so).

Tolerances: integer / index outputs bit-exact; fp32 values within 1e-4 relative (the bound
BASELINE.json's north_star states), most checks far tighter.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import selfrec_oracle as O
from selfrec_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(got, want, floor=1e-7):
    """max |got - want| relative to the largest reference magnitude (floored: an all-zero
    reference, e.g. the InfoNCE gradient at n=1, is compared absolutely)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), floor))


def powerlaw_csr(n_rows, n_cols, nnz, seed, heavy_rows=0, heavy_len=0, empty_rows=0):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_rows + 1) ** 0.8
    rows = rng.choice(n_rows, size=nnz, p=w / w.sum())
    cols = rng.integers(0, n_cols, size=nnz)
    for h in range(heavy_rows):
        extra = rng.choice(n_cols, size=min(heavy_len, n_cols), replace=False)
        rows = np.concatenate([rows, np.full(extra.size, n_rows - 1 - h)])
        cols = np.concatenate([cols, extra])
    vals = rng.standard_normal(rows.size).astype(np.float32)
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n_rows, n_cols), dtype=np.float32)
    if empty_rows:
        lil = m.tolil()
        for r in rng.choice(n_rows, size=empty_rows, replace=False):
            lil.rows[r], lil.data[r] = [], []
        m = lil.tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return m


# ------------------------------------------------------------------------------------------
# (a-3/a-4) SpMM and its epilogues
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("shape", [(1, 1, 1, 0, 0, 0), (97, 130, 700, 0, 0, 5), (3000, 2500, 40000, 3, 1500, 40)])
def test_spmm_matches_scipy(d, shape):
    n_rows, n_cols, nnz, heavy, heavy_len, empty = shape
    m = powerlaw_csr(n_rows, n_cols, nnz, seed=d + n_rows, heavy_rows=heavy, heavy_len=heavy_len, empty_rows=empty)
    x = np.random.default_rng(1).standard_normal((n_cols, d)).astype(np.float32)
    want = (m.astype(np.float64) @ x.astype(np.float64))
    csr = ops.DeviceCSR.from_scipy(m)
    got = ops.spmm(csr, torch.from_numpy(x).to(DEV)).cpu().numpy()
    assert got.shape == want.shape
    assert rel_err(got, want) < 2e-6
    if heavy:   # split rows: explicit small split length exercises the partial-sum path harder
        csr2 = ops.DeviceCSR.from_scipy(m, split_len=64)
        got2 = ops.spmm(csr2, torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert rel_err(got2, want) < 2e-6
        # fixed reduction order: bitwise reproducible
        assert np.array_equal(got2, ops.spmm(csr2, torch.from_numpy(x).to(DEV)).cpu().numpy())


def test_spmm_epilogues_match_oracle():
    d, n = 64, 1200
    m = powerlaw_csr(n, n, 20000, seed=3, heavy_rows=2, heavy_len=900, empty_rows=10)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[:, :3] = 0.0                                        # exact zeros in y -> sign(0) = 0 branch
    noise = rng.random((n, d)).astype(np.float32)
    p0, p1 = rng.standard_normal((n, d)).astype(np.float32), rng.standard_normal((n, d)).astype(np.float32)
    csr = ops.DeviceCSR.from_scipy(m)
    tx, tn = torch.from_numpy(x).to(DEV), torch.from_numpy(noise).to(DEV)
    tp0, tp1 = torch.from_numpy(p0).to(DEV), torch.from_numpy(p1).to(DEV)
    # oracle: XSimGCL.py:88-96 on CPU
    y = torch.sparse.mm(O.to_torch_sparse(m), torch.from_numpy(x))
    y = O.perturb_(y, torch.from_numpy(noise), 0.2)
    mean = torch.mean(torch.stack([torch.from_numpy(p0), torch.from_numpy(p1), y], dim=1), dim=1)
    mean_out = torch.empty((n, d), device=DEV)
    ep = ops.make_epilogue(perturb_eps=0.2, noise=tn, prev=[tp0, tp1], mean_div=3.0, mean_out=mean_out)
    got = ops.spmm(csr, tx, epilogue=ep)
    assert rel_err(got.cpu().numpy(), y.numpy()) < 2e-6
    assert rel_err(mean_out.cpu().numpy(), mean.numpy()) < 2e-6
    # backward-style AXPY with alpha, addend aliasing the output
    out = tp0.clone()
    ep = ops.make_epilogue(add=[out, tp1], add_scale=[1.0, 0.25], alpha=0.5)
    ops.spmm(csr, tx, out=out, epilogue=ep)
    want = 0.5 * (m.astype(np.float64) @ x.astype(np.float64)) + p0 + 0.25 * p1
    assert rel_err(out.cpu().numpy(), want) < 2e-6


@pytest.mark.selfcheck
def test_spmm_fanout_equals_separate_launches():
    """One product, three outputs (SimGCL's clean pass + two perturbed views share A.E0 in layer 1): each
    output is bit-identical to a launch of its own, with injected noise and with the counter RNG."""
    d, n = 64, 900
    m = powerlaw_csr(n, n, 15000, seed=21, heavy_rows=2, heavy_len=800, empty_rows=4)
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV)
    na, nb = (torch.from_numpy(rng.random((n, d)).astype(np.float32)).to(DEV) for _ in range(2))
    csr = ops.DeviceCSR.from_scipy(m)
    clean = ops.spmm(csr, x)
    for injected in (True, False):
        kw = dict(perturb_eps=0.2, rng_seed=77, rng_stride=n * 16)
        if injected:
            sep = [ops.spmm(csr, x, epilogue=ops.make_epilogue(noise=t, **kw)) for t in (na, nb)]
            extra = dict(extra_noise=[na, nb])
        else:
            sep = [ops.spmm(csr, x, epilogue=ops.make_epilogue(rng_offset=off, **kw)) for off in (5 * n, 9 * n)]
            extra = dict(extra_rng_offset=[5 * n, 9 * n])
        ya, yb, y = torch.empty_like(clean), torch.empty_like(clean), torch.empty_like(clean)
        ops.spmm(csr, x, out=y, epilogue=ops.make_epilogue(main_clean=True, extra_out=[ya, yb], **extra, **kw))
        assert torch.equal(y, clean) and torch.equal(ya, sep[0]) and torch.equal(yb, sep[1])
        assert not torch.equal(ya, yb)


@pytest.mark.selfcheck
def test_spmm3_equals_three_launches():
    """Three value arrays over one structure, one traversal (SGL's first layer): bit-identical to three
    launches, split rows and dropped (zero) entries included; other widths are refused."""
    d, n = 64, 1100
    m = powerlaw_csr(n, n, 18000, seed=23, heavy_rows=3, heavy_len=900, empty_rows=6)
    rng = np.random.default_rng(6)
    x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV)
    base = ops.DeviceCSR.from_scipy(m)
    views = [base]
    for _ in range(2):
        keep = torch.from_numpy((rng.random(m.nnz) > 0.3).astype(np.float32)).to(DEV)
        views.append(base.with_values(base.vals * keep * float(rng.uniform(0.5, 2.0))))
    want = [ops.spmm(v, x) for v in views]
    outs = [torch.full((n, d), 3.0, device=DEV) for _ in range(3)]
    ops.spmm3(views, x, outs)
    for got, ref in zip(outs, want):
        assert torch.equal(got, ref)
    with pytest.raises(ops.SelfrecHipError):
        x128 = torch.zeros((n, 128), device=DEV)
        ops.spmm3(views, x128, [torch.empty((n, 128), device=DEV) for _ in range(3)])


@pytest.mark.parametrize("d", [64, 128])
def test_spmm_xcd_shares_and_probe(d):
    """srh_spmm_plan_set_xcd_shares / srh_spmm_f32_probe: unequal numbers of workgroups per XCD (empty records pad the
    list) change where tasks run, never what they compute -- same product bit for bit, also with split heavy rows, the
    pattern form and row marks -- and the probe launch computes the product too and stamps every task once."""
    n, U = 3000, 1200
    m = powerlaw_csr(n, n, 60000, seed=37, heavy_rows=4, heavy_len=1800, empty_rows=11)
    perm, row_mid = ops.column_class_order(m.indptr, m.indices, 64)
    csr = ops.DeviceCSR(m.indptr, m.indices[perm], m.data[perm], m.shape, xcd_split_row=U, row_mid=row_mid)
    rng = np.random.default_rng(8)
    tx = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV)
    mark = torch.from_numpy(np.where(rng.random(n) < 0.2, 7, 1).astype(np.int32)).to(DEV)
    stamp = torch.tensor([7], dtype=torch.int64, device=DEV)
    ep_rows = lambda: ops.make_epilogue(row_mark=mark, mark_stamp=stamp)          # noqa: E731
    ep_cols = lambda: ops.make_epilogue(col_mark=mark, mark_stamp=stamp)          # noqa: E731
    ref, ref_pat = ops.spmm(csr, tx), ops.spmm(csr, tx, pattern=True)
    ref_cols = ops.spmm(csr, tx, epilogue=ep_cols())
    ref_rows = torch.zeros((n, d), device=DEV); ops.spmm(csr, tx, out=ref_rows, epilogue=ep_rows())
    # the row-masked launch (d = 64: its own instantiation, the next chunk's indices issued AFTER the gathers from inline
    # asm) adds the same entries in the same order as the unmasked kernel: identical bits on the marked rows, with a value
    # stream and as a pattern launch; unmarked rows are not touched
    live = mark == 7
    assert torch.equal(ref_rows[live], ref[live]) and not ref_rows[~live].any()
    pat_rows = torch.zeros((n, d), device=DEV); ops.spmm(csr, tx, out=pat_rows, epilogue=ep_rows(), pattern=True)
    assert torch.equal(pat_rows[live], ref_pat[live]) and not pat_rows[~live].any()
    n_canon = ops.spmm_plan_run_tasks(csr, d)
    nb = (n_canon + 3) // 4
    canon = np.array([len(range(k, nb, 8)) for k in range(8)])
    a, b, few = int(canon.min()) // 2, int(canon.min()) // 3, max(1, int(canon.min()) // 4)
    for shares in (canon, canon + np.array([a, -a, b, -b, 0, 0, 1, -1]), np.array([nb - 7 * few] + [few] * 7)):
        ops.spmm_set_xcd_shares(csr, d, shares)
        assert ops.spmm_plan_run_tasks(csr, d) == int(shares.max()) * 32
        assert torch.equal(ops.spmm(csr, tx), ref) and torch.equal(ops.spmm(csr, tx, pattern=True), ref_pat)
        assert torch.equal(ops.spmm(csr, tx, epilogue=ep_cols()), ref_cols)
        out = torch.zeros((n, d), device=DEV); ops.spmm(csr, tx, out=out, epilogue=ep_rows())
        assert torch.equal(out, ref_rows)
        got = torch.empty((n, d), device=DEV)
        finish, begin, end, xcd = ops.spmm_probe(csr, tx, got)
        assert torch.equal(got, ref)
        assert begin.size == int(shares.max()) * 32 and (end >= begin).all() and set(np.unique(xcd)) <= set(range(8))
        assert finish.shape == (8,) and finish.max() < 1e4                         # us: a sane clock
    with pytest.raises(ops.SelfrecHipError):
        ops.spmm_set_xcd_shares(csr, d, canon + 1)                                 # does not add up
    ops.spmm_set_xcd_shares(csr, d, None)
    assert ops.spmm_plan_run_tasks(csr, d) == n_canon and torch.equal(ops.spmm(csr, tx), ref)
    # srh_spmm_gather_bound: the same schedule and gathers with nothing after them -- runs, touches no operand, and (a strict
    # subset of the product's work) is not slower than the product beyond timer noise on this small graph
    before = tx.clone()
    bound_us = ops.spmm_gather_bound(csr, tx, iters=20)
    assert 0.0 < bound_us < 1e4 and torch.equal(tx, before) and torch.equal(ops.spmm(csr, tx), ref)


@pytest.mark.parametrize("d", [64, 128])
def test_spmm_under_another_plan_of_the_same_arrays(d):
    """DeviceCSR.replanned: the same device arrays under a second schedule (the engine runs the column-masked launch of a
    step on a class-free plan with shorter segments): same product in every flavour -- plain, pattern (no value stream),
    row-masked, column-masked -- split heavy rows included, and the two objects share indices and values."""
    n, U = 2600, 1000
    m = powerlaw_csr(n, n, 52000, seed=31, heavy_rows=4, heavy_len=1700, empty_rows=9)
    perm, row_mid = ops.column_class_order(m.indptr, m.indices, 64)
    base = ops.DeviceCSR(m.indptr, m.indices[perm], m.data[perm], m.shape, xcd_split_row=U, row_mid=row_mid)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((n, d)).astype(np.float32)
    tx = torch.from_numpy(x).to(DEV)
    want = m.astype(np.float64) @ x.astype(np.float64)
    ones = sp.csr_matrix((np.ones_like(m.data), m.indices, m.indptr), shape=m.shape)
    want_pattern = ones.astype(np.float64) @ x.astype(np.float64)
    live = rng.random(n) < 0.25
    mark = torch.from_numpy(np.where(live, 7, 3).astype(np.int32)).to(DEV)
    stamp = torch.tensor([7], dtype=torch.int64, device=DEV)
    xz = x.copy(); xz[~live] = 0.0
    want_cols = m.astype(np.float64) @ xz.astype(np.float64)
    for split, split_row in ((256, 0), (128, 0), (512, U)):
        other = base.replanned(split_len=split, xcd_split_row=split_row)
        assert other.indices.data_ptr() == base.indices.data_ptr() and other.vals.data_ptr() == base.vals.data_ptr()
        assert rel_err(ops.spmm(other, tx).cpu().numpy(), want) < 2e-6
        assert rel_err(ops.spmm(other, tx, pattern=True).cpu().numpy(), want_pattern) < 2e-6
        got = ops.spmm(other, tx, epilogue=ops.make_epilogue(col_mark=mark, mark_stamp=stamp))
        assert rel_err(got.cpu().numpy(), want_cols) < 2e-6
        out = torch.full((n, d), -5.0, device=DEV)
        ops.spmm(other, tx, out=out, epilogue=ops.make_epilogue(row_mark=mark, mark_stamp=stamp))
        out = out.cpu().numpy()
        assert rel_err(out[live], want[live]) < 2e-6 and (out[~live] == -5.0).all()
        assert torch.equal(ops.spmm(other, tx), ops.spmm(other, tx))          # fixed reduction order
    del other
    assert rel_err(ops.spmm(base, tx).cpu().numpy(), want) < 2e-6            # the first plan is untouched


@pytest.mark.parametrize("d,min_len", [(64, 64), (64, 8), (128, 16), (32, 16)])
def test_spmm_with_column_class_schedule_matches_scipy(d, min_len):
    """Rows of >= min_len non-zeros stored [even columns | odd columns] and scheduled as two segments on
    different XCDs (srh_spmm_plan_create h_row_mid): same product, split heavy rows and masks included."""
    n, U = 1300, 500
    m = powerlaw_csr(n, n, 26000, seed=17, heavy_rows=3, heavy_len=1100, empty_rows=7)
    perm, row_mid = ops.column_class_order(m.indptr, m.indices, min_len)
    assert sorted(perm.tolist()) == list(range(m.nnz)) and (row_mid >= 0).sum() == (np.diff(m.indptr) >= min_len).sum()
    csr = ops.DeviceCSR(m.indptr, m.indices[perm], m.data[perm], m.shape, xcd_split_row=U, row_mid=row_mid)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, d)).astype(np.float32)
    tx = torch.from_numpy(x).to(DEV)
    want = m.astype(np.float64) @ x.astype(np.float64)
    got = ops.spmm(csr, tx)
    assert rel_err(got.cpu().numpy(), want) < 2e-6
    assert torch.equal(ops.spmm(csr, tx), got)                      # reproducible bit for bit
    # column marks: only the live columns contribute
    live = rng.random(n) < 0.3
    mark = torch.from_numpy(np.where(live, 5, 0).astype(np.int32)).to(DEV)
    stamp = torch.tensor([5], dtype=torch.int64, device=DEV)
    xm = tx * torch.from_numpy(live.astype(np.float32)).to(DEV).unsqueeze(1)
    got_m = ops.spmm(csr, xm, epilogue=ops.make_epilogue(col_mark=mark, mark_stamp=stamp))
    want_m = m.astype(np.float64) @ (x * live[:, None]).astype(np.float64)
    assert rel_err(got_m.cpu().numpy(), want_m) < 2e-6


def test_spmm_rng_perturbation_properties():
    d, n = 64, 500
    m = powerlaw_csr(n, n, 6000, seed=9)
    x = np.random.default_rng(2).standard_normal((n, d)).astype(np.float32)
    csr = ops.DeviceCSR.from_scipy(m)
    tx = torch.from_numpy(x).to(DEV)
    base = ops.spmm(csr, tx)
    a = ops.spmm(csr, tx, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=7, rng_offset=0))
    b = ops.spmm(csr, tx, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=7, rng_offset=0))
    c = ops.spmm(csr, tx, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=7, rng_offset=n))
    assert torch.equal(a, b) and not torch.equal(a, c)          # counter-based: reproducible, offset-dependent
    delta = (a - base).cpu().numpy()
    nz = np.abs(base.cpu().numpy()).sum(1) > 0
    # each perturbed row moved by a vector of norm eps along sign(y) (XSimGCL.py:91)
    assert np.allclose(np.linalg.norm(delta[nz], axis=1), 0.2, rtol=1e-4)
    assert np.all(delta * np.sign(base.cpu().numpy()) >= -1e-7)
    step = torch.tensor([3], dtype=torch.int64, device=DEV)
    e = ops.spmm(csr, tx, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=7, rng_offset=0,
                                                     rng_step=step, rng_stride=n))
    f = ops.spmm(csr, tx, epilogue=ops.make_epilogue(perturb_eps=0.2, rng_seed=7, rng_offset=3 * n))
    assert torch.equal(e, f)


def test_value_free_product_with_row_scaling(tiny_data):
    """srh_spmm_f32 with d_vals == NULL (pattern) + SRH_SCALE_IN / SRH_SCALE_OUT / prev-unscale / addend row scale:
    D^-1/2 A D^-1/2 x without the value stream (graph.py:10-24), as the engine chains it through pre-scaled tables."""
    g = tiny_data.device_graph()
    N, d = g.n_nodes, 64
    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, d)).astype(np.float32)
    a = rng.standard_normal((N, d)).astype(np.float32)
    p0 = rng.standard_normal((N, d)).astype(np.float32)
    A = sp.csr_matrix((g.adj.vals.cpu().numpy(), g.adj.indices.cpu().numpy(), g.adj.indptr.cpu().numpy()), shape=(N, N)).astype(np.float64)
    dinv = g.dinv.cpu().numpy().astype(np.float64)
    want = A @ x.astype(np.float64)                                            # true product
    xs = torch.from_numpy((dinv[:, None] * x).astype(np.float32)).to(DEV)      # the pre-scaled input table
    y = torch.empty((N, d), device=DEV)
    ops.spmm(g.adj, xs, out=y, epilogue=ops.make_epilogue(row_scale=g.dinv, scale_in=True), pattern=True)
    assert rel_err(y.cpu().numpy(), want) < 2e-6
    # stored pre-scaled for the next layer, plus an addend that enters times d_i^-1/2 and one that does not
    ta = torch.from_numpy(a).to(DEV)
    ops.spmm(g.adj, xs, out=y, pattern=True,
             epilogue=ops.make_epilogue(row_scale=g.dinv, scale_in=True, scale_out=True, add=[ta, ta], add_scale=[0.5, 2.0],
                                        alpha=0.25, add_rowscale=[True, False]))
    want2 = dinv[:, None] * (0.25 * want + 0.5 * dinv[:, None] * a + 2.0 * a)
    assert rel_err(y.cpu().numpy(), want2) < 3e-6
    # last layer: mean over a true table, a pre-scaled table and this product
    tp = torch.from_numpy(p0).to(DEV)
    tps = torch.from_numpy((dinv[:, None] * p0).astype(np.float32)).to(DEV)
    mean = torch.empty((N, d), device=DEV)
    ops.spmm(g.adj, xs, out=y, pattern=True,
             epilogue=ops.make_epilogue(row_scale=g.dinv, scale_in=True, prev=[tp, tps], mean_div=3.0, mean_out=mean,
                                        prev_unscale=[False, True]))
    live = dinv > 0
    assert rel_err(mean.cpu().numpy()[live], ((p0 + p0 + want) / 3.0)[live]) < 3e-6
    # the value array path with the same scaling flags (first product of a chain: true input, pre-scaled output)
    tx = torch.from_numpy(x).to(DEV)
    ops.spmm(g.adj, tx, out=y, epilogue=ops.make_epilogue(row_scale=g.dinv, scale_out=True))
    assert rel_err(y.cpu().numpy(), dinv[:, None] * want) < 2e-6


@pytest.mark.selfcheck
def test_product_with_batch_fetch_rider_equals_the_two_launches(tiny_data):
    """srh_spmm_f32_with_fetch: the product and the staged batch are those of srh_spmm_f32 + srh_batch_fetch."""
    g = tiny_data.device_graph()
    N, d, B, E = g.n_nodes, 64, 32, 100
    rng = np.random.default_rng(11)
    U = g.n_users
    ep = {"u": rng.integers(0, U, E), "i": rng.integers(U, N, E), "j": rng.integers(U, N, E),
          "uniq_u": rng.integers(0, U, E), "uniq_i": rng.integers(U, N, E),
          "n_uniq_u": rng.integers(1, B, (E + B - 1) // B), "n_uniq_i": rng.integers(1, B, (E + B - 1) // B)}
    ep = {k: torch.from_numpy(v.astype(np.int32)).to(DEV) for k, v in ep.items()}
    x = torch.from_numpy(rng.standard_normal((N, d)).astype(np.float32)).to(DEV)
    cursor = torch.tensor([3, 17], dtype=torch.int64, device=DEV)          # the last, short batch: 4 of 32 rows

    def buffers():
        stage = {k: torch.full((B,), -1, dtype=torch.int32, device=DEV) for k in ("u", "i", "j", "uniq_u", "uniq_i")}
        return dict(stage=stage, meta=torch.full((4,), -1, dtype=torch.int32, device=DEV),
                    mark=torch.zeros(N, dtype=torch.int32, device=DEV), zero4=torch.ones(4, dtype=torch.float64, device=DEV),
                    cat=torch.full((2 * B,), -1, dtype=torch.int32, device=DEV), n_cat=torch.zeros(1, dtype=torch.int32, device=DEV),
                    now=torch.zeros(2, dtype=torch.int64, device=DEV))

    def args(b):
        return ops.batch_fetch_args(ep, E, B, cursor, b["stage"], b["meta"], row_mark=b["mark"], zero4=b["zero4"],
                                    stage_cat=b["cat"], cat_item_offset=5, n_cat=b["n_cat"], now=b["now"])
    one, two = buffers(), buffers()
    epi = lambda: ops.make_epilogue(perturb_eps=0.2, rng_seed=3, rng_offset=0)      # noqa: E731
    y1 = ops.spmm(g.adj, x, epilogue=epi(), fetch=args(one))
    ops.batch_fetch(args(two))
    y2 = ops.spmm(g.adj, x, epilogue=epi())
    assert torch.equal(y1, y2)
    assert int(one["meta"][0]) == E - 3 * B and torch.equal(one["now"], cursor) and float(one["zero4"].abs().sum()) == 0.0
    assert int((one["mark"] == 17).sum()) > 0
    for k in ("meta", "mark", "zero4", "cat", "n_cat", "now"):
        assert torch.equal(one[k], two[k]), k
    for k in one["stage"]:
        assert torch.equal(one["stage"][k], two["stage"][k]), k
    # the product may not depend on what the fetch writes
    with pytest.raises(ops.SelfrecHipError):
        ops.spmm(g.adj, x, epilogue=ops.make_epilogue(row_mark=one["mark"], mark_stamp=cursor[1:2]), fetch=args(one))


def test_batch_fetch_over_two_epochs_back_to_back():
    """srh_batch_fetch_args_t::half_batches: the arrays hold two epochs of nb batch slots each; batch no b >= nb is batch
    b - nb of the second epoch -- cut against n_edges with b - nb (the last batch of EITHER epoch is short), read at b.
    Every batch of both halves against the CPU stand-in of the same call (tests/cpu_ops.py)."""
    from tests import cpu_ops
    rng = np.random.default_rng(21)
    B, E, N = 32, 100, 500
    nb = (E + B - 1) // B
    host = {k: rng.integers(0, N, 2 * nb * B).astype(np.int32) for k in ("u", "i", "j", "uniq_u", "uniq_i")}
    host["n_uniq_u"] = rng.integers(1, B, 2 * nb).astype(np.int32)
    host["n_uniq_i"] = rng.integers(1, B, 2 * nb).astype(np.int32)
    dev_ep = {k: torch.from_numpy(v).to(DEV) for k, v in host.items()}
    cpu_ep = {k: torch.from_numpy(v) for k, v in host.items()}

    def buffers(dev):
        return dict(stage={k: torch.full((B,), -1, dtype=torch.int32, device=dev) for k in ("u", "i", "j", "uniq_u", "uniq_i")},
                    meta=torch.full((4,), -1, dtype=torch.int32, device=dev), mark=torch.zeros(N, dtype=torch.int32, device=dev))
    for b in range(2 * nb + 1):                                       # (+ one batch past the end: 0 rows)
        got, want = buffers(DEV), buffers("cpu")
        ops.batch_fetch(dev_ep, E, B, torch.tensor([b, 9], dtype=torch.int64, device=DEV), got["stage"], got["meta"],
                        row_mark=got["mark"], half_batches=nb)
        cpu_ops.batch_fetch(cpu_ep, E, B, torch.tensor([b, 9], dtype=torch.int64), want["stage"], want["meta"],
                            row_mark=want["mark"], half_batches=nb)
        rows = int(want["meta"][0])
        assert rows == (E - (nb - 1) * B if b % nb == nb - 1 and b < 2 * nb else (B if b < 2 * nb else 0)), (b, rows)
        assert torch.equal(got["meta"].cpu(), want["meta"]) and torch.equal(got["mark"].cpu(), want["mark"]), b
        for k in ("u", "i", "j"):
            assert torch.equal(got["stage"][k][:rows].cpu(), want["stage"][k][:rows]), (b, k)
        a, c = int(want["meta"][1]), int(want["meta"][2])
        assert torch.equal(got["stage"]["uniq_u"][:a].cpu(), want["stage"]["uniq_u"][:a])
        assert torch.equal(got["stage"]["uniq_i"][:c].cpu(), want["stage"]["uniq_i"][:c])


# ------------------------------------------------------------------------------------------
# (a-2) normalisation and edge-dropped views
# ------------------------------------------------------------------------------------------
def test_device_normalisation_matches_reference(golden_ops, tiny_data):
    g = tiny_data.device_graph()
    assert np.array_equal(g.adj.indptr.cpu().numpy(), golden_ops["norm_adj_indptr"])
    # (long rows are stored [even columns | odd columns] for the SpMM schedule: compare in sorted order)
    mine = sp.csr_matrix((g.adj.vals.cpu().numpy(), g.adj.indices.cpu().numpy(), g.adj.indptr.cpu().numpy()),
                         shape=(g.n_nodes, g.n_nodes))
    mine.sort_indices()
    assert np.array_equal(mine.indices, golden_ops["norm_adj_indices"])
    assert np.array_equal(mine.data, golden_ops["norm_adj_data"])    # bit-exact (host pow table)
    # edge-dropped, re-normalised view (SGL.py:89-96) from the reference keep-set
    keep = np.zeros(g.n_edges, dtype=np.uint8)
    keep[golden_ops["edge_dropout_keep"]] = 1
    view = g.dropped_view(torch.from_numpy(keep).to(DEV))
    dense = sp.csr_matrix((view.vals.cpu().numpy(), g.adj.indices.cpu().numpy(), g.adj.indptr.cpu().numpy()),
                          shape=(g.n_nodes, g.n_nodes))
    dense.eliminate_zeros()
    dense.sort_indices()
    assert np.array_equal(dense.indptr, golden_ops["edge_dropout_lap_indptr"])
    assert np.array_equal(dense.indices, golden_ops["edge_dropout_lap_indices"])
    assert np.array_equal(dense.data, golden_ops["edge_dropout_lap_data"])


def test_normalisation_isolated_nodes_and_weights():
    # node 2 isolated after dropping its only edge; duplicate interaction -> weight 2 in the full adjacency
    # (scipy sums it, ui_graph.py:47-56) but 1 in a dropped view (augmentor.py:36-39 rebuilds with ones)
    r = sp.csr_matrix((np.array([1, 2, 1, 1], dtype=np.float32), ([0, 0, 1, 2], [0, 1, 1, 2])), shape=(3, 3))
    from selfrec_amd.data.device_graph import DeviceGraph
    g = DeviceGraph(r)
    want = O.laplacian_of(r).tocsr(); want.sort_indices()
    np.testing.assert_allclose(g.adj.vals.cpu().numpy(), want.data, rtol=3e-7)
    keep = torch.tensor([1, 1, 1, 0], dtype=torch.uint8, device=DEV)
    v = g.dropped_view(keep).vals.cpu().numpy()
    r2 = r.copy(); r2.data[:] = 1.0; r2.data[3] = 0; r2.eliminate_zeros()
    with np.errstate(divide="ignore"):
        want2 = O.laplacian_of(r2).toarray()
    got2 = sp.csr_matrix((v, g.adj.indices.cpu().numpy(), g.adj.indptr.cpu().numpy()), shape=(6, 6)).toarray()
    np.testing.assert_allclose(got2, want2, rtol=3e-7)
    assert np.isfinite(v).all()


# ------------------------------------------------------------------------------------------
# (a-5..a-8) losses
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 130, 515])
def test_loss_functions_match_reference_outputs(golden_ops, n):
    from selfrec_amd.util import loss_torch as L
    g = golden_ops
    u, p, q = (torch.tensor(x, device=DEV, requires_grad=True) for x in g[f"ops_{n}_in"])
    bpr, reg, nce = L.bpr_loss(u, p, q), L.l2_reg_loss(1e-4, u, p, q), L.InfoNCE(u, p, 0.2)
    np.testing.assert_allclose([bpr.item(), reg.item()], g[f"ops_{n}_loss"][:2], rtol=2e-6)
    # InfoNCE evaluates its similarity products in split-bf16 (3-term) MFMA: logits carry <= 2e-5 absolute
    # error, which only shows at tiny n where nothing averages out (budget: 1e-4 relative)
    np.testing.assert_allclose(nce.item(), g[f"ops_{n}_loss"][2], rtol=2e-5, atol=5e-6)
    gb = torch.stack(torch.autograd.grad(bpr, (u, p, q))).cpu().numpy()
    gr = torch.stack(torch.autograd.grad(reg, (u, p, q))).cpu().numpy()
    gn = torch.stack(torch.autograd.grad(nce, (u, p))).cpu().numpy()
    assert rel_err(gb, g[f"ops_{n}_g_bpr"]) < 1e-5
    assert rel_err(gr, g[f"ops_{n}_g_reg"]) < 1e-5
    if n == 1:      # a single row: the loss is identically 0 and so is its gradient; ours is rounding noise
        assert np.abs(gn).max() < 1e-5 and not g[f"ops_{n}_g_nce"].any()
    else:
        assert rel_err(gn, g[f"ops_{n}_g_nce"]) < 1e-5


@pytest.mark.parametrize("rows,d", [(1, 64), (2048, 64), (777, 32), (300, 100), (4099, 128)])
def test_single_launch_losses_equal_the_torch_expressions(rows, d):
    """bpr_loss / l2_reg_loss finish their scalars on the device (one launch forward, one backward, a shared 64-byte
    workspace that every call leaves zero): values and gradients against the reference's expressions (loss_torch.py:6-22)
    evaluated by ATen on the same device, with a NON-unit upstream gradient, an all-zero block (zero gradient, as
    torch.norm's), four blocks of different sizes in one call, and the same call repeated."""
    from selfrec_amd import ops
    from selfrec_amd.util import loss_torch as L
    gen = torch.Generator().manual_seed(rows * 31 + d)
    mk = lambda r: (torch.randn(r, d, generator=gen) * 0.3).to(DEV).requires_grad_(True)
    u, p, q = mk(rows), mk(rows), mk(rows)
    z = torch.zeros(5, d, device=DEV, requires_grad=True)
    small = mk(3)

    def expression(u, p, q, z, small):
        pos, neg = torch.mul(u, p).sum(dim=1), torch.mul(u, q).sum(dim=1)
        bpr = torch.mean(-torch.log(10e-6 + torch.sigmoid(pos - neg)))
        emb = 0
        for e in (u, z, p, small):
            emb = emb + torch.norm(e, p=2) / e.shape[0]
        return bpr, emb * 0.37

    for _ in range(3):                                  # (the workspace is shared: repeated calls must not leak into each other)
        bpr, reg = L.bpr_loss(u, p, q), L.l2_reg_loss(0.37, u, z, p, small)
        assert bpr.dtype == torch.float32 and bpr.dim() == 0 and reg.dtype == torch.float32 and reg.dim() == 0
        got = torch.autograd.grad(bpr * 3.7 + reg * -2.5, (u, p, q, z, small))
        wb, wr = expression(u, p, q, z, small)
        want = torch.autograd.grad(wb * 3.7 + wr * -2.5, (u, p, q, z, small))
        np.testing.assert_allclose([bpr.item(), reg.item()], [wb.item(), wr.item()], rtol=2e-6)
        for g, w in zip(got, want):
            assert rel_err(g.cpu().numpy(), w.cpu().numpy()) < 2e-6
        assert not got[3].any()                                         # the zero block
        assert not ops.scalar_ws(u.device).any()
    # five blocks (one more than a launch takes) and a block on the CPU go block by block, same value
    five = L.l2_reg_loss(0.37, u, z, p, small, small)
    assert abs(five.item() - (wr.item() + 0.37 * (torch.norm(small) / 3).item())) < 2e-6 * abs(five.item())
    mixed = L.l2_reg_loss(0.37, u, small.detach().cpu())
    assert abs(mixed.item() - 0.37 * ((torch.norm(u) / rows).item() + (torch.norm(small) / 3).item())) < 2e-6 * abs(mixed.item())
    # the number of launches: what the host pays per call in the op-level tier
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        bpr, reg = L.bpr_loss(u, p, q), L.l2_reg_loss(0.37, u, p)
        (bpr + reg).backward()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ours = [k for k in names if "bpr_plain" in k or "l2_reg" in k]
    assert len(ours) == 4, names                                        # two forward, two backward


@pytest.fixture(params=["split", "f32"])
def nce_precision(request):
    """Both arithmetic modes of InfoNCE's two n x n x d products (srh_infonce_set_precision): the process default is
    switched for the test and restored."""
    before = ops.get_infonce_precision()
    ops.set_infonce_precision(request.param)
    yield request.param
    ops.set_infonce_precision(before)


@pytest.mark.parametrize("n,d,tau", [(2048, 64, 0.2), (1500, 64, 0.15), (700, 128, 0.2), (17, 64, 0.5), (900, 256, 0.2),
                                     (300, 256, 0.05), (1, 64, 0.2), (64, 64, 0.2), (65, 128, 0.3), (3000, 64, 0.2),
                                     (4096, 64, 0.2), (1100, 128, 0.1), (8300, 64, 0.25)])
def test_infonce_gathered_matches_oracle(n, d, tau, nce_precision):
    if nce_precision == "f32" and d == 256:
        pytest.skip("the all-f32 MFMA passes serve d = 64 / 128")
    if n > 8192 and nce_precision != "f32":
        pytest.skip("n > 8192 is here for the all-f32 passes' second form: pass 2 recomputes the logits (no n x n weight array)")
    rng = np.random.default_rng(n)
    rows = max(5000, n + 200)
    t1 = (rng.standard_normal((rows, d)) * 0.4).astype(np.float32)
    t2 = (t1 + rng.standard_normal((rows, d)) * 0.2).astype(np.float32)
    idx = np.sort(rng.choice(rows, size=n, replace=False)).astype(np.int32)
    # the reference's expression evaluated in float64: at tau = 0.05 on these correlated views the loss is ~3e-6 -- the
    # difference of two numbers near 20 -- and the reference's own f32 arithmetic is off by 1 % there (lse - s_ii with
    # 2e-6 of rounding on each); the kernel forms log1p(l' / e_ii) and is held to the exact value
    a = torch.tensor(t1, dtype=torch.float64, requires_grad=True); b = torch.tensor(t2, dtype=torch.float64, requires_grad=True)
    loss = 0.3 * O.info_nce(a[idx.astype(np.int64)], b[idx.astype(np.int64)], tau)
    loss.backward()
    d1, d2 = torch.from_numpy(t1).to(DEV), torch.from_numpy(t2).to(DEV)
    base = 1e-5                          # g1 is accumulated into, not overwritten
    g1 = torch.full((rows, d), base, device=DEV); g2 = torch.zeros((rows, d), device=DEV)
    out = torch.zeros(1, dtype=torch.float64, device=DEV)
    nmax = n + 100                       # device-side count smaller than the launch bound
    ws = ops.infonce_ws(nmax, d, DEV)
    didx = torch.zeros(nmax, dtype=torch.int32, device=DEV); didx[:n] = torch.from_numpy(idx).to(DEV)
    ops.infonce_fwd_bwd(d1, d2, didx, nmax, n_dev=torch.tensor([n], dtype=torch.int32, device=DEV), tau=tau,
                        loss_scale=0.3, loss=out, g1=g1, g2=g2, ws=ws)
    if n == 1:                           # (one pair: log_softmax of a 1 x 1 matrix -- the loss and its gradients are exactly 0)
        assert out.item() == 0.0 and not (g1 - base).any() and not g2.any()
        return
    assert abs(out.item() - loss.item()) / abs(loss.item()) < 1e-5
    assert rel_err((g1.double() - base).cpu().numpy(), a.grad.numpy()) < (2e-5 if nce_precision == "split" else 3e-6)
    assert rel_err(g2.cpu().numpy(), b.grad.numpy()) < (2e-5 if nce_precision == "split" else 3e-6)


@pytest.mark.parametrize("ns,d", [((1877,), 64), ((2048,), 64), ((4096,), 64), ((1640,), 128), ((1877, 1640), 64),
                                  ((1500, 1999), 128)])
def test_f32_infonce_is_the_same_bits_fifty_times_over(ns, d):
    """The all-f32 passes (persistent workgroups, an LDS ring fed by buffer_load ... lds, hand-placed waits) at the batch
    sizes the benchmark runs -- n ~ 1877 unique users / 1640 unique items of 2048 pairs, d = 64 -- where round 5 found and
    fixed a stale-register bug that showed up as a DIFFERENT loss on every run.  Fifty calls on the same inputs, one and two
    problems per call as the engine issues them: loss and both gradients identical bit for bit across the calls, and within
    2e-6 (loss) / 3e-6 (gradients) of the float64 evaluation of util/loss_torch.py:35-50."""
    tau, scale = 0.2, 0.2
    rng = np.random.default_rng(sum(ns) + d)
    v1 = [(rng.standard_normal((n, d)) * 0.4).astype(np.float32) for n in ns]
    v2 = [(a + rng.standard_normal(a.shape) * 0.2).astype(np.float32) for a in v1]
    want_loss, want_g1, want_g2 = 0.0, [], []
    for a, b in zip(v1, v2):
        ta = torch.tensor(a, dtype=torch.float64, requires_grad=True); tb = torch.tensor(b, dtype=torch.float64, requires_grad=True)
        l = scale * O.info_nce(ta, tb, tau)
        l.backward()
        want_loss += l.item(); want_g1.append(ta.grad.numpy()); want_g2.append(tb.grad.numpy())
    d1 = [torch.from_numpy(a).to(DEV) for a in v1]; d2 = [torch.from_numpy(b).to(DEV) for b in v2]
    g1 = [torch.zeros_like(t) for t in d1]; g2 = [torch.zeros_like(t) for t in d2]
    out = torch.zeros(1, dtype=torch.float64, device=DEV)
    ws = torch.empty(sum(ops.infonce_ws(n, d, DEV).numel() for n in ns), dtype=torch.uint8, device=DEV)
    first = None
    for rep in range(50):
        for t in g1 + g2:
            t.zero_()
        out.zero_()
        ops.infonce_multi([(d1[k], d2[k], None, n, None, g1[k], g2[k]) for k, n in enumerate(ns)], d=d, tau=tau,
                          loss_scale=scale, loss=out, ws=ws, precision="f32")
        got = [out.clone()] + [t.clone() for t in g1 + g2]
        if first is None:
            first = got
            assert abs(out.item() - want_loss) / abs(want_loss) < 2e-6
            for k in range(len(ns)):
                assert rel_err(g1[k].cpu().numpy(), want_g1[k]) < 3e-6
                assert rel_err(g2[k].cpu().numpy(), want_g2[k]) < 3e-6
        else:
            assert all(torch.equal(x, y) for x, y in zip(first, got)), rep


def dev_segments(u, i, j, pad, rows_are_zero=False):
    from .conftest import host_batch_segments
    seg = {k: torch.from_numpy(v).to(DEV) for k, v in host_batch_segments(u, i, j, pad).items()}
    seg["rows_are_zero"] = rows_are_zero
    return seg


@pytest.mark.parametrize("segmented", [False, True, "store"], ids=["atomics", "segments", "segments-store"])
def test_bpr_l2_fused_matches_oracle_with_duplicates(segmented):
    rng = np.random.default_rng(0)
    U, I, d, B = 300, 400, 64, 1000
    ut = (rng.standard_normal((U, d)) * 0.3).astype(np.float32)
    it = (rng.standard_normal((I, d)) * 0.3).astype(np.float32)
    ui, pi, ni = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    for include_neg, ego in ((False, False), (True, False), (True, True)):
        a = torch.tensor(ut, requires_grad=True); b = torch.tensor(it, requires_grad=True)
        ea = torch.tensor(ut * 0.5, requires_grad=True); eb = torch.tensor(it * 0.5, requires_grad=True)
        ra, rb = (ea, eb) if ego else (a, b)
        regs = [ra[ui], rb[pi]] + ([rb[ni]] if include_neg else [])
        bpr = O.bpr_loss(a[ui], b[pi], b[ni]); reg = O.l2_reg_loss(1e-3, *regs)
        (bpr + reg).backward()
        du, di = torch.from_numpy(ut).to(DEV), torch.from_numpy(it).to(DEV)
        dru, dri = (du * 0.5, di * 0.5) if ego else (du, di)
        gu, gi = torch.zeros_like(du), torch.zeros_like(di)
        gru, gri = (torch.zeros_like(du), torch.zeros_like(di)) if ego else (gu, gi)
        losses = torch.zeros(2, dtype=torch.float64, device=DEV)
        cnt = torch.tensor([B], dtype=torch.int32, device=DEV)
        idx = [torch.zeros(B + 24, dtype=torch.int32, device=DEV) for _ in range(3)]
        for t, src in zip(idx, (ui, pi, ni)):
            t[:B] = torch.from_numpy(src.astype(np.int32)).to(DEV)
        # (segmented: every row written once by the row group that owns it, its slots summed in order -- srh_bpr_l2_fwd_bwd_p)
        kw = dict(seg=dev_segments(ui, pi, ni, B + 24, rows_are_zero=segmented == "store")) if segmented else {}
        runs = []
        for rep in range(3 if segmented else 1):
            for t in (gu, gi, gru, gri):
                t.zero_()
            losses.zero_()
            ops.bpr_l2_fwd_bwd(du, di, dru.contiguous(), dri.contiguous(), *idx, batch=B + 24, n_rows_dev=cnt,
                               reg_coef=1e-3, reg_include_neg=include_neg, loss_scale=1.0, g_user=gu, g_item=gi,
                               greg_user=gru, greg_item=gri, losses=losses, ws=ops.bpr_ws(B + 24, DEV), **kw)
            runs.append([t.clone() for t in (gu, gi, gru, gri, losses)])
        np.testing.assert_allclose(losses.cpu().numpy(), [bpr.item(), reg.item()], rtol=2e-6)
        assert rel_err(gu.cpu().numpy(), a.grad.numpy()) < 2e-5
        assert rel_err(gi.cpu().numpy(), b.grad.numpy()) < 2e-5
        if ego:
            assert rel_err(gru.cpu().numpy(), ea.grad.numpy()) < 2e-5
            assert rel_err(gri.cpu().numpy(), eb.grad.numpy()) < 2e-5
        for other in runs[1:]:                      # no float atomics: the same bits every time
            assert all(torch.equal(x, y) for x, y in zip(runs[0], other))


@pytest.mark.parametrize("segmented", [False, True, "store"], ids=["atomics", "segments", "segments-store"])
@pytest.mark.parametrize("d,B", [(64, 2048), (128, 600), (256, 500)])
def test_bpr_infonce_one_call_matches_oracle(d, B, nce_precision, segmented):
    if nce_precision == "f32" and d == 256:
        pytest.skip("the all-f32 MFMA passes serve d = 64 / 128")
    """srh_bpr_infonce_fwd_bwd = XSimGCL.py:30-35: rec + reg + cl_rate * (user InfoNCE + item InfoNCE), with
    the gradients of the final and the contrast-layer tables, against torch autograd on the oracle losses."""
    rng = np.random.default_rng(d + B)
    U, I, tau, cl_rate, reg = 900, 1300, 0.2, 0.2, 1e-4
    F = (rng.standard_normal((U + I, d)) * 0.3).astype(np.float32)
    CL = (F + rng.standard_normal((U + I, d)) * 0.1).astype(np.float32)
    ui, pi, ni = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    uu, up = np.unique(ui), np.unique(pi)
    f = torch.tensor(F, requires_grad=True); c = torch.tensor(CL, requires_grad=True)
    fu, fi, cu, ci = f[:U], f[U:], c[:U], c[U:]
    rec = O.bpr_loss(fu[ui], fi[pi], fi[ni]); l2 = O.l2_reg_loss(reg, fu[ui], fi[pi])
    cl = cl_rate * (O.info_nce(fu[uu], cu[uu], tau) + O.info_nce(fi[up], ci[up], tau))
    (rec + l2 + cl).backward()

    dF, dC = torch.from_numpy(F).to(DEV), torch.from_numpy(CL).to(DEV)
    gF, gC = torch.zeros_like(dF), torch.zeros_like(dC)
    losses = torch.zeros(3, dtype=torch.float64, device=DEV)
    i32 = lambda a, n: torch.cat([torch.from_numpy(a.astype(np.int32)), torch.zeros(n - a.size, dtype=torch.int32)]).to(DEV)
    cnt = lambda n: torch.tensor([n], dtype=torch.int32, device=DEV)
    nce_ws = torch.empty(2 * ops.infonce_ws(B, d, DEV).numel(), dtype=torch.uint8, device=DEV)
    # segmented: the rows' (slot, role) lists come with the call (srh_batch_segments_t, nce_rows = 1: the user-side problem's
    # row i is user group i, the item-side one's positive-item group i) and every gradient row is written exactly once
    kw = dict(seg=dev_segments(ui, pi, ni, B, rows_are_zero=segmented == "store"), nce_rows=1) if segmented else {}
    runs = []
    for rep in range(3):                     # the workspaces re-arm themselves: a second call gives the same
        gF.zero_(); gC.zero_(); losses.zero_()
        ops.bpr_infonce(dF[:U], dF[U:], dF[:U], dF[U:], i32(ui, B), i32(pi, B), i32(ni, B), batch=B, n_rows_dev=cnt(B),
                        reg_coef=reg, reg_include_neg=False, loss_scale=1.0, g_user=gF[:U], g_item=gF[U:],
                        greg_user=gF[:U], greg_item=gF[U:], losses=losses[0:2], bpr_ws=ops.bpr_ws(B, DEV),
                        problems=[(dF[:U], dC[:U], i32(uu, B), B, cnt(uu.size), gF[:U], gC[:U]),
                                  (dF[U:], dC[U:], i32(up, B), B, cnt(up.size), gF[U:], gC[U:])],
                        tau=tau, cl_scale=cl_rate, cl_loss=losses[2:3], nce_ws=nce_ws, **kw)
        np.testing.assert_allclose(losses.cpu().numpy(), [rec.item(), l2.item(), cl.item()], rtol=1e-5)
        assert rel_err(gF.cpu().numpy(), f.grad.numpy()) < 2e-5
        assert rel_err(gC.cpu().numpy(), c.grad.numpy()) < 2e-5
        runs.append((gF.clone(), gC.clone(), losses.clone()))
    if segmented:                            # no float atomics anywhere in the call: the same bits every time
        for other in runs[1:]:
            assert all(torch.equal(x, y) for x, y in zip(runs[0], other))


@pytest.mark.parametrize("d", [64, 128, 256])
def test_filtered_ranking_equals_exact_ranking(d):
    """srh_score_mask_topk_filtered (scores never stored) against srh_score_mask_topk: identical ids and scores
    on every row whose survivor list fits; tie-heavy rows (all-zero user vectors) report an overflow instead."""
    rng = np.random.default_rng(31 + d)
    U, I, K = 1500, 20000, 20
    ue = torch.from_numpy((rng.standard_normal((U, d)) * 0.3).astype(np.float32)).to(DEV)
    ie = torch.from_numpy((rng.standard_normal((I, d)) * 0.3).astype(np.float32)).to(DEV)
    ue[7] = 0.0; ue[900] = 0.0                          # every score equal: thousands of survivors
    ie[100:140] = ie[100]                                # a block of tied items inside the sample slice
    lens = rng.integers(0, 60, U); lens[3] = 5000        # one user with a huge training row
    indptr = np.zeros(U + 1, dtype=np.int32); np.cumsum(lens, out=indptr[1:])
    indices = np.concatenate([np.sort(rng.choice(I, size=n, replace=False)) for n in lens]).astype(np.int32)
    r_indptr, r_indices = torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV)
    users = torch.from_numpy(rng.permutation(U).astype(np.int32)).to(DEV)
    want_ids, want_sc = ops.score_mask_topk(ue, users, ie, r_indptr, r_indices, K)
    for chunk in (4096, 700):                            # one chunk / several chunks with a ragged tail
        ids, sc, counts, _ = ops.score_mask_topk_filtered(ue, users, ie, r_indptr, r_indices, K, sample_items=2048,
                                                          cap=512, chunk_rows=chunk)
        ok = (counts <= 512)
        bad_users = set(users[~ok].tolist())
        assert bad_users == {7, 900}
        assert (counts[ok] >= K).all()
        assert torch.equal(ids[ok], want_ids[ok]) and torch.equal(sc[ok], want_sc[ok])


@pytest.mark.parametrize("d,scale", [(64, 30.0), (64, 1e-3), (128, 5.0)])
def test_split_bf16_filter_never_loses_a_top_k_item(d, scale):
    """At d = 64 / 128 the filter pass of srh_score_mask_topk_filtered decides on bf16 products against per-item error
    bounds (3.94e-3 |u| |i_j|: csrc/eval.hip), and the survivors are re-scored exactly.  Stress for those margins: large and tiny
    magnitudes, a catalogue of near-duplicates (thousands of items within 1e-6 relative of each other, i.e. inside the
    bf16 products' error of the bound), items of very unequal norms -- ids AND scores must still be bit-identical to the
    plain pipeline's on every row that fits."""
    rng = np.random.default_rng(7 + d)
    U, I, K = 700, 12000, 20
    ue = torch.from_numpy((rng.standard_normal((U, d)) * scale).astype(np.float32)).to(DEV)
    base = (rng.standard_normal((40, d)) * scale).astype(np.float32)
    items = base[rng.integers(0, 40, I)] * (1.0 + 1e-6 * rng.standard_normal((I, 1))).astype(np.float32)   # near-duplicates
    items[::7] = (rng.standard_normal((len(items[::7]), d)) * scale * 0.05).astype(np.float32)             # small-norm items
    items[5] *= 20.0                                                                                        # one huge item
    ie = torch.from_numpy(items.astype(np.float32)).to(DEV)
    r_indptr = torch.zeros(U + 1, dtype=torch.int32, device=DEV)
    r_indices = torch.zeros(1, dtype=torch.int32, device=DEV)
    users = torch.arange(U, dtype=torch.int32, device=DEV)
    want_ids, want_sc = ops.score_mask_topk(ue, users, ie, r_indptr, r_indices, K)
    ids, sc, counts, _ = ops.score_mask_topk_filtered(ue, users, ie, r_indptr, r_indices, K, sample_items=2048, cap=1024,
                                                      chunk_rows=512)
    ok = counts <= 1024
    assert ok.float().mean() > 0.5 and (counts[ok] >= K).all()
    assert torch.equal(ids[ok], want_ids[ok]) and torch.equal(sc[ok], want_sc[ok])


# ------------------------------------------------------------------------------------------
# (a-9) Adam
# ------------------------------------------------------------------------------------------
@pytest.mark.selfcheck
@pytest.mark.parametrize("d", [64, 128, 256])
def test_adam_in_the_product_epilogue_equals_product_then_adam(d):
    """SRH_EPI_ADAM: the last backward product with the optimiser step in its row epilogue leaves the bits of
    srh_spmm_f32 (AXPY) followed by srh_adam_step_reset -- parameters, both moments, the cleared batch rows of the sparse
    gradient buffers, the cursor -- with this step's constants written by srh_batch_fetch."""
    rng = np.random.default_rng(90 + d)
    m = powerlaw_csr(3000, 3000, 40000, seed=5, heavy_rows=3, heavy_len=1500, empty_rows=40)
    N, B, E = 3000, 64, 64
    csr = ops.DeviceCSR.from_scipy(m, split_len=64)            # (split rows: the epilogue runs in the last segment to arrive)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)     # noqa: E731
    x = t(rng.standard_normal((N, d)) * 1e-3)
    ids = {k: torch.from_numpy(rng.integers(0, N, E).astype(np.int32)).to(DEV) for k in ("u", "i", "j")}
    step = 7

    def state():
        r = np.random.default_rng(3)
        gF, gCL = np.zeros((N, d), np.float32), np.zeros((N, d), np.float32)
        rows = np.unique(np.concatenate([v.cpu().numpy() for v in ids.values()]))
        gF[rows] = r.standard_normal((rows.size, d)) * 1e-2
        gCL[rows] = r.standard_normal((rows.size, d)) * 1e-2
        return dict(p=t(r.standard_normal((N, d)) * 0.1), m=t(r.standard_normal((N, d)) * 1e-3),
                    v=t(r.random((N, d)) * 1e-5), gF=t(gF), gCL=t(gCL), g=torch.zeros(N, d, device=DEV),
                    cursor=torch.tensor([0, step], dtype=torch.int64, device=DEV), now=torch.zeros(2, dtype=torch.int64, device=DEV),
                    mark=torch.zeros(N, dtype=torch.int32, device=DEV), coef=torch.zeros(2, device=DEV),
                    stage={k: torch.zeros(B, dtype=torch.int32, device=DEV) for k in ("u", "i", "j")},
                    meta=torch.zeros(4, dtype=torch.int32, device=DEV))

    def fetch(s, **kw):
        ops.batch_fetch(ids, E, B, s["cursor"], s["stage"], s["meta"], row_mark=s["mark"], now=s["now"], **kw)
    a, b = state(), state()
    # the separate pass
    fetch(a)
    ops.spmm(csr, x, out=a["g"], epilogue=ops.make_epilogue(add=[a["gF"], a["gCL"]], add_scale=[0.25, 1.0], alpha=0.25,
                                                            add_mark=a["mark"], mark_stamp=a["now"][1:2], add_sparse=[True, True]))
    ops.adam_step(a["p"], a["g"], a["m"], a["v"], step_dev=a["now"][1:2], lr=1e-3, clear=[a["gF"], a["gCL"]],
                  row_mark=a["mark"], advance_cursor=a["cursor"])
    # the fused launch
    fetch(b, adam_coef=b["coef"], adam_lr=1e-3)
    b1, b2, lr = (float(np.float32(c)) for c in (0.9, 0.999, 1e-3))        # (the C ABI takes the betas as float)
    assert np.allclose(b["coef"].cpu().numpy(), (lr / (1 - b1 ** step), (1 - b2 ** step) ** 0.5), rtol=1e-6)
    ops.spmm(csr, x, out=b["g"], epilogue=ops.make_epilogue(
        add=[b["gF"], b["gCL"]], add_scale=[0.25, 1.0], alpha=0.25, add_mark=b["mark"], mark_stamp=b["now"][1:2],
        add_sparse=[True, True],
        adam=dict(param=b["p"], m=b["m"], v=b["v"], coef=b["coef"], clear=[b["gF"], b["gCL"]], clear_mark=b["mark"],
                  cursor=b["cursor"])))
    assert float(a["g"].abs().max()) > 0 and float(b["g"].abs().max()) == 0         # (the gradient is never stored)
    for k in ("p", "m", "v", "gF", "gCL", "cursor"):
        assert torch.equal(a[k], b[k]), k
    assert b["cursor"].tolist() == [1, step + 1] and float(b["gF"].abs().max()) == 0
    assert float((a["p"] - state()["p"]).abs().max()) > 0
    # the stamp may not be the cursor the launch advances; no perturbation / mean next to the optimiser
    with pytest.raises(ops.SelfrecHipError):
        ops.spmm(csr, x, out=b["g"], epilogue=ops.make_epilogue(
            mark_stamp=b["cursor"][1:2], adam=dict(param=b["p"], m=b["m"], v=b["v"], coef=b["coef"], clear=[b["gF"]],
                                                   clear_mark=b["mark"], cursor=b["cursor"][0:2])))
    with pytest.raises(ops.SelfrecHipError):
        ops.spmm(csr, x, out=b["g"], epilogue=ops.make_epilogue(
            perturb_eps=0.1, rng_seed=1, adam=dict(param=b["p"], m=b["m"], v=b["v"], coef=b["coef"])))


def test_adam_matches_torch_optim():
    rng = np.random.default_rng(4)
    p0 = rng.standard_normal((1000, 64)).astype(np.float32) * 0.1
    ref = torch.nn.Parameter(torch.tensor(p0)); opt = torch.optim.Adam([ref], lr=1e-3)
    p, m, v = torch.tensor(p0, device=DEV), torch.zeros(1000, 64, device=DEV), torch.zeros(1000, 64, device=DEV)
    step_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
    for step in range(1, 6):
        g = (rng.standard_normal((1000, 64)) * 10 ** rng.uniform(-6, 0)).astype(np.float32)
        ref.grad = torch.tensor(g); opt.step()
        step_dev += 1
        if step % 2:
            ops.adam_step(p, torch.tensor(g, device=DEV), m, v, step=step, lr=1e-3)
        else:
            ops.adam_step(p, torch.tensor(g, device=DEV), m, v, step_dev=step_dev, lr=1e-3)
        assert rel_err(p.cpu().numpy(), ref.detach().numpy()) < 1e-6
    pn, mn, vn = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)   # numpy oracle agrees too
    O.adam_step(pn, g, mn, vn, 1, 1e-3)
    assert np.isfinite(pn).all()


# ------------------------------------------------------------------------------------------
# (a-10/a-11) scoring GEMM, mask, top-K
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,d", [(1, 33, 64), (70, 1000, 64), (257, 4099, 128), (64, 96, 32), (100, 777, 256)])
def test_gemm_nt_is_exact_f32(m, n, d):
    rng = np.random.default_rng(m + n)
    a = rng.standard_normal((m, d)).astype(np.float32)
    b = rng.standard_normal((n, d)).astype(np.float32)        # asymmetric operands: catches transposes
    got = ops.gemm_nt(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
    want = a.astype(np.float64) @ b.astype(np.float64).T
    assert got.shape == (m, n) and rel_err(got, want) < 2e-6


def test_topk_rows_order_ties_and_fallback():
    rng = np.random.default_rng(8)
    x = rng.standard_normal((37, 5003)).astype(np.float32)
    x[3, :] = 1.0                                  # all equal: lowest ids win (tie fallback path)
    x[4, ::2] = 7.0                                # 2502 ties for the top: overflows the candidate list
    x[5, 100:130] = -10e8                          # masked block
    ids, sc = ops.topk_rows(torch.from_numpy(x).to(DEV), 20)
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    for r in range(x.shape[0]):
        order = np.lexsort((np.arange(x.shape[1]), -x[r]))[:20]
        assert np.array_equal(ids[r], order), r
        assert np.array_equal(sc[r], x[r][order])


@pytest.mark.parametrize("name", ["MF", "LightGCN", "XSimGCL", "SimGCL", "SGL"])
def test_full_rank_eval_matches_reference(golden_models, tiny_data, name):
    gm, d = golden_models, tiny_data
    ue = torch.from_numpy(gm[f"{name}_final_user"]).to(DEV)
    ie = torch.from_numpy(gm[f"{name}_final_item"]).to(DEV)
    g = d.device_graph()
    users = torch.from_numpy(gm[f"{name}_test_users"]).to(DEV)
    ids, sc = ops.score_mask_topk(ue, users, ie, g.r_indptr, g.r_indices, 20)
    want_ids, want_sc = gm[f"{name}_rec_ids"], gm[f"{name}_rec_scores"]
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    np.testing.assert_allclose(sc, want_sc, rtol=1e-5, atol=1e-7)
    # ids identical except where two neighbouring scores differ by less than fp32 summation noise
    bad = ids != want_ids
    if bad.any():
        gaps = np.abs(np.diff(want_sc, axis=1))
        near = np.zeros_like(bad); near[:, :-1] |= gaps < 1e-6; near[:, 1:] |= gaps < 1e-6
        assert not (bad & ~near).any()


def test_metric_rows_equals_the_host_loops():
    """srh_metric_rows: per-user hits and DCG / IDCG at several cut-offs from the hit flags -- bit-identical to the python
    loops of util/evaluation.py:7-16,66-78 (float64 adds in position order, python's own gain / ideal tables)."""
    import math
    rng = np.random.default_rng(5)
    U, K = 3000, 20
    flags = (rng.random((U, K)) < 0.07).astype(np.uint8)
    flags[3] = 1; flags[4] = 0
    sizes = rng.integers(1, 40, U).astype(np.int32)
    cuts = [5, 10, 20]
    hits, ndcg = ops.metric_rows(torch.from_numpy(flags).to(DEV), torch.from_numpy(sizes).to(DEV), cuts)
    for c, n in enumerate(cuts):
        want_h, want_n = np.zeros(U, dtype=np.int64), np.zeros(U)
        for u in range(U):
            gain = sum(1.0 / math.log(pos + 2, 2) for pos in range(n) if flags[u, pos])
            ideal = sum(1.0 / math.log(pos + 2, 2) for pos in range(min(int(sizes[u]), n)))
            want_h[u], want_n[u] = int(flags[u, :n].sum()), gain / ideal
        assert np.array_equal(hits[c].cpu().numpy(), want_h)
        assert np.array_equal(ndcg[c].cpu().numpy(), want_n)          # bit for bit
