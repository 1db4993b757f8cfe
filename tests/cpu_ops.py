"""Torch-CPU stand-in for the part of ``selfrec_amd.ops`` that ``engine.FusedTrainer`` calls -- TEST
INFRASTRUCTURE ONLY (tests/test_dist_cpu.py swaps it in for ``engine.ops``).

It lets the world_size-2/3 gloo tests run the *product's* step code -- table layout, id -> row mapping,
local slices, epilogue wiring, activity marks, all-gathers, cursor handling -- on a machine without a
GPU.  The arithmetic of each op is the CPU oracle's; the product never imports this module."""
import numpy as np
import scipy.sparse as sp
import torch

from oracle import selfrec_oracle as O
from selfrec_amd import ops as _real_ops

SelfrecHipError = _real_ops.SelfrecHipError
Sampler = _real_ops.Sampler          # host-only C++ (MT19937 replay): needs no GPU
column_class_order = _real_ops.column_class_order    # host-only numpy
NCE_WIDTHS, ROW_WIDTHS, SPMM_WIDTHS, padded_width = _real_ops.NCE_WIDTHS, _real_ops.ROW_WIDTHS, _real_ops.SPMM_WIDTHS, _real_ops.padded_width


def require_gpu():
    pass


class DeviceCSR:
    def __init__(self, indptr, indices, vals, shape, device=None, split_len=0, structure_of=None, xcd_split_row=0,
                 row_mid=None):
        self.shape = (int(shape[0]), int(shape[1]))
        if structure_of is not None:
            self.indptr, self.indices, self._plan = structure_of.indptr, structure_of.indices, structure_of._plan
        else:
            self.indptr = torch.as_tensor(np.asarray(indptr), dtype=torch.int32)
            self.indices = torch.as_tensor(np.asarray(indices), dtype=torch.int32)
            self._plan = object()
        self.vals = vals if isinstance(vals, torch.Tensor) else torch.as_tensor(np.asarray(vals), dtype=torch.float32)
        self.nnz = int(self.indices.numel())

    @property
    def _m(self):        # (values may be rewritten in place by adj_sym_normalize: build on use)
        m = sp.csr_matrix((self.vals.numpy(), self.indices.numpy(), self.indptr.numpy()), shape=self.shape)
        return O.to_torch_sparse(m)

    def with_values(self, vals):
        return DeviceCSR(None, None, vals, self.shape, structure_of=self)


def adj_sym_normalize(indptr, indices, edge_id, keep, n_rows, weight=None, out=None, deg_ws=None, inv_sqrt_table=None,
                      row_offset=0, phase=0):
    ip, ix = indptr.numpy().astype(np.int64), indices.numpy().astype(np.int64)
    rows = np.repeat(np.arange(n_rows), np.diff(ip))
    w = np.ones(ix.size, dtype=np.float32) if weight is None else weight.numpy()
    kept = np.ones(ix.size, dtype=bool) if keep is None else keep.numpy()[edge_id.numpy().astype(np.int64)] != 0
    if phase != 2:
        deg = np.bincount(rows, weights=np.where(kept, w, 0.0), minlength=n_rows).astype(np.float32)
        with np.errstate(divide="ignore"):
            dinv = np.power(deg, -0.5).astype(np.float32)            # graph.py:14-15
        dinv[np.isinf(dinv)] = 0.0
        deg_ws[row_offset:row_offset + n_rows] = torch.from_numpy(dinv)
    if phase != 1:
        d = deg_ws.numpy()
        vals = np.where(kept, (d[row_offset + rows] * w) * d[ix], 0.0).astype(np.float32)
        out.copy_(torch.from_numpy(vals))
    return out


def spmm3(csrs, x, outs):
    for c, o in zip(csrs, outs):
        o.copy_(torch.sparse.mm(c._m, x))


def make_epilogue(**kw):
    return kw


def _live(mark, stamp):
    return mark == int(stamp.item())


def spmm(csr, x, out=None, epilogue=None):
    ep = epilogue or {}
    assert x.shape[0] == csr.shape[1]
    if ep.get("col_mark") is not None:          # columns that are not live hold zeros by contract
        x = x * _live(ep["col_mark"], ep["mark_stamp"]).unsqueeze(1)
    y = torch.sparse.mm(csr._m, x)
    if ep.get("add") or ep.get("alpha", 1.0) != 1.0:
        y = y * ep.get("alpha", 1.0)
        for a, s in zip(ep.get("add") or [], ep.get("add_scale") or []):
            y = y + s * a                       # (batch-sparse addends are zero off the live rows)
    rows = slice(None) if ep.get("row_mark") is None else _live(ep["row_mark"], ep["mark_stamp"])
    if ep.get("perturb_eps") is not None:
        raw = y
        c0, w = int(ep.get("col0") or 0), y.shape[1]

        def perturbed(noise):
            assert noise is not None, "the CPU stand-in needs injected noise"
            if ep.get("d_full"):        # column slice: the unit vector is normalised over the whole row
                unit = torch.nn.functional.normalize(noise, dim=-1)[:, c0:c0 + w]
                return raw + torch.sign(raw) * unit * ep["perturb_eps"]
            return O.perturb_(raw.clone(), noise, ep["perturb_eps"])
        for k, extra in enumerate(ep.get("extra_out") or []):           # FANOUT: more perturbed copies of the product
            extra[rows] = perturbed(ep["extra_noise"][k])[rows]
        if not ep.get("main_clean"):
            y = perturbed(ep.get("noise"))
    out[rows] = y[rows]
    if ep.get("mean_out") is not None:
        mean = torch.stack(list(ep.get("prev") or []) + [y], dim=1).sum(1) / ep["mean_div"]
        ep["mean_out"][rows] = mean[rows]
    return out


def axpby(a, x, b, y):
    y.copy_(a * x + (b * y if b != 0.0 else 0.0))


def batch_fetch(ep, n_edges, batch_size, cursor, stage, meta, row_mark=None, mark_item_offset=0, zero4=None,
                stage_cat=None, cat_item_offset=0, n_cat=None, now=None, half_batches=0):
    b, stamp = int(cursor[0]), int(cursor[1])
    if now is not None:
        now.copy_(cursor)
    lo = b * batch_size
    in_epoch = (b - half_batches if half_batches and b >= half_batches else b) * batch_size
    rows = max(0, min(batch_size, n_edges - in_epoch))
    for k in ("u", "i", "j"):
        stage[k][:rows] = ep[k][lo:lo + rows]
    if row_mark is not None and rows:
        row_mark[stage["u"][:rows].long()] = stamp
        row_mark[stage["i"][:rows].long() + mark_item_offset] = stamp
        row_mark[stage["j"][:rows].long() + mark_item_offset] = stamp
    a = c = 0
    if ep.get("uniq_u") is not None and rows:
        a, c = int(ep["n_uniq_u"][b]), int(ep["n_uniq_i"][b])
        stage["uniq_u"][:a] = ep["uniq_u"][b * batch_size:b * batch_size + a]
        stage["uniq_i"][:c] = ep["uniq_i"][b * batch_size:b * batch_size + c]
    if stage_cat is not None and ep.get("uniq_u") is not None and rows:
        stage_cat[:a] = stage["uniq_u"][:a]
        stage_cat[a:a + c] = stage["uniq_i"][:c] + cat_item_offset
    if n_cat is not None:
        n_cat.fill_(a + c)
    if zero4 is not None:
        zero4.zero_()
    meta.copy_(torch.tensor([rows, a, c, b], dtype=torch.int32))


def cursor_advance(cursor):
    cursor += 1


def zero_rows(lists, d, cursor_advance=None):
    for table, idx, n_dev, n_max, off in lists:
        n = min(int(n_dev), n_max) if n_dev is not None else n_max
        table[idx[:n].long() + off] = 0.0
    if cursor_advance is not None:
        cursor_advance += 1


def bpr_ws(B, device):
    return torch.empty(0)


def infonce_ws(n, d, device):
    return torch.empty(1, dtype=torch.uint8)


def bpr_l2_fwd_bwd(user, item, reg_user, reg_item, u_idx, i_idx, j_idx, *, batch, n_rows_dev=None, reg_coef,
                   reg_include_neg, loss_scale, g_user, g_item, greg_user, greg_item, losses, ws=None):
    rows = min(int(n_rows_dev), batch) if n_rows_dev is not None else batch
    if rows == 0:
        return
    ui, pi, ni = (t[:rows].long() for t in (u_idx, i_idx, j_idx))
    leaves = {}

    def leaf(t):
        if t.data_ptr() not in leaves:
            leaves[t.data_ptr()] = t.detach().clone().requires_grad_()
        return leaves[t.data_ptr()]
    U, I, RU, RI = leaf(user), leaf(item), leaf(reg_user), leaf(reg_item)
    bpr = O.bpr_loss(U[ui], I[pi], I[ni])
    regs = [RU[ui], RI[pi]] + ([RI[ni]] if reg_include_neg else [])
    reg = O.l2_reg_loss(reg_coef, *regs)
    (loss_scale * (bpr + reg)).backward()
    losses[0] += loss_scale * bpr.item()
    losses[1] += loss_scale * reg.item()
    # d/d table goes to the matching gradient table (BPR part -> g_*, regulariser part -> greg_*); when the
    # regularised table IS the scored table the two gradients arrive summed, as in the kernel
    same_u, same_i = reg_user.data_ptr() == user.data_ptr(), reg_item.data_ptr() == item.data_ptr()
    done = set()
    for src, dst in ((user, g_user), (item, g_item), (reg_user, greg_user if not same_u else g_user),
                     (reg_item, greg_item if not same_i else g_item)):
        if src.data_ptr() in done:
            continue
        done.add(src.data_ptr())
        g = leaves[src.data_ptr()].grad
        if g is not None:
            dst += g


def _infonce(problems, tau, scale, loss):
    for v1, v2, idx, n_max, n_dev, g1, g2, *_excl in problems:
        n = min(int(n_dev), n_max) if n_dev is not None else n_max
        if n <= 0:
            continue
        rows = idx[:n].long() if idx is not None else torch.arange(n)
        same = v1.data_ptr() == v2.data_ptr()
        a = v1.detach().clone().requires_grad_()
        b = a if same else v2.detach().clone().requires_grad_()
        l = scale * O.info_nce(a[rows], b[rows], tau)
        l.backward()
        loss += l.item()
        g1 += a.grad
        if not same:
            g2 += b.grad


NCE_PRECISIONS = {"split": 0, "f32": 1, "bf16x3": 0}


def infonce_multi(problems, *, d, tau, loss_scale, loss, ws=None, precision=None):
    _infonce(problems, tau, loss_scale, loss)


def bpr_infonce(user, item, reg_user, reg_item, u_idx, i_idx, j_idx, *, batch, n_rows_dev=None, reg_coef,
                reg_include_neg, loss_scale, g_user, g_item, greg_user, greg_item, losses, bpr_ws=None, problems,
                tau, cl_scale, cl_loss, nce_ws=None, precision=None):
    bpr_l2_fwd_bwd(user, item, reg_user, reg_item, u_idx, i_idx, j_idx, batch=batch, n_rows_dev=n_rows_dev,
                   reg_coef=reg_coef, reg_include_neg=reg_include_neg, loss_scale=loss_scale, g_user=g_user,
                   g_item=g_item, greg_user=greg_user, greg_item=greg_item, losses=losses)
    _infonce(problems, tau, cl_scale, cl_loss)


def adam_step(param, grad, m, v, *, step=0, step_dev=None, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    t = int(step_dev) if step_dev is not None else int(step)
    p, g, mm, vv = (x.contiguous() for x in (param, grad, m, v))
    pn, mn, vn = p.numpy().copy(), mm.numpy().copy(), vv.numpy().copy()
    O.adam_step(pn, g.numpy(), mn, vn, t, lr)
    param.copy_(torch.from_numpy(pn)); m.copy_(torch.from_numpy(mn)); v.copy_(torch.from_numpy(vn))


# ---- column-sharded tables: the batch-row exchange (csrc/exchange.hip) ----
def batch_lists(stage, meta, batch_size):
    return {"stage": stage, "meta": meta, "B": int(batch_size)}


def _live_slots(lists):
    """[(compact slot, node)] of every live slot of [u | i | j | uniq_u | uniq_i]."""
    B, meta, st = lists["B"], lists["meta"], lists["stage"]
    out = []
    for s, (name, cnt) in enumerate((("u", 0), ("i", 0), ("j", 0), ("uniq_u", 1), ("uniq_i", 2))):
        n = min(int(meta[cnt]), B)
        out += [(s * B + k, int(st[name][k])) for k in range(n)]
    return out


def batch_pack(lists, tables, send, cat_idx=None, n_cat=None):
    send.zero_()
    live = _live_slots(lists)
    slots = torch.tensor([k for k, _ in live], dtype=torch.long)
    nodes = torch.tensor([n for _, n in live], dtype=torch.long)
    for t, table in enumerate(tables):
        send[t][slots] = table[nodes]
    if cat_idx is not None:
        B, meta = lists["B"], lists["meta"]
        a, c = min(int(meta[1]), B), min(int(meta[2]), B)
        cat_idx[:a] = torch.arange(3 * B, 3 * B + a, dtype=torch.int32)
        cat_idx[a:a + c] = torch.arange(4 * B, 4 * B + c, dtype=torch.int32)
        if n_cat is not None:
            n_cat.fill_(a + c)


def batch_unpack(lists, recv, world, dl, compact, compact_grads):
    slots = torch.tensor([k for k, _ in _live_slots(lists)], dtype=torch.long)
    for t, table in enumerate(compact):
        whole = recv[:, t].permute(1, 0, 2).reshape(recv.shape[2], world * dl)
        table[slots] = whole[slots]
    for g in compact_grads:
        g[slots] = 0.0


def batch_scatter(lists, pairs, d_full, col0, dl):
    live = _live_slots(lists)
    slots = torch.tensor([k for k, _ in live], dtype=torch.long)
    nodes = torch.tensor([n for _, n in live], dtype=torch.long)
    for cg, local in pairs:
        local.index_add_(0, nodes, cg[slots, col0:col0 + dl])
