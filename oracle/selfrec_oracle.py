"""CPU oracle for the SELFRec hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  Nothing under ``selfrec_amd/`` imports it; the product path fails
loudly when the HIP library is missing instead of falling back to anything here.

It restates, on the CPU (CPython ``random`` / numpy / scipy / torch-CPU fp32 -- the same
third-party arithmetic the reference itself calls: torch==1.13.1 pinned in reference
requirements.txt:6, run here on torch 2.10 CPU; scipy==1.14.1 pinned, 1.15.3 here), the
algorithm of every row of SURVEY.md section 8(a).  Each function cites the reference
file:line it follows.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so parity is pinned the other way round: ``tests/golden/
make_golden.py`` runs the *reference's own Python* from /root/reference (numba stubbed,
``.cuda()`` patched to identity) and commits its outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` holds this oracle to those vectors.

Integer ids: the reference maps user/item strings to ints in first-appearance order
(reference data/ui_graph.py:29-38).  The oracle works on those ints directly; ``edges_u``
/ ``edges_i`` are the id columns of ``training_data`` in file order.
"""
from __future__ import annotations

import heapq
import math
import random

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

MASK_SCORE = -10e8  # reference base/graph_recommender.py:50


# --------------------------------------------------------------------------------------
# a-1  util/sampler.next_batch_pairwise
# --------------------------------------------------------------------------------------
def first_appearance_ids(keys):
    """Map arbitrary hashables to 0.. in first-seen order (reference ui_graph.py:29-38)."""
    table = {}
    out = np.empty(len(keys), dtype=np.int64)
    for n, k in enumerate(keys):
        v = table.get(k)
        if v is None:
            v = len(table)
            table[k] = v
        out[n] = v
    return out, table


class PairwiseSampler:
    """Restates reference util/sampler.py:5-28 on integer ids.

    State carried across epochs is the *order* of the training list: the reference
    shuffles ``data.training_data`` in place (sampler.py:7) and that list is shared with
    the caller, so epoch k+1 shuffles epoch k's order.  ``self.order[p]`` is the index
    (into the original file order) of the triple now sitting at position p.

    RNG: the global CPython ``random`` stream, consumed exactly as the reference does:
    one ``shuffle`` per epoch, then per pair ``n_negs`` x ``choice(item_list)`` redrawn
    while the drawn item is in the user's training set (sampler.py:23-27).  ``item_list``
    is ``list(data.item.keys())`` -- dict order == id order -- so ``choice`` returns the
    item whose id is the drawn index.
    """

    def __init__(self, edges_u, edges_i, n_users, n_items):
        self.edges_u = np.asarray(edges_u, dtype=np.int64)
        self.edges_i = np.asarray(edges_i, dtype=np.int64)
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.order = list(range(len(self.edges_u)))
        self.user_items = [set() for _ in range(self.n_users)]
        for u, i in zip(self.edges_u.tolist(), self.edges_i.tolist()):
            self.user_items[u].add(i)
        self._item_list = list(range(self.n_items))

    def epoch(self, batch_size, n_negs=1):
        random.shuffle(self.order)                       # sampler.py:7
        ptr, size = 0, len(self.order)
        while ptr < size:                                # sampler.py:10-14
            end = ptr + batch_size if ptr + batch_size < size else size
            sel = self.order[ptr:end]
            ptr = end
            u_idx, i_idx, j_idx = [], [], []
            for e in sel:                                # sampler.py:20-27
                u = int(self.edges_u[e])
                i_idx.append(int(self.edges_i[e]))
                u_idx.append(u)
                rated = self.user_items[u]
                for _ in range(n_negs):
                    neg = random.choice(self._item_list)
                    while neg in rated:
                        neg = random.choice(self._item_list)
                    j_idx.append(neg)
            yield u_idx, i_idx, j_idx


# --------------------------------------------------------------------------------------
# a-2 / a-3  data/ui_graph + data/graph: bipartite adjacency and its normalisation
# --------------------------------------------------------------------------------------
def interaction_matrix(edges_u, edges_i, n_users, n_items):
    """R (U x I) CSR fp32; duplicate pairs sum (reference ui_graph.py:67-71)."""
    ones = np.ones(len(edges_u), dtype=np.float32)
    return sp.csr_matrix((ones, (edges_u, edges_i)), shape=(n_users, n_items), dtype=np.float32)


def bipartite_adjacency(edges_u, edges_i, n_users, n_items):
    """A = [[0,R],[R^T,0]] (reference ui_graph.py:47-56)."""
    n = n_users + n_items
    ones = np.ones(len(edges_u), dtype=np.float32)
    upper = sp.csr_matrix((ones, (edges_u, np.asarray(edges_i) + n_users)), shape=(n, n), dtype=np.float32)
    return upper + upper.T


def normalize_graph_mat(adj):
    """D^-1/2 A D^-1/2 for square input, D^-1 A otherwise; inf -> 0; fp32 throughout
    (reference data/graph.py:10-24)."""
    rowsum = np.array(adj.sum(1))
    if adj.shape[0] == adj.shape[1]:
        d_inv = np.power(rowsum, -0.5).flatten()
        d_inv[np.isinf(d_inv)] = 0.0
        dm = sp.diags(d_inv)
        return dm.dot(adj).dot(dm)
    d_inv = np.power(rowsum, -1).flatten()
    d_inv[np.isinf(d_inv)] = 0.0
    return sp.diags(d_inv).dot(adj)


def laplacian_of(r_mat):
    """Reference ui_graph.py:58-65 (used by SGL on dropped interaction matrices)."""
    u, i = r_mat.nonzero()
    nu, ni = r_mat.shape
    tmp = sp.csr_matrix((r_mat.data, (u, i + nu)), shape=(nu + ni, nu + ni), dtype=np.float32)
    return normalize_graph_mat(tmp + tmp.T)


def to_torch_sparse(mat):
    """Reference base/torch_interface.py:8-13 (COO, int64 indices, fp32 values)."""
    coo = mat.tocoo()
    idx = torch.from_numpy(np.vstack([coo.row, coo.col]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data.astype(np.float32)), coo.shape)


# --------------------------------------------------------------------------------------
# a-13  data/augmentor.GraphAugmentor
# --------------------------------------------------------------------------------------
def edge_dropout_keep_idx(n_edges, drop_rate):
    """Keep-set of reference data/augmentor.py:35: indices into ``sp_adj.nonzero()``
    order (CSR row-major), drawn from the global ``random`` stream."""
    return random.sample(range(n_edges), int(n_edges * (1 - drop_rate)))


def edge_dropout(r_mat, drop_rate):
    """Reference data/augmentor.py:29-40."""
    rows, cols = r_mat.nonzero()
    keep = edge_dropout_keep_idx(r_mat.count_nonzero(), drop_rate)
    ones = np.ones(len(keep), dtype=np.float32)
    return sp.csr_matrix((ones, (rows[keep], cols[keep])), shape=r_mat.shape)


def node_dropout(r_mat, drop_rate):
    """Reference data/augmentor.py:10-27.  (scipy's sparse product drops the zeroed entries: the result holds no
    explicit zeros -- pinned by tests/golden/shapes_meta.json N_node_dropout -- which is why the reference's
    convert_to_laplacian_mat, pairing ``adj.nonzero()`` with ``adj.data``, works on it.)"""
    nu, ni = r_mat.shape
    rows, cols = r_mat.nonzero()
    du = random.sample(range(nu), int(nu * drop_rate))
    di = random.sample(range(ni), int(ni * drop_rate))
    iu = np.ones(nu, dtype=np.float32)
    ii = np.ones(ni, dtype=np.float32)
    iu[du] = 0.0
    ii[di] = 0.0
    mat = sp.csr_matrix((np.ones_like(rows, dtype=np.float32), (rows, cols)), shape=(nu, ni))
    return sp.diags(iu).dot(mat).dot(sp.diags(ii))


# --------------------------------------------------------------------------------------
# a-4  LightGCN-family propagation
# --------------------------------------------------------------------------------------
def perturb_(x, noise, eps):
    """x += sign(x) * normalize(noise, dim=-1) * eps   (reference XSimGCL.py:90-91)."""
    x += torch.sign(x) * F.normalize(noise, dim=-1) * eps
    return x


def propagate(adj_t, ego, n_layers, *, include_ego, eps=0.0, noises=None, layer_cl=None):
    """Propagate ``ego`` (N,d) through ``n_layers`` of ``adj_t`` (torch sparse).

    include_ego=True  : LightGCN / SGL, mean over layers 0..L  (LightGCN.py:68-75, SGL.py:98-111)
    include_ego=False : SimGCL / XSimGCL, mean over layers 1..L (SimGCL.py:81-91, XSimGCL.py:83-96)
    noises            : list of L (N,d) U[0,1) tensors or None (no perturbation)
    layer_cl          : XSimGCL's l*; returns the layer-l* tensor as the CL view
                        (XSimGCL.py:85,93-94: l*=0 -> the raw ego embeddings)
    """
    outs = [ego] if include_ego else []
    cl = ego
    x = ego
    for k in range(n_layers):
        x = torch.sparse.mm(adj_t, x)
        if noises is not None:
            x = perturb_(x, noises[k], eps)
        outs.append(x)
        if layer_cl is not None and k == layer_cl - 1:
            cl = x
    final = torch.mean(torch.stack(outs, dim=1), dim=1)
    return (final, cl) if layer_cl is not None else final


# --------------------------------------------------------------------------------------
# a-6 / a-7 / a-8  util/loss_torch
# --------------------------------------------------------------------------------------
def bpr_loss(u, p, n):
    """mean(-log(1e-5 + sigmoid(<u,p> - <u,n>)))   (reference loss_torch.py:6-10)."""
    pos = (u * p).sum(dim=1)
    neg = (u * n).sum(dim=1)
    return torch.mean(-torch.log(10e-6 + torch.sigmoid(pos - neg)))


def l2_reg_loss(reg, *embs):
    """reg * sum_k ||emb_k||_F / rows_k   (reference loss_torch.py:18-22)."""
    total = 0
    for e in embs:
        total = total + torch.norm(e, p=2) / e.shape[0]
    return total * reg


def info_nce(v1, v2, temperature, b_cos=True):
    """-mean(diag(log_softmax(v1n @ v2n.T / tau, dim=1)))   (reference loss_torch.py:35-50)."""
    if b_cos:
        v1, v2 = F.normalize(v1, dim=1), F.normalize(v2, dim=1)
    s = (v1 @ v2.T) / temperature
    return -torch.diag(F.log_softmax(s, dim=1)).mean()


def unique_ids(idx_list):
    """``torch.unique(torch.Tensor(list).type(torch.long))`` -- sorted unique ids via a
    float32 round trip (reference XSimGCL.py:46-47)."""
    return torch.unique(torch.Tensor(idx_list).type(torch.long))


# --------------------------------------------------------------------------------------
# a-9  torch.optim.Adam (dense, default betas/eps, no weight decay)
# --------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """One in-place Adam update on numpy fp32 arrays; ``step`` is 1-based.
    Follows torch.optim.Adam's single-tensor formula (the optimiser the reference
    instantiates at XSimGCL.py:25): denom = sqrt(v)/sqrt(bc2) + eps; p -= lr/bc1 * m/denom."""
    m *= np.float32(b1)
    m += np.float32(1 - b1) * g
    v *= np.float32(b2)
    v += np.float32(1 - b2) * (g * g)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / np.float32(math.sqrt(bc2)) + np.float32(eps)
    p -= np.float32(lr / bc1) * (m / denom)


# --------------------------------------------------------------------------------------
# a-10 / a-11  GraphRecommender.test + util/algorithm.find_k_largest
# --------------------------------------------------------------------------------------
def find_k_largest(k, candidates):
    """Heap top-K with the reference's exact tie behaviour (util/algorithm.py:144-156)."""
    heap = [(float(s), i) for i, s in enumerate(candidates[:k])]
    heapq.heapify(heap)
    for off, s in enumerate(candidates[k:]):
        if s > heap[0][0]:
            heapq.heapreplace(heap, (float(s), off + k))
    heap.sort(key=lambda t: t[0], reverse=True)
    return [t[1] for t in heap], [t[0] for t in heap]


def full_rank_topk(user_emb, item_emb, user_ids, train_items_of, k):
    """Per test user: scores = item_emb @ user_emb[u]; training items -> -1e9; top-k
    (reference graph_recommender.py:46-53 + XSimGCL.py:57-60).  numpy fp32."""
    ids_out, scores_out = [], []
    for u in user_ids:
        cand = (item_emb @ user_emb[u]).astype(np.float32)
        rated = train_items_of(u)
        if len(rated):
            cand[np.asarray(rated, dtype=np.int64)] = MASK_SCORE
        ids, sc = find_k_largest(k, cand)
        ids_out.append(ids)
        scores_out.append(sc)
    return np.asarray(ids_out, dtype=np.int64), np.asarray(scores_out, dtype=np.float32)


def full_rank_topk_fast(user_emb, item_emb, user_ids, r_csr, k):
    """Vectorised variant for larger cases: same scores and mask, top-k by
    (score desc, id asc).  Equals ``full_rank_topk`` whenever the k+1 best scores of a
    user are distinct (ties are measure-zero for trained fp32 embeddings)."""
    user_ids = np.asarray(user_ids, dtype=np.int64)
    scores = (user_emb[user_ids] @ item_emb.T).astype(np.float32)
    sub = r_csr[user_ids]
    rows = np.repeat(np.arange(len(user_ids)), np.diff(sub.indptr))
    scores[rows, sub.indices] = MASK_SCORE
    part = np.argpartition(-scores, k - 1, axis=1)[:, :k]
    ps = np.take_along_axis(scores, part, axis=1)
    order = np.lexsort((part, -ps), axis=1)
    ids = np.take_along_axis(part, order, axis=1)
    return ids, np.take_along_axis(ps, order, axis=1)


# --------------------------------------------------------------------------------------
# a-12  util/evaluation.ranking_evaluation
# --------------------------------------------------------------------------------------
def ranking_evaluation(origin, res, topn):
    """origin: {user: {item: 1}}, res: {user: [(item, score), ...]} -> list of strings
    exactly as reference util/evaluation.py:135-162 formats them (round(...,5))."""
    out = []
    if len(origin) != len(res):
        raise ValueError("The Lengths of test set and predicted set do not match!")
    for n in topn:
        hits, n_test, dcg_sum = {}, 0, 0.0
        for user, truth in origin.items():
            pred = [it for it, _ in res[user][:n]]
            hits[user] = len(set(truth.keys()) & set(pred))
            n_test += len(truth)
            dcg = sum(1.0 / math.log(r + 2, 2) for r, it in enumerate(pred) if it in truth)
            idcg = sum(1.0 / math.log(r + 2, 2) for r in range(min(len(truth), n)))
            dcg_sum += dcg / idcg
        total_hits = sum(hits.values())
        hr = round(total_hits / n_test, 5)
        prec = round(total_hits / (len(hits) * n), 5)
        rec = [hits[u] / len(origin[u]) for u in hits]
        recall = round(sum(rec) / len(rec), 5)
        ndcg = round(dcg_sum / len(res), 5)
        out += [f"Top {n}\n", f"Hit Ratio:{hr}\n", f"Precision:{prec}\n", f"Recall:{recall}\n", f"NDCG:{ndcg}\n"]
    return out


# --------------------------------------------------------------------------------------
# Whole training step, as the model files compose it (the spec of the fused engine and
# bench.py's cpu_baseline "port")
# --------------------------------------------------------------------------------------
class OracleTrainer:
    """torch-CPU restatement of the train() loops of MF / LightGCN / XSimGCL / SimGCL / SGL
    (reference model/graph/MF.py:13-31, LightGCN.py:17-36, XSimGCL.py:23-50,
    SimGCL.py:21-50, SGL.py:24-47,115-125) on integer ids.

    ``noise_fn(shape) -> tensor`` replaces ``torch.rand_like`` so tests can inject noise.
    """

    def __init__(self, model, edges_u, edges_i, n_users, n_items, emb_size, *, n_layers=2,
                 lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1,
                 drop_rate=0.1, aug_type=1, batch_size=2048, user_emb=None, item_emb=None, noise_fn=None):
        self.model, self.n_users, self.n_items, self.d = model, n_users, n_items, emb_size
        self.aug_type = aug_type
        self.L, self.lr, self.reg, self.cl_rate, self.eps, self.tau = n_layers, lr, reg, cl_rate, eps, tau
        self.layer_cl, self.drop_rate, self.batch_size = layer_cl, drop_rate, batch_size
        self.r_mat = interaction_matrix(edges_u, edges_i, n_users, n_items)
        self.norm_adj = normalize_graph_mat(bipartite_adjacency(edges_u, edges_i, n_users, n_items))
        self.adj_t = to_torch_sparse(self.norm_adj)
        init = torch.nn.init.xavier_uniform_
        ue = init(torch.empty(n_users, emb_size)) if user_emb is None else torch.as_tensor(user_emb).clone()
        ie = init(torch.empty(n_items, emb_size)) if item_emb is None else torch.as_tensor(item_emb).clone()
        self.user_emb = torch.nn.Parameter(ue.float())
        self.item_emb = torch.nn.Parameter(ie.float())
        self.opt = torch.optim.Adam([self.user_emb, self.item_emb], lr=lr)
        self.noise_fn = noise_fn or (lambda shape: torch.rand(shape))
        self.dropped = None

    # -- encoders -------------------------------------------------------------------
    def _ego(self):
        return torch.cat([self.user_emb, self.item_emb], 0)

    def _split(self, x):
        return torch.split(x, [self.n_users, self.n_items])

    def encode(self, perturbed=False, adj=None):
        n = self.n_users + self.n_items
        adj = self.adj_t if adj is None else adj
        if self.model == "MF":
            return self.user_emb, self.item_emb
        if self.model in ("LightGCN", "SGL"):
            return self._split(propagate(adj, self._ego(), self.L, include_ego=True))
        noises = [self.noise_fn((n, self.d)) for _ in range(self.L)] if perturbed else None
        if self.model == "SimGCL":
            return self._split(propagate(adj, self._ego(), self.L, include_ego=False, eps=self.eps, noises=noises))
        final, cl = propagate(adj, self._ego(), self.L, include_ego=False, eps=self.eps,
                              noises=noises, layer_cl=self.layer_cl)
        return (*self._split(final), *self._split(cl))

    def resample_views(self):
        """SGL: two dropped, re-normalised graphs per epoch (SGL.py:28-29,89-96): node dropout for aug_type 0,
        edge dropout for 1 and 2."""
        drop = node_dropout if self.aug_type == 0 else edge_dropout
        self.dropped = [to_torch_sparse(laplacian_of(drop(self.r_mat, self.drop_rate))) for _ in range(2)]

    # -- one step -------------------------------------------------------------------
    def losses(self, u_idx, i_idx, j_idx):
        m = self.model
        if m == "XSimGCL":
            ru, ri, cu, ci = self.encode(perturbed=True)
        else:
            ru, ri = self.encode()
        ue, pe, ne = ru[u_idx], ri[i_idx], ri[j_idx]
        rec = bpr_loss(ue, pe, ne)
        cl = torch.zeros(())
        if m == "MF":
            regl = l2_reg_loss(self.reg, ue, pe, ne) / self.batch_size
        elif m == "LightGCN":
            regl = l2_reg_loss(self.reg, self.user_emb[u_idx], self.item_emb[i_idx], self.item_emb[j_idx]) / self.batch_size
        elif m == "SGL":
            regl = l2_reg_loss(self.reg, ue, pe, ne)
        else:
            regl = l2_reg_loss(self.reg, ue, pe)
        if m in ("XSimGCL", "SimGCL", "SGL"):
            uu, ui = unique_ids(u_idx), unique_ids(i_idx)
            if m == "XSimGCL":
                cl = info_nce(ru[uu], cu[uu], self.tau) + info_nce(ri[ui], ci[ui], self.tau)
            elif m == "SimGCL":
                a_u, a_i = self.encode(perturbed=True)
                b_u, b_i = self.encode(perturbed=True)
                cl = info_nce(a_u[uu], b_u[uu], 0.2) + info_nce(a_i[ui], b_i[ui], 0.2)   # SimGCL.py:48-49
            else:
                a_u, a_i = self.encode(adj=self.dropped[0])
                b_u, b_i = self.encode(adj=self.dropped[1])
                cl = info_nce(torch.cat((a_u[uu], a_i[ui]), 0), torch.cat((b_u[uu], b_i[ui]), 0), self.tau)
            cl = self.cl_rate * cl
        return rec, regl, cl

    def step(self, u_idx, i_idx, j_idx):
        rec, regl, cl = self.losses(u_idx, i_idx, j_idx)
        loss = rec + regl + cl
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return float(rec.detach()), float(regl.detach()), float(cl.detach())

    @torch.no_grad()
    def embeddings(self):
        out = self.encode()
        return out[0].detach().numpy().copy(), out[1].detach().numpy().copy()
