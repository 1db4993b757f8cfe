"""BUIR-NB (Lee et al., SIGIR'21; reference model/graph/BUIR.py:13-158), op-level tier: an online and a momentum
target LightGCN encoder, a linear predictor, bootstrapped cosine losses -- no negatives.  The reference's sparse
dropout builds a new torch COO tensor from ``_indices() / _values()`` every forward pass (BUIR.py:118-127); here the
dropped adjacency is a VALUE ARRAY over the resident structure (``SparseAdjHandle.dropout``), so its products stay on
the HIP SpMM.  Config block ``BUIR: {n_layer, tau, drop_rate}``."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...util import torch_rng
from ._oplevel import OpLevelRecommender, PropagationEncoder


class LGCN_Encoder(PropagationEncoder):
    def __init__(self, data, emb_size, n_layers, drop_rate, drop_flag=False):
        super().__init__(data, emb_size, n_layers)
        self.drop_ratio, self.drop_flag = float(drop_rate), bool(drop_flag)

    def forward(self, inputs):
        adj = None
        if self.drop_flag:
            # keep an entry where floor(1 - rate + U[0,1)) == 1, rescale by 1 / (1 - rate); the uniforms are drawn on
            # the host in the entries' (row, column) order, as the reference draws them (BUIR.py:118-121,130-131)
            # (torch_rng: those draws of torch's CPU generator replayed by the library's host code, 10x ATen's kernel)
            rate = np.random.random() * self.drop_ratio
            keep = torch_rng.keep_mask(self.sparse_norm_adj._nnz(), 1 - rate)
            adj = self.sparse_norm_adj.dropout(keep, 1.0 / (1 - rate))
        users, items = super().forward(adj)
        return users[inputs['user']], items[inputs['item']]

    @torch.no_grad()
    def get_embedding(self):
        return super().forward()


class BUIR_NB(nn.Module):
    def __init__(self, data, emb_size, momentum, n_layers, drop_rate, drop_flag=False):
        super().__init__()
        self.emb_size, self.momentum = emb_size, momentum
        self.online_encoder = LGCN_Encoder(data, emb_size, n_layers, drop_rate, drop_flag)
        self.target_encoder = LGCN_Encoder(data, emb_size, n_layers, drop_rate, drop_flag)
        self.predictor = nn.Linear(emb_size, emb_size)
        for online, target in zip(self.online_encoder.parameters(), self.target_encoder.parameters()):
            target.data.copy_(online.data)
            target.requires_grad = False

    def update_target(self, u_idx, i_idx):
        """momentum update of the batch's rows only (BUIR.py:72-75; duplicates in the lists resolve as torch's
        index assignment resolves them)"""
        m = self.momentum
        for key, idx in (("user_emb", u_idx), ("item_emb", i_idx)):
            tgt, src = self.target_encoder.embedding_dict[key].data, self.online_encoder.embedding_dict[key].data
            tgt[idx] = tgt[idx] * m + src[idx] * (1 - m)

    def forward(self, inputs):
        u_on, i_on = self.online_encoder(inputs)
        u_tg, i_tg = self.target_encoder(inputs)
        return self.predictor(u_on), u_tg, self.predictor(i_on), i_tg

    @torch.no_grad()
    def get_embedding(self):
        u, i = self.online_encoder.get_embedding()
        return self.predictor(u), u, self.predictor(i), i

    def get_loss(self, output):
        p_u, t_u, p_i, t_i = (F.normalize(t, dim=-1) for t in output)
        return ((2 - 2 * (p_u * t_i).sum(-1)) + (2 - 2 * (p_i * t_u).sum(-1))).mean()


class _TwoTowerScores(OpLevelRecommender):
    """score(u, i) = p(u).i + u.p(i) (BUIR.py:46-51): ranked on the device as ONE inner product of the concatenated
    (2d) vectors [p(u) ; u] . [i ; p(i)]."""
    verbose_every = 1

    def _set_towers(self, p_u, u, p_i, i):
        """The four tables predict() scores with AND their concatenation, which test() ranks with on the device -- always
        set together, so the final 'Testing...' pass after train() ranks the tables restore_best() put back (BUIR.py:
        36-37 reports the best epoch), not the last epoch's concatenation."""
        self.p_u_online, self.u_online, self.p_i_online, self.i_online = p_u, u, p_i, i
        if 2 * self.emb_size in (64, 128, 256):                  # widths the scoring GEMM serves
            self.user_emb = torch.cat([p_u, u], 1)
            self.item_emb = torch.cat([i, p_i], 1)
        else:                                                    # test() then falls back to predict()
            self.user_emb = self.item_emb = None

    def snapshot(self):
        self._set_towers(*self.model.get_embedding())

    def save(self):
        self.best_p_u, self.best_u, self.best_p_i, self.best_i = self.model.get_embedding()

    def restore_best(self):
        if hasattr(self, 'best_p_u'):
            self._set_towers(self.best_p_u, self.best_u, self.best_p_i, self.best_i)

    def predict(self, u):
        uid = self.data.get_user_id(u)
        with torch.no_grad():
            score = torch.matmul(self.p_u_online[uid], self.i_online.t()) + torch.matmul(self.u_online[uid], self.p_i_online.t())
        return score.cpu().numpy()


class BUIR(_TwoTowerScores):
    def __init__(self, conf, training_set, test_set):
        super().__init__(conf, training_set, test_set)
        block = self.config['BUIR']
        self.momentum, self.n_layers, self.drop_rate = float(block['tau']), int(block['n_layer']), float(block['drop_rate'])
        self.model = BUIR_NB(self.data, self.emb_size, self.momentum, self.n_layers, self.drop_rate, True)

    def batch_loss(self, user_idx, pos_idx, neg_idx):
        return self.model.get_loss(self.model({'user': user_idx, 'item': pos_idx}))

    def after_step(self, user_idx, pos_idx, neg_idx):
        self.model.update_target(user_idx, pos_idx)
