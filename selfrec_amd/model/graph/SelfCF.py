"""SelfCF-he (Zhou et al.; reference model/graph/SelfCF.py:13-91), op-level tier: one LightGCN encoder, a linear
predictor, and target vectors taken from the previous visit of each user / item (history embeddings) mixed with
the current ones -- no negatives, no second encoder.  Config block ``SelfCF: {n_layer, tau}``."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ._oplevel import PropagationEncoder
from .BUIR import _TwoTowerScores


class SelfCF_HE(nn.Module):
    def __init__(self, data, emb_size, momentum, n_layers):
        super().__init__()
        self.user_count, self.item_count, self.latent_size, self.momentum = data.user_num, data.item_num, emb_size, momentum
        self.online_encoder = PropagationEncoder(data, emb_size, n_layers)
        self.predictor = nn.Linear(emb_size, emb_size)
        # history tables start as host-drawn Gaussians (SelfCF.py:62-63)
        self.u_target_his = torch.randn((self.user_count, emb_size), requires_grad=False).cuda()
        self.i_target_his = torch.randn((self.item_count, emb_size), requires_grad=False).cuda()

    def forward(self, inputs):
        u_all, i_all = self.online_encoder()
        users, items = inputs['user'], inputs['item']
        u_now, i_now = u_all[users], i_all[items]
        with torch.no_grad():
            m = self.momentum
            u_target = self.u_target_his[users] * m + u_now.data * (1. - m)
            i_target = self.i_target_his[items] * m + i_now.data * (1. - m)
            self.u_target_his[users, :] = u_now.data.clone()
            self.i_target_his[items, :] = i_now.data.clone()
        return self.predictor(u_now), u_target, self.predictor(i_now), i_target

    @torch.no_grad()
    def get_embedding(self):
        u, i = self.online_encoder()
        return self.predictor(u), u, self.predictor(i), i

    @staticmethod
    def loss_fn(p, z):
        return 1 - F.cosine_similarity(p, z.detach(), dim=-1).mean()

    def get_loss(self, output):
        p_u, t_u, p_i, t_i = output
        return self.loss_fn(p_u, t_i) / 2 + self.loss_fn(p_i, t_u) / 2


class SelfCF(_TwoTowerScores):
    def __init__(self, conf, training_set, test_set):
        super().__init__(conf, training_set, test_set)
        block = self.config['SelfCF']
        self.momentum, self.n_layers = float(block['tau']), int(block['n_layer'])
        self.model = SelfCF_HE(self.data, self.emb_size, self.momentum, self.n_layers)

    def batch_loss(self, user_idx, pos_idx, neg_idx):
        return self.model.get_loss(self.model({'user': user_idx, 'item': pos_idx}))
