"""MixGCF (Huang et al., KDD'21; reference model/graph/MixGCF.py:12-129), op-level tier.  Per pair the sampler
draws ``n_negs`` (64) candidates; for every hop the candidates are mixed with the positive (random alpha), the
hardest one by inner product with the user is kept, and the kept ones are averaged over hops.  Config block
``MixGCF: {n_layer, n_negs}``.  The sampler at 64 negatives per pair is this model's hot spot on the host: the C++
replay produces them at tens of millions of draws per second (tools / DESIGN.md)."""
import torch
import torch.nn as nn

from ...util.loss_torch import bpr_loss, l2_reg_loss
from ._oplevel import OpLevelRecommender, PropagationEncoder


class MixGCF_Encoder(PropagationEncoder):
    def __init__(self, data, emb_size, n_negs, n_layers):
        super().__init__(data, emb_size, n_layers)
        self.n_negs, self.emb_size = int(n_negs), int(emb_size)
        self.dropout = nn.Dropout(0.1)

    def hop_tables(self):
        """user-side mean over hops and the per-hop item tables, with message dropout after each product
        (MixGCF.py:70-82)"""
        hops = self.hops(between=self.dropout)
        n_u = self.data.user_num
        return torch.stack([h[:n_u] for h in hops], dim=1).mean(dim=1), [h[n_u:] for h in hops]

    def negative_mixup(self, user, pos_item, neg_item):
        users, item_hops = self.hop_tables()
        u = users[user]
        picked = []
        for table in item_hops:
            cand = table[neg_item].reshape(-1, self.n_negs, self.emb_size)
            alpha = torch.rand_like(cand)
            cand = alpha * table[pos_item].unsqueeze(1) + (1 - alpha) * cand            # positive mixing
            hardest = (u.unsqueeze(1) * cand).sum(-1).max(dim=1)[1].detach()           # hop-wise hard negative
            picked.append(cand[torch.arange(cand.size(0), device=cand.device), hardest])
        items = torch.stack(item_hops, dim=1).mean(dim=1)
        return u, items[pos_item], torch.stack(picked, dim=1).mean(dim=1)

    def get_embeddings(self):
        return super().forward()


class MixGCF(OpLevelRecommender):
    def __init__(self, conf, training_set, test_set):
        super().__init__(conf, training_set, test_set)
        block = self.config['MixGCF']
        self.n_layers, self.n_negs = int(block['n_layer']), int(block['n_negs'])
        self.model = MixGCF_Encoder(self.data, self.emb_size, self.n_negs, self.n_layers)

    def batch_loss(self, user_idx, pos_idx, neg_idx):
        u, p, n = self.model.negative_mixup(user_idx, pos_idx, neg_idx)
        return bpr_loss(u, p, n) + l2_reg_loss(self.reg, u, p, n) / self.batch_size

    def snapshot(self):
        self.user_emb, self.item_emb = self.model.get_embeddings()
