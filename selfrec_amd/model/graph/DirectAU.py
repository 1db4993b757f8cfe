"""DirectAU (Wang et al., KDD'22; reference model/graph/DirectAU.py:9-59), op-level tier: LightGCN propagation on
the HIP SpMM, alignment + uniformity on the batch rows in torch.  Config block ``DirectAU: {gamma, n_layers}``."""
import torch
import torch.nn.functional as F

from ...util.loss_torch import l2_reg_loss
from ._oplevel import OpLevelRecommender, PropagationEncoder


def alignment(x, y):
    """mean squared distance between the unit vectors of matched rows (DirectAU.py:37-39)"""
    return (F.normalize(x, dim=-1) - F.normalize(y, dim=-1)).norm(p=2, dim=1).pow(2).mean()


def uniformity(x, t=2):
    """log of the mean Gaussian potential over all pairs of unit vectors (DirectAU.py:41-43)"""
    return torch.pdist(F.normalize(x, dim=-1), p=2).pow(2).mul(-t).exp().mean().log()


class DirectAU(OpLevelRecommender):
    def __init__(self, conf, training_set, test_set):
        super().__init__(conf, training_set, test_set)
        block = self.config['DirectAU']
        self.gamma, self.n_layers = float(block['gamma']), int(block['n_layers'])
        self.model = PropagationEncoder(self.data, self.emb_size, self.n_layers)

    def calculate_loss(self, user_emb, item_emb):
        return alignment(user_emb, item_emb) + self.gamma * (uniformity(user_emb) + uniformity(item_emb)) / 2

    def batch_loss(self, user_idx, pos_idx, neg_idx):
        users, items = self.model()
        u, p = users[user_idx], items[pos_idx]
        return self.calculate_loss(u, p) + l2_reg_loss(self.reg, u, p) / self.batch_size

    def snapshot(self):
        self.user_emb, self.item_emb = self.model()
