"""Shared pieces of the OP-LEVEL graph models (DirectAU, MixGCF, BUIR, SelfCF).

Unlike MF / LightGCN / XSimGCL / SimGCL / SGL, whose whole step is the fused engine, these models keep their own
torch code for what is specific to them (alignment / uniformity, hop mixing, bootstrapped targets) and take from
this package what the reference takes from its util / base modules: the bit-exact C++ sampler behind
``next_batch_pairwise``, the resident device adjacency behind ``torch.sparse.mm`` (forward and backward on the HIP
SpMM), the fused loss kernels and device-side ranking.  Attribute names (``embedding_dict``, ``online_encoder``,
``predictor`` ...) and config keys are the reference's, so checkpoints and downstream code line up.
"""
import torch
import torch.nn as nn

from ...base.graph_recommender import GraphRecommender
from ...base.torch_interface import TorchGraphInterface


class PropagationEncoder(nn.Module):
    """E^(k+1) = A_hat E^(k) on the device adjacency; parameters created users first, then items (the reference's
    RNG order, e.g. LightGCN.py:56-63, so ``torch.manual_seed`` reproduces its initial tables)."""

    def __init__(self, data, emb_size, n_layers):
        super().__init__()
        self.data, self.n_layers = data, int(n_layers)
        make = lambda rows: nn.Parameter(nn.init.xavier_uniform_(torch.empty(rows, emb_size)))  # noqa: E731
        self.embedding_dict = nn.ParameterDict({"user_emb": make(data.user_num), "item_emb": make(data.item_num)})
        self.sparse_norm_adj = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj).cuda()

    def hops(self, adj=None, between=None):
        """[E0, A E0, A^2 E0, ...] as (N, d) tensors; ``between`` (optional) is applied to each propagated table."""
        adj = self.sparse_norm_adj if adj is None else adj
        table = torch.cat([self.embedding_dict["user_emb"], self.embedding_dict["item_emb"]], 0)
        out = [table]
        for _ in range(self.n_layers):
            table = torch.sparse.mm(adj, table)
            if between is not None:
                table = between(table)
            out.append(table)
        return out

    def forward(self, adj=None):
        mean = torch.stack(self.hops(adj), dim=1).mean(dim=1)
        return mean[:self.data.user_num], mean[self.data.user_num:]


class OpLevelRecommender(GraphRecommender):
    """train() skeleton shared by the op-level models: Adam over ``self.model``, batches from the sampler mirror,
    ``batch_loss`` supplied by the subclass, per-epoch snapshot + fast_evaluation."""
    n_negs = 1
    verbose_every = 100

    def batch_loss(self, user_idx, pos_idx, neg_idx):
        raise NotImplementedError

    def after_step(self, user_idx, pos_idx, neg_idx):
        pass

    def snapshot(self):
        """Set the tensors test() ranks with (user_emb / item_emb)."""
        raise NotImplementedError

    def train(self):
        from ...util.sampler import next_batch_pairwise
        model = self.model.cuda()
        optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=self.lRate)
        for epoch in range(self.maxEpoch):
            model.train()
            for n, batch in enumerate(next_batch_pairwise(self.data, self.batch_size, self.n_negs, as_arrays=True)):
                # one H2D copy per index stream and step; every table[idx] below then indexes with a device tensor
                user_idx, pos_idx, neg_idx = (torch.from_numpy(a).cuda() for a in batch)
                loss = self.batch_loss(user_idx, pos_idx, neg_idx)
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                self.after_step(user_idx, pos_idx, neg_idx)
                if n % self.verbose_every == 0 and n > 0:
                    print('training:', epoch + 1, 'batch', n, 'batch_loss:', loss.item())
            model.eval()
            with torch.no_grad():
                self.snapshot()
            self.fast_evaluation(epoch)
        self.restore_best()

    def restore_best(self):
        if hasattr(self, 'best_user_emb'):
            self.user_emb, self.item_emb = self.best_user_emb, self.best_item_emb

    def save(self):
        with torch.no_grad():
            self.snapshot()
        self.best_user_emb, self.best_item_emb = self.user_emb.clone(), self.item_emb.clone()

    def predict(self, u):
        uid = self.data.get_user_id(u)
        with torch.no_grad():
            return torch.matmul(self.user_emb[uid], self.item_emb.transpose(0, 1)).cpu().numpy()
