"""The op-level drop-in tier: client code written the way the reference's model files are
(raw torch.sparse.mm on the handle, fancy-index gathers, util.loss_torch calls, torch.optim.Adam)
runs on the HIP kernels through the mirrored interface and matches the CPU oracle."""
import random

import numpy as np
import pytest
import torch

from oracle import selfrec_oracle as O
from selfrec_amd.base.torch_interface import SparseAdjHandle, TorchGraphInterface
from selfrec_amd.data.augmentor import GraphAugmentor
from selfrec_amd.util.loss_torch import InfoNCE, bpr_loss, l2_reg_loss
from selfrec_amd.util.sampler import next_batch_pairwise

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


class ClientEncoder(torch.nn.Module):
    """A LightGCN-style encoder as a user of the library would write it against SELFRec's API."""

    def __init__(self, data, emb, layers, init_u, init_i):
        super().__init__()
        self.data, self.layers = data, layers
        self.embedding_dict = torch.nn.ParameterDict({
            "user_emb": torch.nn.Parameter(torch.tensor(init_u)), "item_emb": torch.nn.Parameter(torch.tensor(init_i))})
        self.sparse_norm_adj = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj).cuda()

    def forward(self, adj=None):
        ego = torch.cat([self.embedding_dict["user_emb"], self.embedding_dict["item_emb"]], 0)
        outs = [ego]
        for _ in range(self.layers):
            ego = torch.sparse.mm(self.sparse_norm_adj if adj is None else adj, ego)
            outs.append(ego)
        mean = torch.mean(torch.stack(outs, dim=1), dim=1)
        return mean[:self.data.user_num], mean[self.data.user_num:]


def test_client_training_loop_matches_oracle(golden_models, golden_meta, fresh_tiny_data):
    gm, m = golden_models, golden_meta["LightGCN"]
    data = fresh_tiny_data
    enc = ClientEncoder(data, m["emb"], 3, gm["LightGCN_init_user"], gm["LightGCN_init_item"]).cuda()
    assert isinstance(enc.sparse_norm_adj, SparseAdjHandle)
    opt = torch.optim.Adam(enc.parameters(), lr=m["lr"])
    random.seed(m["sampler_seed"])
    losses = []
    for u_idx, i_idx, j_idx in next_batch_pairwise(data, m["batch"]):
        ue, ie = enc()
        u, p, n = ue[u_idx], ie[i_idx], ie[j_idx]
        loss = bpr_loss(u, p, n) + l2_reg_loss(m["reg"], enc.embedding_dict["user_emb"][u_idx],
                                               enc.embedding_dict["item_emb"][i_idx],
                                               enc.embedding_dict["item_emb"][j_idx]) / m["batch"]
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.item())
    assert rel_err(enc.embedding_dict["user_emb"].detach().cpu().numpy(), gm["LightGCN_param_user"]) < 1e-4
    assert rel_err(enc.embedding_dict["item_emb"].detach().cpu().numpy(), gm["LightGCN_param_item"]) < 1e-4
    with torch.no_grad():
        fu, fi = enc()
    assert rel_err(fu.cpu().numpy(), gm["LightGCN_final_user"]) < 1e-4


def test_handle_rectangular_backward_uses_transpose():
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    a = sp.random(90, 70, density=0.1, random_state=1, dtype=np.float32).tocsr()
    h = TorchGraphInterface.convert_sparse_mat_to_tensor(a).cuda()
    x = torch.tensor(rng.standard_normal((70, 64)).astype(np.float32), device="cuda", requires_grad=True)
    w = torch.tensor(rng.standard_normal((90, 64)).astype(np.float32), device="cuda")
    (torch.sparse.mm(h, x) * w).sum().backward()
    want = a.T.astype(np.float64) @ w.cpu().numpy().astype(np.float64)
    assert rel_err(x.grad.cpu().numpy(), want) < 2e-6
    with pytest.raises(Exception):
        torch.add(h, 1)


def test_sgl_style_views_stay_on_device(golden_ops, tiny_data):
    """GraphAugmentor.edge_dropout -> convert_to_laplacian_mat -> convert_sparse_mat_to_tensor, as
    SGL.py:89-96 chains them, with the reference's keep-set and global random stream."""
    data = tiny_data
    random.seed(99)
    dropped = GraphAugmentor.edge_dropout(data.interaction_mat, 0.1)
    assert np.array_equal(np.flatnonzero(dropped.keep_mask), np.sort(golden_ops["edge_dropout_keep"]))
    assert random.getrandbits(32) == int(golden_ops["edge_dropout_next_u32"][0])
    lap = data.convert_to_laplacian_mat(dropped)
    h = TorchGraphInterface.convert_sparse_mat_to_tensor(lap).cuda()
    assert isinstance(h, SparseAdjHandle)
    import scipy.sparse as sp
    want = sp.csr_matrix((golden_ops["edge_dropout_lap_data"], golden_ops["edge_dropout_lap_indices"],
                          golden_ops["edge_dropout_lap_indptr"]), shape=(500, 500))
    x = np.random.default_rng(3).standard_normal((500, 64)).astype(np.float32)
    got = torch.sparse.mm(h, torch.from_numpy(x).cuda()).cpu().numpy()
    assert rel_err(got, want.astype(np.float64) @ x.astype(np.float64)) < 2e-6
    # InfoNCE on concatenated user+item rows, as SGL.py:120-125
    v1 = torch.tensor(x[:300], device="cuda", requires_grad=True)
    v2 = torch.tensor(got[:300], device="cuda")
    loss = InfoNCE(v1, v2, 0.2)
    loss.backward()
    a = torch.tensor(x[:300], requires_grad=True)
    ref = O.info_nce(a, torch.tensor(got[:300]), 0.2); ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert rel_err(v1.grad.cpu().numpy(), a.grad.numpy()) < 2e-5


def test_handle_exposes_the_coo_parts_for_sparse_dropout_models(fresh_tiny_data):
    """BUIR.py:118-127 / MixGCF.py:84-94 read ``_nnz() / _indices() / _values()`` of the uploaded adjacency and
    build their own (dropped) torch sparse tensor: the handle serves the same entries as the reference's COO tensor."""
    data = fresh_tiny_data
    h = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj).cuda()
    coo = data.norm_adj.tocsr()
    coo.sort_indices()
    coo = coo.tocoo()
    i, v = h._indices(), h._values()
    assert h._nnz() == coo.nnz and i.shape == (2, coo.nnz) and i.dtype == torch.int64
    assert np.array_equal(i.cpu().numpy(), np.stack([coo.row, coo.col])) and np.array_equal(v.cpu().numpy(), coo.data)
    keep = torch.rand(h._nnz(), device=h.device) < 0.7          # the models' own dropout, then a torch product
    dropped = torch.sparse_coo_tensor(i[:, keep], v[keep], tuple(h.shape)) * (1.0 / 0.7)
    x = torch.randn((h.shape[0], 8), device=h.device)
    want = torch.sparse.mm(dropped, x)
    assert want.shape == (h.shape[0], 8) and torch.isfinite(want).all()
