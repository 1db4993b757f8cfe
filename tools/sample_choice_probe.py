"""How many items reach the exact re-score for different choices of the bound slice (trained XSimGCL tables, Yelp2018 shape):
the first S items of the catalogue (the shipped choice) against the S items of largest norm."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from selfrec_amd.engine import FusedTrainer
args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, 64, model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1, batch_size=2048, use_graph=True)
r = bench.Runner(tr, args.seed); r.run(int(os.environ.get("EVAL_TRAIN_STEPS", "1300"))); r.fence()
ue, ie = (t.contiguous() for t in tr.embeddings())
g = data.device_graph(ie.device)
users = torch.arange(0, data.user_num, 8, device=ie.device)[:2048]
S = ue[users] @ ie.T
indptr, indices = g.r_indptr.long(), g.r_indices.long()
for j, u in enumerate(users.tolist()):
    S[j, indices[indptr[u]:indptr[u + 1]]] = -1e9
K = 20
norms = ie.norm(dim=1)
order = torch.argsort(norms, descending=True)
print("item norms: median %.4f, 99th pct %.4f, max %.4f" % (norms.median(), norms.quantile(0.99), norms.max()))
true_kth = S.topk(K, dim=1).values[:, -1]
for sample in (2048, 4096, 8192):
    for name, cols in (("first", torch.arange(sample, device=ie.device)), ("largest-norm", order[:sample])):
        T = S[:, cols].topk(K, dim=1).values[:, -1]
        surv = (S >= T[:, None]).sum(dim=1).float()
        print(f"slice {sample:5d} {name:13s}: survivors mean {surv.mean():7.1f} median {surv.median():6.0f} max {int(surv.max()):5d}; "
              f"rows whose bound IS the exact K-th best: {(T == true_kth).float().mean() * 100:.1f} %")

# Cauchy-Schwarz cut-off: an item can reach the bound T_u only if |i_j| >= T_u / |u|.  With the catalogue in norm order a
# 32-row block of the filter could stop at the first tile whose largest norm is below the block's smallest T_u / |u|.
un = ue[users].norm(dim=1)
T = S[:, order[:3072]].topk(K, dim=1).values[:, -1]
q = torch.where(T > 0, T / un, torch.zeros_like(T))                 # required norm per user (0: no cut-off possible)
sorted_norms = norms[order]
needed = torch.searchsorted(-sorted_norms, -q)                       # items with norm >= q, per user
print("fraction of the catalogue a user's row has to look at (norm >= T/|u|): mean %.3f median %.3f" %
      (needed.float().mean() / ie.shape[0], needed.float().median() / ie.shape[0]))
for blk in (32, 256):
    qb = q[: (q.numel() // blk) * blk].view(-1, blk).min(dim=1).values
    nb = torch.searchsorted(-sorted_norms, -qb)
    print(f"  per {blk}-row block (its weakest user decides): mean {nb.float().mean() / ie.shape[0]:.3f} median {nb.float().median() / ie.shape[0]:.3f}")
