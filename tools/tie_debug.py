import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from selfrec_amd.base import graph_recommender as gr
from selfrec_amd.engine import FusedTrainer
args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, 64, model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1, batch_size=2048, use_graph=True)
r = bench.Runner(tr, args.seed); r.run(int(os.environ.get("EVAL_TRAIN_STEPS", "1300"))); r.fence()
rec = gr.GraphRecommender.__new__(gr.GraphRecommender)
rec.data, rec.max_N, rec.topN = data, 20, [20]
rec.user_emb, rec.item_emb = (t.contiguous() for t in tr.embeddings())
users, uid, names = rec._test_users()
ue, ie = rec._device_embeddings(); g = data.device_graph(ie.device); uid_dev = rec._device_user_ids(uid, ie.device)
for k in (20, 21):
    ids, sc = rec._rank(ue, uid_dev, ie, g, k)
    tie = (sc[:, 1:] == sc[:, :-1]).any(dim=1)
    rows = torch.nonzero(tie).flatten()
    print("k", k, "tie rows", int(rows.numel()))
    for rr in rows[:5].tolist():
        print(rr, sc[rr].tolist(), ids[rr].tolist())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): rec._rank(ue, uid_dev, ie, g, k)
    torch.cuda.synchronize(); print("  _rank ms", (time.perf_counter() - t0) / 5 * 1e3)

import time as _t
def wall(fn, n=9):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = _t.perf_counter(); r = fn(); torch.cuda.synchronize(); out.append(_t.perf_counter() - t0)
    return sorted(out)[n // 2] * 1e3, r
print("rank_on_device ms", wall(lambda: rec.rank_on_device(uid))[0])
print("_rank k=20 ms", wall(lambda: rec._rank(ue, uid_dev, ie, g, 20))[0], " k=21", wall(lambda: rec._rank(ue, uid_dev, ie, g, 21))[0])
ms, (ids_d, sc_d, tie) = wall(lambda: rec._rank_marking_ties(ue, uid_dev, ie, g, 20)); print("_rank_with_tie_flags ms", ms)
ms, got = wall(lambda: gr._to_host(ids_d, sc_d)); print("_to_host 3 tensors ms", ms)
rows = np.flatnonzero(got[0][:, 0] < 0); print("tied rows", rows)
if rows.size:
    print("_heap_order_rows ms", wall(lambda: rec._heap_order_rows(rows, ue, uid, ie, g, 20))[0])
    u = np.asarray(uid)[rows].astype(np.int64)
    print("  index + gemm_nt + cpu ms", wall(lambda: __import__("selfrec_amd").ops.gemm_nt(ue[torch.from_numpy(u).to(ue.device)].contiguous(), ie).cpu().numpy())[0])
