#!/usr/bin/env python3
"""Where the milliseconds of GraphRecommender.test() + ranking_evaluation go (Yelp2018 shape, 31.5 k test users)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd.base.graph_recommender import GraphRecommender  # noqa: E402
from selfrec_amd.util.evaluation import ranking_evaluation  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(0)
rec = GraphRecommender.__new__(GraphRecommender)
rec.data, rec.max_N, rec.topN = data, 20, [20]
rec.user_emb = torch.randn((data.user_num, 64), device="cuda") * 0.1
rec.item_emb = torch.randn((data.item_num, 64), device="cuda") * 0.1
users, uid, names, _keys, _names_list = rec._test_users()
rec.test(); rec.test()


def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


ue, ie = rec._device_embeddings()
g = data.device_graph(ie.device)
uid_dev = torch.as_tensor(np.asarray(uid, dtype=np.int32), device=ie.device)
ms_upload, _ = t(lambda: torch.as_tensor(np.asarray(uid, dtype=np.int32), device=ie.device))
ms_rank, (ids_dev, sc_dev) = t(lambda: rec._rank(ue, uid_dev, ie, g, 20))
t_indptr, t_indices, _ = rec._test_csr(ie.device)
from selfrec_amd import ops  # noqa: E402
ms_flags, flags = t(lambda: ops.topk_hit_flags(ids_dev, uid_dev, t_indptr, t_indices))
ms_d2h, _ = t(lambda: (ids_dev.cpu().numpy(), sc_dev.cpu().numpy(), flags.cpu().numpy()))
ms_test, out = t(rec.test)
ms_eval, _ = t(lambda: ranking_evaluation(data.test_set, out, [20]))
print(f"users {len(uid)}: upload ids {ms_upload:.2f} ms | _rank (kernels + redo check) {ms_rank:.2f} | hit flags {ms_flags:.2f} | "
      f"D2H x3 {ms_d2h:.2f} | test() total {ms_test:.2f} | ranking_evaluation {ms_eval:.2f}")
