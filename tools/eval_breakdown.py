#!/usr/bin/env python3
"""Where the 1.3 ms of the device ranking (GraphRecommender.rank_on_device, 31.5 k test users, TRAINED embeddings) go: the
kernels' device time (HIP events), the wall time of the same call, the copies to the host, the python around them."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd.base import graph_recommender as gr  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, 64, model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1,
                  batch_size=2048, use_graph=True)
r = bench.Runner(tr, args.seed)
r.run(int(os.environ.get("EVAL_TRAIN_STEPS", "2400")))
r.fence()
rec = gr.GraphRecommender.__new__(gr.GraphRecommender)
rec.data, rec.max_N, rec.topN = data, 20, [20]
rec.user_emb, rec.item_emb = (t.contiguous() for t in tr.embeddings())
users, uid, names, _keys, _names_list = rec._test_users()
for _ in range(3):
    rec.rank_on_device(uid)


def wall(fn, n=7):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = fn()
        torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
    return sorted(out)[n // 2] * 1e3, res


ue, ie = rec._device_embeddings()
g = data.device_graph(ie.device)
uid_dev = rec._device_user_ids(uid, ie.device)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dev = []
for _ in range(7):
    torch.cuda.synchronize(); a.record()
    ids_dev, sc_dev = rec._rank(ue, uid_dev, ie, g, 20)
    b.record(); torch.cuda.synchronize(); dev.append(a.elapsed_time(b))
ms_rank_wall, _ = wall(lambda: rec._rank(ue, uid_dev, ie, g, 20))
ms_host, _ = wall(lambda: gr._to_host(ids_dev, sc_dev))
ms_pageable, _ = wall(lambda: (ids_dev.cpu().numpy(), sc_dev.cpu().numpy()))
ms_all, _ = wall(lambda: rec.rank_on_device(uid))
ms_emb, _ = wall(lambda: rec._device_embeddings())
ms_uid_owned, _ = wall(lambda: rec._device_user_ids(uid, ie.device))
uid_copy = np.array(uid)
ms_uid_foreign, _ = wall(lambda: rec._device_user_ids(uid_copy, ie.device))
ms_all_foreign, _ = wall(lambda: rec.rank_on_device(uid_copy))
ms_chunk, _ = wall(lambda: rec._filter_chunk_rows(ie.device))
ms_ties_wall, marked = wall(lambda: rec._rank_marking_ties(ue, uid_dev, ie, g, 20))
ids_h, sc_h = gr._to_host(marked[0], marked[1])
tie_rows = np.flatnonzero(ids_h[:, 0] < 0)
ms_redo, _ = wall(lambda: rec._heap_order_rows(tie_rows, ue, uid, ie, g, 20)) if tie_rows.size else (0.0, None)
print(f"pieces: uid on the device (owned array, cached) {ms_uid_owned:.3f} ms, (a caller's array: upload) {ms_uid_foreign:.3f} | "
      f"_filter_chunk_rows (hipMemGetInfo) {ms_chunk:.3f} | _rank_marking_ties (K + 1 columns, wall) {ms_ties_wall:.3f} | "
      f"{tie_rows.size} tied rows redone in heap order {ms_redo:.3f} | rank_on_device on a caller's array {ms_all_foreign:.3f} ms")
from selfrec_amd import ops  # noqa: E402
_, _, counts, _ = ops.score_mask_topk_filtered(ue, uid_dev, ie, g.r_indptr, g.r_indices, 20, sample_items=gr.FILTER_SAMPLE_ITEMS,
                                               cap=gr.FILTER_CAP, chunk_rows=gr.FILTER_CHUNK_ROWS)
counts = counts.cpu().numpy()
print(f"survivors of the filter per user: mean {counts.mean():.1f} max {counts.max()} over cap {(counts > gr.FILTER_CAP).sum()}", end=" | ")
print(f"users {len(uid)}: _rank device time (events) {sorted(dev)[3]:.3f} ms | _rank wall {ms_rank_wall:.3f} | pinned D2H of ids + scores "
      f"{ms_host:.3f} (pageable .cpu(): {ms_pageable:.3f}) | _device_embeddings {ms_emb:.3f} | rank_on_device {ms_all:.3f} ms = "
      f"{len(uid) / ms_all / 1e3:.2f} M users/s")
