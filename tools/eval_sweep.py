#!/usr/bin/env python3
"""Full-rank evaluation on TRAINED embeddings (bench.py's XSimGCL after N steps) against the filter's shape constants:
bound-sample size, chunk rows, candidate capacity.  Prints device time of `rank_on_device` (ids + scores to the host)
and the survivors per user for each setting; the ranked ids of every setting must equal the first one's.

    python tools/eval_sweep.py [train_steps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from selfrec_amd.base import graph_recommender as gr  # noqa: E402
from selfrec_amd.engine import FusedTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
args = bench.parse([])
data, raw = bench.build_data(args.shape, args.seed)
torch.manual_seed(args.seed)
tr = FusedTrainer(data, args.emb, model="XSimGCL", n_layers=3, lr=1e-3, reg=1e-4, cl_rate=0.2, eps=0.2, tau=0.2, layer_cl=1,
                  batch_size=2048, use_graph=True)
tr.seed_sampler(args.seed)
done = 0
while done < steps:
    tr.begin_epoch()
    n = min(tr.epoch_batches, steps - done)
    for _ in range(n):
        tr.step()
    done += n
torch.cuda.synchronize()
rec = gr.GraphRecommender.__new__(gr.GraphRecommender)
rec.data, rec.max_N, rec.topN = data, 20, [20]
rec.user_emb, rec.item_emb = (t.contiguous() for t in tr.embeddings())
uid = np.asarray([data.user[u] for u in data.test_set], dtype=np.int32)
ue_p, ie_p = rec._device_embeddings()
g = data.device_graph(ie_p.device)
uid_dev = torch.as_tensor(uid, device=ie_p.device)
first = None
print(f"{len(uid)} test users, embeddings after {steps} steps")
SETTINGS = ((4096, 1024, 4096), (2048, 1024, 4096), (1024, 1024, 4096), (8192, 1024, 4096), (4096, 512, 4096),
            (4096, 1024, 8192), (2048, 1024, 8192), (4096, 1024, 2048))
if os.environ.get("SWEEP"):            # e.g. SWEEP="4096,1024,8192 4096,512,16384"
    SETTINGS = tuple(tuple(int(v) for v in w.split(",")) for w in os.environ["SWEEP"].split())
for sample, cap, chunk in SETTINGS:
    gr.FILTER_SAMPLE_ITEMS, gr.FILTER_CAP, gr.FILTER_CHUNK_ROWS = sample, cap, chunk
    rec._filter_ws = None
    ids, sc = rec.rank_on_device(uid)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ids, sc = rec.rank_on_device(uid)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    _, _, counts, _ = bench.ops_filtered(ue_p, uid_dev, ie_p, g, 20, gr)
    ids = np.asarray(ids)
    if first is None:
        first = ids
    same = bool(np.array_equal(ids, first))
    print(f"sample {sample:5d} cap {cap:5d} chunk {chunk:5d}: {1e3 * min(ts):7.3f} ms best, {1e3 * float(np.median(ts)):7.3f} median "
          f"= {len(uid) / float(np.median(ts)) / 1e6:6.2f} M users/s; survivors mean {float(counts.float().mean()):7.1f} max {int(counts.max())} "
          f"over cap {int((counts > cap).sum())}; ids identical to the first setting: {same}", flush=True)
