"""``GraphRecommender`` with the surface of reference base/graph_recommender.py:10-104.

``test()`` is where the full-catalogue ranking happens.  The reference scores one user at a
time (a device mat-vec, a D2H copy of I floats, a python mask loop and a heap top-K per
user, :46-53).  Here, when the model exposes ``user_emb`` / ``item_emb`` device tensors (every
LightGCN-family model does, e.g. XSimGCL.py:41), all test users go through
``srh_score_mask_topk`` in cache-sized chunks -- fp32 MFMA scores, training items masked to
-1e9, top-K on device -- and only (users x K) ids and scores come back.  Models with a
custom ``predict`` keep the per-user loop.  Either way the return value is the reference's:
``{user: [(item_name, score), ...]}`` best first, length ``max_N``.
"""
from os.path import abspath
from time import localtime, strftime, time

import numpy as np
import torch

from .. import ops
from ..data.loader import FileIO
from ..data.ui_graph import Interaction
from ..util.algorithm import find_k_largest
from ..util.evaluation import RankedLists, ranking_evaluation
from .recommender import Recommender

MASKED_SCORE = -10e8
# users per scoring call: chunk * n_items * 4 B stays well inside the 256 MiB Infinity Cache
SCORE_SLAB_BYTES = 96 << 20
# filtered ranking: items scored exactly to bound each user's K-th score, survivor slots per user, users per chunk
# (tools/eval_sweep.py on trained embeddings, 31.5 k users: chunks of 4096 / 8192 / 16384 / 32768 users rank at 15.8 / 17.9 /
#  19.0 / 19.0 M users/s -- a chunk's six launches fill the chip better and there are fewer of them; its workspace is
#  chunk x (sample + 2 cap) x 4 B = 400 MB at 16384.  A 2048- or 8192-item sample, or 512 slots, move it by < 3 %.)
FILTER_SAMPLE_ITEMS, FILTER_CAP, FILTER_CHUNK_ROWS = 3072, 1024, 16384
FILTER_MIN_ITEMS = 16384            # smaller catalogues: the exact slab pipeline
DEVICE_TOPK_MAX = 128          # srh_topk_rows / srh_score_mask_topk(_filtered): k <= 128
FILTER_WS_KEEP_BYTES = 1 << 30  # the filtered ranking's workspace is cached on the model up to this size


def _to_host(*tensors):
    """Device results -> numpy arrays: every copy into PINNED host memory (torch's caching host allocator: a fresh block per
    call, so an earlier result is never overwritten), all of them asynchronous, ONE synchronisation -- instead of a
    synchronous pageable copy per array (ids + scores of 31.5 k users: 5 MB; 0.25 of the ranking's 1.33 ms)."""
    outs = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
    for o, t in zip(outs, tensors):
        o.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return tuple(o.numpy() for o in outs)


class GraphRecommender(Recommender):
    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        from .. import dropin            # dropin.install(fuse=True): unmodified reference model files get the fused engine
        dropin.maybe_fuse(cls)

    def __init__(self, conf, training_set, test_set, **kwargs):
        super().__init__(conf, training_set, test_set, **kwargs)
        self.data = Interaction(conf, training_set, test_set)
        self.bestPerformance = []
        self.topN = [int(num) for num in self.ranking]
        self.max_N = max(self.topN)

    def print_model_info(self):
        super().print_model_info()
        tr, te = self.data.training_size(), self.data.test_size()
        print(f'Training Set Size: (user number: {tr[0]}, item number: {tr[1]}, interaction number: {tr[2]})')
        print(f'Test Set Size: (user number: {te[0]}, item number: {te[1]}, interaction number: {te[2]})')
        print('=' * 80)

    # ---- ranking ------------------------------------------------------------------------
    def _device_embeddings(self):
        ue, ie = getattr(self, 'user_emb', None), getattr(self, 'item_emb', None)
        ok = all(isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 2 for t in (ue, ie))
        if ok and ue.shape[0] == self.data.user_num and ie.shape[0] == self.data.item_num and ue.shape[1] == ie.shape[1]:
            # any embedding.size (base/recommender.py:16): zero columns up to the next width the scoring GEMM serves
            # add exact zeros to every score's fmaf chain -- same scores, same ranking
            w = ops.padded_width(int(ue.shape[1]), ops.ROW_WIDTHS)
            if w is not None:
                return ops.pad_cols(ue.detach().float(), w), ops.pad_cols(ie.detach().float(), w)
        return None

    def _test_csr(self, device):
        """The test set as a (users x items) CSR with sorted columns on the device (built once)."""
        cached = getattr(self, '_test_csr_cache', None)
        if cached is None or cached[0].device != torch.device(device):
            data = self.data
            rows = [sorted(data.item[i] for i in data.test_set.get(data.id2user[u], ())) for u in range(data.user_num)]
            indptr = np.zeros(data.user_num + 1, dtype=np.int32)
            np.cumsum([len(r) for r in rows], out=indptr[1:])
            indices = np.fromiter((i for r in rows for i in r), dtype=np.int32, count=int(indptr[-1]))
            cached = (torch.from_numpy(indptr).to(device), torch.from_numpy(indices).to(device), indptr)
            self._test_csr_cache = cached
        return cached

    def _rank_exact(self, ue, uid, ie, g, k):
        """Scores through a cache-sized slab, masked, exact top-K (srh_score_mask_topk): one call, the
        users pass through the slab `chunk` at a time inside the library."""
        chunk = max(32, min(int(uid.numel()), SCORE_SLAB_BYTES // (4 * ie.shape[0])))
        slab = getattr(self, '_score_slab', None)
        if slab is None or slab.shape != (chunk, ie.shape[0]) or slab.device != ie.device:
            slab = self._score_slab = torch.empty((chunk, ie.shape[0]), dtype=torch.float32, device=ie.device)
        return ops.score_mask_topk(ue, uid, ie, g.r_indptr, g.r_indices, k, scores_ws=slab)

    def _rank(self, ue, uid, ie, g, k):
        """Large catalogues: the score matrix is never stored (srh_score_mask_topk_filtered: a bound from a
        slice of the catalogue, then a filtering pass); the few rows whose survivor list overflowed
        (tie-heavy scores) are re-ranked by the exact path.  Same ids and scores either way."""
        if ie.shape[0] < FILTER_MIN_ITEMS:
            return self._rank_exact(ue, uid, ie, g, k)
        chunk = self._filter_chunk_rows(ie.device)
        ws = getattr(self, '_filter_ws', None)
        if ws is not None and getattr(self, '_filter_ws_chunk', chunk) != chunk:
            ws = self._filter_ws = None
        ids, sc, counts, ws = ops.score_mask_topk_filtered(
            ue, uid, ie, g.r_indptr, g.r_indices, k, sample_items=FILTER_SAMPLE_ITEMS, cap=FILTER_CAP,
            chunk_rows=chunk, ws=ws)
        # the workspace stays with the model between evaluations only while it is small next to the device's memory: a large
        # one (big catalogues, big chunks) would sit beside the training state for the whole run
        keep = ws is not None and ws.numel() * ws.element_size() <= FILTER_WS_KEEP_BYTES
        self._filter_ws, self._filter_ws_chunk = (ws if keep else None), chunk
        redo = torch.nonzero(counts > FILTER_CAP).flatten()
        if redo.numel():
            ids_r, sc_r = self._rank_exact(ue, uid[redo].contiguous(), ie, g, k)
            ids[redo], sc[redo] = ids_r, sc_r
        return ids, sc

    def _rank_marking_ties(self, ue, uid, ie, g, k):
        """ids, scores (device, k columns); a row whose best K + 1 scores hold two EQUAL neighbours carries ids[row, 0] =
        -1 - id (K + 1 > 128 or > the catalogue: no spare column to rank -- ``_mark_ties_without_spare`` finds them).

        The device kernels order (score desc, id asc); the reference's ``find_k_largest`` (util/algorithm.py:144-156)
        walks a size-K min-heap, and which of several equal scores it keeps, and in what order, is a property of that walk.
        Equal scores among a user's best are not exotic: 20 neighbouring pairs x 31,504 users of scores a few hundred
        thousand ulps apart tie once or twice per ranking at the Yelp2018 shape (and every row of a degenerate table does).
        So the device ranks K + 1 -- an equality anywhere else cannot touch the first K places -- marks those rows in the
        same launch that drops the spare column, and only they are redone the reference's way (``_heap_order_rows``).
        Cost when nothing ties: one more ranked column and one small launch; no synchronisation of its own."""
        if k + 1 > min(128, int(ie.shape[0])):
            ids, sc = self._rank(ue, uid, ie, g, k)
            return self._mark_ties_without_spare(ids, sc, ue, uid, ie, g, k), sc, True
        ids, sc = ops.topk_trim_mark_ties(*self._rank(ue, uid, ie, g, k + 1))
        return ids, sc, True

    def _mark_ties_without_spare(self, ids, sc, ue, uid, ie, g, k, chunk=None):
        """ids with the tied rows marked (ids[row, 0] = -1 - id) when the kernels have no spare column for the (K + 1)-th
        score (K = 128, or K = the catalogue): a row is tied when two neighbours among its K scores are equal, or when MORE
        masked scores of its catalogue row equal its K-th score than its K places hold -- the second test re-scores the
        rows chunk by chunk (srh_gemm_nt_f32, the ranking's own fma chain, training items at -10e8), i.e. it costs a second
        scoring pass, paid only by this one K."""
        tied = (sc[:, 1:] == sc[:, :-1]).any(dim=1) if k > 1 else torch.zeros(sc.shape[0], dtype=torch.bool, device=sc.device)
        n_items = int(ie.shape[0])
        if chunk is None:                      # the re-scored slab stays at 64 M scores (256 MB) whatever the catalogue
            chunk = max(16, min(2048, (64 << 20) // max(1, n_items)))
        if k < n_items:
            kth = sc[:, k - 1]
            held = (sc == kth[:, None]).sum(dim=1)
            uid_l = uid.long()
            lo, cnt = g.r_indptr[:-1].long()[uid_l], (g.r_indptr[1:] - g.r_indptr[:-1]).long()[uid_l]
            for at in range(0, int(uid.numel()), chunk):
                rows = slice(at, min(at + chunk, int(uid.numel())))
                scores = ops.gemm_nt(ue[uid_l[rows]].contiguous(), ie)
                c = cnt[rows]
                owner = torch.repeat_interleave(torch.arange(c.numel(), device=c.device), c)
                within = torch.arange(int(c.sum()), device=c.device) - torch.repeat_interleave(torch.cumsum(c, 0) - c, c)
                scores[owner, g.r_indices.long()[lo[rows][owner] + within]] = MASKED_SCORE
                tied[rows] |= (scores == kth[rows, None]).sum(dim=1) > held[rows]
        ids = ids.clone()
        ids[tied, 0] = -1 - ids[tied, 0]
        return ids

    def _heap_order_rows(self, rows, ue, uid_host, ie, g, k, chunk=256):
        """ids (int32), scores (float32) numpy (len(rows), k) of the given query rows in the reference's heap order: their
        score rows by the ranking's own fma chain (srh_gemm_nt_f32), the training items at -10e8
        (graph_recommender.py:49-50), python's heapq walk restated in C++ (srh_find_k_largest_host, pinned to the
        reference's outputs by tests/golden)."""
        out_ids = np.empty((len(rows), k), dtype=np.int32)
        out_sc = np.empty((len(rows), k), dtype=np.float32)
        users = np.asarray(uid_host)[rows].astype(np.int64)
        for lo in range(0, len(rows), chunk):
            u = users[lo:lo + chunk]
            scores, = _to_host(ops.gemm_nt(ue[torch.from_numpy(u).to(ue.device)].contiguous(), ie))      # (pinned, one sync)
            for j, uu in enumerate(u.tolist()):
                cand = scores[j]
                cand[g.h_r_indices[g.h_r_indptr[uu]:g.h_r_indptr[uu + 1]]] = -10e8
                out_ids[lo + j], out_sc[lo + j] = ops.find_k_largest_host(k, cand)
        return out_ids, out_sc

    @staticmethod
    def _filter_chunk_rows(device):
        """Users per chunk of the filtered ranking: FILTER_CHUNK_ROWS, halved while the chunk's workspace -- chunk x (sample +
        2 cap) x 4 B plus the split images -- would take more than 1/16 of the memory that is FREE on the device right
        now (never below 2048: the launches stop filling the chip)."""
        chunk = FILTER_CHUNK_ROWS
        try:
            free, _ = torch.cuda.mem_get_info(device)
        except Exception:                                    # noqa: BLE001  (no such query: keep the measured default)
            return chunk
        while chunk > 2048 and chunk * (FILTER_SAMPLE_ITEMS + 2 * FILTER_CAP) * 4 > free // 16:
            chunk //= 2
        return chunk

    def rank_on_device(self, user_ids, k=None, with_hits=False, metric_cuts=None):
        """ids, scores (numpy, shape (len(user_ids), k)) for integer user ids; with_hits adds the uint8
        flags 'this ranked item is in the user's test set' (srh_topk_hit_flags); metric_cuts (a list of cut-offs N <= k)
        adds the per-user hits and DCG / IDCG at every cut-off, computed on the device from the flags (srh_metric_rows)."""
        k = self.max_N if k is None else k
        ue, ie = self._device_embeddings()
        g = self.data.device_graph(ie.device)
        uid = self._device_user_ids(user_ids, ie.device)
        ids_dev, sc_dev, marked = self._rank_marking_ties(ue, uid, ie, g, k)
        if with_hits:
            if marked:                                           # (the flags below are taken from the ids: settle ties first)
                rows = torch.nonzero(ids_dev[:, 0] < 0).flatten()
                self._last_tie_rows = int(rows.numel())
                if rows.numel():
                    ids_fix, sc_fix = self._heap_order_rows(rows.cpu().numpy(), ue, user_ids, ie, g, k)
                    ids_dev[rows] = torch.from_numpy(ids_fix).to(ids_dev.device)
                    sc_dev[rows] = torch.from_numpy(sc_fix).to(sc_dev.device)
            t_indptr, t_indices, _ = self._test_csr(ie.device)
            flags = ops.topk_hit_flags(ids_dev, uid, t_indptr, t_indices)
            if metric_cuts:
                sizes = (t_indptr[1:] - t_indptr[:-1])[uid.long()].contiguous()
                hits, ndcg = ops.metric_rows(flags, sizes, metric_cuts)
                got = _to_host(ids_dev, sc_dev, flags, *[t for c in range(len(metric_cuts)) for t in (hits[c], ndcg[c])])
                return (got[0], got[1], got[2], {int(n): (got[3 + 2 * c], got[4 + 2 * c]) for c, n in enumerate(metric_cuts)})
            return _to_host(ids_dev, sc_dev, flags)
        ids, sc = _to_host(ids_dev, sc_dev)
        if marked:
            rows = np.flatnonzero(ids[:, 0] < 0)                 # (the marks ride in the ids: no extra copy, no extra sync)
            self._last_tie_rows = int(rows.size)
            if rows.size:
                ids[rows], sc[rows] = self._heap_order_rows(rows, ue, user_ids, ie, g, k)
        return ids, sc

    def _device_user_ids(self, user_ids, device):
        """The int32 ids on the device.  The upload is cached only for the array this class owns -- the test users'
        ids test() passes every epoch (``_test_users``: never handed out for writing); a caller's array is uploaded on every
        call: whether it was edited in place since the last one cannot be known without reading all of it."""
        owned = getattr(self, '_test_users_cache', None)
        if owned is not None and user_ids is owned[2]:
            cached = getattr(self, '_uid_dev_cache', None)
            if cached is not None and cached[0] is user_ids and cached[1].device == torch.device(device):
                return cached[1]
            uid = torch.as_tensor(user_ids, device=device)
            self._uid_dev_cache = (user_ids, uid)
            return uid
        return torch.as_tensor(np.asarray(user_ids, dtype=np.int32), device=device)

    def _test_users(self):
        """Test users in test-set order with their ids, test-set sizes and the item-name table (built once)."""
        cached = getattr(self, '_test_users_cache', None)
        if cached is None or cached[0] is not self.data.test_set:
            users = list(self.data.test_set)
            uid = np.fromiter((self.data.user[u] for u in users), dtype=np.int32, count=len(users))
            id2item = self.data.id2item
            names = np.array([id2item[i] for i in range(self.data.item_num)], dtype=object)
            cached = self._test_users_cache = (self.data.test_set, users, uid, names, dict.fromkeys(users), names.tolist())
        return cached[1:]

    def test(self):
        users, uid, names, keys, names_list = self._test_users()
        # (the device top-K kernels keep K <= 128 candidates per user in LDS: a longer list, or one longer than the
        # catalogue, takes the reference's per-user loop below -- slow, but every config the reference runs, runs)
        on_device = self.max_N <= min(DEVICE_TOPK_MAX, self.data.item_num)
        if on_device and self._device_embeddings() is not None and users:
            # the cut-offs ranking_evaluation will be asked for (item.ranking.topN): their per-user hits and DCG / IDCG come
            # back with the ranking (at most 8 of them, each <= K)
            cuts = sorted({int(n) for n in getattr(self, 'topN', []) if 1 <= int(n) <= self.max_N})[:8]
            got = self.rank_on_device(uid, with_hits=True, metric_cuts=cuts or None)
            ids, scores, flags = got[:3]
            sizes = np.diff(self._test_csr(self.item_emb.device)[2])[uid]
            # reads like the reference's {user: [(item, score), ...]}; rows are built on access and
            # ranking_evaluation works on the arrays (same strings)
            return RankedLists(users, names, ids, scores, hit_flags=flags, truth_sizes=sizes, origin=self.data.test_set,
                               per_user=got[3] if len(got) > 3 else None, keys=keys, names_list=names_list)
        rec_list = {}
        for user in users:                                   # models with a custom predict()
            candidates = self.predict(user)
            for item in self.data.user_rated(user)[0]:
                candidates[self.data.item[item]] = MASKED_SCORE
            ids, scores = find_k_largest(self.max_N, candidates)
            rec_list[user] = [(self.data.id2item[i], s) for i, s in zip(ids, scores)]
        return rec_list

    # ---- reporting (same files and strings as the reference) ----------------------------
    def evaluate(self, rec_list):
        self.recOutput.append('userId: recommendations in (itemId, ranking score) pairs, * means the item is hit.\n')
        for user, truth in self.data.test_set.items():
            cells = ''.join(f" ({name},{score}){'*' if name in truth else ''}" for name, score in rec_list[user])
            self.recOutput.append(user + ':' + cells + '\n')
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        name = self.config['model']['name']
        FileIO.write_file(self.output, f"{name}@{stamp}-top-{self.max_N}items.txt", self.recOutput)
        print('The result has been output to ', abspath(self.output), '.')
        self.result = ranking_evaluation(self.data.test_set, rec_list, self.topN)
        self.model_log.add('###Evaluation Results###')
        self.model_log.add(self.result)
        FileIO.write_file(self.output, f"{name}@{stamp}-performance.txt", self.result)
        print(f'The result of {self.model_name}:\n{"".join(self.result)}')

    def fast_evaluation(self, epoch):
        print('Evaluating the model...')
        rec_list = self.test()
        measure = ranking_evaluation(self.data.test_set, rec_list, [self.max_N])
        performance = {}
        for line in measure[1:]:
            key, value = line.strip().split(':')
            performance[key] = float(value)
        improved = not self.bestPerformance
        if self.bestPerformance:
            # majority vote over the metrics, as reference graph_recommender.py:88-92
            votes = sum(1 if self.bestPerformance[1][k] > performance[k] else -1 for k in performance)
            improved = votes < 0
        if improved:
            self.bestPerformance = [epoch + 1, performance]
            self.save()
        print('-' * 80)
        print(f'Real-Time Ranking Performance (Top-{self.max_N} Item Recommendation)')
        print(f'*Current Performance*\nEpoch: {epoch + 1}, ' + ', '.join(f'{k}: {v}' for k, v in performance.items()))
        best = ', '.join(f'{k}: {v}' for k, v in self.bestPerformance[1].items())
        print(f'*Best Performance*\nEpoch: {self.bestPerformance[0]}, {best}')
        print('-' * 80)
        return measure
