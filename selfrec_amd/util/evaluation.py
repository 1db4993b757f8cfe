"""Ranking metrics with the output contract of reference util/evaluation.py:135-162:
``ranking_evaluation(origin, res, N)`` -> ``['Top 10\\n', 'Hit Ratio:...\\n', 'Precision:...\\n',
'Recall:...\\n', 'NDCG:...\\n', 'Top 20\\n', ...]`` with every figure ``round(x, 5)``.
``origin``: {user: {item: 1}}; ``res``: {user: [(item, score), ...]} best first.
"""
import gc
import math

import numpy as np


class RankedLists(dict):
    """The ``{user: [(item, score), ...]}`` result of ``GraphRecommender.test()`` (reference
    base/graph_recommender.py:44-58) -- a ``dict`` -- held as the arrays the device ranking returned until somebody
    looks inside: (users x K) item ids and scores plus, when the ranking also produced them, the per-position hit flags
    and the users' test-set sizes.  ``ranking_evaluation`` takes the arrays; anything that treats the object as the
    dictionary it is -- indexing, iteration, ``items()``, ``json.dump``, ``pickle``, ``dict(x)``, ``==`` -- makes it
    build every row once (two bulk conversions, ``materialise``) into its own storage, after which it IS that plain dict.
    From then on ``ranking_evaluation`` reads the rows as well (they are plain lists their holder may have edited), as
    the reference's does; an in-place edit of the mapping drops the arrays outright.
    """

    def __init__(self, users, item_names, ids, scores, hit_flags=None, truth_sizes=None, origin=None, per_user=None,
                 keys=None, names_list=None):
        # The KEYS are in the storage from the start (`keys`: a {user: None} dict of these users to copy -- the caller's
        # cached one -- else built here): C code that sizes a dict without asking it (json's encoder short-cuts an empty
        # one to "{}") sees the right length, and everything that then reads entries goes through the methods below.
        super().__init__(keys if keys is not None else dict.fromkeys(users))
        self.users = users if isinstance(users, list) else list(users)
        self._filled = False                   # the values are still placeholders
        self._names_list = names_list          # item_names as a python list (the caller's cached one, else made on demand)
        self.item_names, self.ids, self.scores = item_names, ids, scores
        self.hit_flags, self.truth_sizes, self.origin = hit_flags, truth_sizes, origin
        # {N: (hits per user int32, DCG / IDCG per user float64)} when the ranking computed them (srh_metric_rows)
        self.per_user = per_user or {}

    # ---- lazily filled storage -----------------------------------------------------------------------------------------
    def materialise(self):
        """The reference's return value as a plain dict -- {user: [(item name, score), ...]} for every user, every tuple
        built (graph_recommender.py:52-53) -- in two bulk conversions instead of one row at a time."""
        if self._filled:
            return dict(dict.items(self))
        # 630 k tuples at the Yelp2018 shape, allocated in one pass of C (selfrec_amd/_reclist: csrc/reclist.c, built by the
        # same make as the HIP library) -- exactly the objects that are returned, no intermediate lists; the cyclic collector,
        # which would run a generation-0 pass every 700 allocations over objects that cannot form a cycle (str, float), is
        # paused for the construction (together: 115 -> 45 ms where the python form was timed)
        try:
            from .. import _reclist                          # (absent = built for another interpreter, or not built at all)
        except ImportError:
            _reclist = None
        if self._names_list is None:
            self._names_list = self.item_names.tolist() if hasattr(self.item_names, "tolist") else list(self.item_names)
        ids = np.ascontiguousarray(self.ids, dtype=np.int32)
        scores = np.ascontiguousarray(self.scores, dtype=np.float32)
        was_on = gc.isenabled()
        gc.disable()
        try:
            if _reclist is not None:
                return _reclist.build(self.users, self._names_list, ids, scores, int(ids.shape[1]))
            # the same rows in python (host-only formatting of results the device already ranked: 2-3 x slower, same objects)
            names = self._names_list
            return {u: [(names[i], s) for i, s in zip(r_ids, r_sc)]
                    for u, r_ids, r_sc in zip(self.users, ids.tolist(), scores.astype(np.float64).tolist())}
        finally:
            if was_on:
                gc.enable()

    def _fill(self):
        if not self._filled:
            rows = self.materialise()
            self._filled = True
            dict.update(self, rows)
        return self

    def _edited(self):
        """the rows no longer are what the device ranked: the arrays (and the fast report they feed) are dropped"""
        self._fill()
        self.hit_flags, self.per_user = None, {}

    @property
    def arrays_valid(self):
        """the arrays still ARE the result: nobody has been handed the rows (a row is a plain list its holder may edit)"""
        return self.hit_flags is not None and not self._filled

    # (len(), `in`: dict's own -- the keys are there)
    def __getitem__(self, user):
        return dict.__getitem__(self._fill(), user)

    def __iter__(self):
        return dict.__iter__(self._fill())

    def __reversed__(self):
        return dict.__reversed__(self._fill())

    def keys(self):
        return dict.keys(self._fill())

    def values(self):
        return dict.values(self._fill())

    def items(self):
        return dict.items(self._fill())

    def get(self, user, default=None):
        return dict.get(self._fill(), user, default)

    def copy(self):
        return dict(dict.items(self._fill()))

    def __eq__(self, other):
        return dict.__eq__(self._fill(), other)

    def __ne__(self, other):
        return dict.__ne__(self._fill(), other)

    __hash__ = None

    def __repr__(self):
        return dict.__repr__(self._fill())

    def __or__(self, other):
        return dict.__or__(self._fill(), other)

    def __ror__(self, other):
        return dict.__ror__(self._fill(), other)

    def __reduce__(self):                                    # pickle / copy: the plain dict the reference returns
        return (dict, (self.materialise(),))

    # writes
    def __setitem__(self, user, row):
        self._edited()
        dict.__setitem__(self, user, row)

    def __delitem__(self, user):
        self._edited()
        dict.__delitem__(self, user)

    def __ior__(self, other):
        self._edited()
        dict.update(self, other)
        return self

    def pop(self, *args):
        self._edited()
        return dict.pop(self, *args)

    def popitem(self):
        self._edited()
        return dict.popitem(self)

    def setdefault(self, user, default=None):
        self._edited()
        return dict.setdefault(self, user, default)

    def update(self, *args, **kwargs):
        self._edited()
        dict.update(self, *args, **kwargs)

    def clear(self):
        self._edited()
        dict.clear(self)


def _left_to_right_sum(x):
    """0 + x[0] + x[1] + ... in that order, each add rounded to float64 (what `sum(list)` and `+=` loops do)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    return float(np.add.accumulate(x)[-1]) if x.size else 0


def _fast_report(res, N):
    """ranking_evaluation on a RankedLists carrying hit flags.  Every figure is accumulated in the
    order and precision of the python loops below (left-to-right float adds over users in test-set
    order, positions best-first), so the strings are identical, not just close."""
    need_flags = any(int(n) not in res.per_user for n in N)
    flags = res.hit_flags.astype(np.float64) if need_flags else res.hit_flags      # (users, K) 0/1
    sizes = res.truth_sizes.astype(np.int64)
    n_users = flags.shape[0]
    relevant = int(sizes.sum())
    report = []
    for n in N:
        if int(n) in res.per_user:
            # per-user figures from the device (same float64 adds in the same order, gains and ideal sums as computed
            # below by the host); only the two cross-user sums are left
            hits_i, ndcg_u = res.per_user[int(n)]
            total_hits = int(hits_i.sum())
            report.append('Top ' + str(n) + '\n')
            report.append('Hit Ratio:' + str(round(total_hits / relevant, 5)) + '\n')
            report.append('Precision:' + str(round(total_hits / (n_users * n), 5)) + '\n')
            report.append('Recall:' + str(round(_left_to_right_sum(hits_i.astype(np.float64) / sizes) / n_users, 5)) + '\n')
            report.append('NDCG:' + str(round(_left_to_right_sum(ndcg_u) / n_users, 5)) + '\n')
            continue
        n_eff = min(n, flags.shape[1])
        hit_n = flags[:, :n_eff]
        hits = hit_n.sum(axis=1)                               # small integers: exact in any order
        total_hits = int(hits.sum())
        gains = np.array([1.0 / math.log(pos + 2, 2) for pos in range(n_eff)])
        ideal_prefix = [0.0]
        for pos in range(n):
            ideal_prefix.append(ideal_prefix[-1] + 1.0 / math.log(pos + 2, 2))
        ideal = np.asarray(ideal_prefix)[np.minimum(sizes, n)]
        dcg = np.zeros(n_users)
        for pos in range(n_eff):                              # same order of adds as the generator sum
            dcg = dcg + hit_n[:, pos] * gains[pos]
        report.append('Top ' + str(n) + '\n')
        report.append('Hit Ratio:' + str(round(total_hits / relevant, 5)) + '\n')
        report.append('Precision:' + str(round(total_hits / (n_users * n), 5)) + '\n')
        # `sum(...)` / `sum_NDCG += DCG / IDCG`, user by user: np.add.accumulate adds strictly left to right in float64 --
        # the additions of the python loop, bit for bit, without the loop (1.5 of this function's 2 ms at 31.5 k users)
        report.append('Recall:' + str(round(_left_to_right_sum(hits / sizes) / n_users, 5)) + '\n')
        report.append('NDCG:' + str(round(_left_to_right_sum(dcg / ideal) / n_users, 5)) + '\n')
    return report


class Metric:
    @staticmethod
    def hits(origin, res):
        return {user: len(set(origin[user]).intersection(item for item, _ in res[user])) for user in origin}

    @staticmethod
    def hit_ratio(origin, hits):
        relevant = sum(len(items) for items in origin.values())
        return round(sum(hits.values()) / relevant, 5)

    @staticmethod
    def precision(hits, N):
        return round(sum(hits.values()) / (len(hits) * N), 5)

    @staticmethod
    def recall(hits, origin):
        per_user = [hits[user] / len(origin[user]) for user in hits]
        return round(sum(per_user) / len(per_user), 5)

    @staticmethod
    def NDCG(origin, res, N):
        total = 0
        for user, ranked in res.items():
            truth = origin[user]
            gain = sum(1.0 / math.log(pos + 2, 2) for pos, (item, _) in enumerate(ranked) if item in truth)
            ideal = sum(1.0 / math.log(pos + 2, 2) for pos in range(min(len(truth), N)))
            total += gain / ideal
        return round(total / len(res), 5)


def ranking_evaluation(origin, res, N):
    if len(origin) != len(res):
        print('The Lengths of test set and predicted set do not match!')
        raise SystemExit(-1)
    if isinstance(res, RankedLists) and res.arrays_valid and res.origin is origin:
        return _fast_report(res, N)
    report = []
    for n in N:
        cut = {user: ranked[:n] for user, ranked in res.items()}
        hits = Metric.hits(origin, cut)
        report.append('Top ' + str(n) + '\n')
        report.append('Hit Ratio:' + str(Metric.hit_ratio(origin, hits)) + '\n')
        report.append('Precision:' + str(Metric.precision(hits, n)) + '\n')
        report.append('Recall:' + str(Metric.recall(hits, origin)) + '\n')
        report.append('NDCG:' + str(Metric.NDCG(origin, cut, n)) + '\n')
    return report
