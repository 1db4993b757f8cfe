"""Ranking metrics with the output contract of reference util/evaluation.py:135-162:
``ranking_evaluation(origin, res, N)`` -> ``['Top 10\\n', 'Hit Ratio:...\\n', 'Precision:...\\n',
'Recall:...\\n', 'NDCG:...\\n', 'Top 20\\n', ...]`` with every figure ``round(x, 5)``.
``origin``: {user: {item: 1}}; ``res``: {user: [(item, score), ...]} best first.
"""
import math


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
