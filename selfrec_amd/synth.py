"""Seeded synthetic interaction graphs in the shapes BASELINE.json names.

The reference ships no yelp2018 / iFashion files (reference .MISSING_LARGE_BLOBS:1-7), so
every benchmark and parity case runs on a generated bipartite graph with the same
user / item / edge counts and power-law degrees on both sides.  Output formats are the
ones the reference's loader produces (reference data/loader.py:22-33): a list of
``[user_str, item_str, float_weight]`` triples, or the same as ``train.txt`` /
``test.txt`` lines ``"user item weight"``.

Guarantees: no duplicate (user, item) pairs, and every
user and every item owns at least one *training* edge, so ``Interaction`` sees exactly
``n_users`` x ``n_items``.
"""
from __future__ import annotations

import numpy as np

# name -> (n_users, n_items, n_edges_total)   [train+test]
SHAPES = {
    # LightGCN split of Yelp2018: 1,237,259 train + 324,147 test
    "yelp2018": (31668, 38048, 1561406),
    # iFashion (SGL paper): 300,000 x 81,614, 1,607,813 interactions
    "ifashion": (300000, 81614, 1607813),
    # shipped dataset/douban-book/test.txt stands in for the missing train file
    "douban-book": (10882, 19075, 119690),
    # BASELINE.json config 4; E chosen here (avg user degree 50), stated in DESIGN.md
    "1m-500k": (1000000, 500000, 50000000),
    "tiny": (300, 500, 6000),
    "small": (2000, 3000, 60000),
}


def _zipf_weights(n: int, exponent: float, rng: np.random.Generator) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), exponent)
    rng.shuffle(w)  # popularity must not correlate with id
    return w / w.sum()


def generate_edges(n_users: int, n_items: int, n_edges: int, seed: int = 2024,
                   user_exp: float = 0.6, item_exp: float = 0.8):
    """Return (users, items) int64 arrays of unique pairs; each node has >= 2 edges
    where possible so that an 80/20 split can leave one in train."""
    rng = np.random.default_rng(seed)
    if n_edges > n_users * n_items // 2:
        raise ValueError("graph too dense for rejection sampling")
    pu = _zipf_weights(n_users, user_exp, rng)
    pi = _zipf_weights(n_items, item_exp, rng)
    cu, ci = np.cumsum(pu), np.cumsum(pi)
    cu[-1] = ci[-1] = 1.0
    # seed edges: every user and item appears at least once
    base_u = np.concatenate([np.arange(n_users), np.searchsorted(cu, rng.random(n_items), side="right")])
    base_i = np.concatenate([np.searchsorted(ci, rng.random(n_users), side="right"), np.arange(n_items)])
    keys = np.unique(base_u.astype(np.int64) * n_items + base_i.astype(np.int64))
    while keys.size < n_edges:
        need = n_edges - keys.size
        m = int(need * 1.15) + 1024
        u = np.searchsorted(cu, rng.random(m), side="right").astype(np.int64)
        i = np.searchsorted(ci, rng.random(m), side="right").astype(np.int64)
        new = np.setdiff1d(np.unique(u * n_items + i), keys, assume_unique=True)
        if new.size > need:
            new = rng.choice(new, size=need, replace=False)
        keys = np.union1d(keys, new)
    if keys.size > n_edges:
        # only possible when the seed edges alone exceed n_edges
        keys = keys[:n_edges]
    users = keys // n_items
    items = keys % n_items
    order = rng.permutation(keys.size)  # file order is not sorted in real datasets
    return users[order], items[order]


def split_train_test(users: np.ndarray, items: np.ndarray, n_users: int, n_items: int,
                     test_frac: float = 0.2, seed: int = 2024):
    """Random per-edge split, then repair: the first edge of every user and of every
    item (in file order) is forced into train."""
    rng = np.random.default_rng(seed + 1)
    is_test = rng.random(users.size) < test_frac
    first_u = np.full(n_users, -1, dtype=np.int64)
    first_i = np.full(n_items, -1, dtype=np.int64)
    idx = np.arange(users.size - 1, -1, -1)
    first_u[users[idx]] = idx
    first_i[items[idx]] = idx
    is_test[first_u[first_u >= 0]] = False
    is_test[first_i[first_i >= 0]] = False
    tr = ~is_test
    return (users[tr], items[tr]), (users[is_test], items[is_test])


def make_dataset(shape: str = "tiny", seed: int = 2024, test_frac: float = 0.2,
                 n_edges: int | None = None):
    """Return ``(train_u, train_i, test_u, test_i, n_users, n_items)`` int64 id arrays."""
    n_users, n_items, e = SHAPES[shape]
    if n_edges is not None:
        e = n_edges
    u, i = generate_edges(n_users, n_items, e, seed)
    (tu, ti), (su, si) = split_train_test(u, i, n_users, n_items, test_frac, seed)
    return tu, ti, su, si, n_users, n_items


def as_triples(users: np.ndarray, items: np.ndarray, weight: float = 1.0):
    """In-memory format of reference data/loader.py:26-33: [[user_str, item_str, float]]."""
    us = users.astype(str).tolist()
    its = items.astype(str).tolist()
    return [[a, b, weight] for a, b in zip(us, its)]


def write_text(path: str, users: np.ndarray, items: np.ndarray, weight: int = 1) -> None:
    """``"user item weight"`` lines, the on-disk format of reference data/loader.py:26-32."""
    with open(path, "w") as f:
        f.writelines(f"{a} {b} {weight}\n" for a, b in zip(users.tolist(), items.tolist()))
