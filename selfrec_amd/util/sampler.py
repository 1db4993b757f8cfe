"""``next_batch_pairwise`` with the generator protocol of reference util/sampler.py:5-28,
backed by the C++ MT19937 replay (csrc/sampler.cpp).

Bit-exact contract: for the same ``random.getstate()`` on entry, this yields the same
``(u_idx, i_idx, j_idx)`` lists as the reference, leaves the global ``random`` state where
the reference would leave it (after every batch, so interleaved ``random`` calls by the
caller see the same stream), and shuffles ``data.training_data`` in place into the same
order.
"""
import numpy as np

from .. import ops
from . import fastpath

_PROBES = 64


def _record_ids(td):
    """identity of every record object of the list, in list order (one C-speed pass: ~40 ms per million records)"""
    seq = td._list() if hasattr(td, "_list") else td          # (data/loader.TripleFile keeps its rows in a plain list)
    return np.fromiter(map(id, seq), dtype=np.int64, count=len(seq))


def _unread(td):
    """data/loader.TripleFile that no one has indexed yet: its order exists only as a pending permutation"""
    return getattr(td, "unread", None) is not None and td.unread()


def _list_matches(data, smp, edge_u, edge_i, expected_ids=None):
    """Does ``data.training_data`` still have the order the C++ sampler believes it has?  Two checks:
    a FULL fingerprint -- the identities of the record objects in list order against the order this module left them in
    (any re-ordering, insertion, deletion or replacement of records by the caller, anywhere in the list) -- and the
    VALUES of 64 probed records (first, last, evenly spaced: fields edited in place without replacing the record).
    A mismatch rebuilds the sampler from the list, i.e. from what the reference's generator would read."""
    td, n = data.training_data, smp.n_edges
    if n == 0 or _unread(td):                                 # (a lazy list nobody has materialised cannot have been edited)
        return True
    if expected_ids is not None and not np.array_equal(_record_ids(td), expected_ids):
        return False
    order = smp.order()
    for k in np.unique(np.linspace(0, n - 1, num=min(_PROBES, n)).astype(np.int64)).tolist():
        rec, e = td[k], int(order[k])
        if data.user.get(rec[0]) != int(edge_u[e]) or data.item.get(rec[1]) != int(edge_i[e]):
            return False
    return True


def _sampler_for(data):
    """One C++ sampler per data object, rebuilt if the training list was replaced or re-ordered by the caller."""
    cached = getattr(data, '_srh_sampler', None)
    if cached is not None and cached[1] is data.training_data and cached[2] == len(data.training_data) and \
            _list_matches(data, cached[0], cached[3], cached[4], getattr(data, '_srh_record_ids', None)):
        return cached[0]
    edge_u, edge_i = data._edge_ids_in_list_order() if hasattr(data, '_edge_ids_in_list_order') else (
        [data.user[r[0]] for r in data.training_data], [data.item[r[1]] for r in data.training_data])
    edge_u, edge_i = np.asarray(edge_u, dtype=np.int32), np.asarray(edge_i, dtype=np.int32)
    smp = ops.Sampler(edge_u, edge_i, len(data.user), len(data.item))
    data._srh_sampler = (smp, data.training_data, len(data.training_data), edge_u, edge_i)
    data._srh_record_ids = None if _unread(data.training_data) else _record_ids(data.training_data)
    return smp


def next_batch_pairwise(data, batch_size, n_negs=1, as_arrays=False):
    """``as_arrays`` (not in the reference's signature): yield the three index streams as int64 numpy arrays instead of
    python lists -- same values, same order, same RNG state; the package's own train() loops take them so that a batch
    of 2048 x 64 MixGCF candidates is not boxed into 131 k python ints and unboxed again by every ``table[idx]``."""
    smp = _sampler_for(data)
    smp.set_state_from_python()
    before = smp.order()
    smp.shuffle()
    after = smp.order()
    # replay the in-place shuffle on the caller-visible list (sampler.py:7): position p of the new order holds the record
    # that sat where edge after[p] was -- one numpy inverse permutation and one C-speed pass over the list (a dict of
    # positions and a python loop cost 1.5 s per epoch at 1.24 M records: more than an epoch of SelfCF steps)
    td = data.training_data
    where = np.empty(len(before), dtype=np.int64)
    where[before] = np.arange(len(before), dtype=np.int64)
    take = where[after]
    if _unread(td):
        td.permute_unread(take)                            # nobody has read the list: the shuffle stays a permutation
        data._srh_record_ids = None
    else:
        seq = td._list() if hasattr(td, "_list") else td
        old_ids = data._srh_record_ids if data._srh_record_ids is not None else _record_ids(td)
        seq[:] = list(map(seq.__getitem__, take.tolist()))
        # the order this call leaves the list in (checked by the next one): the identities in the old order were just
        # verified / recorded by _sampler_for, so the new fingerprint is that array permuted -- no second pass over them
        data._srh_record_ids = old_ids[take]
    smp.push_state_to_python()
    ptr, size = 0, smp.n_edges
    while ptr < size:
        smp.set_state_from_python()
        u, i, j = smp.next_batch(ptr, batch_size, n_negs)
        smp.push_state_to_python()
        ptr += len(u)
        if as_arrays:
            yield u.astype(np.int64), i.astype(np.int64), j.astype(np.int64)
        else:
            lists = (u.tolist(), i.tolist(), j.tolist())
            if fastpath.active():             # (dropin.install(): the lists' device copies, uploaded once -- util/fastpath.py)
                fastpath.register_batch(lists, (u, i, j))
            yield lists
