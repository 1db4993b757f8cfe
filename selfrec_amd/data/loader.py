"""Dataset files -> in-memory triples, in the formats of reference data/loader.py:22-41.

Graph data: one interaction per line, ``user item weight`` separated by single spaces,
returned as ``[[user_str, item_str, float_weight], ...]`` in file order (the order matters:
ids are assigned by first appearance, reference data/ui_graph.py:29-38).
"""
import os
from collections.abc import MutableSequence


class TripleFile(MutableSequence):
    """Lazy stand-in for the list ``FileIO.load_data_set(path, 'graph')`` returns.

    ``Interaction`` recognises it and builds its id arrays with the native loader
    (``srh_dataset_load``) without ever creating the python triples; any other consumer that
    indexes, iterates or mutates it gets the ordinary list, materialised on first touch.

    While nobody has touched it, ``len()`` is answered from the native loader's line count and the sampler's in-place
    shuffles (reference util/sampler.py:7) are kept as ONE pending permutation of the file's rows, applied when the
    list is first materialised: a run that never reads ``data.training_data`` never builds 1.2 M python lists, and
    one that does finds them in the order the reference's ``shuffle`` would have left."""

    def __init__(self, path):
        self.path = path
        self._rows = None
        self._known_len = None        # set by Interaction._init_native (lines of the file)
        self._order = None            # pending permutation: position p holds file row _order[p]

    def _list(self):
        if self._rows is None:
            rows = _read_triples(self.path)
            if self._order is not None:
                rows = list(map(rows.__getitem__, self._order.tolist()))
                self._order = None
            self._rows = rows
        return self._rows

    def unread(self):
        return self._rows is None

    def permute_unread(self, take):
        """new[p] = old[take[p]] on a list nobody has materialised yet"""
        assert self._rows is None
        self._order = take.copy() if self._order is None else self._order[take]

    def __len__(self):
        if self._rows is None and self._known_len is not None:
            return self._known_len
        return len(self._list())

    def __getitem__(self, k):
        return self._list()[k]

    def __setitem__(self, k, v):
        self._list()[k] = v

    def __delitem__(self, k):
        del self._list()[k]

    def insert(self, k, v):
        self._list().insert(k, v)


def _read_triples(file):
    """reference data/loader.py:22-33: every line -> [user, item, float(weight)], in file order"""
    triples = []
    with open(file) as src:
        for raw in src:
            parts = raw.strip().split(' ')
            triples.append([parts[0], parts[1], float(parts[2])])
    return triples


# dropin.install(fast=True) sets this: ``load_data_set(path, 'graph')`` -- what the reference's own SELFRec.py:12-13 calls --
# then hands out the lazy TripleFile (native parallel parse, shuffles kept as a pending permutation) instead of 1.2 M python
# lists that nobody reads: 1.9 s of parsing at the Yelp2018 shape, and 0.14 s PER EPOCH of replaying shuffle() on them.
LAZY_GRAPH_FILES = [False]


class FileIO:
    @staticmethod
    def load_data_set(file, rec_type):
        if rec_type == 'graph':
            if LAZY_GRAPH_FILES[0] and os.path.isfile(file):
                return TripleFile(file)
            return _read_triples(file)
        if rec_type == 'sequential':
            sequences = {}
            with open(file) as src:
                for raw in src:
                    key, _, tail = raw.strip().partition(':')
                    sequences[key] = tail.split()
            return sequences
        raise ValueError(f"unknown recommender type {rec_type!r}")

    @staticmethod
    def open_data_set(file, rec_type):
        """Like load_data_set, but graph data is parsed natively and lazily (see TripleFile)."""
        if rec_type == 'graph':
            if not os.path.exists(file):
                raise FileNotFoundError(file)
            return TripleFile(file)
        return FileIO.load_data_set(file, rec_type)

    @staticmethod
    def write_file(dir, file, content, op='w'):
        os.makedirs(dir, exist_ok=True)
        with open(dir + file, op) as dst:
            dst.writelines(content)

    @staticmethod
    def delete_file(file_path):
        if os.path.exists(file_path):
            os.remove(file_path)
