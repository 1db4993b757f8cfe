"""User-item interaction data with the attribute surface of reference data/ui_graph.py:8-122,
plus device-resident mirrors for the HIP kernels.

Same observable results as the reference's ``Interaction``:
  * ids in first-appearance order of the training list (ui_graph.py:29-38),
  * ``test_set`` restricted to users and items seen in training (:41-45),
  * ``ui_adj`` / ``norm_adj`` / ``interaction_mat`` as scipy CSR fp32 (:47-56, :67-71),
  * ``convert_to_laplacian_mat`` for dropped interaction matrices (:58-65).
Differences are internal: one pass builds flat int32 id arrays (``train_u`` / ``train_i``) that
feed the C++ sampler and the device CSR; the dict-of-dict views (``training_set_u`` ...) and
the scipy matrices are materialised on first use.
"""
import weakref
from collections import defaultdict

import numpy as np
import scipy.sparse as sp

from .data import Data
from .graph import Graph
from .loader import TripleFile


class _LazyTriples(list):
    """[[user, item, 1.0], ...] for id arrays, materialised on first element access; len() is free."""

    def __init__(self, u, i):
        super().__init__()
        self._u, self._i, self._done = u, i, False

    def _fill(self):
        if not self._done:
            self._done = True
            super().extend([str(a), str(b), 1.0] for a, b in zip(self._u.tolist(), self._i.tolist()))

    def __len__(self):
        return len(self._u) if not self._done else super().__len__()

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __setitem__(self, k, v):
        self._fill()
        super().__setitem__(k, v)


class Interaction(Data, Graph):
    def __init__(self, conf, training, test):
        Graph.__init__(self)
        Data.__init__(self, conf, training, test)
        self._cache = {}
        if isinstance(training, TripleFile) and training._rows is None and \
                (isinstance(test, TripleFile) and test._rows is None or (not isinstance(test, TripleFile) and len(test) == 0)):
            self._init_native(training.path, test.path if isinstance(test, TripleFile) else None)
            return
        self.user, self.item = {}, {}
        n = len(self.training_data)
        self.train_u = np.empty(n, dtype=np.int32)
        self.train_i = np.empty(n, dtype=np.int32)
        user, item = self.user, self.item
        for pos, rec in enumerate(self.training_data):
            uid = user.get(rec[0])
            if uid is None:
                uid = user[rec[0]] = len(user)
            iid = item.get(rec[1])
            if iid is None:
                iid = item[rec[1]] = len(item)
            self.train_u[pos] = uid
            self.train_i[pos] = iid
        self.id2user = {v: k for k, v in user.items()}
        self.id2item = {v: k for k, v in item.items()}
        self.user_num, self.item_num = len(user), len(item)
        self.test_set = defaultdict(dict)
        self.test_set_item = set()
        for rec in self.test_data:
            if rec[0] in user and rec[1] in item:
                self.test_set[rec[0]][rec[1]] = 1
                self.test_set_item.add(rec[1])

    @classmethod
    def from_id_arrays(cls, conf, train_u, train_i, test_u, test_i, n_users, n_items):
        """Build from dense integer ids (e.g. a generated graph) without creating python triples: the name
        of node k is str(k).  Every user and item must occur in training (ids are then their own
        first-appearance order only up to relabelling, which no kernel depends on)."""
        self = cls.__new__(cls)
        Graph.__init__(self)
        self.config, self._cache = conf, {}
        self.train_u = np.ascontiguousarray(train_u, dtype=np.int32)
        self.train_i = np.ascontiguousarray(train_i, dtype=np.int32)
        if np.unique(self.train_u).size != n_users or np.unique(self.train_i).size != n_items:
            raise ValueError("from_id_arrays: every user and item needs at least one training interaction")
        self.user_num, self.item_num = int(n_users), int(n_items)
        self.user = {str(k): k for k in range(self.user_num)}
        self.item = {str(k): k for k in range(self.item_num)}
        self.id2user = {k: str(k) for k in range(self.user_num)}
        self.id2item = {k: str(k) for k in range(self.item_num)}
        self.test_set = defaultdict(dict)
        for u, i in zip(np.asarray(test_u).tolist(), np.asarray(test_i).tolist()):
            self.test_set[str(u)][str(i)] = 1
        self.test_set_item = set(str(i) for i in np.unique(np.asarray(test_i)).tolist())
        self.training_data = _LazyTriples(self.train_u, self.train_i)
        self.test_data = _LazyTriples(np.asarray(test_u), np.asarray(test_i))
        return self

    def _init_native(self, train_path, test_path):
        """Same products as the python loops above, parsed and id-mapped by srh_dataset_load."""
        import ctypes as C
        from .. import _lib
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.srh_dataset_load(C.byref(h), train_path.encode(), test_path.encode() if test_path else None),
                   "srh_dataset_load")
        try:
            sizes = (C.c_int64 * 5)()
            _lib.check(lib.srh_dataset_sizes(h, sizes))
            n_users, n_items, n_train, n_test, self._n_test_lines = (int(x) for x in sizes)
            # len(training_data) / len(test_data) (training_size(), test_size(), sampler.py:8) without building the lists
            self.training_data._known_len = n_train
            if isinstance(self.test_data, TripleFile):
                self.test_data._known_len = self._n_test_lines
            self.train_u = np.empty(n_train, dtype=np.int32)
            self.train_i = np.empty(n_train, dtype=np.int32)
            test_u, test_i = np.empty(n_test, dtype=np.int32), np.empty(n_test, dtype=np.int32)
            vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            _lib.check(lib.srh_dataset_copy_ids(h, vp(self.train_u), vp(self.train_i), None, vp(test_u), vp(test_i)))
            names = []
            for which, n in ((0, n_users), (1, n_items)):
                # every name followed by '\n', split in one C-level call (a slice + decode per name was 1 s per million)
                buf = C.create_string_buffer(max(1, int(lib.srh_dataset_names_bytes(h, which)) + n))
                _lib.check(lib.srh_dataset_copy_names(h, which, buf, None))
                got = buf.raw[:-1].decode().split('\n') if n else []
                got = got[:n]
                assert len(got) == n, (len(got), n)
                names.append(got)
        finally:
            lib.srh_dataset_destroy(h)
        user_names, item_names = names
        self.user = dict(zip(user_names, range(n_users)))
        self.item = dict(zip(item_names, range(n_items)))
        self.id2user = dict(enumerate(user_names))
        self.id2item = dict(enumerate(item_names))
        self.user_num, self.item_num = n_users, n_items
        self.test_set = defaultdict(dict)
        for u, i in zip(test_u.tolist(), test_i.tolist()):
            self.test_set[user_names[u]][item_names[i]] = 1
        self.test_set_item = set(item_names[i] for i in np.unique(test_i).tolist())

    # ---- lazily built views ---------------------------------------------------------
    def _lazy(self, key, build):
        if key not in self._cache:
            self._cache[key] = build()
        return self._cache[key]

    def _edge_ids_in_list_order(self):
        """ids of the CURRENT training_data order (the sampler shuffles that list in place)."""
        td = self.training_data
        if isinstance(td, TripleFile) and td.unread():        # file order, or the pending permutation of it
            return (self.train_u, self.train_i) if td._order is None else (self.train_u[td._order], self.train_i[td._order])
        u = np.fromiter((self.user[r[0]] for r in self.training_data), dtype=np.int32, count=len(self.training_data))
        i = np.fromiter((self.item[r[1]] for r in self.training_data), dtype=np.int32, count=len(self.training_data))
        return u, i

    @property
    def training_set_u(self):
        def build():
            out = defaultdict(dict)
            id2user, id2item = self.id2user, self.id2item
            for u, i in zip(self.train_u.tolist(), self.train_i.tolist()):
                out[id2user[u]][id2item[i]] = 1
            return out
        return self._lazy('tsu', build)

    @property
    def training_set_i(self):
        def build():
            out = defaultdict(dict)
            id2user, id2item = self.id2user, self.id2item
            for u, i in zip(self.train_u.tolist(), self.train_i.tolist()):
                out[id2item[i]][id2user[u]] = 1
            return out
        return self._lazy('tsi', build)

    @property
    def interaction_mat(self):
        def build():
            ones = np.ones(self.train_u.size, dtype=np.float32)
            return sp.csr_matrix((ones, (self.train_u, self.train_i)), shape=(self.user_num, self.item_num),
                                 dtype=np.float32)
        return self._lazy('R', build)

    @property
    def ui_adj(self):
        def build():
            n = self.user_num + self.item_num
            ones = np.ones(self.train_u.size, dtype=np.float32)
            upper = sp.csr_matrix((ones, (self.train_u, self.train_i.astype(np.int64) + self.user_num)),
                                  shape=(n, n), dtype=np.float32)
            return upper + upper.T
        return self._lazy('A', build)

    @property
    def norm_adj(self):
        def build():
            m = self.normalize_graph_mat(self.ui_adj)
            # convert_sparse_mat_to_tensor(data.norm_adj) -- what every model file does (e.g. XSimGCL.py:73) -- can then
            # hand out the RESIDENT device adjacency (same values bit for bit, XCD-aware schedule) instead of
            # uploading a second copy with a plain schedule
            m._srh_owner = weakref.ref(self)
            return m
        return self._lazy('Ahat', build)

    def convert_to_laplacian_mat(self, adj_mat):
        lazy = getattr(adj_mat, 'to_device_laplacian', None)
        if lazy is not None:           # device-resident dropped view (data/augmentor.py fast path)
            return lazy(self)
        nu, ni = adj_mat.shape
        rows, cols = adj_mat.nonzero()
        lifted = sp.csr_matrix((adj_mat.data, (rows, cols + nu)), shape=(nu + ni, nu + ni), dtype=np.float32)
        return self.normalize_graph_mat(lifted + lifted.T)

    def device_graph(self, device=None, column_classes=True):
        """Device CSR mirrors (built once per flavour): see data/device_graph.py.  column_classes=False: the
        plain row order, for tables narrow enough to sit in every L2 (the column-sharded layout)."""
        from .device_graph import DeviceGraph
        key = 'dev' if column_classes else 'dev_plain'
        return self._lazy(key, lambda: DeviceGraph(self.interaction_mat, device=device, column_classes=column_classes))

    # ---- accessors of the reference surface -----------------------------------------
    def get_user_id(self, u):
        return self.user.get(u)

    def get_item_id(self, i):
        return self.item.get(i)

    def training_size(self):
        return len(self.user), len(self.item), len(self.training_data)

    def test_size(self):
        return len(self.test_set), len(self.test_set_item), len(self.test_data)

    def contain(self, u, i):
        return u in self.user and i in self.training_set_u[u]

    def contain_user(self, u):
        return u in self.user

    def contain_item(self, i):
        return i in self.item

    def user_rated(self, u):
        row = self.training_set_u[u]
        return list(row.keys()), list(row.values())

    def item_rated(self, i):
        col = self.training_set_i[i]
        return list(col.keys()), list(col.values())

    def row(self, u):
        vec = np.zeros(self.item_num, dtype=np.float32)
        names, ratings = self.user_rated(self.id2user[u])
        vec[[self.item[n] for n in names]] = ratings
        return vec

    def col(self, i):
        vec = np.zeros(self.user_num, dtype=np.float32)
        names, ratings = self.item_rated(self.id2item[i])
        vec[[self.user[n] for n in names]] = ratings
        return vec

    def matrix(self):
        return np.minimum(self.interaction_mat.toarray(), 1.0).astype(np.float32)
