"""Torch-tensor front end of the C ABI: pointer extraction, shape checks, stream plumbing.

PyTorch is the memory container here and nothing else: every function hands raw device
addresses (``tensor.data_ptr()``) and the current HIP stream to libselfrec_hip.so.
"""
from __future__ import annotations

import ctypes as C
import random as _pyrandom

import numpy as np
import torch

from . import _lib
from ._lib import SelfrecHipError, SpmmEpilogue, check

require_gpu = _lib.require_gpu


def gpu_available() -> bool:
    return torch.cuda.is_available()

__all__ = ["Sampler", "DeviceCSR", "column_class_order", "spmm", "spmm_any", "pad_cols", "padded_width", "spmm3", "spmm_probe", "spmm_set_xcd_shares", "spmm_plan_run_tasks", "adj_sym_normalize", "bpr_l2_fwd_bwd", "bpr_fwd", "bpr_bwd",
           "sumsq", "set_infonce_precision", "get_infonce_precision", "infonce_fwd_bwd", "infonce_multi", "bpr_infonce", "infonce_ws", "adam_step", "score_mask_topk", "score_mask_topk_filtered", "gemm_nt", "topk_rows", "topk_hit_flags", "metric_rows",
           "axpby", "batch_fetch", "zero_rows", "cursor_advance", "batch_lists", "batch_pack", "batch_unpack", "batch_scatter",
           "SelfrecHipError"]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t, dtype=None, name="tensor"):
    """Device address of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise SelfrecHipError(f"{name}: expected a HIP device tensor (the product path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise SelfrecHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise SelfrecHipError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


def find_k_largest_host(k: int, candidates) -> tuple[np.ndarray, np.ndarray]:
    """ids (int64), scores (float32) of reference util/algorithm.py:144-156 ``find_k_largest`` -- the heap's own order among
    equal scores -- for a float32 vector on the host (srh_find_k_largest_host: python's heapq restated in C++)."""
    cand = np.ascontiguousarray(candidates, dtype=np.float32)
    m = min(int(k), int(cand.size))
    ids, sc = np.empty(m, dtype=np.int64), np.empty(m, dtype=np.float32)
    n_out = C.c_int64()
    check(_lib.load().srh_find_k_largest_host(int(k), cand.ctypes.data_as(C.c_void_p), int(cand.size),
                                              ids.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), C.byref(n_out)),
          "srh_find_k_largest_host")
    assert n_out.value == m
    return ids, sc


# ----------------------------------------------------------------------------------------
# (a-1) sampler
# ----------------------------------------------------------------------------------------
class Sampler:
    """Host-side bit-exact replay of util/sampler.py:5-28 (see csrc/sampler.cpp)."""

    def __init__(self, edge_user, edge_item, n_users: int, n_items: int):
        self._lib = _lib.load()
        eu, pu = _np(edge_user, np.int32)
        ei, pi = _np(edge_item, np.int32)
        if eu.shape != ei.shape or eu.ndim != 1:
            raise SelfrecHipError("Sampler: edge arrays must be 1-D and of equal length")
        self.n_users, self.n_items, self.n_edges = int(n_users), int(n_items), int(eu.size)
        h = C.c_void_p()
        check(self._lib.srh_sampler_create(C.byref(h), n_users, n_items, eu.size, pu, pi), "srh_sampler_create")
        self._h = h
        self._pushed = None            # the words this object last handed to python's generator (set_state_from_python)
        self._epoch_slots = {}         # epoch(slot=...): arrays the sampler keeps and overwrites

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.srh_sampler_destroy(h)
            self._h = None

    # -- RNG state ------------------------------------------------------------------
    def set_state_from_python(self, state=None):
        """Adopt ``random.getstate()`` (version 3 tuple: 624 words + position).  When python's generator still holds the
        very state this object pushed last (nobody drew from ``random`` in between -- the per-batch case of
        ``next_batch_pairwise``), the C++ generator is already there and nothing is converted: boxing 624 words into
        a numpy array and back cost ~150 us per batch, a tenth of an op-level LightGCN step."""
        state = _pyrandom.getstate() if state is None else state
        if state[0] != 3 or len(state[1]) != 625:
            raise SelfrecHipError("unsupported random.getstate() layout")
        self._gauss = state[2]
        if self._pushed is not None and state[1] == self._pushed:
            return
        words = np.array(state[1][:624], dtype=np.uint32)
        check(self._lib.srh_sampler_set_state(self._h, words.ctypes.data_as(C.c_void_p), int(state[1][624])))
        self._pushed = None

    def python_state(self):
        """The generator state as a tuple ``random.setstate`` accepts."""
        words = np.empty(624, dtype=np.uint32)
        pos = C.c_int32()
        check(self._lib.srh_sampler_get_state(self._h, words.ctypes.data_as(C.c_void_p), C.byref(pos)))
        return (3, tuple(words.tolist()) + (int(pos.value),), getattr(self, "_gauss", None))

    def push_state_to_python(self):
        state = self.python_state()
        _pyrandom.setstate(state)
        self._pushed = state[1]

    def seed(self, seed: int):
        check(self._lib.srh_sampler_seed(self._h, int(seed)), "srh_sampler_seed")
        self._gauss = None
        self._pushed = None

    # -- draws ----------------------------------------------------------------------
    def shuffle(self):
        self._pushed = None                      # (the C++ generator moves on: python's copy is behind until the next push)
        check(self._lib.srh_sampler_shuffle(self._h), "srh_sampler_shuffle")

    def order(self) -> np.ndarray:
        perm = np.empty(self.n_edges, dtype=np.int64)
        check(self._lib.srh_sampler_get_order(self._h, perm.ctypes.data_as(C.c_void_p)))
        return perm

    def next_batch(self, ptr: int, batch_size: int, n_negs: int = 1):
        cnt = min(batch_size, self.n_edges - ptr)
        u = np.empty(cnt, dtype=np.int32)
        i = np.empty(cnt, dtype=np.int32)
        j = np.empty(cnt * n_negs, dtype=np.int32)
        out = C.c_int64()
        self._pushed = None                      # (the C++ generator moves on: python's copy is behind until the next push)
        check(self._lib.srh_sampler_next_batch(self._h, ptr, batch_size, n_negs, u.ctypes.data_as(C.c_void_p),
                                               i.ctypes.data_as(C.c_void_p), j.ctypes.data_as(C.c_void_p),
                                               C.byref(out)), "srh_sampler_next_batch")
        assert out.value == cnt
        return u, i, j

    def epoch(self, batch_size: int, n_negs: int = 1, with_unique: bool = False, slot=None, with_segments: bool = False):
        """shuffle + all batches.  Returns dict of numpy arrays (see srh_sampler_epoch).
        with_segments (True | (user_row0, item_row0)): also the row -> slot lists of every batch (srh_sampler_epoch_segments:
        n_uniq_n, seg_rows, seg_end, seg, seg_a, seg_b -- table rows = ids + the offsets) behind the fixed-order
        batch-gradient reduction.
        slot (None | 0 | 1 | ...): None -- fresh arrays, the caller's to keep.  An integer -- the arrays of that slot, owned
        by the sampler and OVERWRITTEN by the next call with the same slot: a training loop that alternates two slots never
        allocates or frees an epoch's 25 MB (freeing them -- munmap -- beside a thread that is enqueueing GPU work was
        measured to stall it: 0.4 ms per epoch boundary, profiles/r05_o_epoch_boundary_host_cost.txt)."""
        e = self.n_edges
        nb = (e + batch_size - 1) // batch_size
        key = (slot, int(batch_size), int(n_negs), bool(with_unique))
        held = None if slot is None else self._epoch_slots.get(key)
        if held is not None:
            u, i, j = held["u"], held["i"], held["j"]
            res = {"u": u, "i": i, "j": j, "n_batches": nb}
            uu = ui = nuu = nui = None
            if with_unique:
                uu, ui, nuu, nui = held["uniq_u"], held["uniq_i"], held["n_uniq_u"], held["n_uniq_i"]
                for a in (uu, ui, nuu, nui):
                    a.fill(0)
                res.update(uniq_u=uu, uniq_i=ui, n_uniq_u=nuu, n_uniq_i=nui)
        else:
            u = np.empty(e, dtype=np.int32)
            i = np.empty(e, dtype=np.int32)
            j = np.empty(e * n_negs, dtype=np.int32)
            res = {"u": u, "i": i, "j": j, "n_batches": nb}
            uu = ui = nuu = nui = None
            if with_unique:
                uu = np.zeros(nb * batch_size, dtype=np.int32)
                ui = np.zeros(nb * batch_size, dtype=np.int32)
                nuu = np.zeros(nb, dtype=np.int32)
                nui = np.zeros(nb, dtype=np.int32)
                res.update(uniq_u=uu, uniq_i=ui, n_uniq_u=nuu, n_uniq_i=nui)
            if slot is not None:
                self._epoch_slots[key] = {k: v for k, v in res.items() if k != "n_batches"}
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._pushed = None                      # (the C++ generator moves on: python's copy is behind until the next push)
        check(self._lib.srh_sampler_epoch(self._h, batch_size, n_negs, vp(u), vp(i), vp(j), vp(uu), vp(nuu),
                                          vp(ui), vp(nui)), "srh_sampler_epoch")
        if with_segments:
            if not with_unique or n_negs != 1:
                raise SelfrecHipError("Sampler.epoch: with_segments needs with_unique and n_negs = 1")
            user_row0, item_row0 = (0, 0) if with_segments is True else (int(with_segments[0]), int(with_segments[1]))
            seg = None if slot is None else self._epoch_slots.get(key + ("seg",))
            if seg is None:
                seg = {"n_uniq_n": np.zeros(nb, dtype=np.int32), "seg_b": np.zeros(nb * batch_size, dtype=np.int32)}
                seg.update({k: np.zeros(3 * nb * batch_size, dtype=np.int32) for k in ("seg_rows", "seg_end", "seg", "seg_a")})
                if slot is not None:
                    self._epoch_slots[key + ("seg",)] = seg
            check(self._lib.srh_sampler_epoch_segments(self._h, batch_size, vp(u), vp(i), vp(j), vp(uu), vp(nuu), vp(ui), vp(nui),
                                                       user_row0, item_row0, vp(seg["n_uniq_n"]), vp(seg["seg_rows"]),
                                                       vp(seg["seg_end"]), vp(seg["seg"]), vp(seg["seg_a"]), vp(seg["seg_b"])),
                       "srh_sampler_epoch_segments")
            res.update(seg)
        return res

    def sample_range(self, n: int, k: int) -> np.ndarray:
        out = np.empty(k, dtype=np.int64)
        self._pushed = None                      # (the C++ generator moves on: python's copy is behind until the next push)
        check(self._lib.srh_sampler_sample_range(self._h, n, k, out.ctypes.data_as(C.c_void_p)),
              "srh_sampler_sample_range")
        return out

    def next_u32(self) -> int:
        v = C.c_uint32()
        self._pushed = None                      # (the C++ generator moves on: python's copy is behind until the next push)
        check(self._lib.srh_sampler_next_u32(self._h, C.byref(v)))
        return int(v.value)


# ----------------------------------------------------------------------------------------
# (a-2..a-4) device CSR, normalisation, SpMM
# ----------------------------------------------------------------------------------------
class DeviceCSR:
    """CSR matrix resident in HBM (int32 structure, fp32 values) plus its SpMM schedule.

    ``structure_of`` shares indptr/indices/plan with another DeviceCSR (edge-dropped views
    are new value arrays over the same structure).
    """

    def __init__(self, indptr, indices, vals, shape, device=None, split_len: int = 0, structure_of=None,
                 xcd_split_row: int = 0, row_mid=None):
        self._lib = _lib.load()
        _lib.require_gpu()
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.shape = (int(shape[0]), int(shape[1]))
        if structure_of is not None:
            self.indptr, self.indices, self._plan_owner = structure_of.indptr, structure_of.indices, structure_of
            self.h_indptr = structure_of.h_indptr
            self._plan = structure_of._plan
        else:
            h_indptr = np.ascontiguousarray(indptr, dtype=np.int32)
            if h_indptr.size != self.shape[0] + 1:
                raise SelfrecHipError("DeviceCSR: indptr length != n_rows + 1")
            self.h_indptr = h_indptr
            self.indptr = torch.from_numpy(h_indptr).to(device)
            self.indices = torch.as_tensor(np.ascontiguousarray(indices, dtype=np.int32)).to(device)
            h = C.c_void_p()
            h_mid = None if row_mid is None else np.ascontiguousarray(row_mid, dtype=np.int32)
            if h_mid is not None and h_mid.size != self.shape[0]:
                raise SelfrecHipError("DeviceCSR: row_mid needs one entry per row")
            check(self._lib.srh_spmm_plan_create(C.byref(h), self.shape[0], self.shape[1],
                                                 h_indptr.ctypes.data_as(C.c_void_p), split_len, int(xcd_split_row),
                                                 None if h_mid is None else h_mid.ctypes.data_as(C.c_void_p)),
                  "srh_spmm_plan_create")
            self._plan = h
            self._plan_owner = None
        if isinstance(vals, torch.Tensor):
            self.vals = vals.to(device=device, dtype=torch.float32).contiguous()
        else:
            self.vals = torch.as_tensor(np.ascontiguousarray(vals, dtype=np.float32)).to(device)
        self.nnz = int(self.indices.numel())
        if self.vals.numel() != self.nnz:
            raise SelfrecHipError("DeviceCSR: values / indices length mismatch")

    def __del__(self):
        if getattr(self, "_plan_owner", 1) is None and getattr(self, "_plan", None):
            self._lib.srh_spmm_plan_destroy(self._plan)
            self._plan = None

    @classmethod
    def from_scipy(cls, mat, device=None, split_len: int = 0):
        m = mat.tocsr()
        m.sort_indices()
        return cls(m.indptr, m.indices, m.data, m.shape, device=device, split_len=split_len)

    def with_values(self, vals):
        return DeviceCSR(None, None, vals, self.shape, device=self.vals.device, structure_of=self)

    def replanned(self, split_len: int = 0, xcd_split_row: int = 0, row_mid=None):
        """The same matrix (the device arrays are shared) under ANOTHER SpMM schedule: its own segment length and
        task-to-XCD dealing.  The engine runs the column-masked launch of a step on a class-free plan."""
        other = DeviceCSR.__new__(DeviceCSR)
        other._lib = self._lib
        other.shape, other.h_indptr, other.indptr, other.indices = self.shape, self.h_indptr, self.indptr, self.indices
        h = C.c_void_p()
        h_mid = None if row_mid is None else np.ascontiguousarray(row_mid, dtype=np.int32)
        check(self._lib.srh_spmm_plan_create(C.byref(h), self.shape[0], self.shape[1],
                                             self.h_indptr.ctypes.data_as(C.c_void_p), int(split_len), int(xcd_split_row),
                                             None if h_mid is None else h_mid.ctypes.data_as(C.c_void_p)),
              "srh_spmm_plan_create")
        other._plan, other._plan_owner = h, None
        other.vals, other.nnz = self.vals, self.nnz
        other._arrays_of = self                    # (keeps the shared tensors' owner alive)
        return other


def column_class_order(indptr, indices, min_len: int, bit: int = 0):
    """Host helper for DeviceCSR(row_mid=...): reorder the entries of every row with >= min_len non-zeros
    as [class-0 columns | class-1 columns] (stable), leave the others alone; a column's class is bit `bit` of its
    id (0: even / odd).  Returns (perm, row_mid): apply perm to indices / values / anything aligned with them;
    row_mid[r] = number of class-0 entries of a reordered row, or -1 - c for a row left whole whose columns are
    mostly of class c."""
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices)
    n = indptr.size - 1
    lens = np.diff(indptr)
    row_of = np.repeat(np.arange(n, dtype=np.int64), lens)
    odd = ((indices >> int(bit)) & 1).astype(np.int64)
    split = lens >= int(min_len)
    key = row_of * 2 + np.where(split[row_of], odd, 0)
    perm = np.argsort(key, kind="stable")
    n_odd = np.bincount(row_of, weights=odd, minlength=n).astype(np.int64)
    n_even = lens - n_odd
    row_mid = np.where(split, n_even, -1 - (n_odd > n_even).astype(np.int64)).astype(np.int32)
    return perm, row_mid


ADAM_EPILOGUE = True      # make_epilogue(adam=...) / batch_fetch_args(adam_coef=...) exist (engine.py: fuse_adam)


def make_epilogue(*, perturb_eps=None, noise=None, rng_seed=0, rng_offset=0, rng_step=None,
                  rng_stride=0, prev=None, mean_div=None, mean_out=None, add=None, add_scale=None, alpha=1.0,
                  row_mark=None, col_mark=None, mark_stamp=None, add_mark=None, add_sparse=None,
                  extra_out=None, extra_noise=None, extra_rng_offset=None, main_clean=False, d_full=0, col0=0,
                  row_scale=None, scale_in=False, scale_out=False, prev_unscale=None, add_rowscale=None, d_valid=0,
                  adam=None):
    """row_scale / scale_in / scale_out / prev_unscale / add_rowscale: per-row scaling for value-free products
    (include/selfrec_hip.h: SRH_SCALE_*); prev_unscale / add_rowscale are lists of booleans aligned with prev / add.
    adam: dict(param, m, v, coef, beta1, beta2, eps, clear=[tables], clear_mark, cursor) -- SRH_EPI_ADAM: the product (after
    AXPY) is the gradient of `param`, which takes the optimiser's step in the epilogue instead of the gradient being stored;
    mark_stamp must then be batch_fetch's COPY of the step."""
    ep = SpmmEpilogue()
    if row_scale is not None:
        ep.d_row_scale = _p(row_scale, torch.float32, "row_scale")
        ep.scale_flags = (_lib.SRH_SCALE_IN if scale_in else 0) | (_lib.SRH_SCALE_OUT if scale_out else 0)
        ep.prev_unscale_mask = sum(1 << t for t, f in enumerate(prev_unscale or []) if f)
        ep.add_rowscale_mask = sum(1 << t for t, f in enumerate(add_rowscale or []) if f)
    ep.noise_d_full, ep.noise_col0 = int(d_full), int(col0)      # column-sharded tables (0 = whole rows)
    ep.noise_d_valid = int(d_valid)                               # zero-padded rows: PERTURB's noise ends here (0 = all)
    keep = []
    flags = 0
    if perturb_eps is not None:
        flags |= _lib.SRH_EPI_PERTURB
        ep.eps = float(perturb_eps)
        ep.d_noise = _p(noise, torch.float32, "noise")
        ep.rng_seed, ep.rng_offset = int(rng_seed), int(rng_offset)
        ep.d_rng_step = _p(rng_step, torch.int64, "rng_step")
        ep.rng_stride = int(rng_stride)
        keep += [noise, rng_step]
        extra_out = list(extra_out or [])
        if extra_out or main_clean:        # further perturbed copies of the same product (SimGCL layer 1)
            if len(extra_out) > _lib.SRH_MAX_EXTRA:
                raise SelfrecHipError(f"at most {_lib.SRH_MAX_EXTRA} extra perturbed outputs")
            ep.n_extra, ep.main_clean = len(extra_out), int(bool(main_clean))
            for k, t in enumerate(extra_out):
                ep.d_extra_out[k] = _p(t, torch.float32, "extra_out")
                ep.d_extra_noise[k] = _p((extra_noise or [None] * len(extra_out))[k], torch.float32, "extra_noise")
                ep.extra_rng_offset[k] = int((extra_rng_offset or [0] * len(extra_out))[k])
            keep += extra_out + list(extra_noise or [])
    if mean_out is not None:
        flags |= _lib.SRH_EPI_MEAN
        prev = list(prev or [])
        if len(prev) > _lib.SRH_MAX_PREV:
            raise SelfrecHipError(f"at most {_lib.SRH_MAX_PREV} earlier layers can be averaged in the epilogue")
        ep.n_prev = len(prev)
        for t, x in enumerate(prev):
            ep.d_prev[t] = _p(x, torch.float32, "prev")
        ep.mean_div = float(mean_div)
        ep.d_mean_out = _p(mean_out, torch.float32, "mean_out")
        keep += prev + [mean_out]
    if add or alpha != 1.0:
        add = list(add or [])
        flags |= _lib.SRH_EPI_AXPY
        ep.alpha = float(alpha)
        if len(add) > _lib.SRH_MAX_ADD:
            raise SelfrecHipError(f"at most {_lib.SRH_MAX_ADD} addends")
        ep.n_add = len(add)
        for t, x in enumerate(add):
            ep.d_add[t] = _p(x, torch.float32, "add")
            ep.add_scale[t] = float(add_scale[t])
        keep += list(add)
    if row_mark is not None or col_mark is not None or add_mark is not None:
        ep.d_row_mark = _p(row_mark, torch.int32, "row_mark")
        ep.d_col_mark = _p(col_mark, torch.int32, "col_mark")
        ep.d_mark_stamp = _p(mark_stamp, torch.int64, "mark_stamp")
        ep.d_add_mark = _p(add_mark, torch.int32, "add_mark")
        ep.add_sparse_mask = sum(1 << t for t, f in enumerate(add_sparse or []) if f)
        keep += [row_mark, col_mark, mark_stamp, add_mark]
    if adam is not None:
        flags |= _lib.SRH_EPI_ADAM
        ep.d_adam_param, ep.d_adam_m, ep.d_adam_v = (_p(adam[k], torch.float32, k) for k in ("param", "m", "v"))
        ep.d_adam_coef = _p(adam["coef"], torch.float32, "coef")
        ep.adam_beta1, ep.adam_beta2 = float(adam.get("beta1", 0.9)), float(adam.get("beta2", 0.999))
        ep.adam_eps = float(adam.get("eps", 1e-8))
        clear = list(adam.get("clear") or [])
        if len(clear) > _lib.SRH_MAX_ADAM_CLEAR:
            raise SelfrecHipError(f"at most {_lib.SRH_MAX_ADAM_CLEAR} tables to clear")
        ep.adam_n_clear = len(clear)
        for k, t in enumerate(clear):
            ep.d_adam_clear[k] = _p(t, torch.float32, "clear")
        ep.d_adam_clear_mark = _p(adam.get("clear_mark"), torch.int32, "clear_mark")
        ep.d_adam_cursor = _p(adam.get("cursor"), torch.int64, "cursor")
        if mark_stamp is not None and not ep.d_mark_stamp:
            ep.d_mark_stamp = _p(mark_stamp, torch.int64, "mark_stamp")
        keep += [adam, mark_stamp]
    ep.flags = flags
    ep._keepalive = keep + [row_scale]
    return ep


def spmm(csr: DeviceCSR, x: torch.Tensor, out: torch.Tensor | None = None, epilogue: SpmmEpilogue | None = None,
         pattern: bool = False, fetch=None):
    """out = csr @ x (+ fused epilogue).  x: (n_cols, d) fp32.  pattern=True: the structure with every stored entry 1
    (no value stream; d >= 64) -- with row scaling in the epilogue this is the value-free form of D^-1/2 A D^-1/2.
    fetch: a BatchFetchArgs (batch_fetch_args) -- the launch also stages the step's batch (srh_spmm_f32_with_fetch)."""
    if x.dim() != 2 or x.shape[0] != csr.shape[1]:
        raise SelfrecHipError(f"spmm: x has shape {tuple(x.shape)}, expected ({csr.shape[1]}, d)")
    d = int(x.shape[1])
    if out is None:
        out = torch.empty((csr.shape[0], d), dtype=torch.float32, device=x.device)
    common = (csr._plan, _p(csr.indptr, torch.int32), _p(csr.indices, torch.int32),
              None if pattern else _p(csr.vals, torch.float32, "vals"), _p(x, torch.float32, "x"),
              _p(out, torch.float32, "out"), d, C.byref(epilogue) if epilogue is not None else None)
    if fetch is not None:
        check(_lib.load().srh_spmm_f32_with_fetch(*common, C.byref(fetch), _stream()), "srh_spmm_f32_with_fetch")
    else:
        check(_lib.load().srh_spmm_f32(*common, _stream()), "srh_spmm_f32")
    return out


# ---- any table width through the boundary (reference base/recommender.py:16: `embedding.size` is any integer) ----
SPMM_WIDTHS = (8, 16, 32, 64, 128, 256)      # row widths srh_spmm_f32 serves (csrc/spmm.hip)
ROW_WIDTHS = (32, 64, 128, 256)              # LPR kernels: BPR / L2 / scoring GEMM (csrc/common.h: dim_supported)
NCE_WIDTHS = (64, 128, 256)                  # srh_infonce_fwd_bwd (256: the split path only)


def padded_width(d: int, widths) -> int | None:
    """The narrowest width of `widths` that holds d columns (None: wider than the widest)."""
    return next((w for w in widths if w >= int(d)), None)


def pad_cols(t: torch.Tensor, width: int) -> torch.Tensor:
    """(rows, d) -> contiguous (rows, width) with zero columns on the right (a no-op view when d == width).  Zero columns
    change none of the path's results: products, inner products, norms and F.normalize ignore them, their gradients
    are exactly zero."""
    d = int(t.shape[1])
    if d == width:
        return t.contiguous()
    out = torch.zeros((t.shape[0], width), dtype=t.dtype, device=t.device)
    out[:, :d] = t
    return out


def spmm_any(csr: DeviceCSR, x: torch.Tensor) -> torch.Tensor:
    """csr @ x for x of ANY width: 256-column blocks, the last one zero-padded to the next width the kernels serve.
    (Widths the kernels serve directly take the plain call: no copy.)"""
    d = int(x.shape[1])
    if d in SPMM_WIDTHS:
        return spmm(csr, x.contiguous())
    out = torch.empty((csr.shape[0], d), dtype=torch.float32, device=x.device)
    for c0 in range(0, d, SPMM_WIDTHS[-1]):
        wb = min(SPMM_WIDTHS[-1], d - c0)
        y = spmm(csr, pad_cols(x[:, c0:c0 + wb], padded_width(wb, SPMM_WIDTHS)))
        out[:, c0:c0 + wb] = y[:, :wb]
    return out


def spmm_plan_run_tasks(csr: DeviceCSR, d: int) -> int:
    """Records in the task list a launch of `csr` on d-column tables runs now (srh_spmm_plan_run_tasks)."""
    n = int(_lib.load().srh_spmm_plan_run_tasks(csr._plan, int(d)))
    if n < 0:
        raise SelfrecHipError(f"spmm_plan_run_tasks: no task list for d = {d}")
    return n


def spmm_set_xcd_shares(csr: DeviceCSR, d: int, blocks_per_xcd=None):
    """Deal the plan's workgroups of real tasks to the 8 XCDs in these numbers (srh_spmm_plan_set_xcd_shares; None: the
    canonical equal dealing).  Same tasks, same sums; synchronises the device -- never inside a stream capture, and a
    captured launch of this plan must be re-captured afterwards."""
    if blocks_per_xcd is None:
        check(_lib.load().srh_spmm_plan_set_xcd_shares(csr._plan, int(d), None), "srh_spmm_plan_set_xcd_shares")
        return
    h = np.ascontiguousarray(blocks_per_xcd, dtype=np.int32)
    if h.shape != (8,):
        raise SelfrecHipError("spmm_set_xcd_shares: eight shares, one per XCD")
    check(_lib.load().srh_spmm_plan_set_xcd_shares(csr._plan, int(d), h.ctypes.data_as(C.c_void_p)),
          "srh_spmm_plan_set_xcd_shares")


def spmm_probe(csr: DeviceCSR, x: torch.Tensor, out: torch.Tensor, epilogue: SpmmEpilogue | None = None, pattern: bool = False):
    """One srh_spmm_f32_probe launch (no column marks, d = 64 / 128 / 256).  Returns (finish, begin, end, xcd): finish[k]
    = when XCD k's last wave left, in us after the launch's first wave began; begin / end / xcd per stamped wave (us, us,
    0 .. 7) in task order.  A device-to-host sync: calibration and lab work, not the step."""
    d = int(x.shape[1])
    n = spmm_plan_run_tasks(csr, d)
    stamps = torch.zeros(3 * n, dtype=torch.int64, device=x.device)
    check(_lib.load().srh_spmm_f32_probe(csr._plan, _p(csr.indices, torch.int32), None if pattern else _p(csr.vals, torch.float32, "vals"),
                                         _p(x, torch.float32, "x"), _p(out, torch.float32, "out"), d,
                                         C.byref(epilogue) if epilogue is not None else None, stamps.data_ptr(), _stream()),
          "srh_spmm_f32_probe")
    rec = stamps.cpu().numpy().reshape(n, 3)
    rec = rec[rec[:, 1] != 0]
    if rec.shape[0] == 0:
        raise SelfrecHipError("spmm_probe: no wave left a stamp")
    t0 = rec[:, 0].min()
    begin, end, xcd = (rec[:, 0] - t0) / 100.0, (rec[:, 1] - t0) / 100.0, (rec[:, 2] & 0xff).astype(np.int64)
    finish = np.array([end[xcd == k].max() if (xcd == k).any() else 0.0 for k in range(8)])
    return finish, begin, end, xcd


def spmm_gather_bound(csr, x: torch.Tensor, iters: int = 30) -> float:
    """us per launch of srh_spmm_gather_bound on `csr`'s plan: the product's own gathers on the product's own schedule and
    nothing after them -- a measured lower bound of ``spmm(csr, x)`` (HIP events on the launch stream).  Measurement only."""
    d = int(x.shape[1])
    scratch = torch.zeros(d, dtype=torch.float32, device=x.device)
    lib = _lib.load()

    def once():
        check(lib.srh_spmm_gather_bound(csr._plan, _p(csr.indices, torch.int32), _p(x, torch.float32, "x"), scratch.data_ptr(),
                                        d, _stream()), "srh_spmm_gather_bound")
    for _ in range(5):
        once()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        once()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def gather_floor_probe(indices: torch.Tensor, x: torch.Tensor, blocks: int = 4096, iters: int = 30) -> float:
    """us per pass of the bare gather stream over `indices` into the (rows, d) table `x` (srh_gather_floor_probe: the row
    fetches of one propagation launch and nothing else), HIP events on the launch stream.  Measurement only."""
    d = int(x.shape[1])
    sink = torch.zeros(4, dtype=torch.float32, device=x.device)
    lib = _lib.load()

    def once():
        check(lib.srh_gather_floor_probe(_p(indices, torch.int32, "indices"), int(indices.numel()), _p(x, torch.float32, "x"),
                                         int(x.shape[0]), d, int(blocks), sink.data_ptr(), _stream()), "srh_gather_floor_probe")
    for _ in range(5):
        once()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        once()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def adj_sym_normalize(indptr, indices, edge_id, keep, n_rows: int, weight=None, out=None, deg_ws=None,
                      inv_sqrt_table=None, row_offset: int = 0, phase: int = 0):
    dev = indices.device
    if out is None:
        out = torch.empty(indices.numel(), dtype=torch.float32, device=dev)
    if deg_ws is None:
        deg_ws = torch.empty(n_rows, dtype=torch.float32, device=dev)
    check(_lib.load().srh_adj_sym_normalize(n_rows, _p(indptr, torch.int32), _p(indices, torch.int32),
                                            _p(edge_id, torch.int32), _p(weight, torch.float32),
                                            _p(keep, torch.uint8), _p(inv_sqrt_table, torch.float32),
                                            0 if inv_sqrt_table is None else int(inv_sqrt_table.numel()),
                                            _p(deg_ws, torch.float32),
                                            _p(out, torch.float32), int(row_offset), int(phase), _stream()),
          "srh_adj_sym_normalize")
    return out


def spmm3(csrs, x, outs):
    """outs[v] = csrs[v] @ x for three DeviceCSRs over ONE structure (srh_spmm3_f32: x rows gathered once).
    Raises SelfrecHipError (unsupported) unless d == 64."""
    base = csrs[0]
    if any(c._plan is not base._plan for c in csrs) or len(csrs) != 3 or len(outs) != 3:
        raise SelfrecHipError("spmm3: needs three value arrays over one structure and three outputs")
    check(_lib.load().srh_spmm3_f32(base._plan, _p(base.indices, torch.int32), _p(csrs[0].vals, torch.float32),
                                    _p(csrs[1].vals, torch.float32), _p(csrs[2].vals, torch.float32),
                                    _p(x, torch.float32, "x"), _p(outs[0], torch.float32), _p(outs[1], torch.float32),
                                    _p(outs[2], torch.float32), int(x.shape[1]), _stream()), "srh_spmm3_f32")


# ----------------------------------------------------------------------------------------
# (a-5..a-8) losses
# ----------------------------------------------------------------------------------------
def bpr_ws(batch: int, device):
    return torch.empty(int(_lib.load().srh_bpr_ws_bytes(batch)), dtype=torch.uint8, device=device)


def _segments(seg, nce_rows):
    """srh_batch_segments_t from a dict of device int32 tensors: n_uniq_u, n_uniq_i, n_uniq_n, seg_rows, seg_end, seg, seg_a, seg_b
    [, batch_no: the arrays are then EPOCH arrays] [, rows_are_zero: bool]."""
    g = _lib.BatchSegments()
    for k in ("n_uniq_u", "n_uniq_i", "n_uniq_n", "seg_rows", "seg_end", "seg", "seg_a", "seg_b"):
        setattr(g, "d_" + k, _p(seg[k], torch.int32))
    g.d_batch_no = _p(seg.get("batch_no"), torch.int32)
    g.nce_rows = int(nce_rows)
    g.rows_are_zero = int(bool(seg.get("rows_are_zero", False)))
    return g


def _bpr_problem(user, item, reg_user, reg_item, u_idx, i_idx, j_idx, batch, n_rows_dev, reg_coef, reg_include_neg, loss_scale,
                 g_user, g_item, greg_user, greg_item, losses, ws, seg, nce_rows):
    b = _lib.BprProblem()
    b.d_user, b.d_item = _p(user, torch.float32), _p(item, torch.float32)
    b.d_reg_user, b.d_reg_item = _p(reg_user, torch.float32), _p(reg_item, torch.float32)
    b.d_u_idx, b.d_i_idx, b.d_j_idx = _p(u_idx, torch.int32), _p(i_idx, torch.int32), _p(j_idx, torch.int32)
    b.B, b.d_n_rows = int(batch), _p(n_rows_dev, torch.int32)
    b.reg_coef, b.reg_include_neg, b.loss_scale = float(reg_coef), int(bool(reg_include_neg)), float(loss_scale)
    b.d_g_user, b.d_g_item = _p(g_user, torch.float32), _p(g_item, torch.float32)
    b.d_greg_user, b.d_greg_item = _p(greg_user, torch.float32), _p(greg_item, torch.float32)
    b.d_losses, b.d_ws = _p(losses, torch.float64), _p(ws)
    if seg is not None:
        b._seg = _segments(seg, nce_rows)          # (kept alive by the struct object)
        b.seg = C.pointer(b._seg)
    return b


def bpr_l2_fwd_bwd(user, item, reg_user, reg_item, u_idx, i_idx, j_idx, *, batch, n_rows_dev=None, reg_coef,
                   reg_include_neg, loss_scale, g_user, g_item, greg_user, greg_item, losses, ws, seg=None):
    """seg: the batch's row -> slot lists (see _segments): the gradients are summed row by row in slot order, no float atomics."""
    d = int(user.shape[1])
    if seg is not None:
        b = _bpr_problem(user, item, reg_user, reg_item, u_idx, i_idx, j_idx, batch, n_rows_dev, reg_coef, reg_include_neg,
                         loss_scale, g_user, g_item, greg_user, greg_item, losses, ws, seg, 0)
        check(_lib.load().srh_bpr_l2_fwd_bwd_p(C.byref(b), d, _stream()), "srh_bpr_l2_fwd_bwd_p")
        return
    check(_lib.load().srh_bpr_l2_fwd_bwd(
        _p(user, torch.float32), _p(item, torch.float32), _p(reg_user, torch.float32), _p(reg_item, torch.float32),
        _p(u_idx, torch.int32), _p(i_idx, torch.int32), _p(j_idx, torch.int32), int(batch),
        _p(n_rows_dev, torch.int32), d, float(reg_coef), int(bool(reg_include_neg)), float(loss_scale),
        _p(g_user, torch.float32), _p(g_item, torch.float32), _p(greg_user, torch.float32),
        _p(greg_item, torch.float32), _p(losses, torch.float64), _p(ws), _stream()), "srh_bpr_l2_fwd_bwd")


_scalar_ws = {}


def scalar_ws(device) -> torch.Tensor:
    """The 64-byte workspace the single-launch loss kernels finish their scalars in (SRH_SCALAR_WS_BYTES): one per device and
    stream, zero before its first use, left zero by every call."""
    # (one per device AND stream: the calls that share a workspace must be ordered, and a stream is what orders them)
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = _scalar_ws.get(key)
    if ws is None:
        ws = _scalar_ws[key] = torch.zeros(_lib.SCALAR_WS_BYTES // 8, dtype=torch.float64, device=device)
    return ws


def bpr_fwd(u, p, n, loss, coef):
    """loss (0-dim f32) = mean BPR loss of the rows; coef[b] = d loss / d (pos_b - neg_b).  One launch."""
    check(_lib.load().srh_bpr_fwd(_p(u, torch.float32), _p(p, torch.float32), _p(n, torch.float32), u.shape[0],
                                  int(u.shape[1]), scalar_ws(u.device).data_ptr(), _p(loss, torch.float32),
                                  _p(coef, torch.float32), _stream()), "srh_bpr_fwd")


def bpr_bwd(u, p, n, coef, gout, gu, gp, gn):
    """gout: the upstream gradient as a 0-dim f32 DEVICE tensor (read by the kernel: no host synchronisation)."""
    check(_lib.load().srh_bpr_bwd(_p(u, torch.float32), _p(p, torch.float32), _p(n, torch.float32),
                                  _p(coef, torch.float32), u.shape[0], int(u.shape[1]), _p(gout, torch.float32, "gout"),
                                  _p(gu, torch.float32), _p(gp, torch.float32), _p(gn, torch.float32), _stream()),
          "srh_bpr_bwd")


def _l2_blocks(xs, gxs=None):
    arr = (_lib.L2Block * len(xs))()
    for k, x in enumerate(xs):
        arr[k].d_x, arr[k].rows, arr[k].cols = _p(x, torch.float32, "emb"), int(x.shape[0]), int(x.shape[1])
        arr[k].d_gx = _p(gxs[k], torch.float32, "grad") if gxs is not None else None
    return arr


def l2_reg_fwd(xs, reg, norms, loss):
    """loss (0-dim f32) = reg * sum_k ||xs[k]||_F / rows_k; norms[k] = ||xs[k]||_F.  1..4 blocks of rows, one launch."""
    check(_lib.load().srh_l2_reg_fwd(_l2_blocks(xs), len(xs), float(reg), scalar_ws(xs[0].device).data_ptr(),
                                     _p(norms, torch.float32), _p(loss, torch.float32), _stream()), "srh_l2_reg_fwd")


def l2_reg_bwd(xs, reg, norms, gout, gxs):
    check(_lib.load().srh_l2_reg_bwd(_l2_blocks(xs, gxs), len(xs), float(reg), _p(norms, torch.float32),
                                     _p(gout, torch.float32, "gout"), _stream()), "srh_l2_reg_bwd")


# SRH_NCE_SPLIT16, SRH_NCE_F32 (include/selfrec_hip.h); "bf16x3" is the round-1/2 name of the split mode, kept as an alias
NCE_PRECISIONS = {"split": 0, "f32": 1, "bf16x3": 0}
NCE_DEFAULT = -1               # SRH_NCE_DEFAULT: the process default (f32 unless srh_infonce_set_precision / SRH_NCE_SPLIT16 says otherwise)


def _nce_mode(precision):
    """None -> the process default; 'split' | 'f32' -> that arithmetic for this call only."""
    if precision is None:
        return NCE_DEFAULT
    if precision not in NCE_PRECISIONS:
        raise SelfrecHipError(f"InfoNCE precision {precision!r}: one of {sorted(NCE_PRECISIONS)}")
    return NCE_PRECISIONS[precision]


def set_infonce_precision(mode: str):
    """'f32' (the default: every multiply-add of InfoNCE's two n x n x d products on the f32 MFMA, the reference's
    arithmetic) or 'split' (operands as short sums of 16-bit pieces on the 16-bit MFMA pipe -- logits on scaled f16 hi + lo,
    accurate to 2^-22 like an f32 dot product; P.V on bf16 pieces: ~10 us per step faster at the Yelp2018 shape).
    The process DEFAULT: what calls that name no precision run on (the loss mirrors of the op-level tier); a trainer
    carries its own mode and passes it with every call (engine.FusedTrainer.nce_precision)."""
    if mode not in NCE_PRECISIONS:
        raise SelfrecHipError(f"InfoNCE precision {mode!r}: one of {sorted(NCE_PRECISIONS)}")
    check(_lib.load().srh_infonce_set_precision(NCE_PRECISIONS[mode]), "srh_infonce_set_precision")


def get_infonce_precision() -> str:
    got = int(_lib.load().srh_infonce_get_precision())
    return {0: "split", 1: "f32"}[got]


def infonce_ws(n: int, d: int, device):
    return torch.empty(int(_lib.load().srh_infonce_ws_bytes(n, d)), dtype=torch.uint8, device=device)


def infonce_fwd_bwd(v1, v2, idx, n, *, n_dev=None, tau, loss_scale, loss, g1, g2, ws):
    d = int(v1.shape[1])
    need = int(_lib.load().srh_infonce_ws_bytes(n, d))
    if ws.numel() * ws.element_size() < need:
        raise SelfrecHipError(f"infonce workspace too small: {ws.numel() * ws.element_size()} < {need}")
    check(_lib.load().srh_infonce_fwd_bwd(_p(v1, torch.float32), _p(v2, torch.float32), _p(idx, torch.int32), int(n),
                                          _p(n_dev, torch.int32), d, float(tau), float(loss_scale),
                                          _p(loss, torch.float64), _p(g1, torch.float32), _p(g2, torch.float32),
                                          _p(ws), _stream()), "srh_infonce_fwd_bwd")


def infonce_multi(problems, *, d, tau, loss_scale, loss, ws, precision=None):
    """problems: [(v1, v2, idx, n_max, n_dev, g1, g2[, g2_exclusive]), ...] evaluated by one set of launches."""
    lib = _lib.load()
    arr = (_lib.InfonceProblem * len(problems))()
    need = 0
    for k, (v1, v2, idx, n, n_dev, g1, g2, *rest) in enumerate(problems):
        arr[k].d_v1, arr[k].d_v2 = _p(v1, torch.float32), _p(v2, torch.float32)
        arr[k].d_idx, arr[k].n, arr[k].d_n = _p(idx, torch.int32), int(n), _p(n_dev, torch.int32)
        arr[k].d_g1, arr[k].d_g2 = _p(g1, torch.float32), _p(g2, torch.float32)
        arr[k].g2_exclusive = int(bool(rest[0])) if rest else 0
        need += int(lib.srh_infonce_ws_bytes(n, d))
    if ws.numel() * ws.element_size() < need:
        raise SelfrecHipError(f"infonce workspace too small: {ws.numel() * ws.element_size()} < {need}")
    check(lib.srh_infonce_fwd_bwd_multi(arr, len(problems), int(d), float(tau), float(loss_scale),
                                        _p(loss, torch.float64), _p(ws), _nce_mode(precision), _stream()),
          "srh_infonce_fwd_bwd_multi")


def bpr_infonce(user, item, reg_user, reg_item, u_idx, i_idx, j_idx, *, batch, n_rows_dev=None, reg_coef,
                reg_include_neg, loss_scale, g_user, g_item, greg_user, greg_item, losses, bpr_ws,
                problems, tau, cl_scale, cl_loss, nce_ws, precision=None, seg=None, nce_rows=0):
    """bpr_l2_fwd_bwd + infonce_multi with their O(batch) kernels sharing launches (srh_bpr_infonce_fwd_bwd).
    seg / nce_rows: the batch's row -> slot lists and how the problems name those rows (srh_batch_segments_t): every
    gradient row is then written once, by the row group that owns it, in slot order -- no float atomics."""
    lib = _lib.load()
    d = int(user.shape[1])
    b = _bpr_problem(user, item, reg_user, reg_item, u_idx, i_idx, j_idx, batch, n_rows_dev, reg_coef, reg_include_neg,
                     loss_scale, g_user, g_item, greg_user, greg_item, losses, bpr_ws, seg, nce_rows)
    arr = (_lib.InfonceProblem * len(problems))()
    need = 0
    for k, (v1, v2, idx, n, n_dev, g1, g2, *rest) in enumerate(problems):
        arr[k].d_v1, arr[k].d_v2 = _p(v1, torch.float32), _p(v2, torch.float32)
        arr[k].d_idx, arr[k].n, arr[k].d_n = _p(idx, torch.int32), int(n), _p(n_dev, torch.int32)
        arr[k].d_g1, arr[k].d_g2 = _p(g1, torch.float32), _p(g2, torch.float32)
        arr[k].g2_exclusive = int(bool(rest[0])) if rest else 0
        need += int(lib.srh_infonce_ws_bytes(n, d))
    if nce_ws.numel() * nce_ws.element_size() < need:
        raise SelfrecHipError(f"infonce workspace too small: {nce_ws.numel() * nce_ws.element_size()} < {need}")
    check(lib.srh_bpr_infonce_fwd_bwd(C.byref(b), arr, len(problems), d, float(tau), float(cl_scale),
                                      _p(cl_loss, torch.float64), _p(nce_ws), _nce_mode(precision), _stream()),
          "srh_bpr_infonce_fwd_bwd")


# ----------------------------------------------------------------------------------------
# (a-9) optimiser, (a-10/11) evaluation, utilities
# ----------------------------------------------------------------------------------------
def adam_step(param, grad, m, v, *, step=0, step_dev=None, lr, beta1=0.9, beta2=0.999, eps=1e-8, clear=None,
              row_mark=None, advance_cursor=None):
    """clear / row_mark / advance_cursor: the fused end-of-step reset (srh_adam_step_reset)."""
    if clear is not None or advance_cursor is not None:
        clear = list(clear or [])
        tables = (C.c_void_p * max(1, len(clear)))(*[_p(t, torch.float32, "clear") for t in clear])
        check(_lib.load().srh_adam_step_reset(_p(param, torch.float32), _p(grad, torch.float32), _p(m, torch.float32),
                                              _p(v, torch.float32), int(param.shape[0]), int(param.shape[1]),
                                              _p(step_dev, torch.int64), float(lr), float(beta1), float(beta2), float(eps),
                                              _p(row_mark, torch.int32), len(clear), tables,
                                              _p(advance_cursor, torch.int64), _stream()), "srh_adam_step_reset")
        return
    check(_lib.load().srh_adam_step(_p(param, torch.float32), _p(grad, torch.float32), _p(m, torch.float32),
                                    _p(v, torch.float32), param.numel(), int(step), _p(step_dev, torch.int64),
                                    float(lr), float(beta1), float(beta2), float(eps), _stream()), "srh_adam_step")


def score_mask_topk(user_emb, user_ids, item_emb, r_indptr, r_indices, k, scores_ws=None):
    """ids, scores (device, (n_query, k)).  scores_ws: (rows, n_items) slab the queries pass through
    `rows` at a time inside the call (default: one slab for all queries)."""
    nq = int(user_ids.numel()) if user_ids is not None else int(user_emb.shape[0])
    n_items, d = int(item_emb.shape[0]), int(item_emb.shape[1])
    dev = item_emb.device
    if scores_ws is None:
        scores_ws = torch.empty((nq, n_items), dtype=torch.float32, device=dev)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
    check(_lib.load().srh_score_mask_topk(_p(user_emb, torch.float32), _p(user_ids, torch.int32), nq,
                                          _p(item_emb, torch.float32), n_items, d, _p(r_indptr, torch.int32),
                                          _p(r_indices, torch.int32), int(k), _p(scores_ws, torch.float32),
                                          int(scores_ws.shape[0]), _p(ids, torch.int32), _p(sc, torch.float32),
                                          _stream()), "srh_score_mask_topk")
    return ids, sc


def score_mask_topk_filtered(user_emb, user_ids, item_emb, r_indptr, r_indices, k, *, sample_items=3072, cap=1024,
                             chunk_rows=4096, ws=None):
    """ids, scores, counts (device).  Rows with counts > cap are not valid (see include/selfrec_hip.h):
    rank those with score_mask_topk."""
    lib = _lib.load()
    nq = int(user_ids.numel()) if user_ids is not None else int(user_emb.shape[0])
    n_items, d = int(item_emb.shape[0]), int(item_emb.shape[1])
    dev = item_emb.device
    sample_items = max(int(k), min(int(sample_items), n_items))
    chunk_rows = max(1, min(int(chunk_rows), nq))
    need = int(lib.srh_score_mask_topk_filtered_ws_bytes(chunk_rows, sample_items, int(k), int(cap), n_items, d))
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
    counts = torch.empty(nq, dtype=torch.int32, device=dev)
    check(lib.srh_score_mask_topk_filtered(_p(user_emb, torch.float32), _p(user_ids, torch.int32), nq,
                                           _p(item_emb, torch.float32), n_items, d, _p(r_indptr, torch.int32),
                                           _p(r_indices, torch.int32), int(k), sample_items, int(cap), chunk_rows,
                                           _p(ws), _p(ids, torch.int32), _p(sc, torch.float32),
                                           _p(counts, torch.int32), _stream()), "srh_score_mask_topk_filtered")
    return ids, sc, counts, ws


def topk_trim_mark_ties(ids_k1, scores_k1):
    """(rows, K + 1) ranked ids / scores -> (rows, K) with rows holding a tie among their K + 1 scores marked ids[:, 0] < 0
    (= -1 - id): one launch (srh_topk_trim_mark_ties)."""
    rows, k1 = int(ids_k1.shape[0]), int(ids_k1.shape[1])
    ids = torch.empty((rows, k1 - 1), dtype=torch.int32, device=ids_k1.device)
    sc = torch.empty((rows, k1 - 1), dtype=torch.float32, device=ids_k1.device)
    check(_lib.load().srh_topk_trim_mark_ties(_p(ids_k1, torch.int32), _p(scores_k1, torch.float32), rows, k1,
                                              _p(ids, torch.int32), _p(sc, torch.float32), _stream()), "srh_topk_trim_mark_ties")
    return ids, sc


def gemm_nt(a, b, out=None):
    m, d = int(a.shape[0]), int(a.shape[1])
    n = int(b.shape[0])
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    check(_lib.load().srh_gemm_nt_f32(_p(a, torch.float32), _p(b, torch.float32), _p(out, torch.float32), m, n, d,
                                      _stream()), "srh_gemm_nt_f32")
    return out


def topk_rows(scores, k):
    rows, n = int(scores.shape[0]), int(scores.shape[1])
    ids = torch.empty((rows, k), dtype=torch.int32, device=scores.device)
    sc = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
    check(_lib.load().srh_topk_rows(_p(scores, torch.float32), rows, n, int(k), _p(ids, torch.int32),
                                    _p(sc, torch.float32), _stream()), "srh_topk_rows")
    return ids, sc


def topk_hit_flags(ids, user_ids, t_indptr, t_indices):
    """uint8 (n_query, k): 1 where the ranked id is one of the user's test items."""
    flags = torch.empty(ids.shape, dtype=torch.uint8, device=ids.device)
    check(_lib.load().srh_topk_hit_flags(_p(ids, torch.int32), int(ids.shape[0]), int(ids.shape[1]),
                                         _p(user_ids, torch.int32), _p(t_indptr, torch.int32),
                                         _p(t_indices, torch.int32), _p(flags, torch.uint8), _stream()),
          "srh_topk_hit_flags")
    return flags


def metric_rows(flags, sizes, cuts):
    """Per-user hits (int32) and DCG / IDCG (float64) at every cut-off of `cuts` (<= 8, each <= K) from the (users, K) hit
    flags and the users' test-set sizes (int32 device tensor): srh_metric_rows.  The gain table and the ideal prefix
    sums are computed HERE with python's math.log exactly as util/evaluation.py:66-78 computes them and handed to the
    kernel, so every quotient is the reference's bit for bit.  Returns (hits, ndcg) of shape (len(cuts), users)."""
    import math
    n, k = int(flags.shape[0]), int(flags.shape[1])
    cuts = [int(c) for c in cuts]
    gains = [1.0 / math.log(pos + 2, 2) for pos in range(k)]
    ideal = np.zeros((len(cuts), k + 1), dtype=np.float64)
    for c, cut in enumerate(cuts):
        acc = 0.0
        for m in range(1, cut + 1):
            acc = acc + 1.0 / math.log(m - 1 + 2, 2)        # sum(... for pos in range(m)): left to right
            ideal[c, m] = acc
    dev = flags.device
    d_gains = torch.tensor(gains, dtype=torch.float64, device=dev)
    d_ideal = torch.from_numpy(ideal).to(dev)
    hits = torch.empty((len(cuts), n), dtype=torch.int32, device=dev)
    ndcg = torch.empty((len(cuts), n), dtype=torch.float64, device=dev)
    arr = (C.c_int32 * len(cuts))(*cuts)
    check(_lib.load().srh_metric_rows(_p(flags, torch.uint8), _p(sizes, torch.int32), n, k, arr, len(cuts),
                                      _p(d_gains, torch.float64), _p(d_ideal, torch.float64), _p(hits, torch.int32),
                                      _p(ndcg, torch.float64), _stream()), "srh_metric_rows")
    return hits, ndcg


def axpby(a, x, b, y):
    check(_lib.load().srh_axpby(float(a), _p(x, torch.float32), float(b), _p(y, torch.float32), x.numel(), _stream()),
          "srh_axpby")
    return y


def cursor_advance(cursor):
    check(_lib.load().srh_cursor_advance(_p(cursor, torch.int64), _stream()), "srh_cursor_advance")


def zero_rows(lists, d, cursor_advance=None):
    """lists: [(table, idx, count_dev_or_None, n_max, row_offset), ...] (at most 8)."""
    n = len(lists)
    vp = C.c_void_p * n
    tables = vp(*[_p(t, torch.float32, "table") for t, *_ in lists])
    idx = vp(*[_p(i, torch.int32, "idx") for _, i, *_ in lists])
    cnt = vp(*[_p(c, torch.int32, "count") for _, _, c, *_ in lists])
    n_max = (C.c_int32 * n)(*[int(m) for *_, m, _ in lists])
    off = (C.c_int32 * n)(*[int(o) for *_, o in lists])
    check(_lib.load().srh_zero_rows(n, tables, idx, cnt, n_max, off, int(d), _p(cursor_advance, torch.int64),
                                    _stream()), "srh_zero_rows")


def batch_fetch_args(ep, n_edges, batch_size, cursor, stage, meta, row_mark=None, mark_item_offset=0, zero4=None,
                     stage_cat=None, cat_item_offset=0, n_cat=None, now=None, half_batches=0, adam_coef=None, adam_lr=0.0,
                     adam_beta1=0.9, adam_beta2=0.999):
    """srh_batch_fetch_args_t.  ep: dict of device int32 arrays for the epoch (two epochs back to back when half_batches >
    0); stage: dict of staging buffers.  adam_coef: float32[2] that receives this step's Adam constants (SRH_EPI_ADAM)."""
    a = _lib.BatchFetchArgs()
    a.d_epoch_u, a.d_epoch_i, a.d_epoch_j = (_p(ep[k], torch.int32) for k in ("u", "i", "j"))
    a.d_epoch_uniq_u, a.d_epoch_uniq_i = _p(ep.get("uniq_u"), torch.int32), _p(ep.get("uniq_i"), torch.int32)
    a.d_n_uniq_u, a.d_n_uniq_i = _p(ep.get("n_uniq_u"), torch.int32), _p(ep.get("n_uniq_i"), torch.int32)
    a.n_edges, a.batch_size, a.d_cursor = int(n_edges), int(batch_size), _p(cursor, torch.int64)
    a.d_stage_u, a.d_stage_i, a.d_stage_j = (_p(stage[k], torch.int32) for k in ("u", "i", "j"))
    a.d_stage_uniq_u, a.d_stage_uniq_i = _p(stage.get("uniq_u"), torch.int32), _p(stage.get("uniq_i"), torch.int32)
    a.d_meta, a.d_row_mark = _p(meta, torch.int32), _p(row_mark, torch.int32)
    a.mark_item_offset, a.cat_item_offset = int(mark_item_offset), int(cat_item_offset)
    a.d_zero4, a.d_stage_cat, a.d_n_cat = _p(zero4, torch.float64), _p(stage_cat, torch.int32), _p(n_cat, torch.int32)
    a.d_now = _p(now, torch.int64)
    a.half_batches = int(half_batches)
    a.d_adam_coef = _p(adam_coef, torch.float32)
    a.adam_lr, a.adam_beta1, a.adam_beta2 = float(adam_lr), float(adam_beta1), float(adam_beta2)
    a._keepalive = [ep, cursor, stage, meta, row_mark, zero4, stage_cat, n_cat, now, adam_coef]
    return a


def batch_fetch(*args, **kwargs):
    """srh_batch_fetch: same arguments as batch_fetch_args (or one prebuilt BatchFetchArgs)."""
    a = args[0] if len(args) == 1 and isinstance(args[0], _lib.BatchFetchArgs) else batch_fetch_args(*args, **kwargs)
    check(_lib.load().srh_batch_fetch(C.byref(a), _stream()), "srh_batch_fetch")


# ----------------------------------------------------------------------------------------
# (e) column-sharded tables: the batch-row exchange around the loss section (csrc/exchange.hip)
# ----------------------------------------------------------------------------------------
def batch_lists(stage, meta, batch_size):
    """struct srh_batch_lists over the staging buffers of batch_fetch: (u, i, j, uniq_u, uniq_i) with
    their device-side counts (meta[0] for the pair lists, meta[1], meta[2])."""
    bl = _lib.BatchLists()
    keep = []
    for s, (name, cnt) in enumerate((("u", 0), ("i", 0), ("j", 0), ("uniq_u", 1), ("uniq_i", 2))):
        c = meta[cnt:cnt + 1]
        bl.d_idx[s] = _p(stage[name], torch.int32, name)
        bl.d_count[s] = _p(c, torch.int32, "count")
        keep += [stage[name], c]
    bl.B = int(batch_size)
    bl._keepalive = keep
    return bl


def _ptr_array(tensors, name):
    return (C.c_void_p * len(tensors))(*[_p(t, torch.float32, name) for t in tensors])


def batch_pack(lists, tables, send, cat_idx=None, n_cat=None):
    dl = int(tables[0].shape[1])
    rows = 5 * lists.B
    if send.numel() < len(tables) * rows * dl or any(int(t.shape[1]) != dl for t in tables):
        raise SelfrecHipError("batch_pack: send buffer too small / tables of different widths")
    check(_lib.load().srh_batch_pack(C.byref(lists), len(tables), _ptr_array(tables, "table"), dl,
                                     _p(send, torch.float32, "send"), _p(cat_idx, torch.int32, "cat_idx"),
                                     _p(n_cat, torch.int32, "n_cat"), _stream()), "srh_batch_pack")


def batch_unpack(lists, recv, world, dl, compact, compact_grads):
    check(_lib.load().srh_batch_unpack(C.byref(lists), len(compact), int(world), int(dl),
                                       _p(recv, torch.float32, "recv"), _ptr_array(compact, "compact"),
                                       len(compact_grads), _ptr_array(compact_grads, "compact_grad") if compact_grads else None,
                                       _stream()), "srh_batch_unpack")


def batch_scatter(lists, pairs, d_full, col0, dl):
    """pairs: [(compact gradient (5B, d_full), local gradient (N, dl)), ...]"""
    check(_lib.load().srh_batch_scatter(C.byref(lists), len(pairs), _ptr_array([c for c, _ in pairs], "compact_grad"),
                                        _ptr_array([g for _, g in pairs], "local_grad"), int(d_full), int(col0), int(dl),
                                        _stream()), "srh_batch_scatter")
