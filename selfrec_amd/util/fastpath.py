"""Host-side fast paths of the OP-LEVEL tier: what a model file written the reference's way pays per step beyond its kernels.

A step of the unmodified XSimGCL.py on this package's kernels spent 2.2 ms on the host and 1.46 ms on the device
(profiles/r02_e_dropin_torch_profiler.txt, 189 launches).  Three idioms of the reference's model files (and one of its entry class) account for the
avoidable part, and none of them needs the file to change (``dropin.install()`` switches these on, ``uninstall()`` off):

1. ``table[python_list]`` (XSimGCL.py:30, LightGCN.py:24-25, MF.py:20): torch boxes the list into a CPU tensor, copies it to
   the device and gathers with advanced indexing, whose backward is ``index_put_(accumulate=True)`` -- a sort plus several
   launches (0.24 ms per step).  The lists ``next_batch_pairwise`` yields are REAL python lists (the reference's protocol),
   but the generator also uploads all three streams -- and, for models that ask for them, their sorted unique ids -- in ONE pinned copy and registers the
   device tensors under the lists' identities; ``Tensor.__getitem__`` with a registered list (or with one of the device index
   tensors this module handed out: a stream, its unique ids) on a 2-D HIP tensor becomes ``index_select``: one gather launch, backward = ``index_add_`` (atomics, no sort).  Same rows,
   same values; gradients equal up to fp32 summation order of duplicate rows.
2. ``torch.unique(torch.Tensor(idx).type(torch.long)).cuda()`` (XSimGCL.py:46-47, SGL.py:116-117): a float round trip, a
   host sort and a synchronous H2D copy per side.  The C++ sampler already knows the sorted unique ids of the batch:
   ``torch.unique`` of a CPU tensor that IS a registered stream returns them as a tensor whose ``.cuda()`` is the copy
   already on the device.
3. ``torch.optim.Adam`` over the embedding tables (XSimGCL.py:25,37): ~12 foreach launches and their python per step.
   ``Adam`` (below) is a ``torch.optim.Adam`` whose ``step()`` runs ``srh_adam_step`` -- one launch per table, the
   arithmetic the engine's tests hold to ``torch.optim.Adam`` -- whenever the group uses the plain algorithm on fp32 HIP
   parameters, and torch's own step otherwise.

4. ``FileIO.load_data_set(path, 'graph')`` (the reference's SELFRec.py:12-13) building 1.2 M python triples that only
   ``Interaction`` reads -- and that ``shuffle()`` (util/sampler.py:7) then permutes every epoch.  With the fast paths on it
   returns ``data/loader.TripleFile``: a list-like object the native loader parses, materialised as the ordinary list the
   moment anything indexes, iterates or mutates it, its shuffles kept as one pending permutation until then.

Everything here is an optimisation of identical semantics; a list the caller edited after it was yielded (compared element
by element), an index that is not registered, a CPU tensor that does not hold exactly a yielded stream, a parameter group
with weight decay -- each takes torch's own path.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops

_state = {"on": False, "orig_getitem": None, "orig_unique": None, "orig_adam": None, "batch": {}, "pins": [], "pin_at": 0,
          "dev": None, "safe_idx": {}}
PINNED_SLOTS = 3      # staging buffers in rotation: a slot is rewritten only after the copy out of it has run (its event)
hits = {"gather_list": 0, "gather_index": 0, "unique": 0, "adam": 0}       # how often each fast path was taken (tests, profiles)


class _Stream:
    """one yielded list and what the generator knows about it"""
    __slots__ = ("lst", "n", "as_yielded", "host", "host_t", "dev", "uniq_host", "uniq_dev", "_max")

    def __init__(self, lst, host, dev):
        self.lst, self.n = lst, len(lst)
        self.as_yielded = lst.copy()            # (3.7 us for 2048 ids; compared in full below: 2 us)
        self.host, self.dev, self.uniq_host, self.uniq_dev = host, dev, None, None
        self.host_t = torch.from_numpy(host)    # int64 view of the same ids: what torch.unique's argument is compared with
        self._max = None

    def still(self, lst):
        """the caller has not edited the list since it was yielded: every element compared (the generator's lists are
        fresh objects, so only an in-place edit by the caller could change them -- anywhere in the list)"""
        return lst == self.as_yielded

    @property
    def max_id(self):
        if self._max is None:
            self._max = int(self.host.max()) if self.n else 0
        return self._max

    def set_unique(self, ids_host, ids_dev):
        hu = torch.from_numpy(ids_host).as_subclass(HostIds)
        hu._srh_dev = ids_dev
        self.uniq_host, self.uniq_dev = hu, ids_dev
        _state["safe_idx"][id(ids_dev)] = ids_dev


class HostIds(torch.Tensor):
    """sorted unique ids of a batch stream as ``torch.unique`` returns them (CPU int64), carrying the copy that is already
    on the device: ``.cuda()`` / ``.to('cuda')`` hand that out instead of copying again."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        me = args[0] if args else None
        if isinstance(me, HostIds) and getattr(me, "_srh_dev", None) is not None:
            if func is torch.Tensor.cuda:
                return me._srh_dev
            if func is torch.Tensor.to and len(args) > 1 and isinstance(args[1], (str, torch.device)) \
                    and torch.device(args[1]).type == "cuda" and not kwargs.get("dtype"):
                return me._srh_dev
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return out


def active() -> bool:
    return _state["on"]


def _pinned(n):
    """(buffer, slot) of the next staging slot, safe to overwrite.  The copy out of a pinned buffer is asynchronous and the
    op-level training loop does not synchronise per step (the loss is read every 100 batches), so on a device-bound run
    the host is a step or more ahead: the copy of batch N may still be queued when batch N + 1 is staged.  Each slot
    carries the event recorded after its last copy; the slot is reused only once that event has completed."""
    pins = _state["pins"]
    at = _state["pin_at"] = (_state["pin_at"] + 1) % PINNED_SLOTS
    while len(pins) <= at:
        pins.append([None, None])
    slot = pins[at]
    if slot[1] is not None:
        slot[1].synchronize()                  # (normally long complete: two batches have been staged since)
    if slot[0] is None or slot[0].numel() < n:
        slot[0] = torch.empty(max(n, 1 << 15), dtype=torch.int64).pin_memory()
    return slot[0], slot


def register_batch(lists, arrays, device=None):
    """Called by ``next_batch_pairwise`` right before it yields ``lists`` = (u, i, j) python lists built from the int32
    ``arrays``: one pinned host buffer [u | i | j (| unique(u) | unique(i))] (int64), one asynchronous copy, views of the
    device buffer registered under the lists' identities.  The sorted unique ids ride along only when the model asked for
    them on the previous batch (``torch.unique``: XSimGCL, SGL -- LightGCN and MF never do, and two host sorts per batch
    were 3 % of their step); a first request is served on demand (``_unique_of``).  The previous batch's registrations
    are dropped."""
    if not _state["on"] or not torch.cuda.is_available():
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    parts = [np.asarray(a, dtype=np.int64) for a in arrays]
    if _state.get("want_unique"):
        parts += [np.unique(parts[0]), np.unique(parts[1])]          # sorted: torch.unique's order
    total = sum(p.size for p in parts)
    pin, slot = _pinned(total)
    host = pin.numpy()
    at, views = 0, []
    for p in parts:
        host[at:at + p.size] = p
        views.append((at, at + p.size))
        at += p.size
    on_dev = torch.empty(total, dtype=torch.int64, device=dev)
    on_dev.copy_(pin[:total], non_blocking=True)
    if slot[1] is None:
        slot[1] = torch.cuda.Event()
    slot[1].record()                             # on the copy's stream: complete = the slot's bytes have been read
    streams = [_Stream(lst, parts[k], on_dev[views[k][0]:views[k][1]]) for k, lst in enumerate(lists)]
    _state["safe_idx"] = {id(st.dev): st.dev for st in streams}       # (the previous batch's tensors are dropped with it)
    for k in range(len(parts) - 3):
        streams[k].set_unique(parts[3 + k].copy(), on_dev[views[3 + k][0]:views[3 + k][1]])
    _state["batch"] = {id(st.lst): st for st in streams}
    _state["want_unique"] = False                                   # (set again by the first torch.unique of this batch)


def _unique_of(s):
    """sorted unique ids of a registered stream, host and device: from the batch's one copy when they rode along, else
    computed and uploaded now (the model's first ``torch.unique``; later batches carry them)"""
    _state["want_unique"] = True
    if s.uniq_host is None:
        ids = np.unique(s.host)
        s.set_unique(ids, torch.from_numpy(ids).to(s.dev.device))
    return s.uniq_host


def _lookup(lst):
    s = _state["batch"].get(id(lst))
    return s if s is not None and s.lst is lst and s.still(lst) else None


def _getitem(self, idx):
    """``Tensor.__getitem__`` with the two row-gather idioms of the model files routed to ``index_select``."""
    orig = _state["orig_getitem"]
    plain = type(self) is torch.Tensor or type(self) is torch.nn.Parameter      # (other subclasses keep torch's dispatch)
    if not plain or not self.is_cuda or self.dim() != 2:
        return orig(self, idx)
    if type(idx) is list:
        s = _lookup(idx)
        if s is not None and s.dev.device == self.device:
            hits["gather_list"] += 1
            return torch.index_select(self, 0, s.dev)
        return orig(self, idx)
    # a device index is taken only when it IS one of the tensors this module handed out (a batch stream or its sorted unique
    # ids: non-negative by construction) -- advanced indexing wraps negative indices, index_select does not, and whether an
    # arbitrary device tensor holds one cannot be asked without a synchronisation
    if type(idx) is torch.Tensor and _state["safe_idx"].get(id(idx)) is idx and idx.device == self.device:
        hits["gather_index"] += 1
        return torch.index_select(self, 0, idx)
    return orig(self, idx)


def _unique(input, *args, **kwargs):
    """``torch.unique`` of a CPU tensor that holds one of the current batch's streams -> the sampler's sorted unique ids"""
    orig = _state["orig_unique"]
    if not args and not kwargs and type(input) is torch.Tensor and input.device.type == "cpu" and input.dim() == 1 \
            and input.dtype == torch.int64 and _state["batch"]:
        n = input.numel()
        for k, s in enumerate(_state["batch"].values()):
            # (ids above 2^24 do not survive the reference's torch.Tensor(list) float round trip: those keep torch's path,
            # so that this module never computes anything else than the model file's own expression would)
            if k < 2 and s.n == n and n > 0 and input.dtype == torch.int64 and s.max_id < (1 << 24) \
                    and s.still(s.lst) and torch.equal(input, s.host_t):      # (every id compared: 2048 int64s, ~5 us)
                hits["unique"] += 1
                return _unique_of(s)
    return orig(input, *args, **kwargs)


class Adam(torch.optim.Adam):
    """``torch.optim.Adam`` whose step on fp32 HIP parameters is ONE ``srh_adam_step`` launch per parameter (betas, eps,
    lr as given; bias-corrected; the state keys are torch's: ``step``, ``exp_avg``, ``exp_avg_sq``).  Groups with weight
    decay, amsgrad, maximize, a tensor lr, or parameters off the device / of another dtype / with sparse gradients take
    torch's own step."""

    def _fusable(self, group):
        if group.get("weight_decay", 0) or group.get("amsgrad") or group.get("maximize") or group.get("capturable") \
                or group.get("differentiable") or isinstance(group["lr"], torch.Tensor):
            return False
        for p in group["params"]:
            if p.grad is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse
                    and p.numel() % 4 == 0):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if not all(self._fusable(g) for g in self.param_groups):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                hits["adam"] += 1
                ops.adam_step(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], step=int(st["step"]),
                              lr=float(group["lr"]), beta1=float(b1), beta2=float(b2), eps=float(group["eps"]))
        return loss


def install():
    """Switch the fast paths on (idempotent).  ``torch.Tensor.__getitem__``, ``torch.unique`` and
    ``torch.optim.Adam`` are wrapped process-wide until ``uninstall()``; every wrapper falls through to torch's own
    implementation for anything it does not recognise."""
    if _state["on"]:
        return
    _state["orig_getitem"] = torch.Tensor.__getitem__
    _state["own_getitem"] = "__getitem__" in torch.Tensor.__dict__
    _state["orig_unique"] = torch.unique
    _state["orig_adam"] = torch.optim.Adam
    torch.Tensor.__getitem__ = _getitem
    torch.unique = _unique
    torch.optim.Adam = Adam
    from ..data import loader
    loader.LAZY_GRAPH_FILES[0] = True
    _state["on"] = True


def uninstall():
    if not _state["on"]:
        return
    if _state.get("own_getitem"):
        torch.Tensor.__getitem__ = _state["orig_getitem"]
    else:
        del torch.Tensor.__getitem__                         # (back to the inherited slot of torch._C.TensorBase)
    torch.unique = _state["orig_unique"]
    torch.optim.Adam = _state["orig_adam"]
    from ..data import loader
    loader.LAZY_GRAPH_FILES[0] = False
    _state.update(on=False, batch={}, pins=[], pin_at=0, want_unique=False, safe_idx={})
