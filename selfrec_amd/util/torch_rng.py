"""Draws of torch's default CPU generator, replayed by the C++ host code (csrc/sampler.cpp, srh_mt19937_uniform_f32).

model/graph/BUIR.py:118-121 of the reference draws ``torch.rand(nnz)`` on the HOST for every forward pass of both
encoders (5 M uniforms per step on the Yelp2018 shape); ATen's CPU kernel is serial at ~15 ns per draw, 75 ms per
step -- 25x everything else in that step.  The stream is fixed by ATen's CPUGeneratorImpl: mt19937, one 32-bit word
per float32, ``(word & 0xFFFFFF) * 2**-24``.  These functions read the generator's words out of
``torch.get_rng_state()``, let the library produce the same numbers (or the keep mask computed from them), and put the
advanced state back with ``torch.set_rng_state``: the values AND the generator afterwards are what ``torch.rand`` would
have left (tests/test_host_logic.py compares both bit for bit).  This is host code on both sides -- no device involved.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib

_TORCH_RAND = torch.rand
# CPUGeneratorImplState (ATen/CPUGeneratorImpl.cpp): u64 seed | i32 left | i32 seeded | u64 next | u64 state[624] | ...
_STATE_BYTES, _LEFT, _SEEDED, _NEXT, _WORDS, _N = 5056, 8, 12, 16, 24, 624


def _read_state():
    raw = torch.get_rng_state().numpy()
    if raw.size != _STATE_BYTES:
        return None
    left = int(raw[_LEFT:_LEFT + 4].view(np.int32)[0])
    seeded = int(raw[_SEEDED:_SEEDED + 4].view(np.int32)[0])
    nxt = int(raw[_NEXT:_NEXT + 8].view(np.uint64)[0])
    # ATen's engine: `if (--left == 0) next_state();  y = state[next++]` -- left == 625 - next between draws, and
    # left == 1 (next == 0) on a freshly seeded generator: regenerate before the first word
    if seeded != 1 or not (left == _N + 1 - nxt or (left == 1 and nxt == 0)) or not 0 <= nxt <= _N:
        return None
    words = raw[_WORDS:_WORDS + 8 * _N].view(np.uint64).astype(np.uint32)
    return raw, words, (_N if left == 1 else nxt)


def _write_state(raw, words, pos):
    raw = raw.copy()
    raw[_WORDS:_WORDS + 8 * _N].view(np.uint64)[:] = words
    raw[_NEXT:_NEXT + 8].view(np.uint64)[0] = pos
    raw[_LEFT:_LEFT + 4].view(np.int32)[0] = _N + 1 - pos
    torch.set_rng_state(torch.from_numpy(raw))


def _draw(n, want_values, addend):
    st = None if torch.rand is not _TORCH_RAND else _read_state()      # (a caller who patched torch.rand is obeyed)
    if st is None or n == 0:
        u = torch.rand(n)
        return u if want_values else torch.floor(addend + u).type(torch.bool)
    raw, words, pos = st
    out = np.empty(n, dtype=np.float32) if want_values else None
    keep = None if want_values else np.empty(n, dtype=np.uint8)
    cpos = C.c_int32(pos)
    _lib.check(_lib.load().srh_mt19937_uniform_f32(
        words.ctypes.data_as(C.c_void_p), C.byref(cpos), n,
        None if out is None else out.ctypes.data_as(C.c_void_p), float(np.float32(addend)),
        None if keep is None else keep.ctypes.data_as(C.c_void_p)), "srh_mt19937_uniform_f32")
    _write_state(raw, words, cpos.value)
    return torch.from_numpy(out) if want_values else torch.from_numpy(keep).view(torch.bool)


def rand(n: int) -> torch.Tensor:
    """``torch.rand(n)`` (float32, default CPU generator)."""
    return _draw(int(n), True, 0.0)


def keep_mask(n: int, keep_prob: float) -> torch.Tensor:
    """``torch.floor(keep_prob + torch.rand(n)).type(torch.bool)`` without materialising the uniforms."""
    return _draw(int(n), False, keep_prob)
