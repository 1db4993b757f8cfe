"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/selfrec_hip.h
declares, and its host entry points (the sampler) reproduce the reference's streams."""
import os
import random
import re

import numpy as np
import pytest

from selfrec_amd import _lib, ops

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "selfrec_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libselfrec_hip.so does not export {n}"
        assert n in _lib.SIGNATURES, f"ctypes binding lacks {n}"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.srh_abi_version() == _lib.ABI_VERSION


def test_error_reporting_does_not_throw():
    lib = _lib.load()
    rc = lib.srh_sampler_shuffle(None)
    assert rc == -1 and b"null handle" in lib.srh_last_error_string()
    with pytest.raises(_lib.SelfrecHipError):
        _lib.check(rc, "srh_sampler_shuffle")


def test_device_ops_refuse_cpu_tensors():
    import torch
    with pytest.raises(ops.SelfrecHipError):
        ops.axpby(1.0, torch.zeros(8), 0.0, torch.zeros(8))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_sampler_matches_reference_stream(golden_ops, tag):
    g = golden_ops
    bs, negs, seed = (int(x) for x in g[f"sampler_{tag}_meta"])
    s = ops.Sampler(g["graph_train_u_ids"], g["graph_train_i_ids"], 200, 300)
    s.seed(seed)
    us, is_, js, uq = [], [], [], []
    for _ in range(2):
        r = s.epoch(bs, negs, with_unique=True)
        us.append(r["u"]); is_.append(r["i"]); js.append(r["j"]); uq.append(r)
    assert np.array_equal(np.concatenate(us), g[f"sampler_{tag}_u"])
    assert np.array_equal(np.concatenate(is_), g[f"sampler_{tag}_i"])
    assert np.array_equal(np.concatenate(js), g[f"sampler_{tag}_j"])
    assert s.next_u32() == int(g[f"sampler_{tag}_next_u32"][0])
    order = s.order()
    assert np.array_equal(g["graph_train_u_ids"][order], g[f"sampler_{tag}_final_order_u"])
    # unique ids == torch.unique of each batch
    r = uq[-1]
    for b in range(r["n_batches"]):
        lo, hi = b * bs, min((b + 1) * bs, s.n_edges)
        assert np.array_equal(r["uniq_u"][b * bs:b * bs + r["n_uniq_u"][b]], np.unique(r["u"][lo:hi]))
        assert np.array_equal(r["uniq_i"][b * bs:b * bs + r["n_uniq_i"][b]], np.unique(r["i"][lo:hi]))


def test_sampler_per_batch_and_python_state_roundtrip(golden_ops):
    g = golden_ops
    bs, negs, seed = (int(x) for x in g["sampler_b_meta"])
    random.seed(seed)
    s = ops.Sampler(g["graph_train_u_ids"], g["graph_train_i_ids"], 200, 300)
    s.set_state_from_python()
    s.shuffle()
    us, js, ptr = [], [], 0
    while ptr < s.n_edges:
        u, _, j = s.next_batch(ptr, bs, negs)
        ptr += len(u); us.append(u); js.append(j)
    n = s.n_edges
    assert np.array_equal(np.concatenate(us), g["sampler_b_u"][:n])
    assert np.array_equal(np.concatenate(js), g["sampler_b_j"][:n * negs])
    s.push_state_to_python()
    # the python generator continues exactly where the reference's would
    s2 = ops.Sampler(g["graph_train_u_ids"], g["graph_train_i_ids"], 200, 300)
    random.seed(seed)
    s2.set_state_from_python(); s2.epoch(bs, negs)
    assert random.getstate() != s2.python_state()  # reference state unchanged until pushed
    s2.push_state_to_python()
    assert random.getstate()[1] == s.python_state()[1]


def test_sampler_edge_cases():
    s = ops.Sampler([0, 0, 1], [0, 1, 2], 2, 4)
    with pytest.raises(ops.SelfrecHipError):       # unseeded generator must not silently run
        s.shuffle()
    s.seed(0)
    r = s.epoch(2, 2)
    assert r["n_batches"] == 2 and len(r["j"]) == 6
    assert all(j not in ({0, 1} if u == 0 else {2}) for u, j in zip(np.repeat(r["u"], 2), r["j"]))
    with pytest.raises(ops.SelfrecHipError):
        ops.Sampler([0, 5], [0, 1], 2, 4)          # user id out of range
    # random.sample replay: pool path (k large) and set path (k small)
    for n, k, seed in ((1000, 900, 3), (100000, 50, 5), (7, 7, 1), (10, 0, 2)):
        random.seed(seed)
        ref = random.sample(range(n), k)
        s.seed(seed)
        assert s.sample_range(n, k).tolist() == ref
    with pytest.raises(ops.SelfrecHipError):
        s.sample_range(5, 6)


def test_seed_matches_cpython_for_wide_seeds():
    s = ops.Sampler([0], [0], 1, 2)
    for seed in (0, 1, 2**31, 2**32 + 17, 2**63 - 1):
        random.seed(seed)
        want = [random.getrandbits(32) for _ in range(3)]
        s.seed(seed)
        assert [s.next_u32() for _ in range(3)] == want


def test_column_class_order_by_another_bit_of_the_column_id():
    """`bit` picks which bit of the column id defines the two classes (0 = parity, the engine's): still a stable
    partition of the long rows, short rows untouched, and row_mid counts the class-0 entries."""
    import numpy as np
    from selfrec_amd import ops
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 50, 150)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(4096, size=n, replace=False)) for n in lens]).astype(np.int32)
    for bit in (1, 3, 7):
        perm, row_mid = ops.column_class_order(indptr, indices, 16, bit=bit)
        assert sorted(perm.tolist()) == list(range(indices.size))
        new = indices[perm]
        for r, n in enumerate(lens):
            old_row, new_row = indices[indptr[r]:indptr[r + 1]], new[indptr[r]:indptr[r + 1]]
            cls = (old_row >> bit) & 1
            if n >= 16:
                assert row_mid[r] == int((cls == 0).sum())
                assert np.array_equal(new_row, np.concatenate([old_row[cls == 0], old_row[cls == 1]]))
            else:
                assert np.array_equal(new_row, old_row) and row_mid[r] == -1 - int((cls == 1).sum() > (cls == 0).sum())


def test_column_class_order_is_a_stable_partition_of_long_rows():
    """ops.column_class_order (host side of srh_spmm_plan_create's h_row_mid): a permutation that leaves
    short rows alone and stores long rows [even columns | odd columns], each part in its old order."""
    import numpy as np
    from selfrec_amd import ops
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 40, 200)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(5000, size=n, replace=False)) for n in lens]).astype(np.int32)
    perm, row_mid = ops.column_class_order(indptr, indices, 12)
    assert sorted(perm.tolist()) == list(range(indices.size)) and row_mid.dtype == np.int32
    new = indices[perm]
    for r, n in enumerate(lens):
        old_row, new_row = indices[indptr[r]:indptr[r + 1]], new[indptr[r]:indptr[r + 1]]
        if n >= 12:
            even, odd = old_row[old_row % 2 == 0], old_row[old_row % 2 == 1]
            assert row_mid[r] == even.size and np.array_equal(new_row, np.concatenate([even, odd]))
        else:
            assert np.array_equal(new_row, old_row)
            assert row_mid[r] == (-2 if (old_row % 2 == 1).sum() > (old_row % 2 == 0).sum() else -1)


def test_exchange_entry_points_validate_their_arguments_before_any_launch():
    """srh_batch_pack / unpack / scatter (column-sharded layout) reject bad arguments with a message instead of
    launching: checked here without a GPU (nothing below reaches a kernel)."""
    import ctypes as C
    lib = _lib.load()
    assert lib.srh_batch_pack(None, 1, None, 8, None, None, None, None) == -1
    assert b"null lists" in lib.srh_last_error_string()
    bl = _lib.BatchLists()
    bl.B = 0
    assert lib.srh_batch_pack(C.byref(bl), 1, None, 8, None, None, None, None) == -1
    assert b"bad batch size" in lib.srh_last_error_string()
    bl.B = 16
    assert lib.srh_batch_pack(C.byref(bl), 1, None, 8, None, None, None, None) == -1
    assert b"null index list" in lib.srh_last_error_string()
    fake = (C.c_int32 * 16)()
    for s in range(5):
        bl.d_idx[s] = C.cast(fake, C.c_void_p)
    tabs = (C.c_void_p * 1)(C.cast(fake, C.c_void_p))
    assert lib.srh_batch_pack(C.byref(bl), 1, tabs, 6, C.cast(fake, C.c_void_p), None, None, None) == -1
    assert b"multiple of 4" in lib.srh_last_error_string()
    assert lib.srh_batch_pack(C.byref(bl), _lib.SRH_MAX_EXCHANGE + 1, tabs, 8, C.cast(fake, C.c_void_p), None, None, None) == -1
    assert lib.srh_batch_unpack(C.byref(bl), 5, 2, 8, C.cast(fake, C.c_void_p), tabs, 4, tabs, None) == -1
    assert b"tables + gradient tables" in lib.srh_last_error_string()
    assert lib.srh_batch_scatter(C.byref(bl), 1, tabs, tabs, 64, 60, 8, None) == -1
    assert b"outside 64 columns" in lib.srh_last_error_string()
    # the epilogue's column-slice fields are validated the same way
    ep = _lib.SpmmEpilogue()
    ep.noise_d_full, ep.noise_col0 = 64, 4
    plan = C.c_void_p()
    indptr = np.array([0, 1, 2], dtype=np.int32)
    if lib.srh_device_count() < 1:
        return                     # (creating a plan allocates device memory)
    _lib.check(lib.srh_spmm_plan_create(C.byref(plan), 2, 2, indptr.ctypes.data_as(C.c_void_p), 0, 0, None))
    assert lib.srh_spmm_f32(plan, None, C.cast(fake, C.c_void_p), C.cast(fake, C.c_void_p), C.cast(fake, C.c_void_p),
                            C.cast((C.c_int32 * 16)(), C.c_void_p), 8, C.byref(ep), None) == -1
    assert b"column slice" in lib.srh_last_error_string()
    lib.srh_spmm_plan_destroy(plan)


def test_integration_md_structs_match_the_binding():
    """INTEGRATION.md spells the ctypes structures a maintainer would paste into the reference tree: they must be the
    layouts of selfrec_amd/_lib.py (which the GPU tests exercise against include/selfrec_hip.h) -- a field dropped from the
    document (round 3 found `g2_exclusive` missing) makes an array of problems read garbage."""
    import ctypes as C
    import re
    from selfrec_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    found = {}
    for b in blocks:
        for m in re.finditer(r"^class (\w+)\(C\.Structure\):.*?\n((?:    .*\n|\n)+)", b, flags=re.M):
            ns = {"C": C}
            exec(m.group(0), ns)                                  # (class statements only: ctypes field lists)
            found[m.group(1)] = ns[m.group(1)]
    pairs = {"BprProblem": _lib.BprProblem, "InfonceProblem": _lib.InfonceProblem, "BatchLists": _lib.BatchLists,
             "BatchSegments": _lib.BatchSegments}
    assert set(pairs) <= set(found), sorted(found)
    for name, ref in pairs.items():
        doc = found[name]
        assert C.sizeof(doc) == C.sizeof(ref), name
        assert [(n, getattr(doc, n).offset) for n, _ in doc._fields_] == [(n, getattr(ref, n).offset) for n, _ in ref._fields_], name


def test_integration_md_names_only_declared_entry_points():
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "selfrec_hip.h")).read()
    declared = set(re.findall(r"\b(srh_\w+)\s*\(", header))
    named = set(re.findall(r"\b_lib\.(srh_\w+)\(", text)) | set(re.findall(r"`(srh_\w+)`", text))
    named = {n for n in named if not n.endswith("_") and not n.endswith("_t")}     # (`srh_sampler_*` prefixes, type names)
    assert named and named <= declared, sorted(named - declared)


def test_ctypes_structures_have_the_layout_gcc_gives_the_header(tmp_path):
    """Every structure that crosses the boundary by pointer: size and the offset of every field as a C compiler lays out
    include/selfrec_hip.h, against the ctypes mirror in selfrec_amd/_lib.py -- same field names, same order.  (A field added
    to one side only shifts everything behind it: the library would read pointers out of floats.)"""
    import ctypes as C
    import shutil
    import subprocess
    from selfrec_amd import _lib
    gcc = shutil.which("gcc") or shutil.which("cc")
    if gcc is None:
        pytest.skip("no C compiler")
    pairs = {"srh_batch_fetch_args_t": _lib.BatchFetchArgs, "srh_infonce_problem_t": _lib.InfonceProblem,
             "srh_l2_block_t": _lib.L2Block, "srh_bpr_problem_t": _lib.BprProblem, "srh_spmm_epilogue_t": _lib.SpmmEpilogue,
             "srh_batch_lists_t": _lib.BatchLists, "srh_batch_segments_t": _lib.BatchSegments}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "selfrec_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf("{cname} {field} %zu\\n", offsetof({cname}, {field}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run([gcc, "-std=c99", "-I", inc, str(src), "-o", str(exe)], check=True)     # (the header is plain C)
    got = {}
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, field, value = line.split()
        got[(cname, field)] = int(value)
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for field, _ in cls._fields_:
            assert got[(cname, field)] == getattr(cls, field).offset, (cname, field)


def test_sampler_epoch_slots_hold_the_same_epochs_in_arrays_that_are_reused(golden_ops):
    """ops.Sampler.epoch(slot=k): the epochs a training loop draws into two alternating slots are the epochs of fresh arrays,
    and a slot's arrays are the SAME objects every time (nothing allocated or freed per epoch: engine.EpochPrefetcher)."""
    g = golden_ops
    fresh, slots = (ops.Sampler(g["graph_train_u_ids"], g["graph_train_i_ids"], 200, 300) for _ in range(2))
    fresh.seed(11); slots.seed(11)
    keys = ("u", "i", "j", "uniq_u", "uniq_i", "n_uniq_u", "n_uniq_i")
    held = {}
    for e in range(5):
        want = fresh.epoch(64, 1, with_unique=True)
        got = slots.epoch(64, 1, with_unique=True, slot=e & 1)
        assert got["n_batches"] == want["n_batches"]
        for k in keys:
            assert np.array_equal(got[k], want[k]), (e, k)
            assert held.setdefault((e & 1, k), got[k]) is got[k]          # the slot's own array, again
    assert held[(0, "u")] is not held[(1, "u")]
    assert slots.epoch(64, 1, with_unique=True)["u"] is not held[(0, "u")]          # slot=None: the caller's to keep
    assert slots.epoch(32, 1, with_unique=True, slot=0)["uniq_u"] is not held[(0, "uniq_u")]      # (another batch size: other arrays)


@pytest.mark.parametrize("bs", [64, 2048, 7])
def test_epoch_segments_are_the_row_slot_lists_of_every_batch(golden_ops, bs):
    """srh_sampler_epoch_segments: for every batch, the touched rows -- sorted unique users | sorted unique positive items |
    sorted unique negatives that are nobody's positive -- as table rows, and for each the (slot, role) entries that name it
    with the rows their terms read, slots ascending: the lists behind the fixed-order (atomic-free) batch-gradient reduction,
    against an independent numpy construction (tests/conftest.py).  Draws nothing from the generator."""
    from .conftest import host_batch_segments
    g = golden_ops
    U = 200
    plain, seg = (ops.Sampler(g["graph_train_u_ids"], g["graph_train_i_ids"], U, 300) for _ in range(2))
    plain.seed(7); seg.seed(7)
    want = plain.epoch(bs, 1, with_unique=True)
    r = seg.epoch(bs, 1, with_unique=True, with_segments=(0, U), slot=0)
    for k in ("u", "i", "j", "uniq_u", "uniq_i", "n_uniq_u", "n_uniq_i"):
        assert np.array_equal(r[k], want[k]), k
    assert plain.next_u32() == seg.next_u32()                        # the same amount of the stream was consumed
    for b in range(r["n_batches"]):
        lo, hi = b * bs, min((b + 1) * bs, seg.n_edges)
        ref = host_batch_segments(r["u"][lo:hi], r["i"][lo:hi], r["j"][lo:hi], bs, 0, U)
        n_groups = int(ref["n_uniq_u"][0] + ref["n_uniq_i"][0] + ref["n_uniq_n"][0])
        n_ent = 3 * (hi - lo)
        assert r["n_uniq_n"][b] == ref["n_uniq_n"][0]
        assert np.array_equal(r["seg_rows"][3 * b * bs:3 * (b + 1) * bs], ref["seg_rows"])          # (-1 past the last group)
        assert np.array_equal(r["seg_end"][3 * b * bs:3 * b * bs + n_groups], ref["seg_end"][:n_groups])
        for k in ("seg", "seg_a"):
            assert np.array_equal(r[k][3 * b * bs:3 * b * bs + n_ent], ref[k][:n_ent]), (b, k)
        assert np.array_equal(r["seg_b"][b * bs:b * bs + hi - lo], ref["seg_b"][:hi - lo])
        # the definition, spelled out for one batch: group g's entries are the slots that name its row, ascending
        if b == 0:
            u, i, j = r["u"][lo:hi], r["i"][lo:hi], r["j"][lo:hi]
            start = 0
            for gno in range(n_groups):
                row, end = ref["seg_rows"][gno], ref["seg_end"][gno]
                if gno < ref["n_uniq_u"][0]:
                    expect = [4 * s for s in np.nonzero(u == row)[0]]
                else:
                    expect = sorted([4 * s + 1 for s in np.nonzero(i + U == row)[0]] + [4 * s + 2 for s in np.nonzero(j + U == row)[0]])
                assert ref["seg"][start:end].tolist() == expect, gno
                start = end
    again = seg.epoch(bs, 1, with_unique=True, with_segments=(0, U), slot=0)
    assert again["seg"] is r["seg"] and again["seg_rows"] is r["seg_rows"]         # the slot's own arrays, refilled


def test_collective_entry_points_resolve_rccl_and_refuse_bad_arguments():
    """srh_comm_* / srh_allgather_rows / srh_reducescatter_rows / srh_allreduce_sum_f32 (SURVEY 8b's ABI list): RCCL is looked
    up at first use (here: the copy torch has loaded), a unique id can be made without a GPU, and null / empty arguments come
    back as error codes with a message -- nothing is launched, nothing throws."""
    import ctypes as C
    lib = _lib.load()
    uid = (C.c_uint8 * 128)()
    assert lib.srh_comm_unique_id(uid) == 0 and any(uid)
    other = (C.c_uint8 * 128)()
    assert lib.srh_comm_unique_id(other) == 0 and bytes(other) != bytes(uid)
    assert lib.srh_comm_unique_id(None) == -1 and b"null" in lib.srh_last_error_string()
    h = C.c_void_p()
    assert lib.srh_comm_init_rank(C.byref(h), 2, 5, uid) == -1 and b"rank 5 of 2" in lib.srh_last_error_string()
    fake = C.c_void_p(4096)
    assert lib.srh_allgather_rows(fake, fake, 4, 64, None, None) == -1 and b"null" in lib.srh_last_error_string()
    assert lib.srh_reducescatter_rows(fake, fake, 0, 64, fake, None) == -1 and b"bad shape" in lib.srh_last_error_string()
    assert lib.srh_allreduce_sum_f32(None, 8, fake, None) == -1
    assert lib.srh_comm_destroy(None) == 0
