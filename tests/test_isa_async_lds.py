"""The built library's own machine code (no GPU needed: llvm-objdump over the gfx950 code objects inside
selfrec_amd/lib/libselfrec_hip.so).  nce_tile_f32 issues its LDS reads as opaque instructions and settles them with a
hand-placed s_waitcnt (csrc/losses.hip: lds_read_f4_async / lds_wait); nothing but that wait orders a consumer behind the
read, so a compiler-made copy of a destination register between the two would copy stale bytes.  That happened once
(round 5: operands that crossed the loop edge unsettled; wrong sums in 1 of ~5000 weights).  This test reads the
instruction stream and fails if any instruction touches a register whose read may still be in flight."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import device_isa  # noqa: E402


def test_checker_sees_a_copy_of_a_pending_register():
    ok = ["ds_read_b128 v[4:7], v10", "v_mfma_f32_16x16x4_f32 v[20:23], v30, v31, v[20:23]", "s_waitcnt lgkmcnt(0)",
          "v_mfma_f32_16x16x4_f32 v[20:23], v4, v31, v[20:23]"]
    assert device_isa.async_lds_violations(ok) == []
    copied = ["ds_read_b128 v[4:7], v10", "v_mov_b64_e32 v[62:63], v[6:7]", "s_waitcnt lgkmcnt(0)"]
    assert [k for k, _, _ in device_isa.async_lds_violations(copied)] == [1]
    partial = ["ds_read_b128 v[4:7], v10", "s_waitcnt vmcnt(2)", "v_add_f32_e32 v1, v5, v2", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    assert [k for k, _, _ in device_isa.async_lds_violations(partial)] == [2]
    clobber = ["ds_read_b128 v[4:7], v10", "v_mov_b32_e32 v7, 0", "s_waitcnt lgkmcnt(0)"]
    assert [k for k, _, _ in device_isa.async_lds_violations(clobber)] == [1]


@pytest.mark.skipif(not os.path.exists(device_isa.LIB), reason="library not built")
def test_f32_infonce_passes_never_touch_an_unsettled_lds_read():
    ks = device_isa.kernels(name_filter="nce_tile_f32")
    assert len(ks) == 8, sorted(ks)                                   # d = 64 / 128 x pass 1 / 2 x weights kept / recomputed
    for name, body in ks.items():
        assert sum(i.startswith("ds_read_b128") for i in body) >= 16, name
        assert sum(i.startswith("v_mfma_f32_16x16x4_f32") for i in body) >= 64, name
        bad = device_isa.async_lds_violations(body)
        assert not bad, (name, bad[:5])
        # the ring's loads are buffer_load ... lds (not the FLAT-encoded global_load_lds, which turns every compiler-made
        # vmcnt into vmcnt(0) and drains the run-ahead), and no compiler-made vmcnt(0) sits inside the key loop
        assert any("buffer_load_dwordx4" in i and "lds" in i for i in body), name
        assert not any(i.startswith("global_load_lds") for i in body), name
