"""ctypes binding of libselfrec_hip.so (the C ABI declared in include/selfrec_hip.h).

This is the only place the shared library is loaded.  There is no fallback: if the
library is missing or a call fails, ``SelfrecHipError`` is raised.  ``import torch`` comes
first on purpose -- torch loads the ROCm runtime (libamdhip64.so.7) and the library's own
DT_NEEDED entry then resolves to that same copy, so device pointers and streams are shared.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libselfrec_hip.so")
ABI_VERSION = 30

SRH_EPI_PERTURB, SRH_EPI_MEAN, SRH_EPI_AXPY, SRH_EPI_ADAM = 1, 2, 4, 8
SRH_MAX_ADAM_CLEAR = 4
SRH_SCALE_IN, SRH_SCALE_OUT = 1, 2
SRH_MAX_PREV, SRH_MAX_ADD, SRH_MAX_EXTRA = 8, 2, 2


class SelfrecHipError(RuntimeError):
    pass


class BatchFetchArgs(C.Structure):
    """srh_batch_fetch_args_t (include/selfrec_hip.h)."""
    _fields_ = [("d_epoch_u", C.c_void_p), ("d_epoch_i", C.c_void_p), ("d_epoch_j", C.c_void_p),
                ("d_epoch_uniq_u", C.c_void_p), ("d_epoch_uniq_i", C.c_void_p), ("d_n_uniq_u", C.c_void_p),
                ("d_n_uniq_i", C.c_void_p), ("n_edges", C.c_int64), ("batch_size", C.c_int64), ("d_cursor", C.c_void_p),
                ("d_stage_u", C.c_void_p), ("d_stage_i", C.c_void_p), ("d_stage_j", C.c_void_p),
                ("d_stage_uniq_u", C.c_void_p), ("d_stage_uniq_i", C.c_void_p), ("d_meta", C.c_void_p),
                ("d_row_mark", C.c_void_p), ("mark_item_offset", C.c_int32), ("cat_item_offset", C.c_int32),
                ("d_zero4", C.c_void_p), ("d_stage_cat", C.c_void_p), ("d_n_cat", C.c_void_p), ("d_now", C.c_void_p),
                ("half_batches", C.c_int64), ("d_adam_coef", C.c_void_p), ("adam_lr", C.c_float),
                ("adam_beta1", C.c_float), ("adam_beta2", C.c_float)]


class InfonceProblem(C.Structure):
    """struct srh_infonce_problem (include/selfrec_hip.h)."""
    _fields_ = [("d_v1", C.c_void_p), ("d_v2", C.c_void_p), ("d_idx", C.c_void_p), ("n", C.c_int64),
                ("d_n", C.c_void_p), ("d_g1", C.c_void_p), ("d_g2", C.c_void_p), ("g2_exclusive", C.c_int32)]


class L2Block(C.Structure):
    """struct srh_l2_block (include/selfrec_hip.h)."""
    _fields_ = [("d_x", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("d_gx", C.c_void_p)]


SCALAR_WS_BYTES = 64         # SRH_SCALAR_WS_BYTES


class BatchSegments(C.Structure):
    """struct srh_batch_segments (include/selfrec_hip.h)."""
    _fields_ = [("d_n_uniq_u", C.c_void_p), ("d_n_uniq_i", C.c_void_p), ("d_n_uniq_n", C.c_void_p),
                ("d_seg_rows", C.c_void_p), ("d_seg_end", C.c_void_p), ("d_seg", C.c_void_p), ("d_seg_a", C.c_void_p),
                ("d_seg_b", C.c_void_p), ("d_batch_no", C.c_void_p), ("nce_rows", C.c_int32), ("rows_are_zero", C.c_int32)]


class BprProblem(C.Structure):
    """struct srh_bpr_problem (include/selfrec_hip.h)."""
    _fields_ = [("d_user", C.c_void_p), ("d_item", C.c_void_p), ("d_reg_user", C.c_void_p), ("d_reg_item", C.c_void_p),
                ("d_u_idx", C.c_void_p), ("d_i_idx", C.c_void_p), ("d_j_idx", C.c_void_p), ("B", C.c_int64),
                ("d_n_rows", C.c_void_p), ("reg_coef", C.c_float), ("reg_include_neg", C.c_int32),
                ("loss_scale", C.c_float), ("d_g_user", C.c_void_p), ("d_g_item", C.c_void_p),
                ("d_greg_user", C.c_void_p), ("d_greg_item", C.c_void_p), ("d_losses", C.c_void_p), ("d_ws", C.c_void_p),
                ("seg", C.POINTER(BatchSegments))]


class SpmmEpilogue(C.Structure):
    """struct srh_spmm_epilogue (include/selfrec_hip.h)."""
    _fields_ = [
        ("flags", C.c_int32), ("eps", C.c_float), ("d_noise", C.c_void_p),
        ("rng_seed", C.c_uint64), ("rng_offset", C.c_uint64),
        ("d_rng_step", C.c_void_p), ("rng_stride", C.c_uint64),
        ("n_prev", C.c_int32), ("n_add", C.c_int32),
        ("d_prev", C.c_void_p * SRH_MAX_PREV),
        ("mean_div", C.c_float), ("alpha", C.c_float),
        ("d_mean_out", C.c_void_p),
        ("d_add", C.c_void_p * SRH_MAX_ADD),
        ("add_scale", C.c_float * SRH_MAX_ADD),
        ("d_row_mark", C.c_void_p), ("d_col_mark", C.c_void_p), ("d_mark_stamp", C.c_void_p),
        ("d_add_mark", C.c_void_p), ("add_sparse_mask", C.c_int32),
        ("n_extra", C.c_int32), ("main_clean", C.c_int32), ("d_extra_out", C.c_void_p * SRH_MAX_EXTRA),
        ("d_extra_noise", C.c_void_p * SRH_MAX_EXTRA), ("extra_rng_offset", C.c_uint64 * SRH_MAX_EXTRA),
        ("noise_d_full", C.c_int32), ("noise_col0", C.c_int32),
        ("d_row_scale", C.c_void_p), ("scale_flags", C.c_int32), ("prev_unscale_mask", C.c_int32),
        ("add_rowscale_mask", C.c_int32), ("noise_d_valid", C.c_int32),
        ("d_adam_param", C.c_void_p), ("d_adam_m", C.c_void_p), ("d_adam_v", C.c_void_p), ("d_adam_coef", C.c_void_p),
        ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float), ("adam_n_clear", C.c_int32),
        ("d_adam_clear_mark", C.c_void_p), ("d_adam_clear", C.c_void_p * SRH_MAX_ADAM_CLEAR), ("d_adam_cursor", C.c_void_p),
    ]


SRH_MAX_EXCHANGE = 8


class BatchLists(C.Structure):
    """struct srh_batch_lists (include/selfrec_hip.h)."""
    _fields_ = [("d_idx", C.c_void_p * 5), ("d_count", C.c_void_p * 5), ("B", C.c_int32)]


_vp, _i32, _i64, _f32, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); every symbol include/selfrec_hip.h declares
SIGNATURES = {
    "srh_abi_version": (_i32, []),
    "srh_last_error_string": (C.c_char_p, []),
    "srh_device_count": (_i32, []),
    "srh_topk_trim_mark_ties": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "srh_find_k_largest_host": (_i32, [_i64, _vp, _i64, _vp, _vp, _vp]),
    "srh_sampler_create": (_i32, [C.POINTER(_vp), _i64, _i64, _i64, _vp, _vp]),
    "srh_sampler_destroy": (None, [_vp]),
    "srh_sampler_set_state": (_i32, [_vp, _vp, _i32]),
    "srh_sampler_get_state": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "srh_sampler_seed": (_i32, [_vp, _u64]),
    "srh_sampler_shuffle": (_i32, [_vp]),
    "srh_sampler_get_order": (_i32, [_vp, _vp]),
    "srh_sampler_next_batch": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, C.POINTER(_i64)]),
    "srh_sampler_epoch": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srh_sampler_epoch_segments": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srh_sampler_sample_range": (_i32, [_vp, _i64, _i64, _vp]),
    "srh_sampler_next_u32": (_i32, [_vp, C.POINTER(C.c_uint32)]),
    "srh_mt19937_uniform_f32": (_i32, [_vp, C.POINTER(C.c_int32), _i64, _vp, C.c_float, _vp]),
    "srh_adj_sym_normalize": (_i32, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp]),
    "srh_spmm_plan_create": (_i32, [C.POINTER(_vp), _i64, _i64, _vp, _i32, _i64, _vp]),
    "srh_spmm_plan_set_xcd_shares": (_i32, [_vp, _i32, _vp]),
    "srh_spmm_plan_run_tasks": (_i32, [_vp, _i32]),
    "srh_spmm_f32_probe": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "srh_spmm_gather_bound": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "srh_gather_floor_probe": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
    "srh_spmm_plan_destroy": (None, [_vp]),
    "srh_spmm3_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "srh_spmm_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, C.POINTER(SpmmEpilogue), _vp]),
    "srh_bpr_ws_bytes": (_i64, [_i64]),
    "srh_bpr_l2_fwd_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _f32, _i32, _f32,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srh_bpr_l2_fwd_bwd_p": (_i32, [C.POINTER(BprProblem), _i32, _vp]),
    "srh_bpr_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "srh_bpr_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "srh_l2_reg_fwd": (_i32, [C.POINTER(L2Block), _i32, _f32, _vp, _vp, _vp, _vp]),
    "srh_l2_reg_bwd": (_i32, [C.POINTER(L2Block), _i32, _f32, _vp, _vp, _vp]),
    "srh_infonce_ws_bytes": (_i64, [_i64, _i32]),
    "srh_infonce_set_precision": (_i32, [_i32]),
    "srh_infonce_get_precision": (_i32, []),
    "srh_infonce_fwd_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "srh_infonce_fwd_bwd_multi": (_i32, [C.POINTER(InfonceProblem), _i32, _i32, _f32, _f32, _vp, _vp, _i32, _vp]),
    "srh_bpr_infonce_fwd_bwd": (_i32, [C.POINTER(BprProblem), C.POINTER(InfonceProblem), _i32, _i32, _f32, _f32, _vp, _vp,
                                       _i32, _vp]),
    "srh_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _f32, _f32, _f32, _f32, _vp]),
    "srh_adam_step_reset": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _f32, _f32, _f32, _f32, _vp, _i32, _vp, _vp, _vp]),
    "srh_score_mask_topk": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp]),
    "srh_score_mask_topk_filtered_ws_bytes": (_i64, [_i64, _i64, _i32, _i32, _i64, _i32]),
    "srh_score_mask_topk_filtered": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp,
                                            _vp, _vp, _vp]),
    "srh_gemm_nt_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "srh_topk_rows": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "srh_topk_hit_flags": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "srh_metric_rows": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "srh_axpby": (_i32, [_f32, _vp, _f32, _vp, _i64, _vp]),
    "srh_batch_fetch": (_i32, [_vp, _vp]),
    "srh_spmm_f32_with_fetch": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "srh_zero_rows": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "srh_cursor_advance": (_i32, [_vp, _vp]),
    "srh_batch_pack": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "srh_batch_unpack": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "srh_batch_scatter": (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "srh_comm_unique_id": (_i32, [_vp]),
    "srh_comm_init_rank": (_i32, [C.POINTER(_vp), _i32, _i32, _vp]),
    "srh_comm_destroy": (_i32, [_vp]),
    "srh_comm_world": (_i32, [_vp, C.POINTER(_i32)]),
    "srh_allgather_rows": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "srh_reducescatter_rows": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "srh_allreduce_sum_f32": (_i32, [_vp, _i64, _vp, _vp]),
    "srh_dataset_load": (_i32, [C.POINTER(_vp), C.c_char_p, C.c_char_p]),
    "srh_dataset_destroy": (None, [_vp]),
    "srh_dataset_sizes": (_i32, [_vp, _vp]),
    "srh_dataset_copy_ids": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "srh_dataset_names_bytes": (_i64, [_vp, _i32]),
    "srh_dataset_copy_names": (_i32, [_vp, _i32, _vp, _vp]),
}

_lib = None


def load():
    """Load the library once; raise SelfrecHipError if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SelfrecHipError(
            f"{LIB_PATH} not found. Build it with `make -C selfrec_amd/csrc` or "
            f"`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SelfrecHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SelfrecHipError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.srh_abi_version()
    if got != ABI_VERSION:
        raise SelfrecHipError(f"{LIB_PATH} has ABI version {got}, binding expects {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().srh_last_error_string()
        raise SelfrecHipError(f"{what or 'libselfrec_hip'} failed ({status}): {msg.decode() if msg else '?'}")


def require_gpu() -> None:
    """The product path needs a HIP device; say so instead of computing anything elsewhere."""
    if load().srh_device_count() < 1 or not torch.cuda.is_available():
        raise SelfrecHipError("no HIP device visible: selfrec_amd's compute path is MI355X-only (no CPU fallback)")
