"""ctypes binding of libbmf_hip.so (the C ABI declared in include/bmf_hip.h).

The HIP library is the product: if it is missing or fails to load, importing this module raises -- there is
no CPU fallback anywhere in ``pybmf_amd``.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported BEFORE libbmf_hip.so is loaded: both need "libamdhip64.so.7", and the process must end up with
# ONE HIP runtime -- the one PyTorch ships and has initialised -- or our launches run in a runtime that sees no device
# (hipErrorNoDevice) and torch's stream handles mean nothing to it.
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", os.environ.get("BMF_LIB", "libbmf_hip.so"))  # BMF_LIB: experimental flavours

BMF_OK = 0
ROW_PAD = 512
PANEL_BF16, PANEL_F16, PANEL_I8 = 0, 1, 2
LINK_SIGMOID, LINK_KL = 1, 2
PALM_ELBMF, PALM_PRIMP = 1, 2
NORM_SPECTRAL, NORM_FROBENIUS = 0, 1
RED_PAD = 128
MAX_KP = 64
LOG_COLS = 16
MODE_PREPARE, MODE_PENALTY, MODE_WNMF = 0, 1, 2
(LOG_ITER, LOG_ERROR, LOG_REC, LOG_REG, LOG_REGERR, LOG_RMSE, LOG_MAE, LOG_TP, LOG_FP, LOG_FN, LOG_TN, LOG_VALID,
 LOG_STOP) = range(13)

_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class EpilogueArgs(C.Structure):
    """bmf_epilogue_args"""
    _fields_ = [
        ("F64", _vp), ("F", _vp), ("rows_pad", _i64), ("rows", _i32), ("k", _i32), ("kp", _i32),
        ("num", _vp), ("slab_stride", _i64), ("splits", _i32),
        ("G", _vp), ("reg", _f64), ("mode", _i32), ("thr", _f32), ("terms", _i32),
        ("panel", _vp), ("ldp", _i64), ("rowbits", _vp), ("colbits", _vp), ("ldcb", _i64),
        ("partials", _vp), ("stop", _vp), ("den", _vp), ("blockmax", _vp), ("num_block_stride", _i64),
        ("planes", _vp), ("plane_scale", _vp), ("limbs", _i32), ("_pad0", _i32),
    ]


class MaskedSide(C.Structure):
    """bmf_masked_side"""
    _fields_ = [("ptr", _vp), ("idx", _vp), ("val", _vp), ("wgt", _vp), ("seg_row", _vp), ("seg_beg", _vp), ("row_seg_ptr", _vp), ("part", _vp),
                ("rows", _i32), ("nseg", _i32)]


class MaskedLoop(C.Structure):
    """bmf_masked_loop"""
    _fields_ = [
        ("struct_bytes", _i32), ("m", _i32), ("n", _i32), ("k", _i32), ("kp", _i32), ("link", _i32), ("lamda", _f64),
        ("csr", MaskedSide), ("csc", MaskedSide), ("epiU", EpilogueArgs), ("epiV", EpilogueArgs),
        ("sums", _vp), ("Up64", _vp), ("Vp64", _vp),
        ("Xbits", _vp), ("x_m_pad", _i64), ("ldx", _i64), ("x_n_pad", _i64),
        ("Xreal", _vp), ("r_m_pad", _i64), ("r_n_pad", _i64),
        ("sums2", _vp), ("counts", _vp), ("nbU", _i32), ("nbV", _i32),
    ]


class LinkLoop(C.Structure):
    """bmf_link_loop"""
    _fields_ = [
        ("struct_bytes", _i32), ("m", _i32), ("n", _i32), ("k", _i32), ("kp", _i32), ("link", _i32), ("splitsU", _i32), ("splitsV", _i32),
        ("lamda", _f64), ("Xbits", _vp), ("XTbits", _vp), ("m_pad", _i64), ("n_pad", _i64), ("ldx", _i64), ("ldxt", _i64),
        ("wsU", _vp), ("wsV", _vp), ("numU", _vp), ("numV", _vp), ("denU_slabs", _vp), ("denV_slabs", _vp), ("colsum", _vp),
        ("epiU", EpilogueArgs), ("epiV", EpilogueArgs), ("Up64", _vp), ("Vp64", _vp), ("sums", _vp), ("Obits", _vp), ("counts", _vp),
        ("nbU", _i32), ("nbV", _i32),
    ]


class PalmArgs(C.Structure):
    """bmf_palm_args"""
    _fields_ = [
        ("F64", _vp), ("Fprev64", _vp), ("F", _vp), ("rows_pad", _i64), ("rows", _i32), ("k", _i32), ("kp", _i32),
        ("splits", _i32), ("num", _vp), ("slab_stride", _i64), ("G", _vp), ("norms", _vp),
        ("norm_kind", _i32), ("variant", _i32), ("beta", _f64), ("l1", _f64), ("l2", _f64), ("gap_l1", _f64), ("gap_l2", _f64),
        ("advance_prev", _i32), ("thr", _f32), ("rowbits", _vp), ("colbits", _vp), ("ldcb", _i64),
        ("partials", _vp), ("blockmax", _vp), ("stop", _vp), ("den", _vp), ("planes", _vp), ("ldp", _i64), ("plane_scale", _vp), ("dotpart", _vp),
    ]


class PalmState(C.Structure):
    """bmf_palm_state"""
    _fields_ = [
        ("struct_bytes", _i32), ("m", _i32), ("n", _i32), ("k", _i32), ("kp", _i32), ("variant", _i32), ("norm_kind", _i32),
        ("splits_xv", _i32), ("splits_xtu", _i32), ("gram_blocks", _i32), ("dot_blocks", _i32), ("log_rows", _i32),
        ("m_pad", _i64), ("n_pad", _i64), ("Xbits", _vp), ("ldx", _i64), ("Xtiled", _vp), ("XTtiled", _vp),
        ("U64", _vp), ("V64", _vp), ("Up64", _vp), ("Vp64", _vp), ("U", _vp), ("V", _vp), ("Upanel", _vp), ("Vpanel", _vp),
        ("scaleU", _vp), ("scaleV", _vp), ("wsU", _vp), ("wsV", _vp), ("Mslab", _vp), ("Nslab", _vp), ("gram_slabs", _vp),
        ("GU", _vp), ("GV", _vp), ("GU64", _vp), ("GV64", _vp), ("normsU", _vp), ("normsV", _vp), ("partU", _vp), ("partV", _vp),
        ("dotpart", _vp), ("ubits", _vp), ("vbits", _vp), ("ucolbits", _vp), ("vcolbits", _vp), ("counts", _vp), ("log", _vp),
        ("beta", _f64), ("thr_u", _f32), ("thr_v", _f32),
    ]


class WnmfRealState(C.Structure):
    """bmf_wnmf_real_state"""
    _fields_ = [
        ("struct_bytes", _i32), ("m", _i32), ("n", _i32), ("k", _i32), ("kp", _i32), ("with_mae", _i32),
        ("m_pad", _i64), ("n_pad", _i64), ("X", _vp), ("XT", _vp), ("U64", _vp), ("V64", _vp), ("U", _vp), ("V", _vp), ("UT", _vp), ("VT", _vp),
        ("Mslab", _vp), ("splits_xv", _i32), ("_pad0", _i32), ("Nslab", _vp), ("splits_xtu", _i32), ("_pad1", _i32),
        ("gram_slabs", _vp), ("gram_blocks", _i32), ("_pad2", _i32), ("GU", _vp), ("GV", _vp), ("GU64", _vp), ("GV64", _vp),
        ("partU", _vp), ("partV", _vp), ("rowbits", _vp), ("colbits", _vp), ("ldcb", _i64), ("sums", _vp), ("scal", _vp), ("log", _vp),
        ("log_rows", _i32), ("_pad3", _i32), ("stop", _vp), ("sum_x2", _f64), ("cells", _f64), ("tol", _f64), ("min_diff", _f64),
        ("Xtiled", _vp), ("XTtiled", _vp), ("Vrf", _vp), ("Urf", _vp), ("UT3", _vp), ("VT3", _vp),
    ]


class PenaltyState(C.Structure):
    """bmf_penalty_state"""
    _fields_ = [
        ("struct_bytes", _i32), ("m", _i32), ("n", _i32), ("k", _i32), ("kp", _i32), ("terms", _i32),
        ("mode", _i32), ("with_mae", _i32),
        ("m_pad", _i64), ("n_pad", _i64),
        ("Xbits", _vp), ("ldx", _i64), ("XTbits", _vp), ("ldxt", _i64),
        ("U64", _vp), ("V64", _vp), ("U", _vp), ("V", _vp), ("Upanel", _vp), ("Vpanel", _vp),
        ("Mslab", _vp), ("splits_xv", _i32), ("_pad0", _i32),
        ("Nslab", _vp), ("splits_xtu", _i32), ("_pad1", _i32),
        ("Nred", _vp),
        ("gram_slabs", _vp), ("gram_blocks", _i32), ("_pad2", _i32),
        ("GU", _vp), ("GV", _vp), ("comm", _vp), ("GV64", _vp), ("partU", _vp), ("partV", _vp), ("scal", _vp),
        ("ubits", _vp), ("ucolbits", _vp), ("lduc", _i64),
        ("vbits", _vp), ("vcolbits", _vp), ("ldvc", _i64),
        ("counts", _vp), ("log", _vp), ("log_rows", _i32), ("_pad3", _i32),
        ("stop", _vp),
        ("sum_x", _f64), ("cells", _f64), ("tol", _f64), ("min_diff", _f64),
        ("thr_u", _f32), ("thr_v", _f32),
        ("panel_kind", _i32), ("updates_only", _i32),
        ("scaleU", _vp), ("scaleV", _vp), ("panel_ws", _vp), ("mae_ws", _vp),
        ("Xtiled", _vp), ("XTtiled", _vp), ("nred_blocks", _i32), ("exchange_overlap", _i32),
    ]


# bmf_allreduce_fn: int fn(void* user, void* buf, int64_t count, int32_t dtype, void* stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _i64, _i32, _vp)
COMM_RCCL, COMM_HOST = 1, 2
DTYPE_F32, DTYPE_F64 = 0, 1
COMM_ID_BYTES = 128

# name -> (restype, argtypes); mirrors include/bmf_hip.h one to one (tests/test_abi.py checks the symbol list)
SIGNATURES = {
    "bmf_version": (C.c_int, []),
    "bmf_struct_bytes": (C.c_int, [C.c_int]),
    "bmf_last_error": (C.c_char_p, []),
    "bmf_panel_pos": (C.c_int, [C.c_int]),
    "bmf_pack_rows_u8": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "bmf_popcount": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "bmf_make_panel": (C.c_int, [_vp, _i64, _i64, C.c_int, C.c_int, _vp, _i64, _vp]),
    "bmf_make_panel_f16": (C.c_int, [_vp, _i64, _i64, C.c_int, _vp, _i64, _vp, _vp, _vp]),
    "bmf_xf_bits_f16": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, C.c_int, _vp, _i64, C.c_int, _vp]),
    "bmf_xf_bits_slots": (C.c_int, [_i64, _i64, C.c_int, C.c_int]),
    "bmf_xf_bits": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, C.c_int, C.c_int, _vp, _i64, C.c_int, _vp]),
    "bmf_panel_pos_i8": (C.c_int, [C.c_int]),
    "bmf_xf_bits_i8_slots": (C.c_int, [_i64, _i64, C.c_int]),
    "bmf_xf_bits_i8_occupancy": (C.c_int, [C.c_int]),
    "bmf_xf_bits_i8_variant": (C.c_int, [C.c_int]),
    "bmf_xf_bits_i8": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, C.c_int, _vp, C.c_int, _vp, _i64, C.c_int, C.c_int, _vp]),
    "bmf_tile_bits": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "bmf_s24_bytes": (_i64, [_i64, _i64]),
    "bmf_s24_count": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "bmf_s24_pack": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "bmf_xf_bits_i8s_slots": (C.c_int, [_i64, _i64, C.c_int]),
    "bmf_xf_bits_i8s_occupancy": (C.c_int, []),
    "bmf_xf_bits_i8s_form": (C.c_int, [C.c_int]),
    "bmf_xf_bits_i8s": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, C.c_int, _vp, _i64, C.c_int, _vp, _vp]),
    "bmf_s24_overflow": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, C.c_int, _vp, _vp]),
    "bmf_make_panel_i8": (C.c_int, [_vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp, _i64, _vp, _vp, _vp]),
    "bmf_xf_f32": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, C.c_int, _vp, _i64, C.c_int, _vp]),
    "bmf_gram_partial": (C.c_int, [_vp, _i64, _i64, C.c_int, _vp, C.c_int, _vp]),
    "bmf_reduce_slabs": (C.c_int, [_vp, _i64, C.c_int, _i64, _vp, _vp, _vp]),
    "bmf_mu_epilogue": (C.c_int, [C.POINTER(EpilogueArgs), _vp]),
    "bmf_masked_pass": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "bmf_masked_link_pass": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _f64, _vp]),
    "bmf_masked_link_pass_k": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _f64, _vp]),
    "bmf_masked_counts": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "bmf_masked_pass_wide": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp] + [_vp] * 12),
    "bmf_masked_counts_wide": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bmf_real_confusion": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "bmf_confusion_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "bmf_mae_sum": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "bmf_mae_sum_ex": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "bmf_mae_sum_tiled": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "bmf_cover_count": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, C.c_int, _vp, _vp, _vp]),
    "bmf_boolean_product_bits": (C.c_int, [_vp, _i64, _vp, _i64, C.c_int, _i64, _vp, _i64, _vp]),
    "bmf_real_product": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, C.c_int, _vp, _i64, _vp]),
    "bmf_residual_sums": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "bmf_sqdiff_work": (_i64, []),
    "bmf_sqdiff_sum": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "bmf_residual_sums_f32": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, C.c_int, _vp, _vp]),
    "bmf_penalty_prepare": (C.c_int, [C.POINTER(PenaltyState), _vp]),
    "bmf_penalty_update": (C.c_int, [C.POINTER(PenaltyState), _f64, _vp]),
    "bmf_penalty_update_head": (C.c_int, [C.POINTER(PenaltyState), _f64, _vp]),
    "bmf_penalty_update_xtu": (C.c_int, [C.POINTER(PenaltyState), _i32, _vp]),
    "bmf_penalty_finalize": (C.c_int, [C.POINTER(PenaltyState), _i32, _f64, _i32, _vp]),
    "bmf_penalty_run": (C.c_int, [C.POINTER(PenaltyState), _i32, _i32, C.POINTER(_f64), _i32, _vp]),
    "bmf_comm_available": (C.c_int, []),
    "bmf_comm_unique_id": (C.c_int, [_vp]),
    "bmf_comm_create": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "bmf_comm_create_host": (C.c_int, [ALLREDUCE_FN, _vp, _i32, _i32, C.POINTER(_vp)]),
    "bmf_comm_destroy": (C.c_int, [_vp]),
    "bmf_comm_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "bmf_allreduce": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "bmf_penalty_prepare_sharded": (C.c_int, [C.POINTER(PenaltyState), _vp, _f64, _i32, _vp]),
    "bmf_penalty_run_sharded": (C.c_int, [C.POINTER(PenaltyState), _vp, _i32, _i32, C.POINTER(_f64), _i32, _vp]),
    "bmf_exchange_overlaps": (C.c_int, [_vp, _vp]),
    "bmf_exchange_overlap_rule": (C.c_int, [C.c_int, _i64]),
    "bmf_comm_timing": (C.c_int, [_vp, _i32]),
    "bmf_comm_timing_read": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_f64), C.POINTER(_f64)]),
    "bmf_thresh_eval": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, C.c_int, C.c_int, _f64, _f64, _f64,
                                  C.c_int, _vp, _vp, _vp]),
    "bmf_thresh_transform": (C.c_int, [_vp, _i64, _i32, C.c_int, C.c_int, _f64, _f64, _vp, _vp, _vp]),
    "bmf_masked_thresh": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    "bmf_tile_f32": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "bmf_frag_f32": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp]),
    "bmf_frag_rows_f32": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp]),
    "bmf_xf_f32_tiled": (C.c_int, [_vp, _i64, _i64, _vp, C.c_int, _vp, _i64, C.c_int, _vp]),
    "bmf_xf_f32_tiled_resid": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, C.c_int, _vp, _i64, C.c_int, _vp, _vp]),
    "bmf_frag_rows_bf16": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp]),
    "bmf_frag_bf16x3": (C.c_int, [_vp, _i64, _vp, _vp]),
    "bmf_xf_f32_tiled_bf3": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, C.c_int, _vp]),
    "bmf_xf_f32_tiled_resid_bf3": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, C.c_int, _vp, _vp]),
    "bmf_residual_sums_f32_tiled": (C.c_int, [_vp, _i64, _i64, _vp, _vp, C.c_int, _vp, _vp]),
    "bmf_wnmf_real_prepare": (C.c_int, [C.POINTER(WnmfRealState), _vp]),
    "bmf_wnmf_real_run": (C.c_int, [C.POINTER(WnmfRealState), _i32, _i32, _i32, _vp]),
    "bmf_thresh_eval64_work": (_i64, [_i64, _i64, C.c_int]),
    "bmf_thresh_eval64": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, C.c_int, C.c_int, _f64, _f64, _f64, C.c_int, _vp, _vp, _vp]),
    "bmf_thresh_trace64_max_pairs": (C.c_int, [C.c_int]),
    "bmf_thresh_trace64_work": (_i64, [_i32, _i32, C.c_int, C.c_int]),
    "bmf_thresh_trace64": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _i64, C.c_int, _vp, _i32, _f64, _f64, C.c_int, _vp, _vp, _f64, _vp]),
    "bmf_thresh_transform64": (C.c_int, [_vp, _i64, _i32, C.c_int, C.c_int, _f64, _f64, _vp, _vp, _vp]),
    "bmf_masked_thresh64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, C.c_int, _vp, _i32, _vp, _vp]),
    "bmf_masked_thresh64_k": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _i32, _vp, _vp]),
    "bmf_link_splits": (C.c_int, [_i64, _i64]),
    "bmf_masked_iterate": (C.c_int, [_vp, _f64, C.c_int, _vp, _vp]),
    "bmf_link_iterate": (C.c_int, [_vp, _f64, C.c_int, _vp, _vp]),
    "bmf_link_pass": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, C.c_int, C.c_int, _f64, _vp, _vp, _i64, C.c_int, _vp]),
    "bmf_link_split": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp]),
    "bmf_link_split_pair": (C.c_int, [_vp, _i64, _vp, _i64, C.c_int, _vp, _vp, _vp]),
    "bmf_link_pass16": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, C.c_int, C.c_int, _f64, _vp, _vp, _i64, C.c_int, _vp]),
    "bmf_link_sums16": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, C.c_int, C.c_int, _f64, _vp, _vp, _vp]),
    "bmf_link_sums": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, C.c_int, C.c_int, _f64, _vp, _vp, _vp]),
    "bmf_colsum_fill": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp, _i64, _vp]),
    "bmf_sym_norms": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "bmf_palm_epilogue": (C.c_int, [C.POINTER(PalmArgs), _vp]),
    "bmf_palm_extrapolate": (C.c_int, [_vp, _vp, _f64, _i64, _vp, _vp]),
    "bmf_dot_slabs": (C.c_int, [_vp, _vp, _i64, C.c_int, _i64, _vp, C.c_int, _vp]),
    "bmf_fg_f32": (C.c_int, [_vp, _i64, _vp, C.c_int, _vp, C.c_int, _vp]),
    "bmf_gram_cross": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, _vp]),
    "bmf_cover_count_wide": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "bmf_resid_sums_wide": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "bmf_masked_scalars": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "bmf_palm_scalars": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp]),
    "bmf_palm_iterate": (C.c_int, [_vp, C.c_int, _f64, _f64, _f64, _f64, C.c_int, _vp]),
    "bmf_palm_row_lag": (C.c_int, [_vp]),
    "bmf_palm_finish_row": (C.c_int, [_vp, C.c_int, _vp]),
    "bmf_primp_iterate": (C.c_int, [_vp, C.c_int, _f64, _f64, _vp]),
    "bmf_timer_stride": (C.c_int, [C.c_int]),
    "bmf_timer_enable": (C.c_int, [C.c_int]),
    "bmf_timer_read": (C.c_int, [C.POINTER(C.c_int), C.POINTER(_f64)]),
    "bmf_timer_disable": (C.c_int, []),
}


ABI_VERSION = 501   # BMF_ABI_VERSION of include/bmf_hip.h


class BmfError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pybmf_amd/csrc`).  pybmf_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    # the ctypes mirrors against the library's own sizeof(): a header / binding mismatch fails here, not as a corrupted launch
    if lib.bmf_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.bmf_version()}, this binding was written for {ABI_VERSION} (rebuild the library)")
    for which, mirror in enumerate((EpilogueArgs, PalmArgs, PenaltyState, WnmfRealState, PalmState, MaskedLoop, MaskedSide, LinkLoop)):
        if lib.bmf_struct_bytes(which) != C.sizeof(mirror):
            raise ImportError(f"{LIB_PATH}: sizeof({mirror.__doc__}) is {lib.bmf_struct_bytes(which)} in the library, {C.sizeof(mirror)} in the binding")
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc != BMF_OK:
        msg = lib.bmf_last_error()
        raise BmfError(f"{what or 'libbmf_hip'} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())
