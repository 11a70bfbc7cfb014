"""ctypes binding of libicka_hip.so (C-ABI declared in include/icka_hip.h).

The library is built in-tree by ``icka_amd/csrc/Makefile`` (``python -c 'import __graft_entry__ as g; g.build()'``).
There is NO fallback: if the shared object is missing, ``load()`` raises.  Nothing in this package computes on
the CPU or through eager PyTorch ops in place of a missing kernel.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libicka_hip.so")

c_vp, c_i32, c_i64, c_u64, c_f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float

ABI_VERSION = 6   # include/icka_hip.h: ICKA_ABI_VERSION (load() refuses a library built from another header)
GEMM_NT, GEMM_NN, GEMM_TN, GEMM_TT = 0, 1, 2, 3
ATTN_FP8, ATTN_TILED = 1, 2                               # flags of icka_attn_fwd_ex / icka_attn_bwd
LSTM_PER_STEP, LSTM_TICKETS, LSTM_NO_BATCH_SPLIT = 1, 2, 4   # flags of icka_lstm_fwd / icka_lstm_bwd
EPI_NONE, EPI_GELU, EPI_DGELU, EPI_ADD, EPI_GATE, EPI_TANH, EPI_RELU, EPI_ADD_RELU = 0, 1, 2, 3, 4, 5, 6, 7


class GemmDesc(C.Structure):
    """Mirror of ``icka_gemm_desc`` (include/icka_hip.h)."""
    _fields_ = [
        ("op", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32), ("K1", c_i32),
        ("A", c_vp), ("lda", c_i64), ("B", c_vp), ("ldb", c_i64),
        ("A2", c_vp), ("lda2", c_i64), ("B2", c_vp), ("ldb2", c_i64),
        ("C", c_vp), ("ldc", c_i64), ("c_is_f32", c_i32),
        ("C2", c_vp), ("ldc2", c_i64),
        ("aux", c_vp), ("ldaux", c_i64),
        ("bias", c_vp), ("bias2", c_vp),
        ("alpha", c_f32), ("beta", c_f32),
        ("epilogue", c_i32),
        ("colsum_out", c_vp), ("colsum_accumulate", c_i32),
        ("ab_f16", c_i32), ("C3", c_vp), ("ldc3", c_i64), ("aux_f16", c_i32), ("c3_only", c_i32),
        ("tune", c_u64),
    ]


class SlabReduction(C.Structure):
    """Mirror of ``icka_slab_reduction`` (include/icka_hip.h)."""
    _fields_ = [("partials", c_vp), ("slab_stride", c_i64), ("nslab", c_i32), ("H", c_i32), ("nslots", c_i32),
                ("accumulate", c_i32), ("out", c_vp * 4)]


class XGemmDesc(C.Structure):
    """Mirror of ``icka_xgemm_desc`` (include/icka_hip.h): batched f32 GEMM of the fp32-exact mode."""
    _fields_ = [
        ("op", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32),
        ("A", c_vp), ("lda", c_i64), ("a_bs0", c_i64), ("a_bs1", c_i64),
        ("B", c_vp), ("ldb", c_i64), ("b_bs0", c_i64), ("b_bs1", c_i64),
        ("C", c_vp), ("ldc", c_i64), ("c_bs0", c_i64), ("c_bs1", c_i64),
        ("nb0", c_i32), ("nb1", c_i32),
        ("bias", c_vp),
        ("alpha", c_f32), ("beta", c_f32),
    ]


# name -> (restype, argtypes); must list every function declared in include/icka_hip.h
PROTOTYPES = {
    "icka_abi_version": (c_i32, []),
    "icka_copy_many": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp]),
    "icka_build_arch": (C.c_char_p, []),
    "icka_gemm": (c_i32, [C.POINTER(GemmDesc), c_vp]),
    "icka_gemm_grouped": (c_i32, [C.POINTER(GemmDesc), c_i32, c_vp]),
    "icka_gemm_grouped_ex": (c_i32, [C.POINTER(GemmDesc), c_i32, C.POINTER(SlabReduction), c_i32, c_vp]),
    "icka_gemm_ln": (c_i32, [C.POINTER(GemmDesc), c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_f32,
                             c_f32, c_u64, c_vp, c_vp, c_vp]),
    "icka_gemm_qkv_attn": (c_i32, [C.POINTER(GemmDesc), c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp, c_vp]),
    "icka_gemm_ln_sync_words": (c_i64, []),
    "icka_gemm_ln_test_hooks": (c_i32, [c_i32, c_i32]),
    "icka_ln_fwd": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64,
                            c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_ln_fwd_h": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64,
                            c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_ln_bwd_workspace_floats": (c_i64, [c_i32]),
    "icka_ln_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp,
                            c_vp, c_i32, c_i32, c_f32, c_u64, c_i32, c_vp]),
    "icka_ln_bwd_slabs": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32,
                                  c_i32, c_f32, c_u64, c_vp]),
    "icka_ln_bwd_nslab": (c_i32, [c_i32]),
    "icka_ln_slab_slots": (c_i32, []),
    "icka_embed_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                               c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_embed_fwd_h": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                               c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_embed_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                               c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_i32, c_vp]),
    "icka_embed_bwd_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                                    c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_i32, c_vp]),
    "icka_embed_scatter_rows": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "icka_attn_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32,
                              c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_attn_fwd_ex": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32,
                                 c_i32, c_f32, c_f32, c_u64, c_i32, c_vp, c_vp]),
    "icka_attn_keepbits_words": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "icka_attn_fwd_fp8": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32,
                              c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_cls_head_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_cls_head_fwd_h": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_cls_head_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                  c_vp]),
    "icka_cls_head_bwd_slabs": (c_i32, [c_i32]),
    "icka_cls_head_slab_floats": (c_i64, [c_i32, c_i32]),
    "icka_conv_stem_patches": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "icka_conv_im2col3x3": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "icka_conv3x3_gemm": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32,
                                   c_vp, c_vp]),
    "icka_conv_subsample": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "icka_conv_maxpool3x3s2": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "icka_conv_features_out": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_lstm_fwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_lstm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_lstm_barrier_error": (c_i32, []),
    "icka_lstm_clear_error": (c_i32, []),
    "icka_lstm_set_reserved_cus": (c_i32, [c_i32]),
    "icka_lstm_test_hooks": (c_i32, [c_i32, c_i32]),
    "icka_linear_small_m": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_transpose_bf16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_crf_llh": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_crf_grad": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32,
                              c_i32, c_i32, c_vp]),
    "icka_crf_decode": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_attn_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp,
                              c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64,
                              c_vp, c_i32, c_vp]),
    "icka_cast_f32_to_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "icka_cast_to_f16": (c_i32, [c_vp, c_i32, c_vp, c_i64, c_vp]),
    "icka_cast_f32_to_bf16_f16": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "icka_cast_bf16_to_f32": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "icka_cast_pad_f32_to_bf16": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "icka_additive_mask": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_vp]),
    "icka_dropout": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_f32, c_u64, c_vp]),
    "icka_regions_to_tokens": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_regions_to_tokens_h": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_sample_gate_fwd_h": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "icka_colsum": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_colsum_workspace_floats": (c_i64, [c_i32]),
    "icka_gate_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "icka_sample_gate_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "icka_sample_gate_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp,
                                     c_i32, c_i32, c_i32, c_vp]),
    "icka_crs_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_crs_bwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                             c_vp]),
    "icka_add_bf16": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "icka_tanh_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "icka_dgelu_bf16": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "icka_embed_prompt_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                                      c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_embed_prompt_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                                      c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_i32, c_vp]),
    "icka_token_ce": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "icka_token_ce_workspace_floats": (c_i64, [c_i32]),
    "icka_token_ce_fused": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "icka_zero_f32": (c_i32, [c_vp, c_i64, c_vp]),
    "icka_scale_by_ratio": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "icka_scalar_ratio": (c_i32, [c_vp, c_vp, c_vp, c_vp]),
    "icka_set_dropout_nonce": (c_i32, [c_vp]),
    "icka_bump_dropout_nonce": (c_i32, [c_vp, c_vp]),
    "icka_dropout_mask": (c_i32, [c_vp, c_i64, c_f32, c_u64, c_vp]),
    "icka_attn_dropout_mask": (c_i32, [c_vp, c_i64, c_i32, c_f32, c_u64, c_vp]),
    # ---- data-parallel helpers (csrc/dp.hip)
    "icka_dp_chunk_elems": (c_i64, []),
    "icka_dp_cast_chunks": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp]),
    "icka_dp_cast_back_scaled": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    "icka_dp_init": (c_i32, []),
    "icka_dp_error": (c_i32, []),
    "icka_dp_clear_error": (c_i32, []),
    "icka_dp_step_bump": (c_i32, [c_vp, c_vp]),
    "icka_dp_flag_set": (c_i32, [c_vp, c_vp, c_vp]),
    "icka_dp_flag_wait": (c_i32, [c_vp, C.c_uint32, c_vp, c_i32, c_vp]),
    "icka_dp_poison_if": (c_i32, [c_vp, C.c_uint32, c_vp, c_i32, c_i32, c_vp]),
    "icka_dp_poison_final": (c_i32, [c_vp, c_i32, C.c_uint32, c_vp, c_vp, c_i32, c_vp]),
    # ---- parameter update (csrc/optim.hip)
    "icka_optim_chunk_elems": (c_i64, []),
    "icka_optim_sqnorm": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp]),
    "icka_optim_clip": (c_i32, [c_vp, c_i32, c_f32, c_vp, c_vp]),
    "icka_optim_adamw": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32,
                                 c_i32, c_vp]),
    # ---- fp32 "exact" mode (csrc/exact.hip)
    "icka_x_gemm": (c_i32, [C.POINTER(XGemmDesc), c_vp]),
    "icka_x_ln_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_u64,
                              c_vp]),
    "icka_x_ln_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_u64, c_vp]),
    "icka_x_colsum": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "icka_x_embed_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32,
                                 c_vp]),
    "icka_x_embed_scatter": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_x_softmax_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_x_softmax_bwd": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64, c_vp]),
    "icka_x_act_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "icka_x_act_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "icka_x_dropout": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_u64, c_vp]),
    "icka_x_add": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "icka_x_concat2": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i32, c_vp]),
    "icka_x_regions_to_tokens": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_x_sample_gate_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "icka_x_sample_gate_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp,
                                       c_i32, c_i32, c_i32, c_vp]),
    "icka_x_lstm_cell_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_x_lstm_cell_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "icka_x_embed_prompt_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                        c_i32, c_i32, c_i32, c_f32, c_vp]),
    "icka_x_embed_prompt_scatter": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                            c_i32, c_vp]),
    "icka_x_token_ce": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "icka_x_scale_by_ratio": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
}

_lib: Optional[C.CDLL] = None


class IckaLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libicka_hip.so (once) and attach prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ICKA_HIP_LIB", LIB_PATH)   # diagnostic builds (tools/*_bench.py); default: the in-tree .so
    if not os.path.exists(path):
        raise IckaLibraryError(
            "libicka_hip.so not found at %s: build it with `make -C icka_amd/csrc` "
            "(or __graft_entry__.build()).  icka_amd has no CPU / eager fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.icka_abi_version() != ABI_VERSION:
        raise IckaLibraryError("%s was built for C-ABI version %d, this package binds version %d: rebuild it "
                               "(make -C icka_amd/csrc)" % (path, lib.icka_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_ERR = {-1: "ICKA_E_SHAPE (dimension out of supported range)", -2: "ICKA_E_ALIGN (pointer / leading dimension "
        "alignment)", -3: "ICKA_E_ARG (null pointer or inconsistent descriptor)"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, "hipError_t %d" % rc)))
