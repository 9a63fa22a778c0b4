"""ctypes binding of libhfasr_hip.so (the C ABI declared in include/hfasr_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol cannot be
resolved, importing/using the ops raises immediately (the GPU box must run the HIP kernels).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HFASR_HIP_LIB: an A/B build of the SAME C ABI at another path (tools/*_ab.sh build their variants beside the product library instead of over it: ADVICE r4); unset = the product build
LIB_PATH = os.environ.get("HFASR_HIP_LIB") or os.path.join(_HERE, "libhfasr_hip.so")

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t


class EbfConfig(C.Structure):
    """mirror of mi_ebf_config (include/hfasr_hip.h)"""
    _fields_ = [(n, i32) for n in ("B", "T", "F", "d", "H", "I", "L", "V", "C1", "C2", "K", "stride", "pad",
                                   "is_causal", "pos_type", "csgu_kernel", "merge_kernel", "csgu_act", "use_macaron")] + \
               [("ln_eps", f32), ("logits_f32", i32), ("logits_ld", i32), ("branch_overlap", i32), ("extra_layers", i32), ("layer_mixing", i32), ("csgu_linear", i32),
                ("context_mode", i32), ("gate_blk", i32), ("ln_fold", i32), ("wide_tiles", i32)]


class LnRedDesc(C.Structure):
    """mirror of mi_lnred_desc (include/hfasr_hip.h)"""
    _fields_ = [("partial", vp), ("nblk", i32), ("d", i32), ("dgamma", vp), ("dbeta", vp), ("kind", i32)]


class Gpt2Config(C.Structure):
    """mirror of mi_gpt2_config (include/hfasr_hip.h)"""
    _fields_ = [("d", i32), ("H", i32), ("L", i32), ("V", i32), ("eps", f32), ("step_form", i32)]


GLOBAL_SLOTS, LAYER_SLOTS = 24, 64

# name -> (argtypes); every function returns int (0 = ok), except mi_ebf_workspace_bytes (size_t)
SIGNATURES = {
    "mi_gemm_bf16": [vp, i64, vp, i64, vp, i32, vp, i64, i32, vp, i64, f32, i32, i32, i32, i32, i32, i32, vp],
    "mi_conv2d_cl_bf16": [vp, vp, vp, vp] + [i32] * 13 + [vp],
    "mi_gemm_bf16_v": [vp, i64, vp, i64, vp, i32, vp, i64, i32, vp, i64, f32, i32, i32, i32, i32, i32, i32, i32, vp],
    "mi_conv2d_cl_bf16_v": [vp, vp, vp, vp] + [i32] * 14 + [vp],
    "mi_conv2d_first_gelu": [vp, vp, vp, vp] + [i32] * 10 + [vp],
    "mi_conv2d_first_geo": [vp, vp, vp, vp] + [i32] * 13 + [vp],
    "mi_conv2d_first_gated_gelu": [vp, vp, vp, vp, vp, vp] + [i32] * 10 + [vp],
    "mi_conv2d_cl_geo_bf16": [vp, vp, vp, vp] + [i32] * 15 + [vp],
    "mi_gated_act_bf16": [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp],
    "mi_im2col_cl_geo_bf16": [vp, vp] + [i32] * 12 + [vp],
    "mi_col2im_cl_bf16": [vp, vp] + [i32] * 13 + [vp],
    "mi_gated_act_bwd_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp],
    "mi_conv2d_first_wgrad": [vp, vp, vp, vp] + [i32] * 12 + [vp, vp],
    "mi_gemm_lnfold_bf16": [vp, i64, vp, i64, vp, vp, vp, i32, f32, vp, i64, i32, i32, i32, i32, vp],
    "mi_gemm_resid_stats_f32": [vp, i64, vp, i64, vp, vp, i64, vp, i64, f32, vp, i64, vp, i32, i32, i32, vp],
    "mi_layernorm_fold": [vp, i64, vp, i32, vp, vp, f32, vp, i64, vp, i64, vp, i32, i32, vp],
    "mi_layernorm_chain": [vp, i64, vp, i32, vp, vp, f32, vp, i64, vp, vp, f32, vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, vp],
    "mi_cast_f32_bf16": [vp, i64, vp, i64, i32, i32, vp],
    "mi_rotary_bf16": [vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, vp],
    "mi_attention_bf16": [vp, i64, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, vp],
    "mi_attention_qkv_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i64, i32, i32, f32, i32, vp],
    "mi_attention_qkv_bf16_v": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i64, i32, i32, f32, i32, i32, vp],
    "mi_attention_x_lse_bf16": [vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, i32, f32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_attention_x_bwd_probs": [vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, f32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_attention_qkv_lse_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, f32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_attention_qkv_bwd_probs": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, i32,
                                   vp, i64, vp, vp, i32, i32, i32, i32, f32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_attention_qkv_bwd_probs_qb": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, i32,
                                      vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_attention_qkv_bwd_probs_f": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, i32,
                                     vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, f32, C.c_uint, C.c_uint, i32, vp],
    "mi_row_stats_bf16": [vp, i64, i32, f32, vp, i32, vp],
    "mi_csgu_bf16": [vp, i64, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp],
    "mi_dwconv_residual_bf16": [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp],
    "mi_fbank_f64": [vp, i64, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, f64, f64, vp],
    "mi_trim_zeros_pad_f32": [vp, i64, vp, i32, i32, i32, vp, i64, i32, vp, vp, vp, vp],
    "mi_cmvn_utterance": [vp, vp, i32, i32, i32, i32, i32, f32, vp],
    "mi_cmvn_global": [vp, i64, i32, vp, vp, vp],
    "mi_row_lse": [vp, i64, i32, i32, vp, i32, vp],
    "mi_gemm_lse_f32": [vp, i64, vp, i64, vp, vp, i64, vp, vp, i32, i32, i32, vp],
    "mi_gemm_lse_workspace_floats": [i32, i32],
    "mi_ctc_loss_fwd": [vp, i64, i64, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp],
    "mi_ctc_prefix_prepare": [vp, i64, i64, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp],
    "mi_ctc_prefix_score": [vp, i32, i32, i32, i32, i32, vp, vp, i64, i32, vp, vp, vp, vp],
    "mi_ctc_prefix_select": [vp, i32, i32, i32, i32, i32, vp, vp, i64, i32, vp, vp, i64, i32, vp, vp],
    "mi_embed_tokens": [vp, vp, f32, vp, i32, i32, i32, i32, i32, vp, vp],
    "mi_ce_label_smoothing": [vp, i64, vp, i32, i32, i32, i32, f32, vp, vp, vp],
    "mi_whisper_logmel": [vp, i64, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp],
    "mi_transpose_cast_bct_btc": [vp, vp, i32, i32, i32, vp],
    "mi_add_positions": [vp, vp, vp, i32, i32, i32, vp],
    "mi_transpose_bf16": [vp, i64, vp, i64, i32, i32, i32, vp],
    "mi_transpose_many_bf16": [vp, i32, vp],
    "mi_colsum": [vp, i64, i32, i32, i32, vp, vp, vp],
    "mi_act_fwd_bf16": [vp, i64, vp, i64, i32, i32, i32, vp],
    "mi_act_bwd_bf16": [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp],
    "mi_act_dropout_fwd_bf16": [vp, i64, vp, i64, i32, i32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_act_dropout_bwd_bf16": [vp, i64, vp, i64, vp, i64, i32, i32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_layernorm_bwd_workspace_floats": [i32],
    "mi_layernorm_bwd": [vp, i64, i32, vp, f32, vp, i64, i32, vp, i64, i32, i32, vp, vp, vp, i32, i32, vp],
    "mi_layernorm_bwd_partial": [vp, i64, i32, vp, f32, vp, i64, i32, vp, i64, i32, i32, vp, vp, i32, i32, vp],
    "mi_layernorm_bwd_dual_partial": [vp, i64, i32, f32, vp, vp, i64, i32, vp, vp, i64, i32, vp, i64, i32, i32, vp, vp, vp, vp, i64, f32, f32, C.c_uint, C.c_uint, i32, i32, vp],
    "mi_layernorm_bwd_partial_cast": [vp, i64, i32, vp, f32, vp, i64, i32, vp, i64, i32, i32, vp, vp, vp, i64, f32, f32, C.c_uint, C.c_uint, i32, i32, vp],
    "mi_ln_partial_reduce_many": [vp, i32, vp],
    "mi_ln_apply_bf16": [vp, i64, vp, vp, vp, vp, i64, i32, i32, vp],
    "mi_axpy_f32": [vp, vp, i64, f32, vp],
    "mi_scale_f32": [vp, i64, f32, vp],
    "mi_scale_dev_f32": [vp, i64, vp, vp],
    "mi_add2_cast_bf16": [vp, i64, vp, i64, vp, i64, i32, i32, f32, vp],
    "mi_add_rowvec_bf16": [vp, i64, vp, vp, i64, i32, i32, vp],
    "mi_colsum2_acc_f32": [vp, vp, i64, i32, i32, vp, vp, vp],
    "mi_colsum_cast_bf16": [vp, i64, i32, i32, vp, vp],
    "mi_add_rowvec2_bf16": [vp, i64, vp, vp, vp, vp, i64, i32, i32, vp],
    "mi_gate_bwd_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, vp],
    "mi_subsampled_lengths_i32": [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "mi_mask_rows_f32": [vp, i64, vp, i32, i32, i32, vp],
    "mi_spec_mask_apply": [vp, i64, vp, vp, vp, i32, i32, i32, vp],
    "mi_spec_mask_bwd": [vp, i64, vp, vp, vp, i32, i32, i32, vp, vp],
    "mi_softmax_vec_f32": [vp, i32, vp, vp],
    "mi_softmax_vec_bwd_f32": [vp, vp, i32, vp, vp],
    "mi_axpy_dev_f32": [vp, vp, i64, vp, i32, vp],
    "mi_dot_f32": [vp, vp, i64, vp, vp, vp],
    "mi_sumsq_f32": [vp, i64, vp, vp, vp],
    "mi_clip_coef": [vp, f32, f32, vp, vp],
    "mi_adamw_step": [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp, vp, vp],
    "mi_gemm_tn_workspace_bytes": [i32, i32, i32],
    "mi_colsum_workspace_floats": [i32, i32],
    "mi_conv2d_first_bwd_workspace_floats": [i32, i32, i32, i32],
    "mi_conv2d_first_wgrad_workspace_floats": [i32, i32, i32, i32, i32, i32],
    "mi_gemm_tn_bf16": [vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, i32, vp, sz, i32, vp],
    "mi_gemm_dropout_bf16": [vp, i64, vp, i64, vp, vp, i64, i32, vp, i64, f32, f32, C.c_uint, C.c_uint, i32, i32, i32, vp],
    "mi_gemm_act_fwd_bf16": [vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, f32, C.c_uint, C.c_uint, i32, i32, i32, vp],
    "mi_gemm_act_bwd_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, i32, f32, C.c_uint, C.c_uint, i32, i32, i32, vp],
    "mi_conv2d_wgrad_cl_bf16": [vp, i64, vp, vp, i64, vp] + [i32] * 13 + [vp, C.c_size_t, vp],
    "mi_gemm_resid_stats_f32_v": [vp, i64, vp, i64, vp, vp, i64, vp, i64, f32, vp, i64, vp, i32, i32, i32, i32, vp],
    "mi_gemm_tn_group_bf16": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp],
    "mi_gemm_tn_group_ow_bf16": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "mi_bgemm_bf16": [vp, i64, i64, i64, i64, vp, i64, i64, i64, i64, vp, i64, i64, i64, i32, i32, f32, i32, i32, i32, i32, i32, vp],
    "mi_bgemm_sparse_bf16": [vp, i64, i64, i64, i64, vp, i64, i64, i64, i64, vp, i64, i64, i64, i32, i32, f32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "mi_bgemm_band_bf16": [vp, i64, i64, i64, i64, vp, i64, i64, i64, i64, vp, i64, i64, i64, i32, i32, f32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "mi_attn_softmax_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, f32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_attn_softmax_bwd": [vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, f32, f32, C.c_uint, C.c_uint, vp],
    "mi_dropout": [vp, i64, i32, vp, i64, i32, i32, i32, f32, f32, C.c_uint, C.c_uint, vp],
    "mi_dropout_add_f32": [vp, i64, vp, i64, vp, i64, i32, i32, f32, f32, C.c_uint, C.c_uint, vp],
    "mi_csgu_conv_bf16": [vp, i64, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp],
    "mi_gate_act_mul_bf16": [vp, i64, vp, i64, vp, i64, i64, i32, i32, vp],
    "mi_gate_act_mul_bwd_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, i32, vp],
    "mi_csgu_bwd_bf16": [vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "mi_dwconv_residual_bwd_bf16": [vp, i64, vp, vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "mi_im2col_cl_bf16": [vp, vp] + [i32] * 11 + [vp],
    "mi_conv2d_first_bwd": [vp, vp, vp, vp, vp, vp] + [i32] * 16 + [vp, vp],
    "mi_conv2d_s2k3_dgrad_elems": [i32] * 8,
    "mi_conv2d_s2k3_dgrad_pack_bf16": [vp, i64, vp, i32, i32, vp],
    "mi_conv2d_s2k3_dgrad_bf16": [vp, vp, vp] + [i32] * 9 + [vp],
    "mi_conv2d_first_bwd_phases": [vp, vp, vp, vp, vp, vp] + [i32] * 14 + [vp, vp],
    "mi_ctc_bwd_workspace_bytes": [i32, i32, i32],
    "mi_ctc_loss_bwd": [vp, i64, i64, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, f32, vp, sz, vp, i64, vp],
    "mi_ctc_loss_bwd_nll": [vp, i64, i64, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, f32, vp, sz, vp, i64, vp, vp, vp, vp],
    "mi_ctc_reduce": [vp, vp, i32, i32, i32, vp, vp],
    "mi_ce_label_smoothing_bwd": [vp, i64, vp, i32, i32, i32, i32, f32, f32, vp, vp, i64, vp],
    "mi_embed_tokens_bwd": [vp, vp, f32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp],
    "mi_embed_tokens_bwd_workspace_bytes": [i32, i32, i32],
    "mi_specaug_f32": [vp, vp, i32, i32, i32, vp, i32, i32, f32, vp],
    "mi_speed_resample_f32": [vp, i64, vp, i32, i32, i32, i32, vp, i32, vp, i64, i32, vp, vp],
    "mi_rpq_targets": [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mi_mask_noise_f32": [vp, i64, vp, i32, i32, f32, C.c_uint, C.c_uint, vp],
    "mi_gpt2_step_workspace_bytes": [C.POINTER(Gpt2Config), i32, i32],
    "mi_gpt2_step": [C.POINTER(Gpt2Config), vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, f32, vp, sz, vp, i64, vp],
    "mi_kv_cache_reorder": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "mi_ctc_prefix_advance": [vp, i32, i32, i32, i32, i32, vp, vp, i64, i32, vp, vp, i64, i32, vp, vp, vp, vp],
    "mi_ctc_prefix_score_full": [vp, i32, i32, i32, i32, i32, vp, i32, vp, vp, i64, i32, vp, vp, vp, vp],
    "mi_beam_step": [vp, i64, vp, vp, f32, f32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "mi_ebf_workspace_bytes": [C.POINTER(EbfConfig)],
    "mi_ebf_forward": [C.POINTER(EbfConfig), vp, vp, vp, vp, vp, i32, vp, sz, vp, vp, vp, vp, vp],
    "mi_ebf_forward_hs": [C.POINTER(EbfConfig), vp, vp, vp, vp, vp, i32, vp, sz, vp, vp, vp, vp, vp, vp],
    "mi_ebf_forward_lse": [C.POINTER(EbfConfig), vp, vp, vp, vp, vp, i32, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp],
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises HipLibraryError if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python huggingface_asr_amd/csrc/build.py` "
                "(or __graft_entry__.build()). There is no CPU fallback for the HIP path.")
        # PyTorch-ROCm bundles its own libamdhip64; it must be resident BEFORE our library is dlopen()ed so that
        # both bind to ONE HIP runtime (same SONAME).  Loaded the other way round, the process holds two runtimes
        # and kernels launched from here fail with hipErrorNoDevice.
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError here = header/library mismatch: fail loudly
            fn.argtypes = args
            fn.restype = sz if name in ("mi_ebf_workspace_bytes", "mi_ctc_bwd_workspace_bytes", "mi_gemm_tn_workspace_bytes", "mi_layernorm_bwd_workspace_floats",
                                          "mi_colsum_workspace_floats", "mi_conv2d_first_bwd_workspace_floats", "mi_conv2d_s2k3_dgrad_elems", "mi_conv2d_first_wgrad_workspace_floats", "mi_embed_tokens_bwd_workspace_bytes",
                                          "mi_gpt2_step_workspace_bytes", "mi_gemm_lse_workspace_floats") else i32
        h.mi_profile_create.argtypes = [i32]; h.mi_profile_create.restype = i32
        h.mi_profile_enable.argtypes = [i32]; h.mi_profile_enable.restype = None
        h.mi_profile_reset.argtypes = []; h.mi_profile_reset.restype = None
        h.mi_profile_count.argtypes = []; h.mi_profile_count.restype = i32
        h.mi_profile_summary.argtypes = [C.POINTER(f64), C.POINTER(f64)]; h.mi_profile_summary.restype = i32
        h.mi_profile_calibrate.argtypes = [vp, i32, C.POINTER(f64)]; h.mi_profile_calibrate.restype = i32
        h.mi_profile_summary_family.argtypes = [i32, C.POINTER(f64), C.POINTER(f64), C.POINTER(i32)]; h.mi_profile_summary_family.restype = i32
        h.mi_last_error.argtypes = []
        h.mi_last_error.restype = C.c_char_p
        _lib = h
    return _lib


ERR_ARG, ERR_LAUNCH, ERR_UNSUPPORTED = -1, -2, -3          # MI_ERR_* of include/hfasr_hip.h
_ERR = {ERR_ARG: "invalid argument", ERR_LAUNCH: "kernel launch failed", ERR_UNSUPPORTED: "unsupported configuration"}


def check(rc: int, what: str):
    if rc != 0:
        detail = lib().mi_last_error().decode() if rc == -2 else ""
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)} {detail}")
