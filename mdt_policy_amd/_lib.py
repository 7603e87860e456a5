"""ctypes binding of libmdt_hip.so (include/mdt_hip.h, include/mdt_hip_ops.h).

The product path has NO fallback: if the HIP library cannot be loaded (or built with hipcc) every entry
point raises.  PyTorch-ROCm is only used by callers for device memory and streams; nothing here imports torch.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import build as _build

c_float_p = C.POINTER(C.c_float)


class MDTHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libmdt_hip status {status}: {message}")
        self.status = status


class MDTConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "arch", "embed_dim", "n_heads", "n_enc_layers", "n_dec_layers", "action_dim", "obs_dim", "goal_dim",
        "n_obs_token", "goal_seq_len", "action_seq_len", "use_mlp_goal", "use_modality_encoder", "use_abs_pos_emb",
        "use_rot_embed", "use_ada_conditioning", "use_noise_encoder", "linear_output", "bias")] + [
        ("sigma_data", C.c_float), ("no_goal_conditioning", C.c_int32), ("proprio_dim", C.c_int32),
        ("use_proprio", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64), ("Wp", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("ldo", C.c_int64), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ln", C.c_int32),
        ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("mod", C.c_void_p), ("mod_stride", C.c_int64),
        ("shift_off", C.c_int32), ("scale_off", C.c_int32), ("rows_per_sample", C.c_int32), ("act", C.c_int32),
        ("residual", C.c_int32), ("gate_off", C.c_int32), ("gin", C.c_int32), ("gout", C.c_int32),
        ("goff", C.c_int32), ("rowvec", C.c_void_p), ("batch", C.c_int32), ("bs_a", C.c_int64), ("bs_w", C.c_int64),
        ("bs_out", C.c_int64), ("aux", C.c_void_p), ("aux_mode", C.c_int32), ("a_parts", C.c_int32),
        ("a_part_stride", C.c_int64), ("a_merged", C.c_void_p), ("Wp_split", C.c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", C.c_int64), ("k", C.c_void_p), ("v", C.c_void_p), ("ldkv", C.c_int64),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("hd", C.c_int32),
        ("Tq", C.c_int32), ("Tk", C.c_int32), ("causal", C.c_int32), ("rope", C.c_int32)]


class XFoldArgs(C.Structure):
    _fields_ = [("kv", C.c_void_p), ("ldkv", C.c_int64), ("WqT_p", C.c_void_p), ("bq", C.c_void_p), ("Wo_p", C.c_void_p),
                ("U", C.c_void_p), ("Wf", C.c_void_p), ("c", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32),
                ("hd", C.c_int32), ("D", C.c_int32), ("Te", C.c_int32)]


class XApplyArgs(C.Structure):
    _fields_ = [("y", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("U", C.c_void_p), ("Wf", C.c_void_p),
                ("c", C.c_void_p), ("bo", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("D", C.c_int32),
                ("Te", C.c_int32), ("Ta", C.c_int32), ("y_out", C.c_void_p)]


class HeadArgs(C.Structure):
    _fields_ = [
        ("y", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("Wp", C.c_void_p), ("bp", C.c_void_p),
        ("x", C.c_void_p), ("sigma", C.c_void_p), ("sigma_stride", C.c_int64), ("out", C.c_void_p),
        ("M", C.c_int32), ("D", C.c_int32), ("A", C.c_int32), ("rows_per_sample", C.c_int32), ("mode", C.c_int32),
        ("step", C.c_void_p), ("sigma_data", C.c_float), ("y_next", C.c_void_p), ("Wa", C.c_void_p),
        ("ba", C.c_void_p), ("no_ln", C.c_int32), ("y_parts", C.c_int32), ("y_part_stride", C.c_int64)]


class LnTrainArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("mod", C.c_void_p), ("mod_stride", C.c_int64),
                ("shift_off", C.c_int32), ("scale_off", C.c_int32), ("rows_per_sample", C.c_int32),
                ("out", C.c_void_p), ("stats", C.c_void_p), ("M", C.c_int32), ("D", C.c_int32)]


class LnBwdArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("stats", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("mod", C.c_void_p),
                ("mod_stride", C.c_int64), ("shift_off", C.c_int32), ("scale_off", C.c_int32),
                ("dh", C.c_void_p), ("ld_dh", C.c_int64), ("dx", C.c_void_p), ("accumulate", C.c_int32),
                ("d_mod", C.c_void_p), ("d_mod_stride", C.c_int64), ("pw", C.c_void_p), ("pb", C.c_void_p),
                ("B", C.c_int32), ("rows_per_sample", C.c_int32), ("D", C.c_int32), ("row_chunks", C.c_int32),
                ("accumulate_dmod", C.c_int32)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("ldq", C.c_int64), ("k", C.c_void_p), ("v", C.c_void_p), ("ldkv", C.c_int64),
                ("d_out", C.c_void_p), ("ld_do", C.c_int64), ("dq", C.c_void_p), ("ld_dq", C.c_int64),
                ("dk", C.c_void_p), ("dv", C.c_void_p), ("ld_dkv", C.c_int64), ("accumulate_kv", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("hd", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32),
                ("causal", C.c_int32), ("p", C.c_float), ("site", C.c_uint32), ("seed", C.c_uint64),
                ("rope", C.c_int32), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p)]


class AttnTrainArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("ldq", C.c_int64), ("k", C.c_void_p), ("v", C.c_void_p), ("ldkv", C.c_int64),
                ("out", C.c_void_p), ("ldo", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("hd", C.c_int32),
                ("Tq", C.c_int32), ("Tk", C.c_int32), ("causal", C.c_int32), ("p", C.c_float), ("site", C.c_uint32),
                ("seed", C.c_uint64), ("rope", C.c_int32), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p)]


class MergeArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("a", C.c_void_p), ("gate", C.c_void_p), ("gate_stride", C.c_int64),
                ("out", C.c_void_p), ("dgate", C.c_void_p), ("dgate_stride", C.c_int64), ("B", C.c_int32),
                ("rows_per_sample", C.c_int32), ("D", C.c_int32), ("p", C.c_float), ("site", C.c_uint32),
                ("seed", C.c_uint64)]


class Dropout(C.Structure):
    """mirrors mdt_dropout (include/mdt_hip_train.h)"""
    _fields_ = [("attn_p", C.c_float), ("resid_p", C.c_float), ("mlp_p", C.c_float), ("embed_p", C.c_float),
                ("seed", C.c_uint64)]


class LinearBwdArgs(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("dY", C.c_void_p), ("ldy", C.c_int64), ("Wt", C.c_void_p),
                ("dW", C.c_void_p), ("dbias", C.c_void_p), ("dX", C.c_void_p), ("ldxo", C.c_int64),
                ("accumulate_dw", C.c_int32), ("accumulate_dx", C.c_int32), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("scratch", C.c_void_p), ("dx_act_u", C.c_void_p), ("dx_act", C.c_int32)]


class OptTensor(C.Structure):
    """mirrors mdt_opt_tensor (include/mdt_hip_train.h)"""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p),
                ("numel", C.c_int64)]


class ResamplerConfig(C.Structure):
    """mirrors mdt_resampler_config (include/mdt_resampler.h)"""
    _fields_ = [(n, C.c_int32) for n in ("dim", "depth", "dim_head", "heads", "num_latents", "num_time_embeds",
                                          "ff_mult", "activation")]


class MapPoolConfig(C.Structure):
    """mirrors mdt_map_pool_config (include/mdt_map_pool.h)"""
    _fields_ = [(n, C.c_int32) for n in ("n_latents", "embed_dim", "output_dim", "n_heads", "mlp_hidden")]


class InfoNCEArgs(C.Structure):
    """mirrors mdt_infonce_args (include/mdt_map_pool.h)"""
    _fields_ = [("image_features", C.c_void_p), ("lang_features", C.c_void_p), ("logit_scale", C.c_void_p),
                ("batch", C.c_int32), ("dim", C.c_int32), ("mode", C.c_int32), ("loss", C.c_void_p),
                ("d_image", C.c_void_p), ("d_lang", C.c_void_p), ("d_logit_scale", C.c_void_p), ("scratch", C.c_void_p)]


INFONCE_MODE = {"symmetric": 0, "img_to_text": 1, "text_to_img": 2}
ARCH = {"mdtv": 0, "mdt": 1}
MODALITY = {"vis": 0, "lang": 1}
ACT = {"none": 0, "gelu": 1, "mish": 2, "silu": 3, "swiglu": 4}
HEAD = {"denoised": 0, "ddim": 1, "raw": 2}
RAW_OUTPUT, RAW_INPUT, SIGMA_SCALAR = 1, 2, 4

# every symbol include/*.h declares: (name, restype, argtypes)
_VP, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = [
    ("mdt_last_error", C.c_char_p, []),
    ("mdt_version", C.c_char_p, []),
    ("mdt_set_allocator", _I32, [_VP, _VP, _VP]),
    ("mdt_allocator_detach", _I32, []),
    ("mdt_create", _I32, [C.POINTER(MDTConfig), C.POINTER(_VP)]),
    ("mdt_destroy", _I32, [_VP]),
    ("mdt_param_count", _I64, [_VP]),
    ("mdt_param_name", C.c_char_p, [_VP, _I64]),
    ("mdt_param_numel", _I64, [_VP, _I64]),
    ("mdt_load_param", _I32, [_VP, C.c_char_p, _VP, _I64, _VP]),
    ("mdt_load_params", _I32, [_VP, _I32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(_I64), _VP]),
    ("mdt_reserve", _I32, [_VP, _I64]),
    ("mdt_ws_generation", _I64, [_VP]),
    ("mdt_encode", _I32, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _I64, _VP, _VP]),
    ("mdt_denoise_cached", _I32, [_VP, _VP, _VP, _I64, _I32, _VP, _VP]),
    ("mdt_forward", _I32, [_VP, _VP, _VP, _VP, _I32, _VP, _VP, _I64, _VP, _VP, _VP]),
    ("mdt_sample_ddim", _I32, [_VP, _VP, _VP, _VP, _I32, _VP, C.POINTER(C.c_float), _I32, _I64, _VP, _VP, _VP]),
    ("mdt_sample_ddim_dev", _I32, [_VP, _VP, _VP, _VP, _I32, _VP, _VP, _I32, _I64, _VP, _VP, _VP]),
    ("mdt_op_trace_mlp", None, [_I32]),
    ("mdt_op_trace_mlp_read", _I32, [_VP, _I32]),
    ("mdt_op_trace_mlp_read_empty", _I32, [_VP, _I32]),
    ("mdt_op_clock_stamp", _I32, [_VP, _VP]),
    ("mdt_op_set_ws_split", None, [_I32]),
    ("mdt_op_set_tn_split", None, [_I32]),
    ("mdt_loss_fwd", _I32, [_VP, _VP, _VP, _VP, _I32, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP]),
    ("mdt_flops_per_chunk", C.c_double, [_VP, _I32]),
    ("mdt_fnv1_32", C.c_uint32, [C.c_char_p, C.c_uint64, C.c_uint32]),
    ("mdt_op_packed_numel", _I64, [_I64, _I64]),
    ("mdt_op_pack_weight", _I32, [_VP, _I64, _I64, _VP, _I64, _I64, _VP]),
    ("mdt_op_pack_weight_glu", _I32, [_VP, _I64, _I64, _VP, _VP]),
    ("mdt_op_gemm", _I32, [C.POINTER(GemmArgs), _VP]),
    ("mdt_op_mlp", _I32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), _VP, _I64, C.POINTER(_I32), _VP]),
    ("mdt_op_pack_weight_split", _I32, [_VP, _I64, _I64, _VP, _VP]),
    ("mdt_op_pack_weight_split_rows", _I32, [_VP, _I64, _I64, _VP, _I64, _VP]),
    ("mdt_op_mlp_split", _I32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), _VP, _VP, _VP, _I64, C.POINTER(_I32), _VP]),
    ("mdt_op_set_mlp_split", None, [_I32]),
    ("mdt_op_set_gemm_geometry", None, [_I32]),
    ("mdt_op_set_mlp_fuse_min", None, [_I32]),
    ("mdt_op_set_side_jobs", None, [_I32]),
    ("mdt_op_side_jobs_paired", _I64, []),
    ("mdt_op_set_mlp_skew", None, [_I32]),
    ("mdt_op_set_attn_wide_min", None, [_I32]),
    ("mdt_op_attention", _I32, [C.POINTER(AttnArgs), _VP]),
    ("mdt_op_attn_proj", _I32, [C.POINTER(GemmArgs), _VP, _I64, _I32, _I32, _I32, _VP]),
    ("mdt_op_xattn_fold", _I32, [C.POINTER(XFoldArgs), _VP]),
    ("mdt_op_xattn_apply", _I32, [C.POINTER(XApplyArgs), _VP]),
    ("mdt_op_attn_xattn", _I32, [C.POINTER(GemmArgs), _VP, _I64, C.POINTER(XApplyArgs), _I32, _I32, _VP]),
    ("mdt_op_xattn_gemm", _I32, [C.POINTER(XApplyArgs), C.POINTER(GemmArgs), _VP]),
    ("mdt_op_layernorm", _I32, [_VP, _VP, _VP, _VP, _I64, _I32, _VP]),
    ("mdt_op_head", _I32, [C.POINTER(HeadArgs), _VP]),
    ("mdt_op_action_embed", _I32, [_VP, _VP, _I64, _F, _VP, _VP, _VP, _I64, _I32, _I32, _I32, _VP]),
    # include/mdt_hip_train.h
    ("mdt_train_prepare", _I32, [_VP]),
    ("mdt_grad_numel", _I64, [_VP]),
    ("mdt_grad_offset", _I64, [_VP, _I64]),
    ("mdt_train_loss_fwd", _I32, [_VP, _VP, _VP, _VP, _I32, _VP, _VP, _VP, _I64, C.POINTER(Dropout), _VP, _VP, _VP,
                                  C.POINTER(_I32), _VP]),
    ("mdt_train_loss_bwd", _I32, [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("mdt_train_loss_bwd_stages", _I32, [_VP]),
    ("mdt_train_loss_bwd_stage", _I32, [_VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("mdt_train_param_stage", _I32, [_VP, _I64]),
    ("mdt_train_encode_fwd", _I32, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _I64, C.POINTER(Dropout), _VP, C.POINTER(_I32),
                                    _VP]),
    ("mdt_train_encode_bwd", _I32, [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("mdt_tape_release", _I32, [_VP, _I32]),
    ("mdt_denoise_vjp", _I32, [_VP, _VP, _VP, _VP, _I32, _VP, _VP, _VP, _I64, _VP, _VP, _VP]),
    ("mdt_op_pack_weight_t", _I32, [_VP, _I64, _I64, _I64, _VP, _I64, _I64, _VP]),
    ("mdt_op_pack_many", _I32, [_I32, C.POINTER(C.c_void_p), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), _VP]),
    ("mdt_op_ln_fwd_train", _I32, [C.POINTER(LnTrainArgs), _VP]),
    ("mdt_op_ln_bwd", _I32, [C.POINTER(LnBwdArgs), _VP]),
    ("mdt_op_attn_bwd", _I32, [C.POINTER(AttnBwdArgs), _VP]),
    ("mdt_op_act_fwd", _I32, [_VP, _VP, _I64, _I32, _VP]),
    ("mdt_op_act_bwd", _I32, [_VP, _VP, _VP, _I64, _I32, _VP]),
    ("mdt_op_attn_fwd_train", _I32, [C.POINTER(AttnTrainArgs), _VP]),
    ("mdt_op_merge_fwd", _I32, [C.POINTER(MergeArgs), _VP]),
    ("mdt_op_merge_bwd", _I32, [C.POINTER(MergeArgs), _VP]),
    ("mdt_op_merge_ln_fwd", _I32, [C.POINTER(MergeArgs), C.POINTER(LnTrainArgs), _VP]),
    ("mdt_op_ln_bwd_merge", _I32, [C.POINTER(LnBwdArgs), C.POINTER(MergeArgs), _VP]),
    ("mdt_op_colsum", _I32, [_VP, _I64, _I64, _I64, _VP, _I32, _VP]),
    ("mdt_op_linear_bwd", _I32, [C.POINTER(LinearBwdArgs), _VP]),
    ("mdt_op_linear_bwd_scratch", _I64, [_I64, _I64, _I64]),
    ("mdt_op_linear_bwd_scratch_exact", _I64, [_I64, _I64, _I64]),
    ("mdt_op_multi_adamw", _I32, [C.POINTER(OptTensor), _I32, _F, _F, _F, _F, _F, _I64, _VP]),
    ("mdt_op_multi_ema", _I32, [C.POINTER(OptTensor), _I32, _F, _VP]),
    # include/mdt_resampler.h
    ("mdt_resampler_create", _I32, [C.POINTER(ResamplerConfig), C.POINTER(_VP)]),
    ("mdt_resampler_destroy", _I32, [_VP]),
    ("mdt_resampler_param_count", _I64, [_VP]),
    ("mdt_resampler_param_name", C.c_char_p, [_VP, _I64]),
    ("mdt_resampler_param_numel", _I64, [_VP, _I64]),
    ("mdt_resampler_load_param", _I32, [_VP, C.c_char_p, _VP, _I64, _VP]),
    ("mdt_resampler_forward", _I32, [_VP, _VP, _VP, _I64, _I32, _I32, _VP, _VP]),
    ("mdt_resampler_train_prepare", _I32, [_VP]),
    ("mdt_resampler_grad_numel", _I64, [_VP]),
    ("mdt_resampler_grad_offset", _I64, [_VP, _I64]),
    ("mdt_resampler_forward_train", _I32, [_VP, _VP, _VP, _I64, _I32, _I32, _VP, C.POINTER(_I32), _VP]),
    ("mdt_resampler_backward", _I32, [_VP, _I32, _VP, _VP, _VP, _VP]),
    ("mdt_resampler_tape_release", _I32, [_VP, _I32]),
    ("mdt_resampler_flops", C.c_double, [_VP, _I32, _I32]),
    # include/mdt_map_pool.h
    ("mdt_map_pool_create", _I32, [C.POINTER(MapPoolConfig), C.POINTER(_VP)]),
    ("mdt_map_pool_destroy", _I32, [_VP]),
    ("mdt_map_pool_param_count", _I64, [_VP]),
    ("mdt_map_pool_param_name", C.c_char_p, [_VP, _I64]),
    ("mdt_map_pool_param_numel", _I64, [_VP, _I64]),
    ("mdt_map_pool_load_param", _I32, [_VP, C.c_char_p, _VP, _I64, _VP]),
    ("mdt_map_pool_forward", _I32, [_VP, _VP, _I64, _I32, _VP, _VP]),
    ("mdt_map_pool_train_prepare", _I32, [_VP]),
    ("mdt_map_pool_grad_numel", _I64, [_VP]),
    ("mdt_map_pool_grad_offset", _I64, [_VP, _I64]),
    ("mdt_map_pool_forward_train", _I32, [_VP, _VP, _I64, _I32, _VP, C.POINTER(_I32), _VP]),
    ("mdt_map_pool_backward", _I32, [_VP, _I32, _VP, _VP, _VP, _VP]),
    ("mdt_map_pool_tape_release", _I32, [_VP, _I32]),
    # include/mdt_mae.h
    ("mdt_op_rms_fwd", _I32, [_VP, _VP, _VP, _I64, _I32, _F, _VP]),
    ("mdt_op_rms_bwd_scratch", _I64, [_I64, _I32]),
    ("mdt_op_rms_bwd", _I32, [_VP, _VP, _VP, _VP, _I32, _VP, _I32, _I64, _I32, _F, _VP, _VP]),
    ("mdt_op_rms_bwd_res", _I32, [_VP, _VP, _VP, _VP, _VP, _VP, _I32, _I64, _I32, _F, _VP, _VP]),
    ("mdt_op_scale_residual_fwd", _I32, [_VP, _VP, _VP, _VP, _I64, _I32, _VP]),
    ("mdt_op_scale_residual_bwd_scratch", _I64, [_I64, _I32]),
    ("mdt_op_scale_residual_bwd", _I32, [_VP, _VP, _VP, _VP, _VP, _I64, _I32, _VP, _VP]),
    ("mdt_op_scale_residual_rms_fwd", _I32, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _I32, _F, _VP]),
    ("mdt_op_scale_residual_rms_bwd_scratch", _I64, [_I64, _I32]),
    ("mdt_op_scale_residual_rms_bwd", _I32, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _I32, _F, _VP, _VP]),
    ("mdt_op_swiglu_fwd", _I32, [_VP, _VP, _I64, _I32, _VP]),
    ("mdt_op_swiglu_bwd", _I32, [_VP, _VP, _VP, _I64, _I32, _VP]),
    ("mdt_op_attn_mid_fwd", _I32, [_VP, _I64, _VP, _I64, _I64, _I32, _I32, _I32, _F, _VP]),
    ("mdt_op_attn_mid_bwd", _I32, [_VP, _I64, _VP, _I64, _VP, _I64, _VP, _I64, _I64, _I32, _I32, _I32, _F, _VP]),
    ("mdt_op_patch_mse_fwd", _I32, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _I32, _I32, _I32, _I32, _VP]),
    ("mdt_op_patch_mse_bwd", _I32, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _I32, _I32, _I32, _I32, _VP]),
    ("mdt_op_infonce_scratch", _I64, [_I64, _I64]),
    ("mdt_op_infonce", _I32, [C.POINTER(InfoNCEArgs), _VP]),
]

_lock = threading.Lock()
_lib = None


def library_path() -> str:
    return _build.LIB


def load() -> C.CDLL:
    """Load (building with hipcc if the in-tree .so is missing or stale).  Raises if neither is possible."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.LIB
        override = os.environ.get("MDT_HIP_LIB")  # tuning A/B runs: an experimental build of the same sources
        if override:
            path = override
        elif not os.path.exists(path):
            # Only a MISSING library is built here (atomically); a stale one is rebuilt by __graft_entry__.build(),
            # never implicitly at import time -- N ranks importing at once must not race on the same file.
            try:
                path = _build.build_library()
            except Exception as e:  # no hipcc and no prebuilt library: there is nothing to run
                raise RuntimeError(
                    "libmdt_hip.so is missing and could not be built; the MDT hot path has no CPU/eager "
                    f"fallback ({e})") from e
        # torch FIRST: its wheels bundle their own libamdhip64 / HSA runtime.  If this library is loaded before torch (as
        # __graft_entry__.build() followed by smoke() in ONE process did), the process ends up with two HIP runtimes and the one
        # this library is bound to reports "no ROCm-capable device" at its first hipMalloc.
        try:
            import torch  # noqa: F401
        except ImportError:  # a torch-free client of the C ABI (tests/c_client) has only one runtime anyway
            pass
        lib = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(status: int) -> None:
    if status != 0:
        raise MDTHipError(status, load().mdt_last_error().decode("utf-8", "replace"))


def call(fn, *args) -> None:
    """``check(fn(*args))`` with one retry after ``torch.cuda.empty_cache()`` when a hipMalloc inside the library failed:
    weights, workspace and tapes are raw HIP allocations, so memory that torch's caching allocator holds but does not use
    looks exhausted to them although it is free."""
    status = fn(*args)
    if status != 0:
        msg = load().mdt_last_error().decode("utf-8", "replace")
        if "hipMalloc" in msg or "out of memory" in msg.lower():
            import torch
            torch.cuda.empty_cache()
            status = fn(*args)
            if status == 0:
                return
            msg = load().mdt_last_error().decode("utf-8", "replace")
        raise MDTHipError(status, msg)
