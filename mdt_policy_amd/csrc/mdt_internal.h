// mdt_internal.h -- launcher prototypes shared by mdt_kernels.hip (device code) and mdt_model.hip (host logic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdt_hip.h"
#include "mdt_hip_ops.h"

int mdt_gemm_kchunk(int K, int ln, int cap);
hipError_t mdt_launch_gemm(const mdt_gemm_args& a, hipStream_t s);
hipError_t mdt_launch_attention(const mdt_attn_args& a, const float* rope_cos, const float* rope_sin, hipStream_t s);
hipError_t mdt_launch_layernorm(const float* in, const float* w, const float* b, float* out, int M, int D,
                                hipStream_t s);
hipError_t mdt_launch_sigma_emb(const float* sigma, int64_t sstride, const float* freqs, float* out, int R, int D,
                                hipStream_t s);
hipError_t mdt_launch_action_embed(const float* x, const float* sigma, int64_t sstride, float sd, const float* Wa,
                                   const float* ba, float* y, int M, int A, int D, int rps, hipStream_t s);
hipError_t mdt_launch_head(const mdt_head_args& a, hipStream_t s);
hipError_t mdt_launch_noise_input(const float* act, const float* noise, const float* sigma, float* noised, int64_t n,
                                  int per_sample, hipStream_t s);
hipError_t mdt_launch_loss_reduce(const float* F, const float* act, const float* noised, const float* sigma, float sd,
                                  int64_t n, int per_sample, float* loss, hipStream_t s);
hipError_t mdt_launch_pack_weight(const float* w, int n_rows, int K, float* packed, int n_off, hipStream_t s);
hipError_t mdt_launch_transpose(const float* src, float* dst, int R, int Cc, hipStream_t s);
hipError_t mdt_launch_xattn_fold(const mdt_xfold_args& a, hipStream_t s);
hipError_t mdt_launch_xattn_apply(const mdt_xapply_args& a, hipStream_t s);
bool mdt_xattn_apply_supported(int D, int H, int Te, int Ta);
