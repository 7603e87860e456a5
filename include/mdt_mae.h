/*
 * mdt_mae.h -- kernel-level ops behind the masked generative foresight (MGF) head, the reference's
 * MaskedTransformerImgDecoder (mdt/models/img_generation/masked_transformer_decoder.py:68-262; built by the agent at
 * mdt/models/mdtv_agent.py:99 and called on `latent_encoder_emb` at :411-421; shipped configuration
 * conf/model/img_gen/masked_transformer.yaml: 6 blocks, d = 192, 8 heads of 24, 2 x 49 patch tokens + 4 context tokens).
 *
 * Its transformer blocks are `voltron.models.util.transformer.Block(do_rms_norm, do_swish_glu, do_layer_scale)` -- an
 * un-vendored dependency (requirements.txt:20, no pinned version): restated from the published Voltron
 * code, PARITY UNPINNED for the block internals; the reference's own part (masking, token assembly, loss) is pinned by
 * tests/golden/g15_mae_*.npz.
 *
 * The host mirror (mdt_policy_amd/models/img_generation/masked_transformer_decoder.py) runs every Linear on the fp32-MFMA
 * GEMM (mdt_op_gemm forward, mdt_op_linear_bwd backward) and the row / token kernels below; gathers, residual adds and
 * the loss are PyTorch-ROCm glue.  fp32, row-major, device pointers 16-byte aligned.
 */
#ifndef MDT_MAE_H
#define MDT_MAE_H

#include <stdint.h>
#include "mdt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* voltron RMSNorm (the reference carries the same class at transformer_blocks.py:43-51):
 *   y = x / max(||x||_2 * D^-1/2, eps) * g          rows M, D <= 512 */
mdt_status mdt_op_rms_fwd(const float *x, const float *g, float *out, int64_t M, int32_t D, float eps, void *stream);
/* dx (+)= backward of the above; dg (D) (+)= sum over rows.  scratch: mdt_op_rms_bwd_scratch(M, D) floats. */
int64_t mdt_op_rms_bwd_scratch(int64_t M, int32_t D);
mdt_status mdt_op_rms_bwd(const float *x, const float *g, const float *dy, float *dx, int32_t accumulate_dx, float *dg,
                          int32_t accumulate_dg, int64_t M, int32_t D, float eps, float *scratch, void *stream);
/* The norm at the head of a residual branch (x feeds the norm AND the residual sum, masked_transformer_decoder.py:110-121 /
 * voltron Block): dx = d_res + backward of the norm, d_res = the gradient arriving on the residual path -- the sum autograd
 * would form with a separate elementwise launch. */
mdt_status mdt_op_rms_bwd_res(const float *x, const float *g, const float *dy, const float *d_res, float *dx, float *dg,
                              int32_t accumulate_dg, int64_t M, int32_t D, float eps, float *scratch, void *stream);

/* voltron SwishGLU (transformer_blocks.py:55-62): u = [projected | gate] (M, 2 H) -> projected * silu(gate) (M, H) */
mdt_status mdt_op_swiglu_fwd(const float *u, float *out, int64_t M, int32_t H, void *stream);
mdt_status mdt_op_swiglu_bwd(const float *u, const float *d_out, float *du, int64_t M, int32_t H, void *stream);

/* voltron LayerScale on a residual branch (Block.forward: x + layer_scale(branch)): out = x + gamma * z, rows of D (a multiple
 * of 4, <= 1024); and the branch's backward: dz = gamma * g, dgamma = sum_rows g * z (overwritten).  scratch:
 * mdt_op_scale_residual_bwd_scratch(M, D) floats. */
mdt_status mdt_op_scale_residual_fwd(const float *x, const float *z, const float *gamma, float *out, int64_t M, int32_t D,
                                     void *stream);
int64_t mdt_op_scale_residual_bwd_scratch(int64_t M, int32_t D);
mdt_status mdt_op_scale_residual_bwd(const float *g, const float *z, const float *gamma, float *dz, float *dgamma, int64_t M,
                                     int32_t D, float *scratch, void *stream);

/* The same LayerScale + residual AND the RMSNorm at the head of the next branch, one pass each way (the voltron Block the
 * reference builds its decoder from, masked_transformer_decoder.py:110-121: x = x + ls(branch(norm(x))) twice per block):
 *   x_new = x + gamma * z ;  h = RMSNorm(x_new; g_norm)
 * backward: d_res = gradient arriving at x_new on the residual path (NULL: none), d_h = gradient of h;
 *   d_x = d_res + RMSNorm'(d_h) ;  d_z = gamma * d_x ;  d_gamma = sum_rows d_x * z ;  d_gnorm = sum_rows d_h x_new / n
 * scratch: mdt_op_scale_residual_rms_bwd_scratch(M, D) floats.  D a multiple of 4, <= 512; pointers 16-byte aligned. */
mdt_status mdt_op_scale_residual_rms_fwd(const float *x, const float *z, const float *gamma, const float *g_norm, float *x_new,
                                         float *h, int64_t M, int32_t D, float eps, void *stream);
int64_t mdt_op_scale_residual_rms_bwd_scratch(int64_t M, int32_t D);
mdt_status mdt_op_scale_residual_rms_bwd(const float *x_new, const float *g_norm, const float *d_h, const float *d_res,
                                         const float *z, const float *gamma, float *d_x, float *d_z, float *d_gamma,
                                         float *d_gnorm, int64_t M, int32_t D, float eps, float *scratch, void *stream);

/* Unmasked multi-head self-attention over T <= 128 tokens (voltron Attention.forward):
 *   qkv (B*T, 3*H*hd) = q | k | v column blocks (row stride ld_qkv), head h at columns h*hd;
 *   out (B*T, H*hd) = softmax(q k^T * scale) v.   hd in {16, 24, 32, 48, 64}.  One workgroup per (sample, head). */
mdt_status mdt_op_attn_mid_fwd(const float *qkv, int64_t ld_qkv, float *out, int64_t ld_out, int64_t B, int32_t H,
                               int32_t hd, int32_t T, float scale, void *stream);
/* d_qkv (B*T, 3*H*hd; same layout, overwritten) from d_out; the probabilities are recomputed from qkv, `out` is the
 * forward's output (its rows give sum_j P_ij dP_ij = d_out_i . out_i without a second P V product). */
mdt_status mdt_op_attn_mid_bwd(const float *qkv, int64_t ld_qkv, const float *out, int64_t ld_out, const float *d_out,
                               int64_t ld_do, float *d_qkv, int64_t ld_dqkv, int64_t B, int32_t H, int32_t hd, int32_t T,
                               float scale, void *stream);

/* compute_loss of the decoder (masked_transformer_decoder.py:228-262, symmetric mask) in one pass each way:
 *   loss = 1/2 sum_{x in {0,1}} [ sum_{b,n} mask[b][n] * mean_e (rec[b][x][n][e] - target[b][x][n][e])^2 ] / sum(mask)
 * with target = patchify(imgs): element e = (ph * P + pw) * C + c of patch n = (gh, gw) is imgs[b][x][c][gh P + ph][gw P + pw]
 * (:206-213) -- read straight out of the images, no patchified copy.  rec (B, X, n, P*P*C) contiguous, imgs (B, X, C, R, R)
 * contiguous, mask (B, n) of 0 / 1.  partial: B * X * n floats of scratch; loss: 1 float; mask_sum: 1 float (kept for the
 * backward).  Deterministic (per-patch sums, then one fixed-order reduction). */
mdt_status mdt_op_patch_mse_fwd(const float *rec, const float *imgs, const float *mask, float *partial, float *loss,
                                float *mask_sum, int64_t B, int32_t X, int32_t C_, int32_t R, int32_t P, void *stream);
/* d_rec = g * d loss / d rec, g a device scalar (the incoming gradient); zero on the visible patches. */
mdt_status mdt_op_patch_mse_bwd(const float *rec, const float *imgs, const float *mask, const float *mask_sum, const float *g,
                                float *d_rec, int64_t B, int32_t X, int32_t C_, int32_t R, int32_t P, void *stream);

#ifdef __cplusplus
}
#endif
#endif
