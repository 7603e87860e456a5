/*
 * mdt_hip_train.h -- C ABI of the TRAINING path of the MDT denoiser on MI355X:
 * GCDenoiser.loss with saved activations, its backward (gradients of every parameter in the reference's
 * state_dict layout, and of the encoder inputs), and the kernel-level entry points the parity tests pin against
 * torch.autograd.  Same library and conventions as mdt_hip.h.
 *
 * Reference semantics replaced: torch.autograd through
 *   mdt/models/edm_diffusion/score_wrappers.py:45-63          GCDenoiser.loss
 *   mdt/models/networks/mdtv_transformer.py:208-236           MDTVTransformer.forward (enc + dec)
 *   mdt/models/networks/transformers/transformer_blocks.py    LayerNorm :29-38, Attention :119-158, MLP :161-180,
 *                                                             Block :209-214, ConditionedBlock :291-309
 * as driven by MDTVAgent.training_step -> diffusion_loss (mdt/models/mdtv_agent.py:222-262, :508-521).
 */
#ifndef MDT_HIP_TRAIN_H
#define MDT_HIP_TRAIN_H

#include <stdint.h>

#include "mdt_hip.h"
#include "mdt_hip_ops.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- model level ---------------------------- */

/* Allocate what training needs on top of inference: transposed packed images of every Linear weight (for
 * dX = dY W) and the raw copies the narrow layers read.  Parameters must be (re-)uploaded with mdt_load_param
 * AFTER this call (it invalidates the loaded flags).  Supported: use_ada_conditioning=1, use_noise_encoder=0,
 * use_rot_embed=0 (the shipped configurations); others return MDT_ERR_UNSUPPORTED. */
mdt_status mdt_train_prepare(mdt_model *m);

/* Total floats of the flat gradient buffer = sum of mdt_param_numel(i); parameter i's gradient occupies
 * [mdt_grad_offset(i), +mdt_param_numel(i)) in the parameter's reference (state_dict) layout. */
int64_t mdt_grad_numel(const mdt_model *m);
int64_t mdt_grad_offset(const mdt_model *m, int64_t i);

/* A tape holds the activations one forward saved for its backward.  Several tapes may be alive at once (the
 * reference's training step runs model.loss(...) and model.forward_context_only(...) before one backward). */
typedef int32_t mdt_tape_id;

/* GCDenoiser.loss(state, action, goal, noise, sigma), eval-mode arithmetic (no dropout), keeping a tape.
 * Outputs as mdt_loss_fwd; *tape receives the handle for mdt_train_loss_bwd / mdt_tape_release. */
mdt_status mdt_train_loss_fwd(mdt_model *m, const float *tokens, const float *tokens2, const float *goal,
                              int32_t modality, const float *action, const float *noise, const float *sigma,
                              int64_t batch, float *loss_out, float *model_output, float *ctx_out,
                              mdt_tape_id *tape, void *stream);

/* Backward of mdt_train_loss_fwd.
 *   g_loss : device scalar dL/d(loss) or NULL (= 1)
 *   g_ctx  : (B, Te, d) gradient arriving at latent_encoder_emb from other losses, or NULL
 *   grads  : flat gradient buffer (mdt_grad_numel floats); ACCUMULATED into (zero it first for a fresh step)
 *   d_tokens / d_tokens2 / d_goal : gradients of the encoder inputs (same shapes as the inputs) or NULL
 * The tape stays valid until mdt_tape_release. */
mdt_status mdt_train_loss_bwd(mdt_model *m, mdt_tape_id tape, const float *g_loss, const float *g_ctx,
                              float *grads, float *d_tokens, float *d_tokens2, float *d_goal, void *stream);

/* forward_context_only with a tape (encoder only), and its backward given dL/d(ctx). */
mdt_status mdt_train_encode_fwd(mdt_model *m, const float *tokens, const float *tokens2, const float *goal,
                                int32_t modality, int32_t honour_modality, int64_t batch, float *ctx_out,
                                mdt_tape_id *tape, void *stream);
mdt_status mdt_train_encode_bwd(mdt_model *m, mdt_tape_id tape, const float *g_ctx, float *grads,
                                float *d_tokens, float *d_tokens2, float *d_goal, void *stream);

mdt_status mdt_tape_release(mdt_model *m, mdt_tape_id tape);

/* ---------------------------------------------------------------- kernel level --------------------------- */

/* Pack the TRANSPOSE of a row-major (rows, cols; ld) matrix into a fragment-packed (N' = cols, K' = k_total) image
 * at k offset k_off: src[r][c] -> (n' = c, k' = k_off + r).  cols, k_total multiples of 16. */
mdt_status mdt_op_pack_weight_t(const float *src, int64_t rows, int64_t cols, int64_t ld, float *packed,
                                int64_t k_off, int64_t k_total, void *stream);

/* LayerNorm (+ modulate) forward keeping (mean, rstd) per row. */
typedef struct {
    const float *x;                 /* (M, D) */
    const float *w, *b;             /* (D); b may be NULL */
    const float *mod;               /* NULL or modulation rows: row (m / rows_per_sample) * mod_stride */
    int64_t mod_stride;
    int32_t shift_off, scale_off, rows_per_sample;
    float *out;                     /* (M, D): shift + (xhat * w + b) * scale */
    float *stats;                   /* (M, 2) or NULL */
    int32_t M, D;
} mdt_ln_train_args;
mdt_status mdt_op_ln_fwd_train(const mdt_ln_train_args *a, void *stream);

/* LayerNorm (+ modulate) backward, one workgroup per sample. */
typedef struct {
    const float *x;                 /* (B*rps, D) forward input */
    const float *stats;             /* (B*rps, 2) from the forward */
    const float *w, *b;             /* b may be NULL */
    const float *mod;               /* NULL or per-sample modulation rows (stride mod_stride) */
    int64_t mod_stride;
    int32_t shift_off, scale_off;   /* offsets in the mod row AND in the d_mod row */
    const float *dh; int64_t ld_dh; /* gradient of the forward output */
    float *dx;                      /* (B*rps, D) */
    int32_t accumulate;             /* 1: dx += (the residual path's gradient is already there) */
    float *d_mod; int64_t d_mod_stride; /* per-sample rows receiving d_shift / d_scale, or NULL */
    float *pw, *pb;                 /* (B, D) per-sample partials of d_w / d_b (pb may be NULL) */
    int32_t B, rows_per_sample, D;
} mdt_ln_bwd_args;
mdt_status mdt_op_ln_bwd(const mdt_ln_bwd_args *a, void *stream);

/* softmax attention backward for Tq, Tk <= 16 (no RoPE). */
typedef struct {
    const float *q; int64_t ldq;
    const float *k; const float *v; int64_t ldkv;
    const float *d_out; int64_t ld_do;    /* gradient of the merged-head attention output (B*Tq, H*hd) */
    float *dq; int64_t ld_dq;
    float *dk; float *dv; int64_t ld_dkv;
    int32_t accumulate_kv;                /* 1: dk / dv += */
    int32_t B, H, hd, Tq, Tk, causal;
} mdt_attn_bwd_args;
mdt_status mdt_op_attn_bwd(const mdt_attn_bwd_args *a, void *stream);

/* out = act(u) ; du = dy * act'(u)   (MDT_ACT_*) */
mdt_status mdt_op_act_fwd(const float *u, float *out, int64_t n, int32_t act, void *stream);
mdt_status mdt_op_act_bwd(const float *u, const float *dy, float *du, int64_t n, int32_t act, void *stream);

/* d_a = gate[sample] * d_x ; d_gate[sample] = sum over the sample's rows of d_x * a */
mdt_status mdt_op_gate_bwd(const float *dx, const float *a, const float *gate, int64_t gate_stride,
                           int32_t rows_per_sample, float *da, float *dgate, int64_t dgate_stride, int32_t B,
                           int32_t D, void *stream);

/* out[n] (+)= sum_m X[m][n] */
mdt_status mdt_op_colsum(const float *X, int64_t ldx, int64_t M, int64_t N, float *out, int32_t accumulate,
                         void *stream);

/* Backward of out = X @ W^T (+ bias) through the fp32-MFMA GEMM:
 *   dW (N, K) (+)= dY^T X ;  dbias (N) (+)= colsum(dY) ;  dX (M, K) (+)= dY @ W   (needs Wt = packed image of W^T)
 * scratch: at least (N + K) * round_up(M, 16) floats. */
typedef struct {
    const float *X; int64_t ldx;     /* (M, K) forward input */
    const float *dY; int64_t ldy;    /* (M, N) */
    const float *Wt;                 /* packed (N' = K, K' = N) image of W^T, or NULL when dX is NULL */
    float *dW;                       /* (N, K) row-major or NULL */
    float *dbias;                    /* (N) or NULL */
    float *dX; int64_t ldxo;         /* (M, K) or NULL */
    int32_t accumulate_dw, accumulate_dx;
    int32_t M, N, K;
    float *scratch;
} mdt_linear_bwd_args;
mdt_status mdt_op_linear_bwd(const mdt_linear_bwd_args *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDT_HIP_TRAIN_H */
