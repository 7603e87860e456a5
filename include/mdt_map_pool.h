/*
 * mdt_map_pool.h -- C ABI of the attention-pooling head the contrastive (CLA) auxiliary loss hangs on the denoiser's
 * context tokens: MAPBlock, the `latent_proj` of ClipStyleProjection('map').
 * Same library (libmdt_hip.so), same conventions as mdt_hip.h: fp32, row-major, 16-byte aligned device pointers,
 * work enqueued on the caller's HIP stream, mdt_status + mdt_last_error().
 *
 * Reference interface replaced (paths relative to the reference checkout):
 *   mdt/models/networks/transformers/transformer_blocks.py:746-791  MAPBlock.__init__/forward
 *   mdt/models/networks/transformers/transformer_blocks.py:716-743  MAPAttention
 *   mdt/models/networks/transformers/transformer_blocks.py:42-62    RMSNorm, SwishGLU
 *   call sites: transformer_blocks.py:833-870 (ClipStyleProjection), mdt/models/mdtv_agent.py:133-138 (construction,
 *   clip_style='map'), :440-484 (compute_contrastive_loss: clip_proj on latent_encoder_emb of both modalities)
 */
#ifndef MDT_MAP_POOL_H
#define MDT_MAP_POOL_H

#include <stdint.h>

#include "mdt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdt_map_pool mdt_map_pool; /* opaque */

/* Constructor arguments of MAPBlock (transformer_blocks.py:747-756) with do_rms_norm = do_swish_glu = True (the
 * defaults, the only form the reference constructs).  The reference's MAPBlock runs its attention with
 * 2 * n_heads heads (:759); n_heads below is the CONSTRUCTOR argument (8 in ClipStyleProjection). */
typedef struct {
    int32_t n_latents;   /* learnt seed vectors = output rows per sample; 1 (ClipStyleProjection) .. 16 */
    int32_t embed_dim;   /* width of the pooled tokens (input of `projection`); multiple of 16            */
    int32_t output_dim;  /* width of everything after `projection`; multiple of 16, <= 512               */
    int32_t n_heads;     /* constructor argument; the attention uses 2 * n_heads heads                   */
    int32_t mlp_hidden;  /* int(mlp_ratio * output_dim); multiple of 16                                  */
} mdt_map_pool_config;

/* Parameters by their MAPBlock state_dict names: "latents", "projection.weight|bias", "attn_norm.g",
 * "attn.q.weight", "attn.kv.weight", "attn.proj.weight|bias", "mlp_norm.g", "mlp.0.project.weight|bias",
 * "mlp.1.weight|bias" (registration order of the reference module). */
mdt_status mdt_map_pool_create(const mdt_map_pool_config *cfg, mdt_map_pool **out);
mdt_status mdt_map_pool_destroy(mdt_map_pool *p);
int64_t mdt_map_pool_param_count(const mdt_map_pool *p);
const char *mdt_map_pool_param_name(const mdt_map_pool *p, int64_t i);
int64_t mdt_map_pool_param_numel(const mdt_map_pool *p, int64_t i);
mdt_status mdt_map_pool_load_param(mdt_map_pool *p, const char *name, const float *src, int64_t numel, void *stream);

/* MAPBlock.forward(x) (transformer_blocks.py:787-791):
 *   x   : (B, N, embed_dim) tokens to pool, N <= 16
 *   out : (B, n_latents, output_dim)   (the reference squeezes dim 1 when n_latents == 1: same memory) */
mdt_status mdt_map_pool_forward(mdt_map_pool *p, const float *x, int64_t batch, int32_t n_tokens, float *out,
                                void *stream);

/* ---- training ----
 * mdt_map_pool_train_prepare allocates the transposed weight images; parameters must be uploaded again after it.
 * Gradients come back in one flat buffer: parameter i at [mdt_map_pool_grad_offset(i), + numel), reference layout.
 * Several tapes may be alive (the agent pools the language and the vision context before one backward);
 * mdt_map_pool_backward ACCUMULATES into `grads` and optionally returns d(x). */
mdt_status mdt_map_pool_train_prepare(mdt_map_pool *p);
int64_t mdt_map_pool_grad_numel(const mdt_map_pool *p);
int64_t mdt_map_pool_grad_offset(const mdt_map_pool *p, int64_t i);
mdt_status mdt_map_pool_forward_train(mdt_map_pool *p, const float *x, int64_t batch, int32_t n_tokens, float *out,
                                      int32_t *tape, void *stream);
mdt_status mdt_map_pool_backward(mdt_map_pool *p, int32_t tape, const float *g_out, float *grads, float *d_x,
                                 void *stream);
mdt_status mdt_map_pool_tape_release(mdt_map_pool *p, int32_t tape);

/* ---- loss: InfoNCE between the pooled embeddings ----
 * MDTVAgent.clip_auxiliary_loss(image_features, lang_features, mode) (mdt/models/mdtv_agent.py:774-799): both sets
 * L2-normalised, S = exp(logit_scale) * img @ lang^T, cross entropy against the diagonal over the rows
 * (img_to_text), the columns (text_to_img) or the mean of both (symmetric).  Value and -- when the three gradient
 * pointers are given -- dL/d(image_features), dL/d(lang_features), dL/d(logit_scale) in the same enqueue. */
enum { MDT_INFONCE_SYMMETRIC = 0, MDT_INFONCE_IMG_TO_TEXT = 1, MDT_INFONCE_TEXT_TO_IMG = 2 };
typedef struct {
    const float *image_features;  /* (batch, dim)                                                   */
    const float *lang_features;   /* (batch, dim)                                                   */
    const float *logit_scale;     /* device scalar: the log temperature parameter (mdtv_agent.py:140) */
    int32_t batch, dim;           /* dim: multiple of 16; batch 1..32768 (the GLOBAL batch after the gather) */
    int32_t mode;                 /* MDT_INFONCE_*                                                    */
    float *loss;                  /* device scalar out                                                */
    float *d_image, *d_lang;      /* (batch, dim) out, or all three NULL                              */
    float *d_logit_scale;         /* device scalar out                                                */
    float *scratch;               /* mdt_op_infonce_scratch(batch, dim) floats                        */
} mdt_infonce_args;
int64_t mdt_op_infonce_scratch(int64_t batch, int64_t dim);
mdt_status mdt_op_infonce(const mdt_infonce_args *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDT_MAP_POOL_H */
