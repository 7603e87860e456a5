/*
 * mdt_resampler.h -- C ABI of the Perceiver resampler, the module that turns the two cameras' Voltron patch
 * tokens into the `state_images` tokens the MDT-V denoiser's encoder consumes.
 * Same library (libmdt_hip.so), same conventions as mdt_hip.h: fp32, row-major, 16-byte aligned device
 * pointers, work enqueued on the caller's HIP stream, mdt_status + mdt_last_error().
 *
 * Reference interface replaced (paths relative to the reference checkout):
 *   mdt/models/networks/transformers/perceiver_resampler.py:85-162  PerceiverResampler.__init__/forward
 *   mdt/models/networks/transformers/perceiver_resampler.py:11-82   PerceiverAttentionLayer
 *   mdt/models/networks/transformers/utils.py:16-29                 feed_forward_layer
 *   call site: mdt/models/mdtv_agent.py:90-97 (construction), :392-404 (compute_voltron_embeddings)
 */
#ifndef MDT_RESAMPLER_H
#define MDT_RESAMPLER_H

#include <stdint.h>

#include "mdt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdt_resampler mdt_resampler; /* opaque */

/* Constructor kwargs of PerceiverResampler (perceiver_resampler.py:88-99; conf/model/mdtv_agent.yaml:27-32:
 * dim 384, depth 6, heads 8, dim_head 64, num_time_embeds 1, num_latents 3). */
typedef struct {
    int32_t dim;             /* token width (a multiple of 16, <= 512)                          */
    int32_t depth;           /* number of (attention, feed-forward) layer pairs                 */
    int32_t dim_head;        /* 64 (the reference default and the shipped config)               */
    int32_t heads;
    int32_t num_latents;     /* learnt query tokens = output tokens per sample (1..16)          */
    int32_t num_time_embeds; /* rows of time_pos_emb; forward() accepts up to this many frames  */
    int32_t ff_mult;         /* feed-forward inner width = ff_mult * dim                        */
    int32_t activation;      /* 0 = gelu (exact erf); the other reference choices are not built */
} mdt_resampler_config;

/* PerceiverResampler.__init__: allocates the packed parameter arena.  Parameters are addressed by their
 * reference state_dict names: "latents", "time_pos_emb", "layers.{i}.0.norm_media.weight|bias",
 * "layers.{i}.0.norm_latents.weight|bias", "layers.{i}.0.to_q|to_k|to_v|to_out.weight",
 * "layers.{i}.1.0.weight|bias" (LayerNorm), "layers.{i}.1.1.weight", "layers.{i}.1.3.weight", "norm.weight|bias". */
mdt_status mdt_resampler_create(const mdt_resampler_config *cfg, mdt_resampler **out);
mdt_status mdt_resampler_destroy(mdt_resampler *r);

/* Enumerate / upload parameters (state_dict order of the reference module). src: host or device fp32. */
int64_t mdt_resampler_param_count(const mdt_resampler *r);
const char *mdt_resampler_param_name(const mdt_resampler *r, int64_t i);
int64_t mdt_resampler_param_numel(const mdt_resampler *r, int64_t i);
mdt_status mdt_resampler_load_param(mdt_resampler *r, const char *name, const float *src, int64_t numel,
                                    void *stream);

/* PerceiverResampler.forward(x_f, mask) (perceiver_resampler.py:124-162).
 *   x_f  : (B, T, n, dim) media tokens, T <= num_time_embeds frames of n tokens each; T*n + num_latents <= 4096
 *   mask : (B, T) bytes (torch.bool) or NULL -- scales the frame's time embedding, as the reference does
 *   out  : (B, num_latents, dim) */
mdt_status mdt_resampler_forward(mdt_resampler *r, const float *x_f, const uint8_t *mask, int64_t batch,
                                 int32_t n_frames, int32_t n_tokens, float *out, void *stream);

/* ---- training (PerceiverResampler is a trainable module of the agent, mdtv_agent.py:90-97) ----
 * mdt_resampler_train_prepare: allocates the transposed weight images; parameters must be uploaded again after it.
 * Gradients come back in one flat buffer: parameter i at [mdt_resampler_grad_offset(i), + numel), reference layout.
 * mdt_resampler_forward_train keeps a tape (several may be alive: the agent resamples once per modality batch before
 * its single backward); mdt_resampler_backward ACCUMULATES into `grads` and optionally returns d(x_f). */
mdt_status mdt_resampler_train_prepare(mdt_resampler *r);
int64_t mdt_resampler_grad_numel(const mdt_resampler *r);
int64_t mdt_resampler_grad_offset(const mdt_resampler *r, int64_t i);
mdt_status mdt_resampler_forward_train(mdt_resampler *r, const float *x_f, const uint8_t *mask, int64_t batch,
                                       int32_t n_frames, int32_t n_tokens, float *out, int32_t *tape, void *stream);
mdt_status mdt_resampler_backward(mdt_resampler *r, int32_t tape, const float *g_out, float *grads, float *d_x_f,
                                  void *stream);
mdt_status mdt_resampler_tape_release(mdt_resampler *r, int32_t tape);

/* Algorithmic FLOPs of one forward() per sample (2 per multiply-add). */
double mdt_resampler_flops(const mdt_resampler *r, int32_t n_frames, int32_t n_tokens);

#ifdef __cplusplus
}
#endif
#endif /* MDT_RESAMPLER_H */
