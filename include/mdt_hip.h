/*
 * mdt_hip.h -- C ABI of libmdt_hip.so: the MI355X (gfx950) implementation of MDT's diffusion-transformer
 * action-denoising hot path (EDM-preconditioned score network + multi-step sampler loop).
 *
 * The reference (intuitive-robots/mdt_policy) is pure Python and has NO FFI boundary for this path; the
 * operator API the path sits behind is Python.  Each entry point below therefore cites
 * the reference Python call it stands under; the Python facade in mdt_policy_amd/models/ presents the
 * reference's own class / function signatures on top of this ABI (binding shown in INTEGRATION.md).
 *
 * Conventions
 *   - C linkage, POD arguments only: raw pointers, int64 sizes, an opaque handle, a hipStream_t passed
 *     as void*.  No exceptions cross the boundary: every call returns an mdt_status; the message of the
 *     last failure on the calling thread is available from mdt_last_error().
 *   - All tensors are fp32, row-major, contiguous, batch-major (B, T, C); data pointers are DEVICE
 *     pointers on the handle's device and must be 16-byte aligned, except where marked "host".
 *   - All work is enqueued on the caller's stream (e.g. torch.cuda.current_stream().cuda_stream on
 *     PyTorch-ROCm); nothing synchronises the device except mdt_reserve()/first-use workspace growth,
 *     which may call hipMalloc.  Inputs are caller-owned and never written; outputs are caller-owned.
 *   - A handle is bound to one device and is not thread-safe (one handle per GPU / per process rank).
 */
#ifndef MDT_HIP_H
#define MDT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdt_model mdt_model; /* opaque */

typedef enum {
    MDT_OK = 0,
    MDT_ERR_INVALID_ARG = 1,   /* bad pointer / size / alignment / unknown parameter name            */
    MDT_ERR_UNSUPPORTED = 2,   /* configuration outside what the HIP path implements (stated in msg) */
    MDT_ERR_NOT_LOADED = 3,    /* a parameter needed by the call was never loaded                    */
    MDT_ERR_HIP = 4,           /* a HIP runtime call failed (message carries hipGetErrorString)      */
    MDT_ERR_STATE = 5          /* call sequence error, e.g. mdt_denoise_cached() without mdt_encode()  */
} mdt_status;

enum { MDT_ARCH_MDTV = 0, MDT_ARCH_MDT = 1 };
enum { MDT_MODALITY_VIS = 0, MDT_MODALITY_LANG = 1 };

/*
 * Constructor fields of the score network.  Mirrors the Hydra kwargs of
 *   mdt.models.networks.mdtv_transformer.MDTVTransformer.__init__  (reference mdtv_transformer.py:38-68,
 *   conf/model/model/mdtv_transformer.yaml:6-35) and
 *   mdt.models.networks.mdt_transformer.MDTTransformer.__init__    (reference mdt_transformer.py:39-68),
 * plus GCDenoiser's sigma_data (reference score_wrappers.py:26-29).  Dropout probabilities are not part
 * of the ABI: this path is the eval-mode forward.
 */
typedef struct {
    int32_t arch;                 /* MDT_ARCH_MDTV | MDT_ARCH_MDT                                   */
    int32_t embed_dim;            /* d; multiple of 16, <= 512                                      */
    int32_t n_heads;              /* head dim d/n_heads in {16,32,48,64}                            */
    int32_t n_enc_layers;
    int32_t n_dec_layers;
    int32_t action_dim;           /* <= 16                                                          */
    int32_t obs_dim;              /* multiple of 16                                                 */
    int32_t goal_dim;             /* multiple of 16                                                 */
    int32_t n_obs_token;          /* MDT-V: state tokens per sample (3); MDT: ignored (static+gripper) */
    int32_t goal_seq_len;         /* 1                                                              */
    int32_t action_seq_len;       /* Ta <= 16                                                       */
    int32_t use_mlp_goal;         /* goal_emb / lang_emb are Linear-GELU-Linear                     */
    int32_t use_modality_encoder; /* separate lang_emb                                              */
    int32_t use_abs_pos_emb;      /* MDT only: pos_emb added to the encoder tokens                  */
    int32_t use_rot_embed;        /* RoPE on q/k (rot dim 32, theta 1e4), position_embeddings.py:83 */
    int32_t use_ada_conditioning; /* 1: sigma conditions the decoder blocks (adaLN-Zero, shipped configs);
                                     0: sigma embedding is the first ENCODER token, plain Block decoder
                                        (mdtv_transformer.py:296-297, transformer_blocks.py:460-506)  */
    int32_t use_noise_encoder;    /* with use_ada_conditioning: NoiseBlock (ln(x)+c) instead of
                                     ConditionedBlock (transformer_blocks.py:312-341, :533-544)      */
    int32_t linear_output;        /* 1: action_pred = Linear(d, A); 0: Linear(d, h) -> GELU -> Linear(h, A), h = 100
                                   * (MDTV, mdtv_transformer.py:178-185) / h = d (MDT, mdt_transformer.py:170-177) */
    int32_t bias;                 /* reference 'bias' flag: biases on c_proj / MLP / LayerNorms     */
    float   sigma_data;           /* GCDenoiser.sigma_data                                          */
    int32_t no_goal_conditioning; /* 1: constructor kwarg goal_conditioned=False.  MDTV: the goal token FOLLOWS the
                                     state tokens (mdtv_transformer.py:284-299); MDT: no goal token at all, which
                                     the reference can only run with use_ada_conditioning=0 (mdt_transformer.py:326-334) */
    int32_t proprio_dim;          /* MDT-V: width of state['state_obs'] (1..16; conf/model/model/mdtv_transformer.yaml:11) */
    int32_t use_proprio;          /* MDT-V: 1 = every call carries state['state_obs'] (B, 1, proprio_dim), passed as
                                     `tokens2`: proprio_emb (Linear(p, 2d), Mish, Linear(2d, d)) embeds it into one
                                     more context token BEHIND the state tokens, and a goal_conditioned=False model
                                     then has no goal token at all (mdtv_transformer.py:260-266, 284-299).  The
                                     reference decides per call ('state_obs' in states); a handle is built for one
                                     of the two context layouts, the facade keeps one handle per layout.          */
} mdt_config;

/* Human-readable message of the last failing call on this thread ("" if none). */
const char *mdt_last_error(void);

/* Library version string and the offload architecture it was compiled for ("gfx950"). */
const char *mdt_version(void);

/* hydra.utils.instantiate(cfg.model) -> GCDenoiser.__init__ -> MDTVTransformer.__init__
 * (reference score_wrappers.py:26-29, mdtv_transformer.py:38-195).  Allocates the packed weight arena on
 * the current HIP device.  Parameters start unloaded. */
mdt_status mdt_create(const mdt_config *cfg, mdt_model **out);
mdt_status mdt_destroy(mdt_model *m);

/* Enumeration of the parameters the forward path READS, in the reference's state_dict order (the
 * checkpoint / positional-EMA contract, reference mdt/evaluation/utils.py:92-103).  Names are GCDenoiser
 * state_dict keys ("inner_model.…").  Parameters the reference carries but never reads on this path
 * (pos_emb in MDT-V, proprio_emb.* unless use_proprio, *.rotary_pos_emb.freqs; reference mdtv_transformer.py:105,160-164,
 * 260-266) are not enumerated; mdt_load_param() accepts and ignores them. */
int64_t     mdt_param_count(const mdt_model *m);
const char *mdt_param_name(const mdt_model *m, int64_t index);
int64_t     mdt_param_numel(const mdt_model *m, int64_t index);

/* load_state_dict / optimizer step: copy one fp32 parameter (reference layout, e.g. Linear weight
 * (out,in) row-major) from `src` (host OR device pointer) into the library's MFMA-fragment-packed
 * arena.  Stream-ordered on `stream`; a host `src` must stay valid until the stream has passed. */
mdt_status mdt_load_param(mdt_model *m, const char *name, const float *src, int64_t numel, void *stream);

/* The same for n parameters at once -- a whole load_state_dict, or the re-upload of every parameter an optimizer step
 * changed -- as ONE kernel launch for all device-resident sources (host sources take the mdt_load_param path).  A
 * training loop calls this once per step. */
mdt_status mdt_load_params(mdt_model *m, int32_t n, const char *const *names, const float *const *srcs,
                           const int64_t *numels, void *stream);

/* Pre-size the workspace for batches up to max_batch (avoids hipMalloc later, e.g. before graph capture). */
mdt_status mdt_reserve(mdt_model *m, int64_t max_batch);

/* Number of times the workspace was (re)allocated.  A captured HIP graph of a sampler call holds workspace addresses: it
 * stays valid while this number does not change (mdt_reserve the largest batch first, capture afterwards).  It also moves when a
 * parameter load during training leaves the bf16 split weight images of the large-batch launches stale (round 6): a graph captured
 * before that would replay those launches without the refresh they need. */
int64_t mdt_ws_generation(const mdt_model *m);

/* inner_model.forward_enc_only(state, action, goal, sigma)
 * (reference mdtv_transformer.py:213-222; mdt_transformer.py:211-229 / :257-281): goal/state token
 * embedding, n_enc_layers Blocks, final LayerNorm.  Also projects the per-decoder-block cross-attention
 * K/V once and keeps ctx + K/V cached in the handle for mdt_denoise_cached()/samplers.
 *   sigma  : (B,) device.  Read only when use_ada_conditioning == 0 (the sigma embedding is then the first
 *            context token, so the cached context is only valid for that sigma); may be NULL otherwise.
 *   tokens : MDT-V state['state_images'] (B, n_obs_token, obs_dim); MDT state['static'] (B,1,obs_dim)
 *   tokens2: MDT state['gripper'] (B,1,obs_dim); MDT-V: state['state_obs'] (B,1,proprio_dim) when the handle was
 *            created with use_proprio, else NULL
 *   goal   : (B, 1, goal_dim)
 *   honour_modality: 1 = pick lang_emb when modality==LANG (MDT-V always; MDT forward_enc_only),
 *                    0 = always goal_emb (MDT.forward -> enc_only_forward, mdt_transformer.py:215)
 *   ctx_out: (B, Te, d) or NULL -- the value the reference caches as inner_model.latent_encoder_emb */
mdt_status mdt_encode(mdt_model *m, const float *tokens, const float *tokens2, const float *goal,
                      int32_t modality, int32_t honour_modality, const float *sigma, int64_t batch,
                      float *ctx_out, void *stream);

/* GCDenoiser.forward(state, action, goal, sigma) given the cached context of the last mdt_encode()
 * (reference score_wrappers.py:65-80 -> mdtv_transformer.py:224-236): EDM preconditioning, sigma
 * embedding, adaLN decoder, action head.   x:(B,Ta,A)  sigma:(B,) device  out:(B,Ta,A).
 * flags: MDT_RAW_OUTPUT returns the network output F instead of F*c_out + x*c_skip (GCDenoiser.loss);
 *        MDT_RAW_INPUT feeds x to action_emb without the c_in scaling (inner_model.forward_dec_only);
 *        MDT_SIGMA_SCALAR: `sigma` points to ONE float shared by the whole batch (what every sampler passes:
 *        sigmas[i] * s_in) -- one sigma-embedding / adaLN row instead of B, broadcast by the kernels. */
enum { MDT_RAW_OUTPUT = 1, MDT_RAW_INPUT = 2, MDT_SIGMA_SCALAR = 4 };
mdt_status mdt_denoise_cached(mdt_model *m, const float *x, const float *sigma, int64_t batch,
                              int32_t flags, float *out, void *stream);

/* model(state, action, goal, sigma) exactly as the samplers call it (reference gc_sampling.py:945):
 * mdt_encode + mdt_denoise_cached in one call ("as written": the encoder is re-run). */
mdt_status mdt_forward(mdt_model *m, const float *tokens, const float *tokens2, const float *goal,
                       int32_t modality, const float *x, const float *sigma, int64_t batch,
                       float *out, float *ctx_out, void *stream);

/* sample_ddim(model, state, action, goal, sigmas) (reference gc_sampling.py:922-951) as ONE enqueue:
 * encoder + cross K/V once, sigma-embedding/adaLN vectors for all steps once, then n_steps decoder
 * evaluations with the update x <- (s_{i+1}/s_i) x - expm1(-h_i) den fused into the action-head kernel.
 * (use_ada_conditioning == 0: the encoder depends on sigma and runs inside the step loop, as in the reference;
 * ctx_out then receives the LAST step's context, which is what the reference leaves in latent_encoder_emb.)
 *   x_T    : (B, Ta, A) initial noisy actions (already multiplied by sigma_max, mdtv_agent.py:546)
 *   sigmas : HOST array of n_steps+1 floats (get_sigmas_* output, last entry normally 0)
 *   out    : (B, Ta, A) sampled actions;  ctx_out: optional (B,Te,d) latent_encoder_emb
 * Arguments are checked before anything is enqueued (an invalid-argument return leaves no device work behind).
 * Under stream capture: the HOST schedule travels by value in the first kernel's arguments, so a captured call bakes the
 * schedule of capture time into the graph (a replay does not re-read `sigmas`); capture mdt_sample_ddim_dev, whose schedule is
 * read from device memory at every replay, when the schedule may change between replays. */
mdt_status mdt_sample_ddim(mdt_model *m, const float *tokens, const float *tokens2, const float *goal,
                           int32_t modality, const float *x_T, const float *sigmas_host, int32_t n_steps,
                           int64_t batch, float *out, float *ctx_out, void *stream);

/* The same call with the noise schedule in DEVICE memory, as MDTVAgent.get_noise_schedule builds it
 * (mdtv_agent.py:660-667: `get_sigmas_*(..., self.device)`): the per-step scalars t = -ln(sigma), h, sigma ratio and
 * -expm1(-h) (gc_sampling.py:946-950) are computed by a device kernel, nothing is copied to the host and the call never
 * synchronises.  sigmas_dev: n_steps + 1 floats on the model's device. */
mdt_status mdt_sample_ddim_dev(mdt_model *m, const float *tokens, const float *tokens2, const float *goal,
                               int32_t modality, const float *x_T, const float *sigmas_dev, int32_t n_steps,
                               int64_t batch, float *out, float *ctx_out, void *stream);

/* GCDenoiser.loss(state, action, goal, noise, sigma) forward value, eval mode (reference
 * score_wrappers.py:45-63): noised = a + n*sigma; F = inner(noised*c_in); target = (a - c_skip*noised)/c_out;
 * loss = mean((F - target)^2) over all B*Ta*A elements.  loss_out: 1 float (device); model_output: (B,Ta,A). */
mdt_status mdt_loss_fwd(mdt_model *m, const float *tokens, const float *tokens2, const float *goal,
                        int32_t modality, const float *action, const float *noise, const float *sigma,
                        int64_t batch, float *loss_out, float *model_output, float *ctx_out, void *stream);

/* Algorithmic FLOPs of one sampler call per action chunk (SURVEY.md 8(d) accounting: 2MNK per Linear,
 * full score matrices, encoder + cross K/V once, n_steps decoder evaluations). */
double mdt_flops_per_chunk(const mdt_model *m, int32_t n_steps);

/* Host helper (no GPU work): pyhash.fnv1_32 as the reference's harness uses it to derive deterministic
 * validation window sizes and environment seeds (mdt/datasets/base_dataset.py:20,37; mdt/evaluation/utils.py:17,305).
 * FNV-1, 32 bit: for each byte  h = h * 0x01000193; h ^= byte  (pyhash-0.9.3/src/fnv/hash_32.c:91-113), started from
 * `seed` -- pyhash passes its seed (default 0) as the initial value (src/FNV1.h:36-39, src/Hash.h:113,167), NOT the
 * standard offset basis 0x811c9dc5.  Chaining: pass the previous result as `seed`. */
uint32_t mdt_fnv1_32(const void *buf, uint64_t len, uint32_t seed);

/* Where the library's LARGE, batch-sized buffers come from: workspaces, training tapes and backward scratch (the weight
 * arenas and small tables stay hipMalloc'ed).  By default they are raw hipMalloc's; a host framework with its own caching
 * allocator (PyTorch) would then see "out of memory" while its own pool sits on gigabytes of cached-but-free blocks -- or the
 * other way round.  With an allocator installed the buffers live in the host's pool: `alloc(bytes, user)` returns a device
 * pointer (256-byte aligned, on the current device) or NULL when it cannot; `free_(ptr, user)` releases one -- the library
 * calls it only after hipDeviceSynchronize() (growing a buffer) or from the destroy functions.  NULL, NULL restores
 * hipMalloc / hipFree; buffers are released through whatever allocated them.  The switch is PROCESS-WIDE and takes effect for
 * every later allocation of every component (denoiser handles, Perceiver resampler, contrastive head, training tapes), whenever
 * it is made: buffers that already exist stay with the allocator they came from.  (The Python facade installs
 * torch.cuda.caching_allocator_alloc / _delete on every engine construction -- idempotent.) */
typedef void *(*mdt_alloc_fn)(size_t bytes, void *user);
typedef void (*mdt_free_fn)(void *ptr, void *user);
mdt_status mdt_set_allocator(mdt_alloc_fn alloc, mdt_free_fn free_, void *user);

/* Host shutdown: restore hipMalloc / hipFree AND forget the release callbacks of the buffers the installed allocator handed
 * out so far -- a later destroy call then DROPS such a buffer instead of calling `free_` (the host's pool is going away with
 * the process; its callbacks may already be gone).  The Python facade calls it from an atexit hook, so that a handle whose
 * finaliser runs during interpreter teardown never calls into a freed ctypes closure. */
mdt_status mdt_allocator_detach(void);

#ifdef __cplusplus
}
#endif
#endif /* MDT_HIP_H */
