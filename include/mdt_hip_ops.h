/*
 * mdt_hip_ops.h -- kernel-level entry points of libmdt_hip.so.
 *
 * These launch ONE hand-written gfx950 kernel each on caller-provided device buffers.  The model-level
 * ABI (mdt_hip.h) is built from exactly these launches; they are exported so that the parity tests can
 * pin every kernel against a PyTorch fp32 reference of the same op, and so that bench.py can time the
 * dominant kernel in isolation.  fp32 everywhere, row-major, device pointers 16-byte aligned.
 *
 * Reference ops replaced (all in mdt/models/networks/transformers/transformer_blocks.py unless noted):
 *   mdt_op_gemm       nn.Linear (+ F.layer_norm :38 / nn.LayerNorm :205 prologue, modulate :262,
 *                     nn.GELU :171 / nn.Mish / nn.SiLU :251 epilogue, gated residual :296-307)
 *   mdt_op_attention  F.scaled_dot_product_attention :142 (+ RotaryEmbedding position_embeddings.py:138)
 *   mdt_op_layernorm  LayerNorm.forward :37-38
 *   mdt_op_head       decoder.ln + action_pred (mdtv_transformer.py:233-235) + GCDenoiser.forward
 *                     combine (score_wrappers.py:79-80) + sample_ddim update (gc_sampling.py:948-950)
 *                     + next step's action_emb (mdtv_transformer.py:226)
 */
#ifndef MDT_HIP_OPS_H
#define MDT_HIP_OPS_H

#include <stdint.h>
#include "mdt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { MDT_ACT_NONE = 0, MDT_ACT_GELU = 1, MDT_ACT_MISH = 2, MDT_ACT_SILU = 3,
       MDT_ACT_SWIGLU = 4 /* only as mdt_linear_bwd_args.dx_act: the layer below is a SwishGLU, dX has 2K columns */ };
enum { MDT_HEAD_DENOISED = 0, MDT_HEAD_DDIM = 1, MDT_HEAD_RAW = 2 };

/* Number of floats of the fragment-packed image of an (N, K) Linear weight (N, K multiples of 16). */
int64_t mdt_op_packed_numel(int64_t N, int64_t K);

/* Pack rows [0, n_rows) of a row-major (n_rows, K) weight into rows [n_off, n_off+n_rows) of a packed
 * (N_total, K) image: for every 16(n) x 16(k) block, lane l = n%16 + 16*((k%16)/4) holds the 4
 * consecutive-k values W[n][k0 + 4*(l>>4) .. +3] -- the operand order of v_mfma_f32_16x16x4_f32. */
mdt_status mdt_op_pack_weight(const float *w, int64_t n_rows, int64_t K, float *packed, int64_t n_off,
                              int64_t N_total, void *stream);
/* The packed image of a SwishGLU project weight (2H, K) for mdt_gemm_args.aux_mode 3: 16-row tile 2t holds projected rows
 * [16t, 16t+16), tile 2t+1 the gate rows [H + 16t, H + 16t + 16) (voltron SwishGLU / transformer_blocks.py:54-62:
 * projected, gate = project(x).tensor_split(2, dim=-1)).  H % 16 == 0, K % 16 == 0. */
mdt_status mdt_op_pack_weight_glu(const float *w, int64_t H2, int64_t K, float *packed, void *stream);

typedef struct {
    const float *A;        /* (M, K) activations, row stride lda (multiple of 4)                      */
    int64_t lda;
    const float *Wp;       /* packed (N, K) weight                                                    */
    const float *bias;     /* (N) or NULL                                                             */
    float *out;            /* output rows, row stride ldo                                             */
    int64_t ldo;
    int32_t M, N, K;
    /* prologue: LayerNorm over K (needs K <= 512) and optional modulate shift + x*scale              */
    int32_t ln;            /* 0 = plain A, 1 = layer_norm(A) * ln_w (+ ln_b)                          */
    const float *ln_w, *ln_b;
    const float *mod;      /* NULL, or modulation table: row (m / rows_per_sample) * mod_stride       */
    int64_t mod_stride;    /* 0 broadcasts one row to the whole batch                                 */
    int32_t shift_off, scale_off;
    int32_t rows_per_sample;
    /* epilogue                                                                                        */
    int32_t act;           /* MDT_ACT_*  applied after bias                                           */
    int32_t residual;      /* 1: out = out + gate * value (gate = 1 when gate_off < 0)                */
    int32_t gate_off;      /* offset of the gate vector inside the mod row, or -1                     */
    int32_t gin, gout, goff;   /* output row = (m / gin) * gout + m % gin + goff  (1,1,0 = identity)  */
    const float *rowvec;   /* NULL or (N) vector added to every row after bias (positional embedding) */
    /* batched launch: `batch` independent products in one launch, member z reading A + z*bs_a, Wp + z*bs_w and
     * writing out + z*bs_out (floats); 0 or 1 = a single product.  Used for the split-K partial products of the
     * training path's weight gradients.                                                              */
    int32_t batch;
    int64_t bs_a, bs_w, bs_out;
    /* training: a second (M, N) operand laid out like `out` (row stride ldo, same row remap), plain prologue, no residual:
     *   aux_mode 1: aux = the value BEFORE `act` (after bias): one launch leaves both u and act(u);
     *   aux_mode 2: out = value * act'(aux)  -- `act` names the activation whose derivative is taken, the value itself is
     *               not activated (dX of the Linear that follows an activation: d_u = (dY W) * act'(u)).
     *   aux_mode 3: SwishGLU forward on the producing Linear (N = 2H columns [projected | gate]; Wp packed by
     *               mdt_op_pack_weight_glu, which interleaves the two halves tile by tile so that one wave holds a projected
     *               column and its gate): aux (M, 2H, row stride 2 * ldo) = u in its natural column order, out (M, H, row
     *               stride ldo) = projected * silu(gate).  N % 32 == 0.
     *   aux_mode 4: SwishGLU backward on the input-gradient product of the Linear that FOLLOWS it (N = H columns of
     *               d_out): aux = u (M, 2H, row stride ldo), out (M, 2H, row stride ldo) = [value * silu(gate) |
     *               value * projected * silu'(gate)].                                                              */
    const float *aux;
    int32_t aux_mode;
    /* merge-on-read (consumer side of mdt_op_mlp): a_parts >= 2 means the activation rows are the SUM of a_parts arrays
     * A, A + a_part_stride, ... (floats; same lda), added in that fixed order -- the partial slabs a fused MLP launch left;
     * needs the LayerNorm prologue (whole rows per wave) and the wide tiles (N >= 1024, M > 1400 rows).  a_merged (optional,
     * may not alias any part): the summed rows are also written there (row stride lda) by the column-0 tiles, so that
     * the next residual GEMM finds the residual stream in one place.  0 / 1 = plain A.                              */
    int32_t a_parts;
    int64_t a_part_stride;
    float *a_merged;
    /* optional (round 6): the three-way bf16 split image of the same weight (mdt_op_pack_weight_split; rows in the order of Wp's).
     * Where it is given, the LayerNorm-prologue products on the wide tiles (K = 384 or 512, N a multiple of 384, from 768 rows on
     * -- MDT_HIP_SPLIT_MIN_ROWS --, plain output rows) multiply in the split form -- six bf16 MFMA products per 32-deep step, fp32
     * accumulation: fp32's product accuracy, not the fp32 form's bits -- unless mdt_op_set_mlp_split(0).  NULL: the fp32 form. */
    const void *Wp_split;
} mdt_gemm_args;

mdt_status mdt_op_gemm(const mdt_gemm_args *args, void *stream);

/* The whole MLP sublayer of a (Conditioned)Block as ONE launch (transformer_blocks.py:160-181, 296-309):
 *     x + gate * ( act( prologue(x) @ W1^T + b1 ) @ W2^T + b2 )
 * `fc` describes the first Linear as for mdt_op_gemm (A = x rows with lda, M, K = D, N = 4D, LayerNorm (+ modulate)
 * prologue, act; `out` ignored), `proj` the second (Wp, bias, N = D, K = 4D, mod / mod_stride / gate_off /
 * rows_per_sample for the gate; A / out ignored).  A workgroup owns 32 rows and ONE 512-wide slice of the hidden layer:
 * it keeps its (32 x 512) slice of act(.) in LDS and multiplies it by the matching K-slice of W2, so the hidden layer never
 * reaches memory and the S = 4D / 512 slices of a row tile produce S partial slabs
 *     parts[0] = x + gate * (h_0 W2_0^T + b2),   parts[s] = gate * (h_s W2_s^T)        (slab s at parts + s * part_stride)
 * whose sum IN THE ORDER 0, 1, .., S-1 is the sublayer's output (deterministic: no atomics).  The consumer adds them on
 * read (mdt_gemm_args.a_parts, mdt_head_args.y_parts).  *n_parts returns S.  Needs D a multiple of 128, D <= 512, and
 * parts (S * part_stride floats, part_stride >= M * D) not aliasing x -- except S == 1, where parts may be x itself. */
mdt_status mdt_op_mlp(const mdt_gemm_args *fc, const mdt_gemm_args *proj, float *parts, int64_t part_stride,
                      int32_t *n_parts, void *stream);
/* The same launch with every contraction as a THREE-WAY bf16 SPLIT (round 6): each fp32 operand as three bf16 parts, six
 * v_mfma_f32_16x16x32_bf16 products per 32-deep step, fp32 accumulation -- fp32's product accuracy (not the bits of mdt_op_mlp) at a
 * third of its matrix-pipe time.  fc_split / proj_split: images of the two weights made by mdt_op_pack_weight_split ((4 D, D) and
 * (D, 4 D) row-major sources; 6 bytes per weight); fc->Wp / proj->Wp are not read.  The model-level entry points use it
 * from the row count of mdt_op_set_mlp_fuse_min on unless mdt_op_set_mlp_split(0) / MDT_HIP_MLP_SPLIT=0 (negative: default). */
mdt_status mdt_op_pack_weight_split(const float *w, int64_t n_rows, int64_t K, void *image, void *stream);
/* ... rows [n_off, n_off + n_rows) of a taller image (stacked weights: query | key | value); n_off a multiple of 16 */
mdt_status mdt_op_pack_weight_split_rows(const float *w, int64_t n_rows, int64_t K, void *image, int64_t n_off, void *stream);
mdt_status mdt_op_mlp_split(const mdt_gemm_args *fc, const mdt_gemm_args *proj, const void *fc_split, const void *proj_split,
                            float *parts, int64_t part_stride, int32_t *n_parts, void *stream);
void mdt_op_set_mlp_split(int32_t on);
/* Tuning / test hook: the model-level entry points run the MLP sublayer through mdt_op_mlp from `rows` rows (B * horizon)
 * on; 0 = never (the two-GEMM sequence), -1 = default (1401, or MDT_HIP_MLP_FUSE_MIN from the environment). */
void mdt_op_set_mlp_fuse_min(int32_t rows);
/* Tuning / test hook (round 5): rollout-sized model-level calls queue the products that do not depend on their neighbours in the
 * launch chain (the sigma-MLP / adaLN table of mdt_sample_ddim, the MDTV token embedding) and let each ride as extra workgroups in
 * the next split-K small-M launch.  0 = every product its own launch, 1 = on, -1 = default (on, or MDT_HIP_SIDE_JOBS from the
 * environment).  Same tiles, same order inside each product: results are bit-identical either way.
 * mdt_op_side_jobs_paired: how many launches of this process have taken such a product along so far. */
void mdt_op_set_side_jobs(int32_t on);
int64_t mdt_op_side_jobs_paired(void);
/* Measurement hook: bracket every fused-MLP launch of the model-level calls that follow (mdt_sample_ddim, mdt_forward, ...)
 * with a pair of HIP events on its stream; mdt_op_trace_mlp_read waits for them, writes up to `cap` durations in
 * MICROSECONDS (launch order) to `us`, releases the events and returns how many it wrote.  The duration of the dominant
 * kernel inside its launch chain, as a kernel trace reports it (bench.py's roofline.dominant_kernel).  Process-wide; not
 * for use under stream capture. */
void mdt_op_trace_mlp(int32_t enable);
int32_t mdt_op_trace_mlp_read(float *us, int32_t cap);
/* Behind every traced launch the hook also brackets NOTHING with a second pair of events: mdt_op_trace_mlp_read_empty returns
 * those empty brackets (microseconds; same order and count as the last mdt_op_trace_mlp_read) -- what the bracket itself costs
 * on the stream.  Launch bracket minus empty bracket = the kernel as a kernel trace's row reports it. */
int32_t mdt_op_trace_mlp_read_empty(float *us, int32_t cap);
/* Measurement hook (round 6): in stream order, one wave per XCD writes {shader-clock counter (s_memtime), constant 100 MHz counter
 * (s_memrealtime)} to out16[2 x + 0 .. 1] (device memory, 16 values, zero them first; x = the XCD the wave ran on: the shader-clock
 * counters of the eight XCDs are not synchronised).  Two stamps around a span of launches give, XCD by XCD, the average shader clock
 * the chip sustained over it: (d memtime / d memrealtime) x 100 MHz -- what bench.py reports as roofline.sustained_mhz beside the
 * 2.4 GHz the peak is quoted at. */
mdt_status mdt_op_clock_stamp(uint64_t *out16, void *stream);
/* Tuning / test hook (round 6): 1 = the K = 384 products that take the weight-stationary body (from 8192 rows on: the training step at
 * B = 1024) run its THREE-WAY bf16 SPLIT form -- every fp32 operand as three bf16 parts, six v_mfma_f32_16x16x32_bf16 products per k32
 * step with fp32 accumulation: fp32's product accuracy (not its bits) at 2.7x the fp32 matrix rate; 0 = the fp32 MFMA form;
 * negative = default (MDT_HIP_WS_SPLIT from the environment; unset = 1). */
void mdt_op_set_ws_split(int32_t on);

/* Tuning / test hook: wave schedule inside mdt_op_mlp's kernel.  Low byte = number of k-steps the second wave of every
 * SIMD starts behind the first (0 = lockstep with a workgroup barrier between the two products), | 256 = MFMA loops at
 * raised issue priority; -1 = default (18 | 256, or MDT_HIP_MLP_SKEW from the environment).  Every setting produces the
 * same bits (the K order of both products does not depend on it). */
void mdt_op_set_mlp_skew(int32_t v);

/* Tuning / test hook: force the workgroup geometry of every following GEMM launch in this process.
 * 0 = heuristic (default): up to 15 rows the split-K small-M kernel -- one workgroup per 16 columns, K divided between its
 * 8 waves --; up to 512 rows the same kernel for the products whose half-height tiling would have fewer than 60 (LayerNorm
 * prologue) / 100 (plain, <= 192 rows) / 160 (plain, <= 512 rows) tiles; up to 1400 rows the half-height tiled geometry 6;
 * beyond, the widest row-tile geometry that still fills the chip;
 * 1 = 4 waves 32x64; 2 = 8 waves 32x128; 3 = 8 waves 32x384; 4 = 8 waves 32x512; 5 = 4 waves 32x128; 6 = 4 waves 16x64;
 * 7 = 4 waves 64x128; 8 = 4 waves 32x256; 9 = 4 waves 32x192;
 * 10 / 12 / 16 = the TALL body (128-row tiles, both operands staged in LDS by LDS-DMA; plain prologue, K % 32 == 0;
 * anything else falls back to the heuristic): 4 waves 128x128 / 128x64 / 128x96; 23 = 128x64 with a loader wave, 3 stages;
 * -1 = the split-K small-M kernel wherever it applies.
 * 30 = the weight-stationary body wherever it is supported, whatever the row count.
 * Row-tile, tall and (fp32 form, mdt_op_set_ws_split(0)) weight-stationary geometries compute bit-identical results (same k order
 * per output element); the small-M kernel and the weight-stationary body's bf16 split form (the default) agree to fp32 rounding.  MDT_HIP_SMALLM_MAX / MDT_HIP_SMALLM_TILES / MDT_HIP_SMALLM_ROWS / MDT_HIP_MID_MAX (environment) move the
 * thresholds. */
void mdt_op_set_gemm_geometry(int32_t geometry);

typedef struct {
    const float *q; int64_t ldq;       /* (B*Tq, >= H*hd) query rows                                  */
    const float *k; const float *v;    /* (B*Tk, ...) key / value rows                                */
    int64_t ldkv;
    float *out; int64_t ldo;           /* (B*Tq, H*hd)                                                */
    int32_t B, H, hd, Tq, Tk;          /* hd in {16,32,48,64}; Tq, Tk <= 16                           */
    int32_t causal;                    /* 1: key j visible to query i iff j <= i (top-left aligned)   */
    int32_t rope;                      /* 1: rotate q by its position 0..Tq-1 and k by 0..Tk-1        */
} mdt_attn_args;

mdt_status mdt_op_attention(const mdt_attn_args *args, void *stream);

/* One sample's self-attention fused into its output projection (the rollout batch B = 1, where every launch is
 * latency): out (+)= gate * (softmax(q k^T / sqrt(hd)) v @ W^T + bias) for the T <= 16 rows of qkv (T, 3*K), q | k | v
 * column blocks of K = 8 * hd each.  `proj` describes the projection as for mdt_op_gemm (A is ignored, M = T; bias /
 * residual / gate as there; no LayerNorm prologue, activation or row remap).  8 heads, hd in {16,32,48,64}, no RoPE.
 * Replaces Attention.forward's SDPA + c_proj (transformer_blocks.py:142-157) for one sample.
 * Summation order: since round 5 both attention products run on the MFMA pipe inside this launch (wave = head, k order of the
 * matrix instruction), so its results differ from mdt_op_attention + mdt_op_gemm on the same operands in the last bits: within
 * 2e-5 absolute + 1e-4 relative on values of order one (tests/test_gpu_ops.py::test_fused_attention_projection_against_the_two_
 * launches_at_rollout_batches, B = 1 .. 8); the model-level entry points use it up to 32 samples (MDT_HIP_ATTN_PROJ_MAX). */
mdt_status mdt_op_attn_proj(const mdt_gemm_args *proj, const float *qkv, int64_t ldq, int32_t hd, int32_t T, int32_t causal,
                            void *stream);
/* (More than 64 samples: the same contract for a LARGE batch, M = samples * T rows, causal, residual, hd in {16, 32, 48} --
 * the attention of each 32-row tile is computed in the prologue of the tiled projection GEMM, k_attn_proj_wide; the
 * model-level entry points use it from `rows` rows on: 0 = never (attention launch + projection GEMM), -1 = default 1401.) */
void mdt_op_set_attn_wide_min(int32_t rows);

mdt_status mdt_op_layernorm(const float *in, const float *w, const float *b, float *out, int64_t M, int32_t D,
                            void *stream);

typedef struct {
    const float *y;            /* (M, D) decoder residual stream                                      */
    const float *ln_w;         /* decoder.ln.weight (D), ln_b optional                                */
    const float *ln_b;
    const float *Wp, *bp;      /* action_pred.weight (A, D) row-major UNPACKED, bias (A)              */
    const float *x;            /* (M, A) current noisy actions                                        */
    const float *sigma;        /* sigma of sample b at sigma[b * sigma_stride]                        */
    int64_t sigma_stride;
    float *out;                /* (M, A)                                                              */
    int32_t M, D, A, rows_per_sample;
    int32_t mode;              /* MDT_HEAD_*                                                          */
    const float *step;         /* DDIM: device {ratio, coef, sigma_next}                              */
    float sigma_data;
    /* optional fused embedding of the NEXT step's input: y_next = (out * c_in(sigma_next)) Wa^T + ba */
    float *y_next;             /* NULL or (M, D) (may alias y)                                        */
    const float *Wa, *ba;      /* action_emb.weight TRANSPOSED to (A, D) row-major, bias (D)          */
    int32_t no_ln;             /* 1: y is used as it is (the MLP head's hidden layer), ln_w / ln_b only have to be
                                * readable for D floats                                                */
    int32_t y_parts;           /* >= 2: the rows are the sum of y_parts arrays y, y + y_part_stride, ... (the slabs  */
    int64_t y_part_stride;     /* mdt_op_mlp left), added in that order; 0 / 1 = plain y                             */
} mdt_head_args;

mdt_status mdt_op_head(const mdt_head_args *args, void *stream);

/* Collapsed cross-attention (see mdt_kernels.hip): fold the sigma-independent context K|V into the query and
 * output projections once per sampler call ... */
typedef struct {
    const float *kv; int64_t ldkv; /* (B*Te, ...) rows holding K at column 0 and V at column D of this block   */
    const float *WqT_p;            /* fragment image of cross_att.query.weight TRANSPOSED: mdt_op_pack_weight_t(Wq, D, D, D, .., 0, D / 16) */
    const float *bq;               /* cross_att.query.bias (D)                                                 */
    const float *Wo_p;             /* fragment image of cross_att.c_proj.weight: mdt_op_pack_weight(Wo, D, D, .., 0)               */
    float *U, *Wf;                 /* out: B images of 4 H * D floats each, opaque: MFMA weight-fragment order, every head */
                                   /* padded to 4 context tokens (row p = 4 h + j; mdt_kernels.hip "Collapsed ...")        */
    float *c;                      /* out: (B, 4 H)                                                                        */
    int32_t B, H, hd, D, Te;       /* H in {4, 8}, D = H * hd <= 512 a multiple of 128, 1 <= Te <= 4                       */
} mdt_xfold_args;
mdt_status mdt_op_xattn_fold(const mdt_xfold_args *args, void *stream);

/* ... and apply the whole sublayer  y += c_proj(softmax_causal(q K^T / sqrt(hd)) V) + bo  per denoising step. */
typedef struct {
    float *y;                      /* (B*Ta, D) residual stream, updated in place                              */
    const float *ln_w, *ln_b;      /* ln3 weight / bias (D)                                                    */
    const float *U, *Wf, *c;       /* from mdt_op_xattn_fold                                                   */
    const float *bo;               /* cross_att.c_proj.bias (D) or NULL                                        */
    int32_t B, H, D, Te, Ta;       /* D a multiple of 128, Ta <= 16                                             */
    float *y_out;                  /* NULL: y is updated in place; else the new rows go here and y stays as it is */
} mdt_xapply_args;
mdt_status mdt_op_xattn_apply(const mdt_xapply_args *args, void *stream);

/* Rollout batches: mdt_op_xattn_apply and the LayerNorm(+ modulate)-prologue Linear that follows it on the same rows (mlp.c_fc,
 * transformer_blocks.py:301-307) in ONE launch (k_xattn_gemm_smallm): every 16-column workgroup of the Linear repeats the
 * sample's cross-attention on the MFMA pipe and multiplies its output rows straight out of LDS; the workgroup of column 0
 * writes the new rows to x->y_out, which must be given and differ from x->y (the other workgroups may still be reading x->y).
 * `g` as for mdt_op_gemm with g->A == x->y, lda = K = x->D, M = x->B * x->Ta, rows_per_sample = x->Ta, no residual / row remap /
 * split input.  Results equal the two launches bit for bit.  The model-level entry points use it up
 * to 2 samples (MDT_HIP_XATTN_FC_MAX_B overrides; B = 4 measured slower: 1.67 vs 1.58 ms per call). */
mdt_status mdt_op_xattn_gemm(const mdt_xapply_args *x, const mdt_gemm_args *g, void *stream);

/* The middle of a ConditionedBlock for a batch of at most one sample per compute unit, ONE launch, one workgroup per sample
 * (k_attn_xattn): mdt_op_attn_proj's contract (causal self-attention of the sample's T rows -> c_proj + gate + residual,
 * transformer_blocks.py:296-300) followed by mdt_op_xattn_apply's on the same rows (:301-305); the rows between the two
 * sublayers stay in the workgroup's LDS and x->y == proj->out is written once.  8 heads of 48 (K = N = ldo = 384, ldq = 3 K),
 * T = x->Ta <= 16, proj->M = x->B * T; results equal the two launches up to the summation order of the self-attention
 * (this form runs q k^T and P v on the MFMA pipe, one wave per head).  The model-level entry points use it
 * from 1401 rows up to 512 samples (MDT_HIP_ATTN_XATTN_MIN / MDT_HIP_ATTN_XATTN_MAX_B; mdt_op_set_attn_wide_min(0) switches
 * it off together with the tiled form). */
mdt_status mdt_op_attn_xattn(const mdt_gemm_args *proj, const float *qkv, int64_t ldq, const mdt_xapply_args *x, int32_t hd,
                             int32_t T, void *stream);

/* y = (x * c_in(sigma)) Wa^T + ba   (c_in omitted when sigma == NULL); WaT = action_emb.weight transposed to (A, D) */
mdt_status mdt_op_action_embed(const float *x, const float *sigma, int64_t sigma_stride, float sigma_data,
                               const float *WaT, const float *ba, float *y, int64_t M, int32_t A, int32_t D,
                               int32_t rows_per_sample, void *stream);

#ifdef __cplusplus
}
#endif
#endif
