// mdt_internal.h -- launcher prototypes shared by mdt_kernels.hip (device code) and the host logic
// (mdt_model.hip: denoiser handle; mdt_resampler.hip: Perceiver resampler handle), plus the host helpers both use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "mdt_hip.h"
#include "mdt_hip_ops.h"
#include "mdt_resampler.h"
#include "mdt_hip_train.h"
#include "mdt_mae.h"

// ---- error plumbing: the message behind mdt_last_error() (thread local, defined in mdt_model.hip) ----
mdt_status mdt_fail(mdt_status st, const char* fmt, ...);

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) return mdt_fail(MDT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));  \
    } while (0)

#define MDT_TRY(expr)                      \
    do {                                   \
        mdt_status _s = (expr);            \
        if (_s != MDT_OK) return _s;       \
    } while (0)

#define LAUNCH(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return mdt_fail(MDT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));       \
    } while (0)

// ---- a Linear whose weight lives fragment-packed in a handle's arena ----
struct Lin {
    float* wp = nullptr;    // fragment-packed (N, K)
    float* bias = nullptr;  // (N) or nullptr
    int N = 0, K = 0;
    float* wt = nullptr;    // training only: fragment-packed image of W^T (N' = K, K' = N), for dX = dY W
    void* ws = nullptr;     // the MLP's two Linears: three-way bf16 split fragment image (6 N K bytes; mdt_mlp_split.h)
};

// bump allocator over one hipMalloc'ed block (count pass with base == nullptr, then the real pass)
struct Bump {
    float* base = nullptr;
    size_t off = 0;
    float* take(size_t n) {
        float* p = base ? base + off : nullptr;
        off += (n + 63) & ~(size_t)63;  // 256-byte granules keep every buffer 16-byte aligned
        return p;
    }
};

static inline mdt_gemm_args gemm_args(const float* A, int64_t lda, const Lin& w, float* out, int64_t ldo, int M) {
    mdt_gemm_args g;
    memset(&g, 0, sizeof g);
    g.A = A; g.lda = lda; g.Wp = w.wp; g.Wp_split = w.ws; g.bias = w.bias; g.out = out; g.ldo = ldo;
    g.M = M; g.N = w.N; g.K = w.K;
    g.shift_off = -1; g.scale_off = -1; g.gate_off = -1; g.rows_per_sample = 1;
    g.gin = 1; g.gout = 1; g.goff = 0;
    return g;
}

// batch-sized buffers (workspace, tapes, backward scratch): through the installed allocator (mdt_set_allocator) or hipMalloc
hipError_t mdt_dev_malloc(void** p, size_t bytes);
hipError_t mdt_dev_free(void* p);

int mdt_gemm_kchunk(int K, int ln, int cap);
// the current device's buffer of zeros (stands in for absent bias / LayerNorm-bias vectors); nullptr on failure
const float* mdt_zeros();
hipError_t mdt_launch_gemm(const mdt_gemm_args& a, hipStream_t s);
// the MLP sublayer as one launch (k_mlp): S = mdt_mlp_slices(D) partial slabs at parts + s * part_stride
bool mdt_mlp_supported(const mdt_gemm_args& fc, const mdt_gemm_args& proj);
int mdt_mlp_slices(int D);
hipError_t mdt_launch_mlp(const mdt_gemm_args& fc, const mdt_gemm_args& proj, float* parts, int64_t part_stride, hipStream_t s);
// the same launch in the three-way bf16 split form (mdt_mlp_split.h): w1s / w2s = split images of fc / proj (mdt_launch_pack_weight_split)
bool mdt_mlp_split_supported(const mdt_gemm_args& fc, const mdt_gemm_args& proj);
bool mdt_mlp_split_enabled();
int mdt_split_min_rows();   // rows from which the split forms of the sampler's launches are used (MDT_HIP_SPLIT_MIN_ROWS, default 768: B = 80 2.97 -> 2.89 ms per call, 96 3.08 -> 2.93, 120 3.54 -> 2.96, 128 3.42 -> 2.96; B = 64 loses, 2.44 -> 2.90)
hipError_t mdt_launch_mlp_split(const mdt_gemm_args& fc, const mdt_gemm_args& proj, const void* w1s, const void* w2s, float* parts,
                                int64_t part_stride, hipStream_t s);
// (n_rows, K) row-major fp32 -> split image of 6 n_rows K bytes (n_rows % 16 == 0, K % 32 == 0)
hipError_t mdt_launch_pack_weight_split(const float* w, int n_rows, int K, void* image, hipStream_t s, int n_off = 0);
// the same image from the fp32 fragment image of the weight (mdt_launch_pack_weight's output for the whole (N, K) matrix)
hipError_t mdt_launch_split_from_packed(const float* wp, int N, int K, void* image, hipStream_t s);
hipError_t mdt_launch_attention(const mdt_attn_args& a, const float* rope_cos, const float* rope_sin, hipStream_t s);
// one sample's self-attention fused into its output projection p (rollout batch 1); see mdt_kernels.hip
bool mdt_attn_proj_supported(const mdt_gemm_args& p, int H, int hd, int T, int rope);
hipError_t mdt_launch_attn_proj(const mdt_gemm_args& p, const float* qkv, int64_t ldq, int H, int hd, int T, int causal,
                                hipStream_t s);
// the same for a LARGE batch: the causal attention of a 32-row tile computed in the projection's prologue (k_attn_proj_wide)
bool mdt_attn_proj_wide_supported(const mdt_gemm_args& p, int H, int hd, int T, int causal, int rope);
hipError_t mdt_launch_attn_proj_wide(const mdt_gemm_args& p, const float* qkv, int64_t ldq, int H, int hd, int T, hipStream_t s);
// one workgroup per sample: self-attention -> projection -> collapsed cross-attention (k_attn_xattn); x.y == p.out
bool mdt_attn_xattn_supported(const mdt_gemm_args& p, const mdt_xapply_args& x, int H, int hd, int T, int causal, int rope);
hipError_t mdt_launch_attn_xattn(const mdt_gemm_args& p, const float* qkv, int64_t ldq, const mdt_xapply_args& x, int H, int hd,
                                 int T, hipStream_t s);
// side jobs (mdt_kernels.hip): small-M products that ride in the launches of the small-M products that follow them
hipError_t mdt_gemm_side_push(const mdt_gemm_args& a, hipStream_t s);
hipError_t mdt_gemm_side_push_front(const mdt_gemm_args& a, hipStream_t s);
hipError_t mdt_gemm_side_flush(hipStream_t s);
void mdt_gemm_side_drop();
size_t mdt_gemm_side_pending();                       // jobs still queued on this host thread
hipError_t mdt_gemm_side_launch_front(hipStream_t s);  // the head of the queue as a launch of its own
hipError_t mdt_launch_layernorm(const float* in, const float* w, const float* b, float* out, int M, int D,
                                hipStream_t s, float* out2 = nullptr);
hipError_t mdt_launch_sigma_emb(const float* sigma, int64_t sstride, const float* freqs, float* out, int R, int D,
                                hipStream_t s);
hipError_t mdt_launch_ddim_steps(const float* sigmas_dev, int n, float* steps, hipStream_t s);
// the DDIM sampler's once-per-call scalar work in one launch: per-step scalars, the sigma embeddings of all steps, the first
// action embedding (mdt_kernels.hip: k_sample_prep); the schedule from device memory or, by value, from the host
constexpr int MDT_SCHED_MAX = 64;
struct mdt_sched_arg { float s[MDT_SCHED_MAX + 1]; };
hipError_t mdt_launch_sample_prep(const float* sigmas_dev, const float* sigmas_host, int n_steps, float* steps, const float* freqs,
                                  float* sig_e, int D, const float* x, float sd, const float* Wa, const float* ba, float* y, int M,
                                  int A, hipStream_t s);
hipError_t mdt_launch_action_embed(const float* x, const float* sigma, int64_t sstride, float sd, const float* Wa,
                                   const float* ba, float* y, int M, int A, int D, int rps, hipStream_t s);
hipError_t mdt_launch_head(const mdt_head_args& a, hipStream_t s);
hipError_t mdt_launch_noise_input(const float* act, const float* noise, const float* sigma, float* noised, int64_t n,
                                  int per_sample, hipStream_t s);
constexpr int MDT_LOSS_PARTS = 1024;   // floats of scratch mdt_launch_loss_reduce wants (`part`)
hipError_t mdt_launch_loss_reduce(const float* F, const float* act, const float* noised, const float* sigma, float sd,
                                  int64_t n, int per_sample, float* loss, float* part, hipStream_t s);
hipError_t mdt_launch_pack_weight(const float* w, int n_rows, int K, float* packed, int n_off, hipStream_t s);
hipError_t mdt_launch_pack_weight_glu(const float* w, int H, int K, float* packed, hipStream_t s);
// one move of a batched parameter upload (k_multi_load): src is (rows, K) row-major on the device
enum { MDT_LOAD_RAW = 0, MDT_LOAD_PACK = 1, MDT_LOAD_PACK_T = 2, MDT_LOAD_TRANSPOSE = 3, MDT_LOAD_PAD_COLS = 4,
       MDT_LOAD_PACK_SPLIT = 5 };  // PACK_SPLIT: the three-way bf16 split fragment image (mdt_mlp_split.h); dst is a byte image of 6 rows K bytes
struct mdt_load_entry {
    const float* src;
    float* dst;
    int32_t kind, rows, K;
    int32_t p0;  // PACK: first packed row (n_off);  PACK_T: k offset in the W^T image (n_off);  PAD_COLS: row pitch
    int32_t p1;  // PACK_T: 16-blocks along the image's k (= Lin.N / 16)
    int32_t pad_;
};
hipError_t mdt_launch_multi_load(const mdt_load_entry* tab, const int2* blocks, int n_blocks, hipStream_t s);
hipError_t mdt_launch_transpose(const float* src, float* dst, int R, int Cc, hipStream_t s);
hipError_t mdt_launch_xattn_fold(const mdt_xfold_args& a, hipStream_t s);
hipError_t mdt_launch_xattn_fold_n(const mdt_xfold_args* sets, int n, hipStream_t s);  // equal shapes; one launch per 8 sets
hipError_t mdt_launch_xattn_apply(const mdt_xapply_args& a, hipStream_t s);
// the collapsed cross-attention + the LayerNorm-prologue Linear on its output rows in one launch (k_xattn_gemm_smallm)
bool mdt_xattn_gemm_supported(const mdt_xapply_args& x, const mdt_gemm_args& g);
hipError_t mdt_launch_xattn_gemm(const mdt_xapply_args& x, const mdt_gemm_args& g, hipStream_t s);
bool mdt_xattn_apply_supported(int D, int H, int Te, int Ta);
size_t mdt_xattn_lds_floats(int D, int H);  // LDS floats of the collapsed cross-attention body (xattn_tile)
// ---- Perceiver resampler kernels ----
// media (B, T, n, D) + time_pos_emb[t] * mask[b][t] -> out (same shape); mask may be nullptr
hipError_t mdt_launch_add_time_emb(const float* media, const float* tpe, const uint8_t* mask, float* out, int64_t B,
                                   int T, int n, int D, hipStream_t s);
// out[b*R + r][:] = src[r][:]
hipError_t mdt_launch_bcast_rows(const float* src, float* out, int64_t B, int R, int D, hipStream_t s);
// softmax(q k^T * scale) v for few queries over many keys: q (B*Tq, H*64) ld ldq; k / v rows (B*Tk) ld ldkv
hipError_t mdt_launch_attention_long(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                     float* out, int64_t ldo, int B, int H, int hd, int Tq, int Tk, float scale,
                                     hipStream_t s);
bool mdt_attention_long_supported(int hd, int Tq, int Tk);
// ---- training-path kernels (mdt_train_kernels.hip) ----
hipError_t mdt_launch_pack_weight_t(const float* src, int rows, int cols, int64_t ld, float* packed, int k_off, int K16,
                                    hipStream_t s, int slice_len = 0);
hipError_t mdt_launch_transpose_ld(const float* src, int64_t lds_, float* dst, int64_t ldd, int R, int Cc, float* part,
                                   hipStream_t s, int slice_len = 0);
// floats of scratch mdt_linear_bwd needs for an (M, N, K) layer
int64_t mdt_linear_bwd_scratch(int64_t M, int64_t N, int64_t K);
// dW[n][k] = sum_m dY[m][n] X[m][k] from the row-major operands (mdt_train_kernels.hip: k_gemm_tn), S row slices of L rows
hipError_t mdt_launch_gemm_tn(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* out, int64_t slice_stride, int M, int N,
                              int K, int S, int L, int accumulate, float* bpart, hipStream_t s);
int mdt_gemm_tn_ktile(int K);  // 128 or 192: the k-tile width k_gemm_tn picks for K columns
// the same product as three-way bf16 splits of both operands (k_gemm_tn_split: 128 x 128 / 128 x 192 tiles, 8 waves, 123 / 154 KB of LDS)
bool mdt_gemm_tn_split_on();
int mdt_gemm_tn_split_ktile(int K);
void mdt_gemm_tn_split_tile(int N, int K, int* tn, int* tk);   // its (n, k) tile: 128 x 128 / 128 x 192 / 192 x 128
hipError_t mdt_launch_gemm_tn_split(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* out, int64_t slice_stride, int M, int N,
                                    int K, int S, int L, int accumulate, float* bpart, hipStream_t s);
void mdt_gemm_tn_tile(int64_t M, int N, int K, int* tn, int* tk);  // its (n, k) tile for an M-deep product: n 64 / 128 / 192 (4 / 8 / 12 waves)
hipError_t mdt_launch_colsum2(const float* X0, const float* X1, int64_t ldx, int M, int N, float* out0, float* out1,
                              int accumulate, hipStream_t s);
hipError_t mdt_launch_ln_fwd_train(const mdt_ln_train_args& a, hipStream_t s);
hipError_t mdt_launch_ln_bwd(const mdt_ln_bwd_args& a, hipStream_t s);
// fused pairs of the training path (round 6): LayerNorm backward + the merge backward behind it; merge + the LayerNorm of its result
bool mdt_attn_train_mfma_supported(int hd, int H, int rope);  // the MFMA form of the training attention kernels takes this shape
hipError_t mdt_launch_ln_bwd_merge(const mdt_ln_bwd_args& a, const mdt_merge_args& g, hipStream_t s);
hipError_t mdt_launch_merge_ln_fwd(const mdt_merge_args& g, const mdt_ln_train_args& l, hipStream_t s);
hipError_t mdt_launch_act_fwd(const float* u, float* out, int64_t n, int act, hipStream_t s);
hipError_t mdt_launch_act_bwd(const float* u, const float* dy, float* du, int64_t n, int act, hipStream_t s);
hipError_t mdt_launch_merge_fwd(const mdt_merge_args& a, hipStream_t s);
hipError_t mdt_launch_merge_bwd(const mdt_merge_args& a, hipStream_t s);
hipError_t mdt_launch_attn_fwd_train(const mdt_attn_train_args& a, hipStream_t s);
hipError_t mdt_launch_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, hipStream_t s);
// a table of independent column sums run as one launch: dst[n] (+)= sum_m src[m * ld + n], m < M, n < N
struct mdt_colsum_entry {
    const float* src;
    float* dst;
    int64_t ld;
    int32_t M, N, accumulate;
};
enum { MDT_COLSUM_TABLE = 80 };  // entries per launch (passed by value: 80 x 40 B of kernel arguments)
struct mdt_colsum_table { mdt_colsum_entry e[MDT_COLSUM_TABLE]; };
hipError_t mdt_launch_colsum_batched(const mdt_colsum_entry* entries, int n, hipStream_t s);
hipError_t mdt_launch_attn_bwd(const mdt_attn_bwd_args& a, hipStream_t s);
hipError_t mdt_launch_loss_grad(const float* F, const float* act, const float* noised, const float* sigma, float sd,
                                int64_t n, int per_sample, const float* gscale, float* dF, hipStream_t s);
hipError_t mdt_launch_narrow_dx(const float* G, const float* W, float* out, int M, int A, int D, hipStream_t s);
hipError_t mdt_launch_vjp_seed(const float* F, const float* x, const float* sigma, const float* v, float sd, int64_t n,
                               int per_sample, float* den, float* dF, hipStream_t s);
hipError_t mdt_launch_vjp_finish(const float* dxin, const float* sigma, const float* v, float sd, int64_t n, int per_sample,
                                 float* out, hipStream_t s);
// out[m][n] = act(b[n] + sum_a X[m][a] WT[a][n]), A <= 16; pre (optional): the pre-activation rows
hipError_t mdt_launch_narrow_linear(const float* X, const float* WT, const float* b, float* pre, float* out, int M, int A,
                                    int N, int act, hipStream_t s);
// out[m][a] = sum_d G[m][d] WT[a][d], A <= 16   (input gradient of the narrow-input Linear; WT is (A, D))
hipError_t mdt_launch_narrow_out(const float* G, int64_t ldg, const float* WT, float* out, int M, int A, int D, hipStream_t s);
hipError_t mdt_launch_narrow_dw(const float* G, const float* Y, int64_t ldy, float* partial, int n_slices, int M, int A,
                                int D, int transposed, hipStream_t s);
hipError_t mdt_launch_scaled_input(const float* x, const float* sigma, float sd, int64_t n, int per_sample, float* out,
                                   hipStream_t s);
hipError_t mdt_launch_gather_rows(const float* src, float* dst, int M, int D, int gin, int gout, int goff, hipStream_t s);
hipError_t mdt_launch_dropout_rows(float* x, int64_t rows, int D, int rows_per_sample, int row_lo, float p, uint32_t site,
                                   uint64_t seed, hipStream_t s);
hipError_t mdt_launch_add_inplace(const float* x, float* y, int64_t n, hipStream_t s);
// dst[r][c] += src[r][c] for r < rows, c < cols (leading dimensions lds_ / ldd)
hipError_t mdt_launch_add_2d(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int cols, hipStream_t s);
// backward of a Linear through the forward GEMM kernel (mdt_train.hip); see mdt_linear_bwd_args
// defer_bias (optional, with bias_space of at least 256 * N floats that stay valid until the caller runs the entry): the
// per-slice bias partials are left in bias_space and *defer_bias describes their final column sum instead of launching it
// (defer_bias->src == nullptr on return: the bias gradient was produced another way, nothing to run).
mdt_status mdt_linear_bwd(const mdt_linear_bwd_args& a, hipStream_t s, mdt_colsum_entry* defer_bias = nullptr,
                          float* bias_space = nullptr);
hipError_t mdt_launch_attention_long_bwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                         const float* d_out, int64_t ld_do, float* dq, int64_t ld_dq, float* dk, float* dv,
                                         int64_t ld_dkv, int B, int H, int hd, int Tq, int Tk, float scale, hipStream_t s,
                                         float* dkl = nullptr, int F = 0);  // F > 0: keys < F -> dk / dv (B*F rows), the rest -> dkl
bool mdt_attention_long_bwd_supported(int hd, int Tq, int Tk);
hipError_t mdt_launch_time_emb_grad(const float* dxf, const uint8_t* mask, float* out, float* partial, int64_t B, int T, int n,
                                    int D, int accumulate, hipStream_t s);  // partial: B*T*D floats of scratch
hipError_t mdt_launch_multi_adamw(const mdt_opt_tensor* tab, const int2* blocks, int n_blocks, float lr, float beta1,
                                  float beta2, float eps, float wd, float bc1, float bc2_sqrt, hipStream_t s);
hipError_t mdt_launch_multi_axpby(const mdt_opt_tensor* tab, const int2* blocks, int n_blocks, float a, float b,
                                  hipStream_t s);
// RMSNorm / SwishGLU row kernels (mdt_map_pool.hip), shared with the masked-image decoder's ops (mdt_mae.hip)
hipError_t mdt_launch_rms_fwd(const float* x, const float* g, float* out, int64_t M, int D, float eps, hipStream_t s);
hipError_t mdt_launch_rms_bwd(const float* x, const float* g, const float* dy, float* dx, int accumulate, float* pg, int64_t M,
                              int D, float eps, hipStream_t s, const float* res = nullptr);
hipError_t mdt_launch_swiglu_fwd(const float* u, float* out, int64_t M, int Hm, hipStream_t s);
hipError_t mdt_launch_swiglu_bwd(const float* u, const float* d_out, float* du, int64_t M, int Hm, hipStream_t s);
struct mdt_model;
void mdt_train_free(mdt_model* m);  // releases what mdt_train_prepare() and the tapes allocated (mdt_train.hip)
