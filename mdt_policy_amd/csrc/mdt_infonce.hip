// mdt_infonce.hip -- the InfoNCE loss of the contrastive (CLA) head, value AND gradients in one enqueue
// (include/mdt_map_pool.h, "loss" section).
//
// Reference replaced: MDTVAgent.clip_auxiliary_loss (mdt/models/mdtv_agent.py:774-799): F.normalize of both
// embedding sets, S = exp(logit_scale) * img_n @ lang_n^T, cross entropy with the diagonal as labels over the rows
// ('img_to_text'), the columns ('text_to_img') or both halves averaged ('symmetric'; the reference's second matmul
// lang_n @ img_n^T is S^T, so its row cross-entropy is the column cross-entropy of S).
//
//   forward : k_rownorm (both sets) -> pack lang_n as the "weight" -> S on the fp32-MFMA GEMM -> row / column
//             log-sum-exp -> loss
//   backward: G = dL/dS from the two LSE vectors in one pass over S (w_r softmax_rows + w_c softmax_cols - (w_r+w_c) I) / B,
//             d_img_n = G lang_n, d_lang_n = G^T img_n on the GEMM (the other set's TRANSPOSED image as the weight),
//             then through the normalisation; d(logit_scale) = sum(G * S).
// Batch sizes that are not multiples of 16 are padded with zero rows / columns that the LSE kernels skip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "mdt_device.h"
#include "mdt_internal.h"
#include "mdt_map_pool.h"

#define fail mdt_fail

// x (M, D) -> x / max(||x||, 1e-12)   (F.normalize, p = 2, eps = 1e-12); inv[m] = 1 / max(||x||, eps); one wave per row
__global__ __launch_bounds__(256) void k_rownorm_fwd(const float* __restrict__ x, float* __restrict__ xn, float* __restrict__ inv,
                                                     int M, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = x[(int64_t)row * D + c]; ss = fmaf(v, v, ss); }
    const float r = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int c = lane; c < D; c += 64) xn[(int64_t)row * D + c] = x[(int64_t)row * D + c] * r;
    if (lane == 0) inv[row] = r;
}

// dx = (dn - xn (xn . dn)) * inv   (rows whose norm was clamped never occur for non-zero embeddings)
__global__ __launch_bounds__(256) void k_rownorm_bwd(const float* __restrict__ xn, const float* __restrict__ inv,
                                                     const float* __restrict__ dn, float* __restrict__ dx, int M, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot = fmaf(xn[(int64_t)row * D + c], dn[(int64_t)row * D + c], dot);
    dot = wave_sum(dot);
    const float r = inv[row];
    for (int c = lane; c < D; c += 64)
        dx[(int64_t)row * D + c] = (dn[(int64_t)row * D + c] - xn[(int64_t)row * D + c] * dot) * r;
}

// lse_r[i] = log sum_{j < B} exp(scale * S[i][j]);  one wave per row
__global__ __launch_bounds__(256) void k_row_lse(const float* __restrict__ S, int64_t ld, const float* __restrict__ logit_scale,
                                                 float* __restrict__ lse, int B) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float sc = expf(logit_scale[0]);
    float mx = -INFINITY;
    for (int j = lane; j < B; j += 64) mx = fmaxf(mx, sc * S[(int64_t)row * ld + j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < B; j += 64) sum += expf(sc * S[(int64_t)row * ld + j] - mx);
    sum = wave_sum(sum);
    if (lane == 0) lse[row] = mx + logf(sum);
}

// lse_c[j] = log sum_{i < B} exp(scale * S[i][j]);  64 columns per workgroup, 4 row groups combined through LDS
__global__ __launch_bounds__(256) void k_col_lse(const float* __restrict__ S, int64_t ld, const float* __restrict__ logit_scale,
                                                 float* __restrict__ lse, int B) {
    __shared__ float pm[4][64], ps[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, col = blockIdx.x * 64 + cl;
    const float sc = expf(logit_scale[0]);
    float mx = -INFINITY, sum = 0.f;
    if (col < B)
        for (int i = rg; i < B; i += 4) {  // online max / sum
            const float v = sc * S[(int64_t)i * ld + col];
            if (v > mx) { sum = sum * expf(mx - v) + 1.f; mx = v; }
            else sum += expf(v - mx);
        }
    pm[rg][cl] = mx; ps[rg][cl] = sum;
    __syncthreads();
    if (rg == 0 && col < B) {
        float m = fmaxf(fmaxf(pm[0][cl], pm[1][cl]), fmaxf(pm[2][cl], pm[3][cl]));
        float t = 0.f;
        for (int g = 0; g < 4; ++g)
            if (pm[g][cl] > -INFINITY) t += ps[g][cl] * expf(pm[g][cl] - m);
        lse[col] = m + logf(t);
    }
}

// G = dL/dS (in place over S, (Bp, Bp) zero outside B x B) and per-row partials of the loss and of d(logit_scale):
//   L = sum_i [ w_r (lse_r[i] - s_ii) + w_c (lse_c[i] - s_ii) ] / B          with s = scale * S
//   dL/ds_ij = ( w_r exp(s_ij - lse_r[i]) + w_c exp(s_ij - lse_c[j]) - (w_r + w_c) [i == j] ) / B
//   G = scale * dL/ds  (gradient with respect to the UNSCALED product);  d(logit_scale) = sum_ij dL/ds_ij * s_ij
__global__ __launch_bounds__(256) void k_infonce_grad(float* __restrict__ S, int64_t ld, const float* __restrict__ logit_scale,
                                                      const float* __restrict__ lse_r, const float* __restrict__ lse_c, float w_r,
                                                      float w_c, float* __restrict__ part, int B, int Bp) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= Bp) return;
    const float sc = expf(logit_scale[0]), invB = 1.0f / (float)B;
    float dsc = 0.f, loss = 0.f;
    for (int j = lane; j < Bp; j += 64) {
        float g = 0.f;
        if (row < B && j < B) {
            const float s = sc * S[(int64_t)row * ld + j];
            float d = w_r * expf(s - lse_r[row]) + w_c * expf(s - lse_c[j]);
            if (j == row) { d -= w_r + w_c; loss = w_r * (lse_r[row] - s) + w_c * (lse_c[row] - s); }
            d *= invB;
            dsc = fmaf(d, s, dsc);
            g = d * sc;
        }
        S[(int64_t)row * ld + j] = g;
    }
    dsc = wave_sum(dsc);
    loss = wave_sum(loss);
    if (lane == 0) { part[2 * row] = loss * invB; part[2 * row + 1] = dsc; }
}

// out[0] = sum_m part[2m], out[1] = sum_m part[2m+1]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_pair_sum(const float* __restrict__ part, int M, float* __restrict__ out0,
                                                  float* __restrict__ out1) {
    __shared__ float r0[256], r1[256];
    float a = 0.f, b = 0.f;
    for (int m = threadIdx.x; m < M; m += 256) { a += part[2 * m]; b += part[2 * m + 1]; }
    r0[threadIdx.x] = a; r1[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out0[0] = r0[0]; if (out1) out1[0] = r1[0]; }
}

namespace {
struct Carve {
    float *img_n, *lang_n, *inv_i, *inv_l, *pk, *S, *St, *lse_r, *lse_c, *part, *dn_i, *dn_l;
};
int64_t carve(Carve& c, float* base, int64_t B, int64_t D) {
    const int64_t Bp = (B + 15) & ~(int64_t)15;
    Bump b;
    b.base = base;
    c.img_n = b.take(Bp * D); c.lang_n = b.take(Bp * D); c.inv_i = b.take(Bp); c.inv_l = b.take(Bp);
    c.pk = b.take(std::max(Bp * D, D * Bp)); c.S = b.take(Bp * Bp); c.St = b.take(Bp * Bp);
    c.lse_r = b.take(Bp); c.lse_c = b.take(Bp); c.part = b.take(2 * Bp); c.dn_i = b.take(Bp * D); c.dn_l = b.take(Bp * D);
    return (int64_t)b.off;
}
}  // namespace

extern "C" int64_t mdt_op_infonce_scratch(int64_t batch, int64_t dim) {
    if (batch < 1 || dim < 1) return -1;
    Carve c;
    return carve(c, nullptr, batch, dim);
}

extern "C" mdt_status mdt_op_infonce(const mdt_infonce_args* a, void* stream) {
    if (!a || !a->image_features || !a->lang_features || !a->logit_scale || !a->loss || !a->scratch)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_infonce: null argument");
    const int B = a->batch, D = a->dim;
    if (B < 1 || B > 32768) return fail(MDT_ERR_INVALID_ARG, "mdt_op_infonce: batch must be 1..32768");
    if (D < 16 || D % 16) return fail(MDT_ERR_UNSUPPORTED, "mdt_op_infonce: dim must be a multiple of 16");
    float w_r, w_c;
    switch (a->mode) {
        case MDT_INFONCE_SYMMETRIC: w_r = 0.5f; w_c = 0.5f; break;
        case MDT_INFONCE_IMG_TO_TEXT: w_r = 1.f; w_c = 0.f; break;
        case MDT_INFONCE_TEXT_TO_IMG: w_r = 0.f; w_c = 1.f; break;
        default: return fail(MDT_ERR_INVALID_ARG, "mdt_op_infonce: invalid mode %d (symmetric / img_to_text / text_to_img)", a->mode);
    }
    const bool grads = a->d_image || a->d_lang || a->d_logit_scale;
    if (grads && !(a->d_image && a->d_lang && a->d_logit_scale))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_infonce: gradients come together (d_image, d_lang, d_logit_scale) or not at all");
    hipStream_t s = (hipStream_t)stream;
    const int Bp = (B + 15) & ~15;
    Carve c;
    carve(c, a->scratch, B, D);
    const int rb = (B + 3) / 4;
    // normalised embeddings (padded rows zero)
    if (Bp > B) {
        HIP_TRY(hipMemsetAsync(c.img_n + (int64_t)B * D, 0, (size_t)(Bp - B) * D * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(c.lang_n + (int64_t)B * D, 0, (size_t)(Bp - B) * D * sizeof(float), s));
    }
    hipLaunchKernelGGL(k_rownorm_fwd, dim3(rb), dim3(256), 0, s, a->image_features, c.img_n, c.inv_i, B, D);
    hipLaunchKernelGGL(k_rownorm_fwd, dim3(rb), dim3(256), 0, s, a->lang_features, c.lang_n, c.inv_l, B, D);
    LAUNCH(hipGetLastError());
    // S = img_n @ lang_n^T
    Lin w;
    w.wp = c.pk; w.N = Bp; w.K = D;
    LAUNCH(mdt_launch_pack_weight(c.lang_n, Bp, D, c.pk, 0, s));
    LAUNCH(mdt_launch_gemm(gemm_args(c.img_n, D, w, c.S, Bp, Bp), s));
    hipLaunchKernelGGL(k_row_lse, dim3(rb), dim3(256), 0, s, c.S, (int64_t)Bp, a->logit_scale, c.lse_r, B);
    hipLaunchKernelGGL(k_col_lse, dim3((B + 63) / 64), dim3(256), 0, s, c.S, (int64_t)Bp, a->logit_scale, c.lse_c, B);
    hipLaunchKernelGGL(k_infonce_grad, dim3(Bp / 4), dim3(256), 0, s, c.S, (int64_t)Bp, a->logit_scale, c.lse_r, c.lse_c, w_r, w_c,
                       c.part, B, Bp);
    hipLaunchKernelGGL(k_pair_sum, dim3(1), dim3(256), 0, s, c.part, B, a->loss, a->d_logit_scale);
    LAUNCH(hipGetLastError());
    if (!grads) return MDT_OK;
    // d_img_n = G @ lang_n : weight image of lang_n^T (N' = D, K' = Bp)
    Lin wt;
    wt.wp = c.pk; wt.N = D; wt.K = Bp;
    LAUNCH(mdt_launch_pack_weight_t(c.lang_n, Bp, D, D, c.pk, 0, Bp / 16, s));
    LAUNCH(mdt_launch_gemm(gemm_args(c.S, Bp, wt, c.dn_i, D, Bp), s));
    // d_lang_n = G^T @ img_n
    LAUNCH(mdt_launch_transpose(c.S, c.St, Bp, Bp, s));
    LAUNCH(mdt_launch_pack_weight_t(c.img_n, Bp, D, D, c.pk, 0, Bp / 16, s));
    LAUNCH(mdt_launch_gemm(gemm_args(c.St, Bp, wt, c.dn_l, D, Bp), s));
    hipLaunchKernelGGL(k_rownorm_bwd, dim3(rb), dim3(256), 0, s, c.img_n, c.inv_i, c.dn_i, a->d_image, B, D);
    hipLaunchKernelGGL(k_rownorm_bwd, dim3(rb), dim3(256), 0, s, c.lang_n, c.inv_l, c.dn_l, a->d_lang, B, D);
    LAUNCH(hipGetLastError());
    return MDT_OK;
}
