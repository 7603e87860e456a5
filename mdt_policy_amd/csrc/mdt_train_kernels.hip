// mdt_train_kernels.hip -- gfx950 kernels of the TRAINING path (GCDenoiser.loss forward with saved activations,
// and its backward): the row-local / per-sample pieces.  Every dense contraction of the backward (dX = dY W,
// dW = dY^T X) runs on the same fp32-MFMA GEMM as the forward (mdt_kernels.hip: k_gemm) -- dX with a transposed
// packed image of the weight, dW with the transposed gradient as the row operand and the transposed activation
// packed as the "weight" (k_pack_weight_t) -- so this file holds no GEMM.
//
// Reference semantics: autograd through mdt/models/networks/transformers/transformer_blocks.py (LayerNorm :29-38,
// Attention :119-158, MLP :161-180, ConditionedBlock :291-309) and mdt/models/edm_diffusion/score_wrappers.py:45-63.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "mdt_internal.h"

#include "mdt_device.h"

// ------------------------------------------------------------------------------------------------
// packing the TRANSPOSE of a row-major (rows, cols) matrix part into a fragment-packed (N' = cols, K' = K16*16)
// image at k-offset k_off:  src[r][c] -> logical (n' = c, k' = k_off + r).
//   * weights:      W (N, K) part at row offset n_off -> image of W^T (N' = K, K' = N_total), k_off = n_off
//   * activations:  X (M, K)                           -> image of X^T (N' = K, K' = M padded to 16)
// ------------------------------------------------------------------------------------------------
// slice_len > 0: the rows are cut into slices of slice_len (a multiple of 16); slice s becomes its own packed image
// (N' = cols, K' = slice_len) at packed + s * cols * slice_len (the per-slice operands of a split-K product).
__global__ void k_pack_weight_t(const float* __restrict__ src, int rows, int cols, int64_t lds_, float* __restrict__ packed,
                                int k_off, int K16, int slice_len) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rows * cols) return;
    const int r = (int)(idx / cols), c = (int)(idx % cols);
    int k = k_off + r;
    int64_t base = 0;
    if (slice_len > 0) {
        const int sl = k / slice_len;
        k -= sl * slice_len;
        base = (int64_t)sl * cols * slice_len;
    }
    const int n = c;
    const int nt = n >> 4, ni = n & 15, kc = k >> 4, h = (k & 15) >> 2, j = k & 3;
    packed[base + (((int64_t)nt * K16 + kc) * 64 + (ni + 16 * h)) * 4 + j] = src[(int64_t)r * lds_ + c];
}

// The same packing with one thread per (4 consecutive source rows, column): the four values are the j = 0..3 floats of
// one fragment slot, so a wave writes 1 KiB contiguously (the scalar kernel writes 4-byte pieces 16 bytes apart) and
// reads four coalesced row segments.  Needs k_off % 4 == 0; rows past `rows` inside the last group are written as 0.
__global__ void k_pack_weight_t4(const float* __restrict__ src, int rows, int cols, int64_t lds_, float* __restrict__ packed,
                                 int k_off, int K16, int slice_len) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int rows4 = (rows + 3) >> 2;
    if (idx >= (int64_t)rows4 * cols) return;
    const int r4 = (int)(idx / cols), c = (int)(idx % cols);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * r4 + j;
        v[j] = r < rows ? src[(int64_t)r * lds_ + c] : 0.f;
    }
    int k = k_off + 4 * r4;
    int64_t base = 0;
    if (slice_len > 0) {
        const int sl = k / slice_len;
        k -= sl * slice_len;
        base = (int64_t)sl * cols * slice_len;
    }
    const int nt = c >> 4, ni = c & 15, kc = k >> 4, h = (k & 15) >> 2;
    *(f32x4*)(packed + base + (((int64_t)nt * K16 + kc) * 64 + (ni + 16 * h)) * 4) = v;
}

hipError_t mdt_launch_pack_weight_t(const float* src, int rows, int cols, int64_t ld, float* packed, int k_off, int K16,
                                    hipStream_t s, int slice_len) {
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return hipSuccess;
    if ((k_off & 3) == 0 && (slice_len & 3) == 0) {
        const int64_t n4 = (int64_t)((rows + 3) >> 2) * cols;
        hipLaunchKernelGGL(k_pack_weight_t4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, src, rows, cols, ld, packed,
                           k_off, K16, slice_len);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_pack_weight_t, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, rows, cols, ld, packed,
                       k_off, K16, slice_len);
    return hipGetLastError();
}

// dst (C, ldd) = src (R, C; lds)^T, through a 32x33 LDS tile so both sides are coalesced.  Optionally also emits
// per-row-block column sums  part[blockIdx.y][c] = sum of the block's 32 rows of column c  (the bias gradient is
// then a column sum over R/32 partial rows instead of R rows).
// slice_len > 0 (a multiple of 32): the R rows are cut into slices; slice s lands as its own (C, slice_len) matrix at
// dst + s * C * slice_len (ldd is ignored then).
__global__ __launch_bounds__(256) void k_transpose_ld(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst,
                                                      int64_t ldd, int R, int Cc, float* __restrict__ part, int slice_len) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < Cc) ? src[(int64_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < Cc && r < R) {
            if (slice_len > 0) {
                const int sl = r / slice_len;
                dst[((int64_t)sl * Cc + c) * slice_len + (r - sl * slice_len)] = tile[tx][ty + 8 * i];
            } else {
                dst[(int64_t)c * ldd + r] = tile[tx][ty + 8 * i];
            }
        }
    }
    if (part && threadIdx.x < 32 && c0 + tx < Cc) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) acc += tile[r][tx];
        part[(int64_t)blockIdx.y * Cc + c0 + tx] = acc;
    }
}

hipError_t mdt_launch_transpose_ld(const float* src, int64_t lds_, float* dst, int64_t ldd, int R, int Cc, float* part,
                                   hipStream_t s, int slice_len) {
    hipLaunchKernelGGL(k_transpose_ld, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, s, src, lds_, dst, ldd, R, Cc, part,
                       slice_len);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (+ adaLN modulate) forward that keeps the row statistics        (transformer_blocks.py:37-38,:262)
//   h = shift + (xhat * w + b) * scale ;  stats[row] = (mean, rstd)
// one wave per row; lane owns columns lane, lane+64, ...  (D <= 512)
// ------------------------------------------------------------------------------------------------
#define LN_MAXC 8

__global__ __launch_bounds__(256) void k_ln_fwd_train(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, const float* __restrict__ mod,
                                                      int64_t mod_stride, int shift_off, int scale_off, int rps,
                                                      float* __restrict__ out, float* __restrict__ stats, int M, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (int64_t)row * D;
    float v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        const float d = c < D ? v[i] - mean : 0.f;
        q = fmaf(d, d, q);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
    const float* mr = mod ? mod + (int64_t)(row / rps) * mod_stride : nullptr;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            float n = (v[i] - mean) * rstd * w[c] + (b ? b[c] : 0.f);
            if (mr) n = fmaf(n, scale_off >= 0 ? mr[scale_off + c] : 1.f, shift_off >= 0 ? mr[shift_off + c] : 0.f);
            out[(int64_t)row * D + c] = n;
        }
    }
    if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// the same with 16-byte column groups (D % 4 == 0, aligned operands): 96 lanes' worth of requests per row of 384 instead of
// 6 x 64 four-byte ones (13.4 us for 31 MB at M = 10240: 2.3 TB/s)
__global__ __launch_bounds__(256) void k_ln_fwd_train4(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, const float* __restrict__ mod,
                                                       int64_t mod_stride, int shift_off, int scale_off, int rps,
                                                       float* __restrict__ out, float* __restrict__ stats, int M, int D4) {
    constexpr int C4 = LN_MAXC / 4;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
    const f32x4* xr = (const f32x4*)x + (int64_t)row * D4;
    const float Df = (float)(4 * D4);
    f32x4 v[C4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D4 ? xr[c] : zero4;
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / Df;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int c = lane + 64 * i;
        const f32x4 d = c < D4 ? v[i] - mean : zero4;
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    const float rstd = rsqrtf(wave_sum(q) / Df + 1e-5f);
    const float* mr = mod ? mod + (int64_t)(row / rps) * mod_stride : nullptr;
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) {
            f32x4 n = (v[i] - mean) * rstd * ((const f32x4*)w)[c] + (b ? ((const f32x4*)b)[c] : zero4);
            if (mr) {
                const f32x4 sc = scale_off >= 0 ? *(const f32x4*)(mr + scale_off + 4 * c) : one4;
                const f32x4 sh = shift_off >= 0 ? *(const f32x4*)(mr + shift_off + 4 * c) : zero4;
                n = (f32x4){fmaf(n.x, sc.x, sh.x), fmaf(n.y, sc.y, sh.y), fmaf(n.z, sc.z, sh.z), fmaf(n.w, sc.w, sh.w)};
            }
            ((f32x4*)out)[(int64_t)row * D4 + c] = n;
        }
    }
    if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// Branch merge + the LayerNorm that reads its result, one launch (round 6): every merge of the training forward --
// x_out = x + gate * dropout(a), k_merge_fwd4 -- is followed by the LayerNorm(+modulate) of the next sublayer on the rows it has
// just written.  One wave per row: the merged row is formed in registers, stored (the residual stream the backward needs), and
// normalised on the spot.  Same arithmetic per element as the two kernels: bit-identical to k_merge_fwd4 + k_ln_fwd_train4.
// l.x is ignored (it IS g.out); gate rows are indexed by row / g.rows_per_sample, modulation rows by row / rps.
__global__ __launch_bounds__(256) void k_merge_ln_fwd4(mdt_merge_args g, const float* __restrict__ w, const float* __restrict__ b,
                                                       const float* __restrict__ mod, int64_t mod_stride, int shift_off,
                                                       int scale_off, int rps, float* __restrict__ out, float* __restrict__ stats,
                                                       int M, int D4) {
    constexpr int C4 = LN_MAXC / 4;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
    const float Df = (float)(4 * D4);
    const float* gr = g.gate ? g.gate + (int64_t)(row / g.rows_per_sample) * g.gate_stride : nullptr;
    f32x4 v[C4];
    float s = 0.f;
    {
        f32x4 av[C4], xv[C4], gv[C4];
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int c = lane + 64 * i;
            av[i] = c < D4 ? ((const f32x4*)g.a)[(int64_t)row * D4 + c] : zero4;
            xv[i] = c < D4 ? ((const f32x4*)g.x)[(int64_t)row * D4 + c] : zero4;
            gv[i] = (c < D4 && gr) ? *(const f32x4*)(gr + 4 * c) : one4;
        }
#pragma unroll
        for (int i = 0; i < C4; ++i) {
            const int c = lane + 64 * i;
            const int64_t e = (int64_t)row * D4 + c;
            v[i] = zero4;
            if (c < D4) {
                const f32x4 ds = dropout_scale4(g.seed, g.site, (uint64_t)e * 4, g.p), t = av[i] * ds;
                v[i] = (f32x4){fmaf(gv[i].x, t.x, xv[i].x), fmaf(gv[i].y, t.y, xv[i].y), fmaf(gv[i].z, t.z, xv[i].z), fmaf(gv[i].w, t.w, xv[i].w)};
                ((f32x4*)g.out)[e] = v[i];
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / Df;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int c = lane + 64 * i;
        const f32x4 d = c < D4 ? v[i] - mean : zero4;
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    const float rstd = rsqrtf(wave_sum(q) / Df + 1e-5f);
    const float* mr = mod ? mod + (int64_t)(row / rps) * mod_stride : nullptr;
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) {
            f32x4 n = (v[i] - mean) * rstd * ((const f32x4*)w)[c] + (b ? ((const f32x4*)b)[c] : zero4);
            if (mr) {
                const f32x4 sc = scale_off >= 0 ? *(const f32x4*)(mr + scale_off + 4 * c) : one4;
                const f32x4 sh = shift_off >= 0 ? *(const f32x4*)(mr + shift_off + 4 * c) : zero4;
                n = (f32x4){fmaf(n.x, sc.x, sh.x), fmaf(n.y, sc.y, sh.y), fmaf(n.z, sc.z, sh.z), fmaf(n.w, sc.w, sh.w)};
            }
            ((f32x4*)out)[(int64_t)row * D4 + c] = n;
        }
    }
    if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

static bool ln_fwd_vec_ok(const mdt_ln_train_args& a) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return a.D % 4 == 0 && al(a.x) && al(a.w) && al(a.b) && al(a.out) &&
           (!a.mod || (al(a.mod) && a.mod_stride % 4 == 0 && (a.shift_off < 0 || a.shift_off % 4 == 0) &&
                       (a.scale_off < 0 || a.scale_off % 4 == 0)));
}
static bool merge_vec_ok(const mdt_merge_args& a, bool bwd);
hipError_t mdt_launch_merge_fwd(const mdt_merge_args& a, hipStream_t s);
// the merge g, then the LayerNorm l of its result (l.x is ignored): one launch where both take 16-byte accesses
hipError_t mdt_launch_merge_ln_fwd(const mdt_merge_args& g, const mdt_ln_train_args& l, hipStream_t s) {
    mdt_ln_train_args ll = l;
    ll.x = g.out;
    if (ll.D > 64 * LN_MAXC || ll.D < 1) return hipErrorInvalidValue;
    if (ln_fwd_vec_ok(ll) && merge_vec_ok(g, false) && g.D == ll.D && (int64_t)g.B * g.rows_per_sample == ll.M && g.out != g.a) {
        hipLaunchKernelGGL(k_merge_ln_fwd4, dim3((ll.M + 3) / 4), dim3(256), 0, s, g, ll.w, ll.b, ll.mod, ll.mod_stride, ll.shift_off,
                           ll.scale_off, ll.rows_per_sample > 0 ? ll.rows_per_sample : 1, ll.out, ll.stats, ll.M, ll.D / 4);
        return hipGetLastError();
    }
    hipError_t e = mdt_launch_merge_fwd(g, s);
    return e != hipSuccess ? e : mdt_launch_ln_fwd_train(ll, s);
}

hipError_t mdt_launch_ln_fwd_train(const mdt_ln_train_args& a, hipStream_t s) {
    if (a.D > 64 * LN_MAXC || a.D < 1) return hipErrorInvalidValue;
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.D % 4 == 0 && al(a.x) && al(a.w) && al(a.b) && al(a.out) &&
        (!a.mod || (al(a.mod) && a.mod_stride % 4 == 0 && (a.shift_off < 0 || a.shift_off % 4 == 0) &&
                    (a.scale_off < 0 || a.scale_off % 4 == 0)))) {
        hipLaunchKernelGGL(k_ln_fwd_train4, dim3((a.M + 3) / 4), dim3(256), 0, s, a.x, a.w, a.b, a.mod, a.mod_stride, a.shift_off,
                           a.scale_off, a.rows_per_sample > 0 ? a.rows_per_sample : 1, a.out, a.stats, a.M, a.D / 4);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_ln_fwd_train, dim3((a.M + 3) / 4), dim3(256), 0, s, a.x, a.w, a.b, a.mod, a.mod_stride,
                       a.shift_off, a.scale_off, a.rows_per_sample > 0 ? a.rows_per_sample : 1, a.out, a.stats, a.M, a.D);
    return hipGetLastError();
}

// LayerNorm (+ modulate) backward.  One workgroup per sample (rows_per_sample rows, a wave per row in turn):
//   dn   = dh * scale                     (scale = 1 without modulation)
//   dxh  = dn * w
//   dx  (+)= rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat))
//   per sample:  d_shift = sum_rows dh ;  d_scale = sum_rows dh * n ;  pw = sum_rows dn * xhat ;  pb = sum_rows dn
// pw / pb are per-sample partials of the weight / bias gradient (summed over samples by k_colsum afterwards).
// Samples with many rows (the Perceiver's ~400 media tokens) are cut into gridDim.y row chunks, each with its own
// partial row (pw / pb hold B * gridDim.y rows then); modulation gradients need the whole sample in one workgroup.
__global__ __launch_bounds__(256) void k_ln_bwd(mdt_ln_bwd_args a) {
    extern __shared__ float red[];  // [4 waves][4 kinds][D]
    const int b = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D = a.D, rps = a.rows_per_sample;
    const int per = (rps + gridDim.y - 1) / gridDim.y, r_lo = blockIdx.y * per, r_hi = min(rps, r_lo + per);
    const int prow = b * gridDim.y + blockIdx.y;  // row of the pw / pb partials
    const float* mr = a.mod ? a.mod + (int64_t)b * a.mod_stride : nullptr;
    float wgt[LN_MAXC], bia[LN_MAXC], sc[LN_MAXC];
    float a_sh[LN_MAXC], a_sc[LN_MAXC], a_w[LN_MAXC], a_b[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        wgt[i] = c < D ? a.w[c] : 0.f;
        bia[i] = (c < D && a.b) ? a.b[c] : 0.f;
        sc[i] = c < D ? ((mr && a.scale_off >= 0) ? mr[a.scale_off + c] : 1.f) : 0.f;
        a_sh[i] = a_sc[i] = a_w[i] = a_b[i] = 0.f;
    }
    for (int r = r_lo + wv; r < r_hi; r += 4) {
        const int64_t row = (int64_t)b * rps + r;
        const float mean = a.stats[2 * row], rstd = a.stats[2 * row + 1];
        float xh[LN_MAXC], dxh[LN_MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            const float xv = c < D ? a.x[row * D + c] : 0.f;
            const float dh = c < D ? a.dh[row * a.ld_dh + c] : 0.f;
            xh[i] = c < D ? (xv - mean) * rstd : 0.f;
            const float n = fmaf(xh[i], wgt[i], bia[i]);
            const float dn = dh * sc[i];
            a_sh[i] += dh;
            a_sc[i] = fmaf(dh, n, a_sc[i]);
            a_w[i] = fmaf(dn, xh[i], a_w[i]);
            a_b[i] += dn;
            dxh[i] = dn * wgt[i];
            s1 += dxh[i];
            s2 = fmaf(dxh[i], xh[i], s2);
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float g = rstd * (dxh[i] - c1 - xh[i] * c2);
                float* p = a.dx + row * D + c;
                *p = a.accumulate ? *p + g : g;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            red[(wv * 4 + 0) * D + c] = a_sh[i];
            red[(wv * 4 + 1) * D + c] = a_sc[i];
            red[(wv * 4 + 2) * D + c] = a_w[i];
            red[(wv * 4 + 3) * D + c] = a_b[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = red[(0 * 4 + k) * D + c] + red[(1 * 4 + k) * D + c] + red[(2 * 4 + k) * D + c] +
                                           red[(3 * 4 + k) * D + c];
        if (a.d_mod) {
            // accumulate_dmod: the same conditioning vector feeds several LayerNorms (NoiseBlock adds c at every
            // attention input), whose launches follow each other on the stream and add into one row
            float* dm = a.d_mod + (int64_t)b * a.d_mod_stride;
            if (a.shift_off >= 0) dm[a.shift_off + c] = a.accumulate_dmod ? dm[a.shift_off + c] + t[0] : t[0];
            if (a.scale_off >= 0) dm[a.scale_off + c] = a.accumulate_dmod ? dm[a.scale_off + c] + t[1] : t[1];
        }
        a.pw[(int64_t)prow * D + c] = t[2];
        if (a.pb) a.pb[(int64_t)prow * D + c] = t[3];
    }
}

// the same with 16-byte column groups; the row of dx an accumulating call adds to is requested with x and dh, not after the
// row's two reductions.
// MG (round 6): the branch merge that FOLLOWS this LayerNorm in the backward order rides along.  Every LayerNorm backward of a
// block is followed by the backward of the merge in front of it -- d_a = mask/(1-p) * gate * d_x, d_gate = sum_rows d_x * drop(a)
// (k_merge_bwd4) -- on exactly the rows this workgroup has just finished (one workgroup per sample in both), so the freshly
// accumulated d_x row goes from registers into both: one launch and one read of d_x less per sublayer (20 a step).
// g.x is ignored (it IS a.dx); g.out must not be a.dh's buffer unless it is the same rows (it is the same thread either way).
// RB = rows a wave has in flight at once: a wave's rows wv, wv + 4, ... were one dependent memory round trip each (the store of a
// row may alias the next row's loads for all the compiler knows, so nothing was requested ahead: three trips per launch at ten
// rows per sample); now every operand of up to RB rows is requested before the first reduction and the first store.
template <bool MG, int RB>
__global__ __launch_bounds__(256) void k_ln_bwd4(mdt_ln_bwd_args a, mdt_merge_args g) {
    extern __shared__ float red[];  // [4 waves][4 (+ 1) kinds][D]
    constexpr int C4 = LN_MAXC / 4, NK = MG ? 5 : 4;
    const int b = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D = a.D, D4 = D >> 2, rps = a.rows_per_sample;
    const int per = (rps + gridDim.y - 1) / gridDim.y, r_lo = blockIdx.y * per, r_hi = min(rps, r_lo + per);
    const int prow = b * gridDim.y + blockIdx.y;
    const float* mr = a.mod ? a.mod + (int64_t)b * a.mod_stride : nullptr;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
    const float Df = (float)D;
    f32x4 wgt[C4], bia[C4], sc[C4], a_sh[C4], a_sc[C4], a_w[C4], a_b[C4], gt[C4], a_g[C4];
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int c = lane + 64 * i;
        wgt[i] = c < D4 ? ((const f32x4*)a.w)[c] : zero4;
        bia[i] = (c < D4 && a.b) ? ((const f32x4*)a.b)[c] : zero4;
        sc[i] = c < D4 ? ((mr && a.scale_off >= 0) ? *(const f32x4*)(mr + a.scale_off + 4 * c) : one4) : zero4;
        a_sh[i] = a_sc[i] = a_w[i] = a_b[i] = a_g[i] = zero4;
        if constexpr (MG) gt[i] = (c < D4 && g.gate) ? *(const f32x4*)(g.gate + (int64_t)b * g.gate_stride + 4 * c) : one4;
    }
    const int64_t ld4 = a.ld_dh >> 2;
    for (int r0 = r_lo + wv; r0 < r_hi; r0 += 4 * RB) {
        f32x4 xv[RB][C4], dhv[RB][C4], old[RB][C4], br[MG ? RB : 1][C4];
        float mean[RB], rstd[RB];
        // ---- every request of the batch (rows past the end: the wave's first row again, never stored) ----
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + 4 * u < r_hi ? r0 + 4 * u : r0;
            const int64_t row = (int64_t)b * rps + r;
            mean[u] = a.stats[2 * row]; rstd[u] = a.stats[2 * row + 1];
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int c = min(lane + 64 * i, D4 - 1);
                xv[u][i] = ((const f32x4*)a.x)[row * D4 + c];
                dhv[u][i] = ((const f32x4*)a.dh)[row * ld4 + c];
                old[u][i] = a.accumulate ? ((const f32x4*)a.dx)[row * D4 + c] : zero4;
                if constexpr (MG) br[u][i] = ((const f32x4*)g.a)[row * D4 + c];
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + 4 * u;
            if (r >= r_hi) continue;  // wave-uniform
            const int64_t row = (int64_t)b * rps + r;
            f32x4 xh[C4], dxh[C4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const bool in = lane + 64 * i < D4;
                const f32x4 dh = in ? dhv[u][i] : zero4;
                xh[i] = in ? (xv[u][i] - mean[u]) * rstd[u] : zero4;
                const f32x4 n = xh[i] * wgt[i] + bia[i];
                const f32x4 dn = dh * sc[i];
                a_sh[i] += dh;
                a_sc[i] += dh * n;
                a_w[i] += dn * xh[i];
                a_b[i] += dn;
                dxh[i] = dn * wgt[i];
                s1 += (dxh[i].x + dxh[i].y) + (dxh[i].z + dxh[i].w);
                const f32x4 t = dxh[i] * xh[i];
                s2 += (t.x + t.y) + (t.z + t.w);
            }
            const float c1 = wave_sum(s1) / Df, c2 = wave_sum(s2) / Df;
#pragma unroll
            for (int i = 0; i < C4; ++i) {
                const int c = lane + 64 * i;
                if (c < D4) {
                    const f32x4 d = old[u][i] + (dxh[i] - c1 - xh[i] * c2) * rstd[u];
                    ((f32x4*)a.dx)[row * D4 + c] = d;
                    if constexpr (MG) {  // k_merge_bwd4's arithmetic on the row just formed
                        const int64_t e = row * D4 + c;
                        const f32x4 ds = dropout_scale4(g.seed, g.site, (uint64_t)e * 4, g.p), t = br[u][i] * ds;
                        a_g[i] = (f32x4){fmaf(d.x, t.x, a_g[i].x), fmaf(d.y, t.y, a_g[i].y), fmaf(d.z, t.z, a_g[i].z), fmaf(d.w, t.w, a_g[i].w)};
                        ((f32x4*)g.out)[e] = gt[i] * d * ds;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < C4; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) {
            *(f32x4*)(red + (wv * NK + 0) * D + 4 * c) = a_sh[i];
            *(f32x4*)(red + (wv * NK + 1) * D + 4 * c) = a_sc[i];
            *(f32x4*)(red + (wv * NK + 2) * D + 4 * c) = a_w[i];
            *(f32x4*)(red + (wv * NK + 3) * D + 4 * c) = a_b[i];
            if constexpr (MG) *(f32x4*)(red + (wv * NK + 4) * D + 4 * c) = a_g[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float t[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) t[k] = red[(0 * NK + k) * D + c] + red[(1 * NK + k) * D + c] + red[(2 * NK + k) * D + c] +
                                            red[(3 * NK + k) * D + c];
        if (a.d_mod) {
            float* dm = a.d_mod + (int64_t)b * a.d_mod_stride;
            if (a.shift_off >= 0) dm[a.shift_off + c] = a.accumulate_dmod ? dm[a.shift_off + c] + t[0] : t[0];
            if (a.scale_off >= 0) dm[a.scale_off + c] = a.accumulate_dmod ? dm[a.scale_off + c] + t[1] : t[1];
        }
        a.pw[(int64_t)prow * D + c] = t[2];
        if (a.pb) a.pb[(int64_t)prow * D + c] = t[3];
        if constexpr (MG) { if (g.dgate) g.dgate[(int64_t)b * g.dgate_stride + c] = t[NK - 1]; }
    }
}

template <bool MG>
static void launch_ln_bwd4(const mdt_ln_bwd_args& a, const mdt_merge_args& g, int chunks, hipStream_t s) {
    static int rb_force = -1;  // MDT_HIP_LNB_ROWS=1..3: rows a wave keeps in flight (A/B runs; default: all of them, up to 3)
    if (rb_force < 0) { const char* e = getenv("MDT_HIP_LNB_ROWS"); rb_force = e ? atoi(e) : 0; }
    const int per = (a.rows_per_sample + chunks - 1) / chunks;
    int rb = per <= 4 ? 1 : (per <= 8 ? 2 : 3);
    if (rb_force >= 1 && rb_force <= 3) rb = rb_force;
    const size_t lds = (size_t)4 * (MG ? 5 : 4) * a.D * sizeof(float);
    if (rb == 1) hipLaunchKernelGGL((k_ln_bwd4<MG, 1>), dim3(a.B, chunks), dim3(256), lds, s, a, g);
    else if (rb == 2) hipLaunchKernelGGL((k_ln_bwd4<MG, 2>), dim3(a.B, chunks), dim3(256), lds, s, a, g);
    else hipLaunchKernelGGL((k_ln_bwd4<MG, 3>), dim3(a.B, chunks), dim3(256), lds, s, a, g);
}

static bool ln_bwd_vec_ok(const mdt_ln_bwd_args& a) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return a.D % 4 == 0 && a.ld_dh % 4 == 0 && al(a.x) && al(a.dh) && al(a.dx) && al(a.w) && al(a.b) &&
           (!a.mod || (al(a.mod) && a.mod_stride % 4 == 0 && (a.scale_off < 0 || a.scale_off % 4 == 0)));
}
hipError_t mdt_launch_ln_bwd(const mdt_ln_bwd_args& a, hipStream_t s) {
    if (a.D > 64 * LN_MAXC || a.D < 1 || a.rows_per_sample < 1) return hipErrorInvalidValue;
    const int chunks = a.row_chunks > 1 ? a.row_chunks : 1;
    if (chunks > 1 && (a.d_mod || chunks > a.rows_per_sample)) return hipErrorInvalidValue;
    if (ln_bwd_vec_ok(a)) {
        launch_ln_bwd4<false>(a, mdt_merge_args{}, chunks, s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_ln_bwd, dim3(a.B, chunks), dim3(256), (size_t)16 * a.D * sizeof(float), s, a);
    return hipGetLastError();
}

static bool merge_vec_ok(const mdt_merge_args& a, bool bwd);
hipError_t mdt_launch_merge_bwd(const mdt_merge_args& a, hipStream_t s);
// LayerNorm backward, then the backward of the branch merge `g` on the gradient it leaves in a.dx (g.x is ignored): one launch
// where both take 16-byte accesses and a sample is one workgroup, else the two kernels one after the other
hipError_t mdt_launch_ln_bwd_merge(const mdt_ln_bwd_args& a, const mdt_merge_args& g, hipStream_t s) {
    mdt_merge_args gg = g;
    gg.x = a.dx;
    const int chunks = a.row_chunks > 1 ? a.row_chunks : 1;
    if (a.D > 64 * LN_MAXC || a.D < 1 || a.rows_per_sample < 1) return hipErrorInvalidValue;
    if (chunks == 1 && ln_bwd_vec_ok(a) && merge_vec_ok(gg, true) && gg.B == a.B && gg.rows_per_sample == a.rows_per_sample &&
        gg.D == a.D) {
        launch_ln_bwd4<true>(a, gg, 1, s);
        return hipGetLastError();
    }
    hipError_t e = mdt_launch_ln_bwd(a, s);
    return e != hipSuccess ? e : mdt_launch_merge_bwd(gg, s);
}

// ------------------------------------------------------------------------------------------------
// elementwise activation forward / backward on a pre-activation buffer      (nn.GELU :171, nn.Mish, nn.SiLU :251)
// ------------------------------------------------------------------------------------------------
__global__ void k_act_fwd(const float* __restrict__ u, float* __restrict__ out, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = apply_act1(u[i], act);
}
__global__ void k_act_bwd(const float* __restrict__ u, const float* dy, float* du, int64_t n, int act) {  // du may alias dy
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) du[i] = dy[i] * apply_act_grad1(u[i], act);
}
hipError_t mdt_launch_act_fwd(const float* u, float* out, int64_t n, int act, hipStream_t s) {
    hipLaunchKernelGGL(k_act_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, out, n, act);
    return hipGetLastError();
}
hipError_t mdt_launch_act_bwd(const float* u, const float* dy, float* du, int64_t n, int act, hipStream_t s) {
    hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, dy, du, n, act);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// branch merge   x_out = x + gate[sample] * dropout(a)                      (transformer_blocks.py:156,176,296-307)
// forward, and backward:  d_a = mask/(1-p) * gate * d_x ;  d_gate[sample] = sum_rows d_x * dropout(a)
// (one workgroup per sample; gate / d_gate optional; the mask is regenerated from (seed, site, row*D + column))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_merge_fwd(mdt_merge_args a) {
    const int b = blockIdx.x, D = a.D, rps = a.rows_per_sample;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float g = a.gate ? a.gate[(int64_t)b * a.gate_stride + c] : 1.f;
        for (int r = 0; r < rps; ++r) {
            const int64_t i = ((int64_t)b * rps + r) * D + c;
            const float av = a.a[i] * dropout_scale(a.seed, a.site, (uint64_t)i, a.p);
            a.out[i] = fmaf(g, av, a.x[i]);
        }
    }
}
__global__ __launch_bounds__(256) void k_merge_bwd(mdt_merge_args a) {
    const int b = blockIdx.x, D = a.D, rps = a.rows_per_sample;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float g = a.gate ? a.gate[(int64_t)b * a.gate_stride + c] : 1.f;
        float acc = 0.f;
        for (int r = 0; r < rps; ++r) {
            const int64_t i = ((int64_t)b * rps + r) * D + c;
            const float d = a.x[i], ds = dropout_scale(a.seed, a.site, (uint64_t)i, a.p);
            acc = fmaf(d, a.a[i] * ds, acc);
            a.out[i] = g * d * ds;
        }
        if (a.dgate) a.dgate[(int64_t)b * a.dgate_stride + c] = acc;
    }
}
// The same two with 16-byte accesses and every load of a thread requested before its first store (round 4).  Above, a thread
// walks the rows of ONE column: load a and x, store out, next row -- and as `out` may alias `a` or `x` for all the compiler knows,
// the next row's loads wait behind the store: rows_per_sample dependent memory round trips per launch (9.2 / 12.4 us each way
// at B = 1024, D = 384, twenty of each per training step).  Here: thread = (row lane rl, column group c4); a thread's rows
// rl, rl + RL, ... (at most MERGE_U of them) are all in flight at once.  Same arithmetic per element, so the forward is bit
// for bit the kernel above; the gate gradient sums a sample's rows as (rows of lane 0) + (rows of lane 1) + ...
constexpr int MERGE_U = 8;
__global__ __launch_bounds__(256) void k_merge_fwd4(mdt_merge_args a) {
    const int b = blockIdx.x, D4 = a.D >> 2, rps = a.rows_per_sample, RL = 256 / D4;
    const int c4 = threadIdx.x % D4, rl = threadIdx.x / D4;
    if (rl >= RL) return;
    const f32x4* A = (const f32x4*)a.a;  // no __restrict__: `out` may be `a` or `x` (in-place merges); the batches below
    const f32x4* X = (const f32x4*)a.x;  // already request every row of a thread before its first store
    f32x4* O = (f32x4*)a.out;
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
    const f32x4 g = a.gate ? *(const f32x4*)(a.gate + (int64_t)b * a.gate_stride + 4 * c4) : one4;
    for (int r0 = rl; r0 < rps; r0 += RL * MERGE_U) {
        f32x4 av[MERGE_U], xv[MERGE_U];
#pragma unroll
        for (int u = 0; u < MERGE_U; ++u) {
            const int r = r0 + u * RL < rps ? r0 + u * RL : r0;  // past the end: the thread's own first row again (never a row another thread stores)
            const int64_t i = ((int64_t)b * rps + r) * D4 + c4;
            av[u] = A[i];
            xv[u] = X[i];
        }
#pragma unroll
        for (int u = 0; u < MERGE_U; ++u) {
            const int r = r0 + u * RL;
            if (r < rps) {
                const int64_t i = ((int64_t)b * rps + r) * D4 + c4;
                const f32x4 ds = dropout_scale4(a.seed, a.site, (uint64_t)i * 4, a.p), t = av[u] * ds;
                O[i] = (f32x4){fmaf(g.x, t.x, xv[u].x), fmaf(g.y, t.y, xv[u].y), fmaf(g.z, t.z, xv[u].z), fmaf(g.w, t.w, xv[u].w)};
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_merge_bwd4(mdt_merge_args a) {
    __shared__ f32x4 red[256];
    const int b = blockIdx.x, D4 = a.D >> 2, rps = a.rows_per_sample, RL = 256 / D4;
    const int c4 = threadIdx.x % D4, rl = threadIdx.x / D4;
    const f32x4* A = (const f32x4*)a.a;  // no __restrict__: `out` may be `a` or `x` (in-place merges); the batches below
    const f32x4* X = (const f32x4*)a.x;  // already request every row of a thread before its first store
    f32x4* O = (f32x4*)a.out;
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rl < RL) {
        const f32x4 g = a.gate ? *(const f32x4*)(a.gate + (int64_t)b * a.gate_stride + 4 * c4) : one4;
        for (int r0 = rl; r0 < rps; r0 += RL * MERGE_U) {
            f32x4 av[MERGE_U], dv[MERGE_U];
#pragma unroll
            for (int u = 0; u < MERGE_U; ++u) {
                const int r = r0 + u * RL < rps ? r0 + u * RL : r0;  // past the end: the thread's own first row again (never a row another thread stores)
                const int64_t i = ((int64_t)b * rps + r) * D4 + c4;
                av[u] = A[i];
                dv[u] = X[i];
            }
#pragma unroll
            for (int u = 0; u < MERGE_U; ++u) {
                const int r = r0 + u * RL;
                if (r < rps) {
                    const int64_t i = ((int64_t)b * rps + r) * D4 + c4;
                    const f32x4 ds = dropout_scale4(a.seed, a.site, (uint64_t)i * 4, a.p), t = av[u] * ds, d = dv[u];
                    acc = (f32x4){fmaf(d.x, t.x, acc.x), fmaf(d.y, t.y, acc.y), fmaf(d.z, t.z, acc.z), fmaf(d.w, t.w, acc.w)};
                    O[i] = g * d * ds;
                }
            }
        }
    }
    if (a.dgate == nullptr) return;
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int q = 1; q < RL; ++q) acc += red[q * D4 + c4];
        *(f32x4*)(a.dgate + (int64_t)b * a.dgate_stride + 4 * c4) = acc;
    }
}
static bool merge_vec_ok(const mdt_merge_args& a, bool bwd) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return a.D % 4 == 0 && a.D >= 4 && a.D <= 1024 && al(a.x) && al(a.a) && al(a.out) &&
           (!a.gate || (al(a.gate) && a.gate_stride % 4 == 0)) && (!bwd || !a.dgate || (al(a.dgate) && a.dgate_stride % 4 == 0));
}
hipError_t mdt_launch_merge_fwd(const mdt_merge_args& a, hipStream_t s) {
    if (merge_vec_ok(a, false)) hipLaunchKernelGGL(k_merge_fwd4, dim3(a.B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_merge_fwd, dim3(a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t mdt_launch_merge_bwd(const mdt_merge_args& a, hipStream_t s) {
    if (merge_vec_ok(a, true)) hipLaunchKernelGGL(k_merge_bwd4, dim3(a.B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_merge_bwd, dim3(a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// column sums  out[n] (+)= sum_m X[m][n]   (bias gradients; reduction of per-sample partials)
// grid.x = column groups of 64, 4 row groups per workgroup meet in LDS; deterministic.  Tall inputs with few columns
// (the action head's (B*Ta, 7) gradient, LayerNorm partials of a large batch) would run on a handful of workgroups:
// they go through gridDim.y row slices into a small scratch and a second pass over the slices.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ X, int64_t ldx, int M, int N,
                                                float* __restrict__ out, int accumulate) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (gridDim.y > 1) {  // row slice blockIdx.y -> row blockIdx.y of the (gridDim.y, N) scratch `out`
        const int per = (M + gridDim.y - 1) / gridDim.y, lo = blockIdx.y * per;
        X += (int64_t)lo * ldx;
        M = max(0, min(per, M - lo));
        out += (int64_t)blockIdx.y * N;
    }
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (c < N) {
        int m = rg;
        for (; m + 12 < M; m += 16) {
            acc0 += X[(int64_t)m * ldx + c];
            acc1 += X[(int64_t)(m + 4) * ldx + c];
            acc2 += X[(int64_t)(m + 8) * ldx + c];
            acc3 += X[(int64_t)(m + 12) * ldx + c];
        }
        for (; m < M; m += 4) acc0 += X[(int64_t)m * ldx + c];
    }
    part[rg][cl] = (acc0 + acc1) + (acc2 + acc3);
    __syncthreads();
    if (rg == 0 && c < N) {
        const float t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}
// two independent column sums of equal shape in one launch (LayerNorm weight / bias partials)
__global__ __launch_bounds__(256) void k_colsum2(const float* __restrict__ X0, const float* __restrict__ X1, int64_t ldx, int M,
                                                 int N, float* __restrict__ out0, float* __restrict__ out1, int accumulate) {
    __shared__ float part[4][64];
    const float* X = blockIdx.y ? X1 : X0;
    float* out = blockIdx.y ? out1 : out0;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < N)
        for (int m = rg; m < M; m += 4) acc += X[(int64_t)m * ldx + c];
    part[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < N) {
        const float t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}
// many independent column sums in ONE launch (the training backward's small reductions -- LayerNorm weight / bias
// partials of every block, per-slice bias partials of every Linear -- were ~60 launches of 6 workgroups per step, each a
// serialized ~5 us): blockIdx.y = table entry, blockIdx.x = group of 32 columns, 8 row groups x 8 independent accumulators
// per thread (64 rows in flight).  Rows are added in a fixed order: deterministic.
__global__ __launch_bounds__(256) void k_colsum_batched(mdt_colsum_table tab) {
    __shared__ float part[8][32];
    const mdt_colsum_entry e = tab.e[blockIdx.y];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    if (blockIdx.x * 32 >= e.N) return;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    if (c < e.N) {
        int m = rg;
        for (; m + 56 < e.M; m += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += e.src[(int64_t)(m + 8 * u) * e.ld + c];
        }
        for (; m < e.M; m += 8) acc[0] += e.src[(int64_t)m * e.ld + c];
    }
    part[rg][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (rg == 0 && c < e.N) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][cl];
        e.dst[c] = e.accumulate ? e.dst[c] + t : t;
    }
}
hipError_t mdt_launch_colsum_batched(const mdt_colsum_entry* entries, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += MDT_COLSUM_TABLE) {
        mdt_colsum_table tab;
        const int cnt = n - i0 < MDT_COLSUM_TABLE ? n - i0 : MDT_COLSUM_TABLE;
        int gx = 1;
        for (int i = 0; i < cnt; ++i) {
            tab.e[i] = entries[i0 + i];
            gx = gx > (entries[i0 + i].N + 31) / 32 ? gx : (entries[i0 + i].N + 31) / 32;
        }
        hipLaunchKernelGGL(k_colsum_batched, dim3(gx, cnt), dim3(256), 0, s, tab);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t mdt_launch_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, hipStream_t s);
hipError_t mdt_launch_colsum2(const float* X0, const float* X1, int64_t ldx, int M, int N, float* out0, float* out1,
                              int accumulate, hipStream_t s) {
    if (M >= 512) {  // deep sums: the sliced two-stage reduction, once per operand (63 -> 2 x 9 us at M = 1024, N = 384)
        hipError_t e = mdt_launch_colsum(X0, ldx, M, N, out0, accumulate, s);
        return e != hipSuccess ? e : mdt_launch_colsum(X1, ldx, M, N, out1, accumulate, s);
    }
    hipLaunchKernelGGL(k_colsum2, dim3((N + 63) / 64, 2), dim3(256), 0, s, X0, X1, ldx, M, N, out0, out1, accumulate);
    return hipGetLastError();
}

// 64 row slices x 4096 columns (1 MiB) of scratch PER STREAM: uses on one stream are ordered, two streams (two
// handles training side by side) must not share it
static const int CS_SLICES = 64, CS_MAXN = 4096;
static std::mutex g_cs_mu;
static std::unordered_map<hipStream_t, float*> g_cs_scratch;

static float* colsum_scratch(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_cs_mu);
    auto it = g_cs_scratch.find(s);
    if (it != g_cs_scratch.end()) return it->second;
    float* p = nullptr;
    if (hipMalloc((void**)&p, (size_t)CS_SLICES * CS_MAXN * sizeof(float)) != hipSuccess) return nullptr;
    g_cs_scratch[s] = p;
    return p;
}

// very many columns (the S slices of a weight gradient, N = the flattened matrix): the kernel above with 16-byte column groups --
// 256 columns per workgroup instead of 64, four loads of a thread in flight (c_fc's gradient: 9216 workgroups of 5 KB each,
// 12 us a launch, 41 such launches a training step).  Same order of the additions as k_colsum.
__global__ __launch_bounds__(256) void k_colsum_wide4(const float* __restrict__ X, int64_t ldx4, int M, int N4,
                                                      float* __restrict__ out, int accumulate) {
    __shared__ f32x4 part[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + cl;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc0 = zero4, acc1 = zero4, acc2 = zero4, acc3 = zero4;
    if (c4 < N4) {
        const f32x4* __restrict__ X4 = (const f32x4*)X + c4;
        int m = rg;
        for (; m + 12 < M; m += 16) {
            const f32x4 v0 = X4[(int64_t)m * ldx4], v1 = X4[(int64_t)(m + 4) * ldx4], v2 = X4[(int64_t)(m + 8) * ldx4],
                        v3 = X4[(int64_t)(m + 12) * ldx4];
            acc0 += v0; acc1 += v1; acc2 += v2; acc3 += v3;
        }
        for (; m < M; m += 4) acc0 += X4[(int64_t)m * ldx4];
    }
    part[rg][cl] = (acc0 + acc1) + (acc2 + acc3);
    __syncthreads();
    if (rg == 0 && c4 < N4) {
        const f32x4 t = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        f32x4* o = (f32x4*)out + c4;
        *o = accumulate ? *o + t : t;
    }
}

hipError_t mdt_launch_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, hipStream_t s) {
    if (N >= 16384 && N % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)X | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(k_colsum_wide4, dim3((N / 4 + 63) / 64), dim3(256), 0, s, X, ldx / 4, M, N / 4, out, accumulate);
        return hipGetLastError();
    }
    if (M >= 512 && N <= CS_MAXN) {  // 64 row slices in parallel, then their sum (a single stage runs on N / 64 workgroups)
        float* scratch = colsum_scratch(s);
        if (!scratch) return hipErrorOutOfMemory;
        hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64, CS_SLICES), dim3(256), 0, s, X, ldx, M, N, scratch, 0);
        hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64), dim3(256), 0, s, scratch, (int64_t)N, CS_SLICES, N, out, accumulate);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64), dim3(256), 0, s, X, ldx, M, N, out, accumulate);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// attention backward for the small sequences of this model (Tq, Tk <= 16): one wave per (sample, head).
//   P recomputed from q, k (same masking as k_attn);  dV = P^T dO ;  dP = dO V^T ;
//   dS = P * (dP - rowsum(dP * P)) * scale ;  dQ = dS K ;  dK = dS^T Q        (F.scaled_dot_product_attention :142)
// ------------------------------------------------------------------------------------------------
// RoPE on rows held in LDS: feature pairs (2i, 2i+1), i < 16, of row r rotate by angle pos(r) * freq_i
// (position_embeddings.py:56-70,138-142; tables cos/sin[pos][i], 16 x 16).  sign = -1 applies the transpose, which is
// what carries a gradient with respect to the rotated vector back to the unrotated one.
// LDS rows of these kernels are HD + 4 floats apart: 16-byte aligned, so that every inner loop below moves four features per
// LDS instruction (the first versions read 4 bytes per FMA operand and were bound by LDS instruction issue).
template <int HD>
__device__ __forceinline__ void rope_rows(float (*rows)[HD + 4], int T, const float* __restrict__ rc,
                                          const float* __restrict__ rs, float sign, int lane) {
    for (int e = lane; e < T * 16; e += 64) {
        const int r = e >> 4, i = e & 15;
        if (2 * i + 1 < HD) {
            const float c = rc[r * 16 + i], sn = sign * rs[r * 16 + i];
            const float x1 = rows[r][2 * i], x2 = rows[r][2 * i + 1];
            rows[r][2 * i] = x1 * c - x2 * sn;
            rows[r][2 * i + 1] = x2 * c + x1 * sn;
        }
    }
}

// NB head slices (rows of HD floats at column h * HD) of one sample: global -> LDS, all 16-byte loads of the wave requested
// before the first is consumed.  T[b] rows each (<= 16), rows beyond are left alone.
template <int HD, int NB>
__device__ __forceinline__ void load_head_rows(const float* const (&src)[NB], const int64_t (&ld)[NB], const int (&T)[NB],
                                               float (*const (&dst)[NB])[HD + 4], int lane) {
    constexpr int H4 = HD / 4, NI = (16 * H4 + 63) / 64;
    f32x4 t[NB][NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = lane + 64 * u, r = i / H4, c = 4 * (i - r * H4);
#pragma unroll
        for (int b = 0; b < NB; ++b) t[b][u] = ldg4(src[b] + (int64_t)min(r, T[b] - 1) * ld[b] + c);
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = lane + 64 * u, r = i / H4, c = 4 * (i - r * H4);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (r < T[b]) *(f32x4*)&dst[b][r][c] = t[b][u];
    }
}

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc);
    return fmaf(a.w, b.w, acc);
}

// Dropout multipliers of one (sample, head)'s Tq x Tk probabilities into LDS, by ALL lanes: element (i, j) has index base0 + i Tk + j
// and takes word idx & 3 of Philox block idx >> 2, so the ~Tq Tk / 4 blocks go one per lane (round 6; before, lane i < Tq ran
// Tk whole blocks one after the other inside its softmax loop -- ~1000 integer instructions on 10 of 64 lanes, more than the
// rest of the kernel).  No barrier inside: the caller's next __syncthreads() publishes the table.
__device__ __forceinline__ void attn_dropout_table(float (*mk)[17], uint64_t seed, uint32_t site, uint64_t base0, int Tq, int Tk,
                                                   float p, int lane) {
    const int n = Tq * Tk;
    const uint64_t blk0 = base0 >> 2;
    const int nblk = (int)(((base0 + n - 1) >> 2) - blk0) + 1;
    for (int q = lane; q < nblk; q += 64) {
        const philox4_t r = philox4(seed, site, blk0 + q);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int64_t e = (int64_t)(((blk0 + q) << 2) + w) - (int64_t)base0;
            if (e >= 0 && e < n) mk[e / Tk][e % Tk] = dropout_keep(r.w[w], p);
        }
    }
}

// training forward with dropout on the probabilities: out = (mask/(1-p) * softmax(q k^T / sqrt(hd))) v
template <int HD>
__global__ __launch_bounds__(64) void k_attn_fwd_train(mdt_attn_train_args a, float scale) {
    __shared__ __attribute__((aligned(16))) float qs[16][HD + 4], ks[16][HD + 4], vs[16][HD + 4];
    __shared__ float P[16][17], Mk[16][17];
    constexpr int H4 = HD / 4;
    const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int Tq = a.Tq, Tk = a.Tk;
    const bool drop = a.p > 0.f && a.seed != 0;
    if (drop) attn_dropout_table(Mk, a.seed, a.site, ((uint64_t)b * a.H + h) * Tq * Tk, Tq, Tk, a.p, lane);
    {
        const float* const src[3] = {a.q + (int64_t)b * Tq * a.ldq + h * HD, a.k + (int64_t)b * Tk * a.ldkv + h * HD,
                                     a.v + (int64_t)b * Tk * a.ldkv + h * HD};
        const int64_t ld[3] = {a.ldq, a.ldkv, a.ldkv};
        const int T[3] = {Tq, Tk, Tk};
        float (*const dst[3])[HD + 4] = {qs, ks, vs};
        load_head_rows<HD, 3>(src, ld, T, dst, lane);
    }
    __syncthreads();
    if (a.rope) {
        rope_rows<HD>(qs, Tq, a.rope_cos, a.rope_sin, 1.f, lane);
        rope_rows<HD>(ks, Tk, a.rope_cos, a.rope_sin, 1.f, lane);
        __syncthreads();
    }
    // one (query, key) pair per lane
    for (int e = lane; e < Tq * Tk; e += 64) {
        const int i = e / Tk, j = e - i * Tk;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < H4; ++c) s = dot4(*(const f32x4*)&qs[i][4 * c], *(const f32x4*)&ks[j][4 * c], s);
        P[i][j] = (!a.causal || j <= i) ? s * scale : -INFINITY;
    }
    __syncthreads();
    if (lane < Tq) {
        const int i = lane;
        float sc[16];
        float mx = -INFINITY;
        for (int j = 0; j < Tk; ++j) { sc[j] = P[i][j]; mx = fmaxf(mx, sc[j]); }
        float sum = 0.f;
        for (int j = 0; j < Tk; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
        const float inv = 1.f / sum;
        for (int j = 0; j < Tk; ++j) P[i][j] = sc[j] * inv * (drop ? Mk[i][j] : 1.f);
    }
    __syncthreads();
    for (int e = lane; e < Tq * H4; e += 64) {  // (row, four features) per lane
        const int i = e / H4, c = 4 * (e - i * H4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Tk; ++j) acc += P[i][j] * *(const f32x4*)&vs[j][c];
        *(f32x4*)(a.out + ((int64_t)b * Tq + i) * a.ldo + h * HD + c) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// The same two kernels on the MFMA pipe (round 6): one workgroup per (sample, group of HG heads), wave = head.
// k_attn_fwd_train / k_attn_bwd above give every (sample, head) its own one-wave workgroup: 8192 of them at B = 1024, each
// loading 192-byte row pieces and walking ~100 (query, key) pairs with vector FMAs out of 4- and 16-byte LDS reads.  Here the
// HG heads of a sample share coalesced row loads (HG * hd contiguous floats per row), and every product is a handful of
// v_mfma_f32_16x16x4_f32 on fragments read from LDS, the way attn_sample_tile (mdt_tiles.h) runs the sampler's attention:
//   scores with TRANSPOSED operand roles (A = key rows, B = query rows): the lane ends with S[query m][keys 4 g .. 4 g + 3],
//   m = lane % 16, g = lane / 16; the masked softmax of a query runs over the lane's four values and the wave's four 16-lane
//   rows (v_permlane swaps); P (times its dropout multipliers), still in those registers, is the B operand of O^T = V^T P^T.
// Backward, same orientation: dP = dO V^T like the scores; dS = P (m dP - delta) scale; dQ = dS K like O; dK = dS^T Q and
// dV = (m P)^T dO contract over the QUERY index, which the registers hold the wrong way round: the two 16 x 16 tiles go through
// LDS once (one 16-byte store, four 4-byte reads per lane).  12 MFMAs per product: 24 forward, 72 backward per head.
// No RoPE (those models keep the kernels above); head dims that are multiples of 16; Tq, Tk <= 16; H a multiple of HG.
// ------------------------------------------------------------------------------------------------
template <int HD, int HG>
struct AttnMfma {
    static constexpr int W = HG * HD, RS = W + 16, W4 = W / 4, NT = 64 * HG, KH = HD / 16;
    static constexpr int NL = (16 * W4 + NT - 1) / NT;   // float4 items of one 16-row set per thread
    // T rows of W floats (row r at src + r * ld) -> dst rows of RS floats; loads first, stores by the caller's second call
    __device__ static __forceinline__ void request(const float* src, int64_t ld, int T, int tid, f32x4 (&t)[NL]) {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int i = min(tid + NT * u, T * W4 - 1), r = i / W4, c = i - r * W4;
            t[u] = ldg4(src + (int64_t)r * ld + 4 * c);
        }
    }
    __device__ static __forceinline__ void commit(float* dst, int T, int tid, const f32x4 (&t)[NL]) {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int i = tid + NT * u, r = i / W4, c = i - r * W4;
            if (i < T * W4) *(f32x4*)(dst + r * RS + 4 * c) = t[u];
        }
    }
    // S^T-form product of two row sets: acc[e] = sum_f A[row 4 g + e][f] * B[row m][f] over the head's HD features
    __device__ static __forceinline__ f32x4 rows_dot(const float* a_rows, int ta, const float* b_rows, int tb, int hoff, int m, int g) {
        const float* ap = a_rows + min(m, ta - 1) * RS + hoff + 4 * g;
        const float* bp = b_rows + min(m, tb - 1) * RS + hoff + 4 * g;
        f32x4 af[KH], bf[KH];
#pragma unroll
        for (int kc = 0; kc < KH; ++kc) { af[kc] = *(const f32x4*)(ap + 16 * kc); bf[kc] = *(const f32x4*)(bp + 16 * kc); }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KH; ++kc)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kc][e], bf[kc][e], acc, 0, 0, 0);
        return acc;
    }
    // out[row m][16 nt + 4 g .. + 3] = sum over rows r = 4 g' + e of `rows` (T of them) of rows[r][16 nt + .] * coef[.][r]:
    // coef[e] is the lane's B operand (value for row 4 g + e against column m)
    __device__ static __forceinline__ void rows_combine(const float* rows, int T, int hoff, int m, int g, const f32x4& coef,
                                                        f32x4 (&o)[KH]) {
        float rt[KH][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* rp = rows + min(4 * g + e, T - 1) * RS + hoff + m;
#pragma unroll
            for (int nt = 0; nt < KH; ++nt) rt[nt][e] = rp[16 * nt];
        }
#pragma unroll
        for (int nt = 0; nt < KH; ++nt) {
            o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(rt[nt][e], coef[e], o[nt], 0, 0, 0);
        }
    }
};

// the lane's four probabilities P[query m][keys 4 g + e] from its four scores; rc = the query row (clamped), invisible keys -> 0
__device__ __forceinline__ f32x4 attn_softmax4(f32x4 sc, float scale, int rc, int g, int Tk, bool causal) {
    float mx = -INFINITY;
    bool vis[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        vis[e] = 4 * g + e < Tk && (!causal || 4 * g + e <= rc);
        sc[e] = vis[e] ? sc[e] * scale : -INFINITY;
        mx = fmaxf(mx, sc[e]);
    }
    mx = xrow_max(mx);   // key 0 is visible to every query: finite
    f32x4 pr;
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pr[e] = vis[e] ? expf(sc[e] - mx) : 0.f; sum += pr[e]; }
    return pr * (1.0f / xrow_sum(sum));
}

template <int HD, int HG>
__global__ __launch_bounds__(64 * HG) void k_attn_fwd_train_mfma(mdt_attn_train_args a, float scale) {
    using AM = AttnMfma<HD, HG>;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Tq = a.Tq, Tk = a.Tk, tid = threadIdx.x, lane = tid & 63, hl = tid >> 6, m = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, h = blockIdx.y * HG + hl, hoff = hl * HD;
    float* qs = sm;
    float* ks = qs + Tq * AM::RS;
    float* vs = ks + Tk * AM::RS;
    float (*mk)[17] = (float (*)[17])(vs + Tk * AM::RS + hl * 16 * 17);
    const bool drop = a.p > 0.f && a.seed != 0;
    const int64_t c0 = (int64_t)blockIdx.y * AM::W;
    f32x4 tq[AM::NL], tk[AM::NL], tv[AM::NL];
    AM::request(a.q + (int64_t)b * Tq * a.ldq + c0, a.ldq, Tq, tid, tq);
    AM::request(a.k + (int64_t)b * Tk * a.ldkv + c0, a.ldkv, Tk, tid, tk);
    AM::request(a.v + (int64_t)b * Tk * a.ldkv + c0, a.ldkv, Tk, tid, tv);
    if (drop) attn_dropout_table(mk, a.seed, a.site, ((uint64_t)b * a.H + h) * Tq * Tk, Tq, Tk, a.p, lane);  // under the loads
    AM::commit(qs, Tq, tid, tq);
    AM::commit(ks, Tk, tid, tk);
    AM::commit(vs, Tk, tid, tv);
    __syncthreads();
    const int rc = min(m, Tq - 1);
    f32x4 pr = attn_softmax4(AM::rows_dot(ks, Tk, qs, Tq, hoff, m, g), scale, rc, g, Tk, a.causal != 0);
    if (drop) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pr[e] *= mk[rc][min(4 * g + e, Tk - 1)];
    }
    f32x4 o[AM::KH];
    AM::rows_combine(vs, Tk, hoff, m, g, pr, o);
    if (m < Tq) {
        float* op = a.out + ((int64_t)b * Tq + m) * a.ldo + (int64_t)h * HD + 4 * g;
#pragma unroll
        for (int nt = 0; nt < AM::KH; ++nt) *(f32x4*)(op + 16 * nt) = o[nt];
    }
}

template <int HD, int HG>
__global__ __launch_bounds__(64 * HG) void k_attn_bwd_mfma(mdt_attn_bwd_args a, float scale) {
    using AM = AttnMfma<HD, HG>;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Tq = a.Tq, Tk = a.Tk, tid = threadIdx.x, lane = tid & 63, hl = tid >> 6, m = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, h = blockIdx.y * HG + hl, hoff = hl * HD;
    float* qs = sm;
    float* ks = qs + Tq * AM::RS;
    float* vs = ks + Tk * AM::RS;
    float* os = vs + Tk * AM::RS;                       // d_out rows
    float* wv = os + Tq * AM::RS + hl * (16 * 17 + 2 * 16 * 20);
    float (*mk)[17] = (float (*)[17])wv;                // this head's dropout multipliers
    float (*tp)[20] = (float (*)[20])(wv + 16 * 17);    // (m P) tile, [query][key]
    float (*td)[20] = tp + 16;                          // dS tile
    const bool drop = a.p > 0.f && a.seed != 0;
    const int64_t c0 = (int64_t)blockIdx.y * AM::W;
    {
        f32x4 tq[AM::NL], tk[AM::NL], tv[AM::NL], to[AM::NL];
        AM::request(a.q + (int64_t)b * Tq * a.ldq + c0, a.ldq, Tq, tid, tq);
        AM::request(a.k + (int64_t)b * Tk * a.ldkv + c0, a.ldkv, Tk, tid, tk);
        AM::request(a.v + (int64_t)b * Tk * a.ldkv + c0, a.ldkv, Tk, tid, tv);
        AM::request(a.d_out + (int64_t)b * Tq * a.ld_do + c0, a.ld_do, Tq, tid, to);
        if (drop) attn_dropout_table(mk, a.seed, a.site, ((uint64_t)b * a.H + h) * Tq * Tk, Tq, Tk, a.p, lane);
        AM::commit(qs, Tq, tid, tq);
        AM::commit(ks, Tk, tid, tk);
        AM::commit(vs, Tk, tid, tv);
        AM::commit(os, Tq, tid, to);
    }
    __syncthreads();
    const int rc = min(m, Tq - 1);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 pr = attn_softmax4(AM::rows_dot(ks, Tk, qs, Tq, hoff, m, g), scale, rc, g, Tk, a.causal != 0);
    f32x4 dp = AM::rows_dot(vs, Tk, os, Tq, hoff, m, g);   // dP[query m][keys 4 g + e] = dO[m] . V[key]
    f32x4 pm = pr, ds;
    {
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mul = drop ? mk[rc][min(4 * g + e, Tk - 1)] : 1.f;
            pm[e] = pr[e] * mul;                            // what multiplied V in the forward
            dp[e] *= mul;                                   // out = (m P) V  =>  dP = m (dO V^T)
            part = fmaf(pr[e], dp[e], part);
        }
        const float delta = xrow_sum(part);
#pragma unroll
        for (int e = 0; e < 4; ++e) ds[e] = pr[e] * (dp[e] - delta) * scale;
    }
    if (m >= Tq) { pm = zero4; ds = zero4; }               // lanes past the last query repeated its row: they contribute nothing
    // dQ[query m][.] = sum_keys dS[m][key] K[key][.]
    {
        f32x4 o[AM::KH];
        AM::rows_combine(ks, Tk, hoff, m, g, ds, o);
        if (m < Tq) {
            float* op = a.dq + ((int64_t)b * Tq + m) * a.ld_dq + (int64_t)h * HD + 4 * g;
#pragma unroll
            for (int nt = 0; nt < AM::KH; ++nt) *(f32x4*)(op + 16 * nt) = o[nt];
        }
    }
    // the two tiles the other way round: lane (m, g) needs [query 4 g + e][key m]
    *(f32x4*)&tp[m][4 * g] = pm;
    *(f32x4*)&td[m][4 * g] = ds;
    __syncthreads();
    f32x4 pmT, dsT;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pmT[e] = tp[4 * g + e][m]; dsT[e] = td[4 * g + e][m]; }
    {
        f32x4 ov[AM::KH], ok[AM::KH];
        AM::rows_combine(os, Tq, hoff, m, g, pmT, ov);     // dV[key m][.] = sum_queries (m P)[query][m] dO[query][.]
        AM::rows_combine(qs, Tq, hoff, m, g, dsT, ok);     // dK[key m][.] = sum_queries dS[query][m] Q[query][.]
        if (m < Tk) {
            float* vp = a.dv + ((int64_t)b * Tk + m) * a.ld_dkv + (int64_t)h * HD + 4 * g;
            float* kp = a.dk + ((int64_t)b * Tk + m) * a.ld_dkv + (int64_t)h * HD + 4 * g;
#pragma unroll
            for (int nt = 0; nt < AM::KH; ++nt) {
                f32x4* pv = (f32x4*)(vp + 16 * nt);
                f32x4* pk = (f32x4*)(kp + 16 * nt);
                *pv = a.accumulate_kv ? *pv + ov[nt] : ov[nt];
                *pk = a.accumulate_kv ? *pk + ok[nt] : ok[nt];
            }
        }
    }
}

// which (hd, H) the MFMA kernels take, and MDT_HIP_ATTN_TRAIN_MFMA=0 (A/B runs: the one-wave-per-head kernels everywhere)
static int attn_train_mfma_group(int hd, int H, int rope) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MDT_HIP_ATTN_TRAIN_MFMA"); on = e ? atoi(e) : 1; }
    if (!on || rope || (hd != 16 && hd != 32 && hd != 48 && hd != 64)) return 0;
    return H % 4 == 0 ? 4 : (H % 2 == 0 ? 2 : 1);
}
template <int HD, int HG>
static hipError_t launch_attn_fwd_mfma(const mdt_attn_train_args& a, float scale, hipStream_t s) {
    using AM = AttnMfma<HD, HG>;
    const size_t lds = ((size_t)(a.Tq + 2 * a.Tk) * AM::RS + (size_t)HG * 16 * 17) * sizeof(float);
    static bool attr_dev[32] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (!attr_dev[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_fwd_train_mfma<HD, HG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_dev[dev] = true;
    }
    hipLaunchKernelGGL((k_attn_fwd_train_mfma<HD, HG>), dim3(a.B, a.H / HG), dim3(64 * HG), lds, s, a, scale);
    return hipGetLastError();
}
template <int HD, int HG>
static hipError_t launch_attn_bwd_mfma(const mdt_attn_bwd_args& a, float scale, hipStream_t s) {
    using AM = AttnMfma<HD, HG>;
    const size_t lds = ((size_t)(2 * a.Tq + 2 * a.Tk) * AM::RS + (size_t)HG * (16 * 17 + 2 * 16 * 20)) * sizeof(float);
    static bool attr_dev[32] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (!attr_dev[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_bwd_mfma<HD, HG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_dev[dev] = true;
    }
    hipLaunchKernelGGL((k_attn_bwd_mfma<HD, HG>), dim3(a.B, a.H / HG), dim3(64 * HG), lds, s, a, scale);
    return hipGetLastError();
}
bool mdt_attn_train_mfma_supported(int hd, int H, int rope) { return attn_train_mfma_group(hd, H, rope) != 0; }
#define ATTN_MFMA_DISPATCH(FN, a, scale, s, hg)                                                         \
    switch ((a).hd * 8 + (hg)) {                                                                         \
        case 16 * 8 + 4: return FN<16, 4>(a, scale, s); case 16 * 8 + 2: return FN<16, 2>(a, scale, s); \
        case 16 * 8 + 1: return FN<16, 1>(a, scale, s);                                                  \
        case 32 * 8 + 4: return FN<32, 4>(a, scale, s); case 32 * 8 + 2: return FN<32, 2>(a, scale, s); \
        case 32 * 8 + 1: return FN<32, 1>(a, scale, s);                                                  \
        case 48 * 8 + 4: return FN<48, 4>(a, scale, s); case 48 * 8 + 2: return FN<48, 2>(a, scale, s); \
        case 48 * 8 + 1: return FN<48, 1>(a, scale, s);                                                  \
        case 64 * 8 + 4: return FN<64, 4>(a, scale, s); case 64 * 8 + 2: return FN<64, 2>(a, scale, s); \
        case 64 * 8 + 1: return FN<64, 1>(a, scale, s);                                                  \
        default: break;                                                                                  \
    }

hipError_t mdt_launch_attn_fwd_train(const mdt_attn_train_args& a, hipStream_t s) {
    if (a.Tq < 1 || a.Tq > 16 || a.Tk < 1 || a.Tk > 16) return hipErrorInvalidValue;
    if (((a.ldq | a.ldkv | a.ldo) & 3) || (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.out) & 15))
        return hipErrorInvalidValue;  // 16-byte loads / stores of the head slices
    if (a.rope && (a.hd < 32 || !a.rope_cos || !a.rope_sin)) return hipErrorInvalidValue;
    const float scale = 1.0f / sqrtf((float)a.hd);
    if (const int hg = attn_train_mfma_group(a.hd, a.H, a.rope)) { ATTN_MFMA_DISPATCH(launch_attn_fwd_mfma, a, scale, s, hg) }
    const dim3 grid(a.B, a.H);
    switch (a.hd) {
        case 16: hipLaunchKernelGGL((k_attn_fwd_train<16>), grid, dim3(64), 0, s, a, scale); break;
        case 32: hipLaunchKernelGGL((k_attn_fwd_train<32>), grid, dim3(64), 0, s, a, scale); break;
        case 48: hipLaunchKernelGGL((k_attn_fwd_train<48>), grid, dim3(64), 0, s, a, scale); break;
        case 64: hipLaunchKernelGGL((k_attn_fwd_train<64>), grid, dim3(64), 0, s, a, scale); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int HD>
__global__ __launch_bounds__(64) void k_attn_bwd(mdt_attn_bwd_args a, float scale) {
    __shared__ __attribute__((aligned(16))) float qs[16][HD + 4], ks[16][HD + 4], vs[16][HD + 4], os[16][HD + 4];
    __shared__ float P[16][17], dS[16][17], Mk[16][17];
    constexpr int H4 = HD / 4;
    const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int Tq = a.Tq, Tk = a.Tk;
    const bool drop = a.p > 0.f && a.seed != 0;
    if (drop) attn_dropout_table(Mk, a.seed, a.site, ((uint64_t)b * a.H + h) * Tq * Tk, Tq, Tk, a.p, lane);
    {   // q / k / v / dO tiles (rows are 16-byte aligned: HD and the leading dimensions are multiples of 4)
        const float* const src[4] = {a.q + (int64_t)b * Tq * a.ldq + h * HD, a.k + (int64_t)b * Tk * a.ldkv + h * HD,
                                     a.v + (int64_t)b * Tk * a.ldkv + h * HD, a.d_out + (int64_t)b * Tq * a.ld_do + h * HD};
        const int64_t ld[4] = {a.ldq, a.ldkv, a.ldkv, a.ld_do};
        const int T[4] = {Tq, Tk, Tk, Tq};
        float (*const dst[4])[HD + 4] = {qs, ks, vs, os};
        load_head_rows<HD, 4>(src, ld, T, dst, lane);
    }
    __syncthreads();
    if (a.rope) {  // the scores were formed on the rotated q / k
        rope_rows<HD>(qs, Tq, a.rope_cos, a.rope_sin, 1.f, lane);
        rope_rows<HD>(ks, Tk, a.rope_cos, a.rope_sin, 1.f, lane);
        __syncthreads();
    }
    // the two HD-deep dot products of every (query, key) pair, one pair per lane (Tq * Tk <= 256 pairs): scores -> P,
    // dO . V -> dS; the row-wise softmax algebra below then only touches Tk values per query
    for (int e = lane; e < Tq * Tk; e += 64) {
        const int i = e / Tk, j = e - i * Tk;
        float s = 0.f, t = 0.f;
#pragma unroll
        for (int c = 0; c < H4; ++c) {
            s = dot4(*(const f32x4*)&qs[i][4 * c], *(const f32x4*)&ks[j][4 * c], s);
            t = dot4(*(const f32x4*)&os[i][4 * c], *(const f32x4*)&vs[j][4 * c], t);
        }
        P[i][j] = (!a.causal || j <= i) ? s * scale : -INFINITY;
        dS[i][j] = t;
    }
    __syncthreads();
    if (lane < Tq) {
        const int i = lane;
        float sc[16], dp[16];
        float mx = -INFINITY;
        for (int j = 0; j < Tk; ++j) {
            sc[j] = P[i][j];
            dp[j] = dS[i][j];
            mx = fmaxf(mx, sc[j]);
        }
        float sum = 0.f;
        for (int j = 0; j < Tk; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
        const float inv = 1.f / sum;
        // with dropout: out = (m * P) V, m = mask/(1-p)  =>  dP = m * (dO V^T), dV uses m * P
        float delta = 0.f;
        for (int j = 0; j < Tk; ++j) {
            const float mk = drop ? Mk[i][j] : 1.f;
            sc[j] *= inv;
            dp[j] *= mk;
            delta = fmaf(sc[j], dp[j], delta);
            P[i][j] = sc[j] * mk;  // what multiplied V in the forward
        }
        for (int j = 0; j < Tk; ++j) dS[i][j] = sc[j] * (dp[j] - delta) * scale;
    }
    __syncthreads();
    // (row, four features) per lane from here on
    float* dqp = a.dq + (int64_t)b * Tq * a.ld_dq + h * HD;
    float* dkp = a.dk + (int64_t)b * Tk * a.ld_dkv + h * HD;
    float* dvp = a.dv + (int64_t)b * Tk * a.ld_dkv + h * HD;
    if (a.rope) {
        // gradients with respect to the ROTATED q / k: computed into the (no longer needed) dO / V tiles, rotated back
        // by the transpose, then stored.  dV first, while dO is still intact.
        for (int e = lane; e < Tk * H4; e += 64) {
            const int j = e / H4, c = 4 * (e - j * H4);
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < Tq; ++i) av += P[i][j] * *(const f32x4*)&os[i][c];
            f32x4* pv = (f32x4*)(dvp + (int64_t)j * a.ld_dkv + c);
            *pv = a.accumulate_kv ? *pv + av : av;
        }
        __syncthreads();
        for (int e = lane; e < Tq * H4; e += 64) {
            const int i = e / H4, c = 4 * (e - i * H4);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < Tk; ++j) acc += dS[i][j] * *(const f32x4*)&ks[j][c];
            *(f32x4*)&os[i][c] = acc;                              // dQ_rot
        }
        for (int e = lane; e < Tk * H4; e += 64) {
            const int j = e / H4, c = 4 * (e - j * H4);
            f32x4 ak = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < Tq; ++i) ak += dS[i][j] * *(const f32x4*)&qs[i][c];
            *(f32x4*)&vs[j][c] = ak;                               // dK_rot
        }
        __syncthreads();
        rope_rows<HD>(os, Tq, a.rope_cos, a.rope_sin, -1.f, lane);
        rope_rows<HD>(vs, Tk, a.rope_cos, a.rope_sin, -1.f, lane);
        __syncthreads();
        for (int e = lane; e < Tq * H4; e += 64) {
            const int i = e / H4, c = 4 * (e - i * H4);
            *(f32x4*)(dqp + (int64_t)i * a.ld_dq + c) = *(const f32x4*)&os[i][c];
        }
        for (int e = lane; e < Tk * H4; e += 64) {
            const int j = e / H4, c = 4 * (e - j * H4);
            f32x4* pk = (f32x4*)(dkp + (int64_t)j * a.ld_dkv + c);
            const f32x4 v = *(const f32x4*)&vs[j][c];
            *pk = a.accumulate_kv ? *pk + v : v;
        }
        return;
    }
    // dQ[i][:] = sum_j dS[i][j] K[j][:]
    for (int e = lane; e < Tq * H4; e += 64) {
        const int i = e / H4, c = 4 * (e - i * H4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Tk; ++j) acc += dS[i][j] * *(const f32x4*)&ks[j][c];
        *(f32x4*)(dqp + (int64_t)i * a.ld_dq + c) = acc;
    }
    // dK[j][:] = sum_i dS[i][j] Q[i][:] ;  dV[j][:] = sum_i P[i][j] dO[i][:]
    for (int e = lane; e < Tk * H4; e += 64) {
        const int j = e / H4, c = 4 * (e - j * H4);
        f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < Tq; ++i) {
            ak += dS[i][j] * *(const f32x4*)&qs[i][c];
            av += P[i][j] * *(const f32x4*)&os[i][c];
        }
        f32x4* pk = (f32x4*)(dkp + (int64_t)j * a.ld_dkv + c);
        f32x4* pv = (f32x4*)(dvp + (int64_t)j * a.ld_dkv + c);
        *pk = a.accumulate_kv ? *pk + ak : ak;
        *pv = a.accumulate_kv ? *pv + av : av;
    }
}

hipError_t mdt_launch_attn_bwd(const mdt_attn_bwd_args& a, hipStream_t s) {
    if (a.Tq < 1 || a.Tq > 16 || a.Tk < 1 || a.Tk > 16) return hipErrorInvalidValue;
    if (((a.ldq | a.ld_do | a.ldkv | a.ld_dq | a.ld_dkv) & 3) ||
        (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.d_out | (uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv) & 15))
        return hipErrorInvalidValue;  // 16-byte loads / stores of the head slices
    if (a.rope && (a.hd < 32 || !a.rope_cos || !a.rope_sin)) return hipErrorInvalidValue;
    const float scale = 1.0f / sqrtf((float)a.hd);
    if (const int hg = attn_train_mfma_group(a.hd, a.H, a.rope)) { ATTN_MFMA_DISPATCH(launch_attn_bwd_mfma, a, scale, s, hg) }
    const dim3 grid(a.B, a.H);
    switch (a.hd) {
        case 16: hipLaunchKernelGGL((k_attn_bwd<16>), grid, dim3(64), 0, s, a, scale); break;
        case 32: hipLaunchKernelGGL((k_attn_bwd<32>), grid, dim3(64), 0, s, a, scale); break;
        case 48: hipLaunchKernelGGL((k_attn_bwd<48>), grid, dim3(64), 0, s, a, scale); break;
        case 64: hipLaunchKernelGGL((k_attn_bwd<64>), grid, dim3(64), 0, s, a, scale); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// loss gradient and the narrow (A <= 16 wide) linears around the action tokens
// ------------------------------------------------------------------------------------------------
// dF = gscale * 2 (F - target) / n ,  target = (action - c_skip * noised) / c_out     (score_wrappers.py:59-63)
__global__ void k_loss_grad(const float* __restrict__ F, const float* __restrict__ act, const float* __restrict__ noised,
                            const float* __restrict__ sigma, float sd, int64_t n, int per_sample,
                            const float* __restrict__ gscale, float* __restrict__ dF) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sg = sigma[i / per_sample];
    const float den2 = sg * sg + sd * sd;
    const float c_skip = sd * sd / den2, c_out = sg * sd / sqrtf(den2);
    const float tgt = (act[i] - c_skip * noised[i]) / c_out;
    dF[i] = (gscale ? *gscale : 1.f) * 2.f * (F[i] - tgt) / (float)n;
}
hipError_t mdt_launch_loss_grad(const float* F, const float* act, const float* noised, const float* sigma, float sd,
                                int64_t n, int per_sample, const float* gscale, float* dF, hipStream_t s) {
    hipLaunchKernelGGL(k_loss_grad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, F, act, noised, sigma, sd, n,
                       per_sample, gscale, dF);
    return hipGetLastError();
}

// mdt_denoise_vjp: D = c_skip x + c_out F (score_wrappers.py:65-80) and the seed of its backward, dF = c_out v
__global__ void k_vjp_seed(const float* __restrict__ F, const float* __restrict__ x, const float* __restrict__ sigma,
                           const float* __restrict__ v, float sd, int64_t n, int per_sample, float* __restrict__ den,
                           float* __restrict__ dF) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sg = sigma[i / per_sample];
    const float den2 = sg * sg + sd * sd;
    const float c_skip = sd * sd / den2, c_out = sg * sd / sqrtf(den2);
    den[i] = fmaf(F[i], c_out, x[i] * c_skip);
    dF[i] = c_out * v[i];
}
hipError_t mdt_launch_vjp_seed(const float* F, const float* x, const float* sigma, const float* v, float sd, int64_t n,
                               int per_sample, float* den, float* dF, hipStream_t s) {
    hipLaunchKernelGGL(k_vjp_seed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, F, x, sigma, v, sd, n, per_sample, den, dF);
    return hipGetLastError();
}
// ... and its end: the network saw c_in x, the skip connection c_skip x
__global__ void k_vjp_finish(const float* __restrict__ dxin, const float* __restrict__ sigma, const float* __restrict__ v,
                             float sd, int64_t n, int per_sample, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sg = sigma[i / per_sample];
    const float den2 = sg * sg + sd * sd;
    out[i] = fmaf(dxin[i], 1.0f / sqrtf(den2), v[i] * (sd * sd / den2));
}
hipError_t mdt_launch_vjp_finish(const float* dxin, const float* sigma, const float* v, float sd, int64_t n, int per_sample,
                                 float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_vjp_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dxin, sigma, v, sd, n, per_sample, out);
    return hipGetLastError();
}

// out[m][d] = sum_a G[m][a] * W[a][d]            (d(ln_out) = dF @ action_pred.weight, W is (A, D))
__global__ void k_narrow_dx(const float* __restrict__ G, const float* __restrict__ W, float* __restrict__ out, int64_t n,
                            int A, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t m = i / D;
    const int d = (int)(i - m * D);
    float acc = 0.f;
    for (int a = 0; a < A; ++a) acc = fmaf(G[m * A + a], W[a * D + d], acc);
    out[i] = acc;
}
hipError_t mdt_launch_narrow_dx(const float* G, const float* W, float* out, int M, int A, int D, hipStream_t s) {
    const int64_t n = (int64_t)M * D;
    hipLaunchKernelGGL(k_narrow_dx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, G, W, out, n, A, D);
    return hipGetLastError();
}

// out[m][a] = sum_d G[m][d] * WT[a][d]   (A <= 16): one wavefront per row m, lanes stride over d
__global__ __launch_bounds__(256) void k_narrow_out(const float* __restrict__ G, int64_t ldg, const float* __restrict__ WT,
                                                    float* __restrict__ out, int M, int A, int D) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float acc[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) acc[a] = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float g = G[(int64_t)m * ldg + d];
#pragma unroll
        for (int a = 0; a < 16; ++a)
            if (a < A) acc[a] = fmaf(g, WT[(int64_t)a * D + d], acc[a]);
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        if (a < A) {
            const float v = wave_sum(acc[a]);
            if (lane == 0) out[(int64_t)m * A + a] = v;
        }
    }
}
hipError_t mdt_launch_narrow_out(const float* G, int64_t ldg, const float* WT, float* out, int M, int A, int D, hipStream_t s) {
    if (A < 1 || A > 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_narrow_out, dim3((M + 3) / 4), dim3(256), 0, s, G, ldg, WT, out, M, A, D);
    return hipGetLastError();
}

// partial[y][a][d] = sum_{m in slice y} G[m][a] * Y[m][d]   (A <= 16); summed over y by k_colsum.
//   transposed = 0: rows of the result are a (action_pred.weight (A, D));  1: result stored (D, A) (action_emb.weight)
__global__ __launch_bounds__(256) void k_narrow_dw(const float* __restrict__ G, const float* __restrict__ Y, int64_t ldy,
                                                   float* __restrict__ partial, int M, int A, int D, int transposed) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int ny = gridDim.y, y = blockIdx.y;
    const int m0 = (int)((int64_t)M * y / ny), m1 = (int)((int64_t)M * (y + 1) / ny);
    float acc[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) acc[a] = 0.f;
    if (d < D) {
        int m = m0;
        for (; m + 4 <= m1; m += 4) {  // four rows' loads in flight per trip
            float yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) yv[u] = Y[(int64_t)(m + u) * ldy + d];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < 16; ++a)
                    if (a < A) acc[a] = fmaf(G[(int64_t)(m + u) * A + a], yv[u], acc[a]);
        }
        for (; m < m1; ++m) {
            const float yv = Y[(int64_t)m * ldy + d];
#pragma unroll
            for (int a = 0; a < 16; ++a)
                if (a < A) acc[a] = fmaf(G[(int64_t)m * A + a], yv, acc[a]);
        }
        float* p = partial + (int64_t)y * A * D;
#pragma unroll
        for (int a = 0; a < 16; ++a)
            if (a < A) p[transposed ? d * A + a : a * D + d] = acc[a];
    }
}
hipError_t mdt_launch_narrow_dw(const float* G, const float* Y, int64_t ldy, float* partial, int n_slices, int M, int A,
                                int D, int transposed, hipStream_t s) {
    if (A < 1 || A > 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_narrow_dw, dim3((D + 255) / 256, n_slices), dim3(256), 0, s, G, Y, ldy, partial, M, A, D,
                       transposed);
    return hipGetLastError();
}

// d_y[m][d] = c_in[sample] ... is not needed (no gradient flows to the noisy actions); the action embedding only
// needs its weight / bias gradient: xin[m][a] = noised[m][a] * c_in(sigma[sample]) recomputed here.
__global__ void k_scaled_input(const float* __restrict__ x, const float* __restrict__ sigma, float sd, int64_t n,
                               int per_sample, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sg = sigma[i / per_sample];
    out[i] = x[i] / sqrtf(sg * sg + sd * sd);
}
hipError_t mdt_launch_scaled_input(const float* x, const float* sigma, float sd, int64_t n, int per_sample, float* out,
                                   hipStream_t s) {
    hipLaunchKernelGGL(k_scaled_input, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, sigma, sd, n, per_sample, out);
    return hipGetLastError();
}

// gather / scatter-add of row groups: dst[(m / gin) * gin + m % gin] = src[(m / gin) * gout + m % gin + goff]
// (the inverse of the GEMM epilogue's output row remap: picks the goal / state rows out of the context gradient)
__global__ void k_gather_rows(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int D, int gin, int gout,
                              int goff) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t m = i / D;
    const int c = (int)(i - m * D);
    dst[i] = src[((m / gin) * gout + m % gin + goff) * D + c];
}
hipError_t mdt_launch_gather_rows(const float* src, float* dst, int M, int D, int gin, int gout, int goff, hipStream_t s) {
    const int64_t n = (int64_t)M * D;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n, D, gin, gout, goff);
    return hipGetLastError();
}

// nn.Dropout on embedded tokens, in place (self.drop: mdtv_transformer.py:104,227; mdt_transformer.py:220-227,234):
// rows whose position inside their sample (row % rows_per_sample) is below row_lo are left alone (the sigma token).
// The backward applies the same call to the gradient.
__global__ void k_dropout_rows(float* __restrict__ x, int64_t n, int D, int rows_per_sample, int row_lo, float p,
                               uint32_t site, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if ((int)((i / D) % rows_per_sample) < row_lo) return;
    x[i] *= dropout_scale(seed, site, (uint64_t)i, p);
}
hipError_t mdt_launch_dropout_rows(float* x, int64_t rows, int D, int rows_per_sample, int row_lo, float p, uint32_t site,
                                   uint64_t seed, hipStream_t s) {
    if (p <= 0.f || seed == 0) return hipSuccess;
    const int64_t n = rows * D;
    hipLaunchKernelGGL(k_dropout_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, D, rows_per_sample, row_lo, p,
                       site, seed);
    return hipGetLastError();
}

// y (+)= x elementwise
__global__ void k_axpy1(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}
__global__ void k_add_2d(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd, int rows, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i - r * cols;
    dst[(int64_t)r * ldd + c] += src[(int64_t)r * lds_ + c];
}
hipError_t mdt_launch_add_2d(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(k_add_2d, dim3((rows * cols + 255) / 256), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols);
    return hipGetLastError();
}
hipError_t mdt_launch_add_inplace(const float* x, float* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_axpy1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Perceiver resampler: backward of k_attn_long (few queries, ~400 keys), one workgroup per (sample, head).
//   s[q][f] = (scale q[q]) . k[f] ;  P = softmax_f(s) ;  o[q] = sum_f P[q][f] v[f]
//   dP[q][f] = dO[q] . v[f] ;  dS = P * (dP - sum_f P dP)
//   dq[q] = scale * sum_f dS[q][f] k[f] ;  dk[f] = sum_q dS[q][f] (scale q[q]) ;  dv[f] = sum_q P[q][f] dO[q]
// Pass 1 (thread per key): scores -> P in LDS.  Pass 2 (thread per key): dP -> LDS, then dS in place.
// Pass 3 (thread per key): dk / dv rows.  Pass 4 (thread = key group x feature): dq through an LDS reduction.
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void k_attn_long_bwd(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                       const float* __restrict__ v, int64_t ldkv,
                                                       const float* __restrict__ d_out, int64_t ld_do,
                                                       float* __restrict__ dq, int64_t ld_dq, float* __restrict__ dk,
                                                       float* __restrict__ dv, int64_t ld_dkv, int Tq, int Tk, float scale,
                                                       float* __restrict__ dkl, int F) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int H4 = HD / 4, G = 256 / HD, QMAX = 16;
    const int tid = threadIdx.x, b = blockIdx.x, h = blockIdx.y;
    const int Tkp = (Tk + 3) & ~3;
    float* qs = lds;                  // [Tq][HD] scaled queries
    float* os = qs + Tq * HD;         // [Tq][HD] dO
    float* P = os + Tq * HD;          // [Tq][Tkp]
    float* dS = P + Tq * Tkp;         // [Tq][Tkp]  dP, then dS
    float* aux = dS + Tq * Tkp;       // [16] delta per query
    float* red = aux + QMAX;          // [G][Tq][HD]
    for (int i = tid; i < Tq * HD; i += 256) {
        const int qi = i / HD, d = i - qi * HD;
        qs[i] = q[((int64_t)b * Tq + qi) * ldq + h * HD + d] * scale;
        os[i] = d_out[((int64_t)b * Tq + qi) * ld_do + h * HD + d];
    }
    __syncthreads();
    const float* kb = k + (int64_t)b * Tk * ldkv + h * HD;
    const float* vb = v + (int64_t)b * Tk * ldkv + h * HD;
    // pass 1 + 2a: scores and dP per key
    for (int f0 = 0; f0 < Tk; f0 += 256) {
        const int f = min(f0 + tid, Tk - 1);
        f32x4 kr[H4], vr[H4];
#pragma unroll
        for (int c = 0; c < H4; ++c) { kr[c] = ldg4(kb + (int64_t)f * ldkv + c * 4); vr[c] = ldg4(vb + (int64_t)f * ldkv + c * 4); }
        for (int qi = 0; qi < Tq; ++qi) {
            const f32x4* qv = reinterpret_cast<const f32x4*>(qs + qi * HD);
            const f32x4* ov = reinterpret_cast<const f32x4*>(os + qi * HD);
            float s = 0.f, t = 0.f;
#pragma unroll
            for (int c = 0; c < H4; ++c) {
                const f32x4 a = qv[c], o = ov[c];
                s = fmaf(a.x, kr[c].x, s); s = fmaf(a.y, kr[c].y, s); s = fmaf(a.z, kr[c].z, s); s = fmaf(a.w, kr[c].w, s);
                t = fmaf(o.x, vr[c].x, t); t = fmaf(o.y, vr[c].y, t); t = fmaf(o.z, vr[c].z, t); t = fmaf(o.w, vr[c].w, t);
            }
            if (f0 + tid < Tk) { P[qi * Tkp + f] = s; dS[qi * Tkp + f] = t; }
        }
    }
    __syncthreads();
    // softmax per query, delta = sum_f P dP, then dS = P (dP - delta)
    const int w = tid >> 6, lane = tid & 63;
    for (int qi = w; qi < Tq; qi += 4) {
        float* pr = P + qi * Tkp;
        float* dr = dS + qi * Tkp;
        float mx = -INFINITY;
        for (int f = lane; f < Tk; f += 64) mx = fmaxf(mx, pr[f]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int f = lane; f < Tk; f += 64) { const float e = expf(pr[f] - mx); pr[f] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        float dl = 0.f;
        for (int f = lane; f < Tk; f += 64) { const float p = pr[f] * inv; pr[f] = p; dl = fmaf(p, dr[f], dl); }
        dl = wave_sum(dl);
        for (int f = lane; f < Tk; f += 64) dr[f] = pr[f] * (dr[f] - dl);
    }
    __syncthreads();
    // pass 3: dk[f] = sum_q dS[q][f] qs[q] ; dv[f] = sum_q P[q][f] dO[q]   (thread per key, rows written as float4)
    for (int f = tid; f < Tk; f += 256) {
        f32x4 ak[H4], av[H4];
#pragma unroll
        for (int c = 0; c < H4; ++c) { ak[c] = f32x4{0.f, 0.f, 0.f, 0.f}; av[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int qi = 0; qi < Tq; ++qi) {
            const float ds = dS[qi * Tkp + f], p = P[qi * Tkp + f];
            const f32x4* qv = reinterpret_cast<const f32x4*>(qs + qi * HD);
            const f32x4* ov = reinterpret_cast<const f32x4*>(os + qi * HD);
#pragma unroll
            for (int c = 0; c < H4; ++c) {
                const f32x4 a = qv[c], o = ov[c];
                ak[c].x = fmaf(ds, a.x, ak[c].x); ak[c].y = fmaf(ds, a.y, ak[c].y);
                ak[c].z = fmaf(ds, a.z, ak[c].z); ak[c].w = fmaf(ds, a.w, ak[c].w);
                av[c].x = fmaf(p, o.x, av[c].x); av[c].y = fmaf(p, o.y, av[c].y);
                av[c].z = fmaf(p, o.z, av[c].z); av[c].w = fmaf(p, o.w, av[c].w);
            }
        }
        // F > 0: the first F keys of a sample (media tokens) and the rest (latents) land in two dense buffers, (B*F, ld)
        // at dk / dv and (B*(Tk-F), ld) at dkl (V at the same offset behind K) -- what the two K|V GEMMs' backwards read
        float* pk = F > 0 ? (f < F ? dk + ((int64_t)b * F + f) * ld_dkv : dkl + ((int64_t)b * (Tk - F) + (f - F)) * ld_dkv) + h * HD
                          : dk + ((int64_t)b * Tk + f) * ld_dkv + h * HD;
        float* pv = pk + (dv - dk);
#pragma unroll
        for (int c = 0; c < H4; ++c) { *(f32x4*)(pk + c * 4) = ak[c]; *(f32x4*)(pv + c * 4) = av[c]; }
    }
    // pass 4: dq[q][d] = scale * sum_f dS[q][f] k[f][d]
    const int g = tid / HD, d = tid - g * HD;
    float acc[QMAX];
#pragma unroll
    for (int qi = 0; qi < QMAX; ++qi) acc[qi] = 0.f;
    for (int f = g; f < Tk; f += G) {
        const float kv_ = kb[(int64_t)f * ldkv + d];
#pragma unroll
        for (int qi = 0; qi < QMAX; ++qi)
            if (qi < Tq) acc[qi] = fmaf(dS[qi * Tkp + f], kv_, acc[qi]);
    }
#pragma unroll
    for (int qi = 0; qi < QMAX; ++qi)
        if (qi < Tq) red[(g * Tq + qi) * HD + d] = acc[qi];
    __syncthreads();
    for (int i = tid; i < Tq * HD; i += 256) {
        const int qi = i / HD, dd = i - qi * HD;
        float t = 0.f;
        for (int gg = 0; gg < G; ++gg) t += red[(gg * Tq + qi) * HD + dd];
        dq[((int64_t)b * Tq + qi) * ld_dq + h * HD + dd] = t * scale;
    }
}

static size_t attn_long_bwd_lds(int hd, int Tq, int Tk) {
    return (size_t)(2 * Tq * hd + 2 * Tq * ((Tk + 3) & ~3) + 16 + 256 * Tq) * sizeof(float);
}

bool mdt_attention_long_bwd_supported(int hd, int Tq, int Tk) {
    return (hd == 16 || hd == 32 || hd == 64) && Tq >= 1 && Tq <= 16 && Tk >= 1 && Tk <= 4096 &&
           attn_long_bwd_lds(hd, Tq, Tk) <= 160 * 1024;
}

template <int HD>
static hipError_t launch_attn_long_bwd_t(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                         const float* d_out, int64_t ld_do, float* dq, int64_t ld_dq, float* dk, float* dv,
                                         int64_t ld_dkv, int B, int H, int Tq, int Tk, float scale, hipStream_t s, float* dkl,
                                         int F) {
    const size_t lds = attn_long_bwd_lds(HD, Tq, Tk);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_long_bwd<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_attn_long_bwd<HD>), dim3(B, H), dim3(256), lds, s, q, ldq, k, v, ldkv, d_out, ld_do, dq, ld_dq, dk,
                       dv, ld_dkv, Tq, Tk, scale, dkl, F);
    return hipGetLastError();
}

hipError_t mdt_launch_attention_long_bwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                         const float* d_out, int64_t ld_do, float* dq, int64_t ld_dq, float* dk, float* dv,
                                         int64_t ld_dkv, int B, int H, int hd, int Tq, int Tk, float scale, hipStream_t s,
                                         float* dkl, int F) {
    if (F < 0 || F >= Tk || (F > 0 && !dkl)) return hipErrorInvalidValue;
    if (!mdt_attention_long_bwd_supported(hd, Tq, Tk)) return hipErrorInvalidValue;
    switch (hd) {
        case 16: return launch_attn_long_bwd_t<16>(q, ldq, k, v, ldkv, d_out, ld_do, dq, ld_dq, dk, dv, ld_dkv, B, H, Tq, Tk, scale, s, dkl, F);
        case 32: return launch_attn_long_bwd_t<32>(q, ldq, k, v, ldkv, d_out, ld_do, dq, ld_dq, dk, dv, ld_dkv, B, H, Tq, Tk, scale, s, dkl, F);
        default: return launch_attn_long_bwd_t<64>(q, ldq, k, v, ldkv, d_out, ld_do, dq, ld_dq, dk, dv, ld_dkv, B, H, Tq, Tk, scale, s, dkl, F);
    }
}

// time_pos_emb gradient: out[t][d] = sum_{b, n} mask[b][t] * dxf[b][t][n][d], in two deterministic stages:
//   partial[b*T + t][d] = mask * sum_n dxf[b][t][n][d]   (one workgroup per frame and 64 columns, 4 row groups)
//   out[t][d] (+)= sum_b partial[b*T + t][d]              (k_colsum over the B rows of frame t, stride T*D)
__global__ __launch_bounds__(256) void k_frame_sums(const float* __restrict__ dxf, const uint8_t* __restrict__ mask,
                                                    float* __restrict__ partial, int n, int D) {
    __shared__ float part[4][64];
    const int64_t frame = blockIdx.y;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < D && (!mask || mask[frame])) {
        const float* base = dxf + frame * (int64_t)n * D + c;
        for (int i = rg; i < n; i += 4) acc += base[(int64_t)i * D];
    }
    part[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < D) partial[frame * D + c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}
hipError_t mdt_launch_time_emb_grad(const float* dxf, const uint8_t* mask, float* out, float* partial, int64_t B, int T, int n,
                                    int D, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(k_frame_sums, dim3((D + 63) / 64, (unsigned)(B * T)), dim3(256), 0, s, dxf, mask, partial, n, D);
    hipError_t e = hipGetLastError();
    for (int t = 0; t < T && e == hipSuccess; ++t)
        e = mdt_launch_colsum(partial + (int64_t)t * D, (int64_t)T * D, (int)B, D, out + (int64_t)t * D, accumulate, s);
    return e;
}

// ------------------------------------------------------------------------------------------------
// multi-tensor optimizer kernels: ONE launch updates a whole list of tensors (torch.optim.AdamW semantics;
// the EMA callback's amp_C.multi_tensor_axpby, mdt/callbacks/ema.py:108-115, which exists only on NVIDIA/apex).
// blocks[i] = (tensor index, first element); a workgroup walks OPT_CHUNK elements of one tensor.
// ------------------------------------------------------------------------------------------------
#define OPT_CHUNK 4096

__global__ __launch_bounds__(256) void k_multi_adamw(const mdt_opt_tensor* __restrict__ tab, const int2* __restrict__ blocks,
                                                     float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                     float bc2_sqrt) {
    const int2 blk = blocks[blockIdx.x];
    const mdt_opt_tensor t = tab[blk.x];
    const int64_t end = min((int64_t)blk.y + OPT_CHUNK, t.numel);
    const float step_size = lr / bc1;
    // whole chunks of 16-byte aligned tensors: 4 float4 per thread and array, every load issued before the first store
    if (end - blk.y == OPT_CHUNK && ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0)) {
        f32x4 *p4 = (f32x4*)(t.p + blk.y), *m4 = (f32x4*)(t.m + blk.y), *v4 = (f32x4*)(t.v + blk.y);
        const f32x4* g4 = (const f32x4*)(t.g + blk.y);
        f32x4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k;
            pv[k] = p4[i]; gv[k] = g4[i]; mv[k] = m4[i]; vv[k] = v4[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = gv[k][e];
                float p = pv[k][e] * (1.0f - lr * wd);
                const float m = fmaf(beta1, mv[k][e], (1.0f - beta1) * g);
                const float v = fmaf(beta2, vv[k][e], (1.0f - beta2) * g * g);
                p -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
                pv[k][e] = p; mv[k][e] = m; vv[k][e] = v;
            }
            p4[i] = pv[k]; m4[i] = mv[k]; v4[i] = vv[k];
        }
        return;
    }
    for (int64_t i = (int64_t)blk.y + threadIdx.x; i < end; i += 256) {
        const float g = t.g[i];
        float p = t.p[i] * (1.0f - lr * wd);                 // decoupled weight decay
        const float m = fmaf(beta1, t.m[i], (1.0f - beta1) * g);
        const float v = fmaf(beta2, t.v[i], (1.0f - beta2) * g * g);
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p -= step_size * (m / denom);
        t.m[i] = m; t.v[i] = v; t.p[i] = p;
    }
}

// ema = a * ema + b * w     (a = decay, b = 1 - decay)
__global__ __launch_bounds__(256) void k_multi_axpby(const mdt_opt_tensor* __restrict__ tab, const int2* __restrict__ blocks,
                                                     float a, float b) {
    const int2 blk = blocks[blockIdx.x];
    const mdt_opt_tensor t = tab[blk.x];
    const int64_t end = min((int64_t)blk.y + OPT_CHUNK, t.numel);
    if (end - blk.y == OPT_CHUNK && ((((uintptr_t)t.p | (uintptr_t)t.ema) & 15) == 0)) {
        f32x4* e4 = (f32x4*)(t.ema + blk.y);
        const f32x4* p4 = (const f32x4*)(t.p + blk.y);
        f32x4 ev[4], pv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { ev[k] = e4[threadIdx.x + 256 * k]; pv[k] = p4[threadIdx.x + 256 * k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ev[k][e] = fmaf(a, ev[k][e], b * pv[k][e]);
            e4[threadIdx.x + 256 * k] = ev[k];
        }
        return;
    }
    for (int64_t i = (int64_t)blk.y + threadIdx.x; i < end; i += 256) t.ema[i] = fmaf(a, t.ema[i], b * t.p[i]);
}

hipError_t mdt_launch_multi_adamw(const mdt_opt_tensor* tab, const int2* blocks, int n_blocks, float lr, float beta1,
                                  float beta2, float eps, float wd, float bc1, float bc2_sqrt, hipStream_t s) {
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_multi_adamw, dim3(n_blocks), dim3(256), 0, s, tab, blocks, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
    return hipGetLastError();
}
hipError_t mdt_launch_multi_axpby(const mdt_opt_tensor* tab, const int2* blocks, int n_blocks, float a, float b,
                                  hipStream_t s) {
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_multi_axpby, dim3(n_blocks), dim3(256), 0, s, tab, blocks, a, b);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_gemm_tn: the weight gradient  dW[n][k] = sum_m dY[m][n] * X[m][k]  straight from the two row-major operands --
// no transposed copy of dY, no packed copy of X^T (those were 0.64 ms of a 12.7 ms B = 1024 step and 6 ms of the
// 48 ms masked-image head).  The reduction index m is the SLOW index of both operands, so both go through LDS in their
// natural layout ([m][columns], 128-/512-byte coalesced row pieces) and the MFMA fragments are read with 4-byte LDS
// reads: for v_mfma_f32_16x16x4_f32 lane l feeds A[i = l%16][kk = l/16] and B[kk = l/16][j = l%16]; with
// A = X (i -> k column), B = dY (j -> n column), kk -> m, the lanes of a 16-lane group read 16 consecutive floats of one
// LDS row and the 4 groups read rows 4h + e: conflict free with a row stride = 4 (mod 8) floats.  D[i][j] leaves every lane
// with 4 consecutive k of one n: 16-byte stores.
//   workgroup = 4 waves, tile = 64 (n) x 128 (k); wave w owns k-tiles 2w, 2w+1 and all 4 n-tiles: 8 accumulators, 32 MFMAs per
//   16 rows of m; chunks of 32 rows double-buffered through registers (next chunk's 6 x 16-byte loads in flight during the
//   MFMAs); 51 KB of LDS -> 3 workgroups per CU.
//   grid.z = row slices (split of the reduction): slice z covers rows [z L, min(M, z L + L)) and writes its own (N, K) partial
//   (summed afterwards in a fixed order by k_colsum: deterministic).  Workgroups of k-tile 0 also leave the column sums of
//   their dY tile (the bias gradient's per-slice partials) in bpart[z][n].
// ------------------------------------------------------------------------------------------------
// KT = 16-wide k-tiles per wave: 2 (tile 64 x 128) or 3 (64 x 192, for K = 192 / 576 / ... where 128-wide tiles would leave
// a quarter of the last one empty: the masked-image head's d = 192 layers).
// WN = 64-column halves of the n-tile (1: 4 waves, 64 x TK; 2: 8 waves, 128 x TK -- wave = (n half, k quarter)).  The wide tile
// reads every X row half as often: a 64 x 192 tile at full MFMA rate wants 10.4 bytes per clock and CU from L2 / Infinity Cache
// (32 KB per 32-row chunk and 96 MFMAs per wave), about what a CU gets; the 128-wide one 6.5.
template <int KT, int WN>
__global__ __launch_bounds__(256 * WN) void k_gemm_tn(const float* __restrict__ dY, int64_t ldy, const float* __restrict__ X, int64_t ldx,
                                                 float* __restrict__ out, int64_t slice_stride, int M, int N, int K, int L,
                                                 int accumulate, float* __restrict__ bpart, int xcd) {
    constexpr int CM = 32, TN_ = 64 * WN, TK = 64 * KT, SY = TN_ + 4, SX = TK + 4, NTHR = 256 * WN, X4 = TK / 4, Y4 = TN_ / 4,
                  XU = CM * X4 / NTHR, YU = CM * Y4 / NTHR;
    static_assert(CM * X4 % NTHR == 0 && CM * Y4 % NTHR == 0, "staging does not split evenly");
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];  // ys[2][CM * SY] | xs[2][CM * SX]  (51 / 68 KB)
    float (*ys)[CM * SY] = (float (*)[CM * SY])tn_lds;
    float (*xs)[CM * SX] = (float (*)[CM * SX])(tn_lds + 2 * CM * SY);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wk = wave & 3, wn = wave >> 2;
    const int gk = (K + TK - 1) / TK;
    // Which (slice, tile) this workgroup takes.  The dispatcher places workgroup b on XCD b % 8 (speed-only assumption).  As
    // launched -- tile fastest -- XCD x got k-tile x of EVERY slice: each of the 8 L2s pulled all of dY and an eighth of X's
    // columns over all rows (k_gemm_tn<3, 3> at 10240 x 384 x 1536: 113 MB from the fabric for 79 MB of operands, L2 hit 0.30).
    // Round 6: every XCD gets a contiguous range of (slice-major) ids, i.e. whole slices -- all tiles that share a slice's rows
    // of X and dY run behind one L2 (xcd = 0: the launch order, A/B runs).
    int id = blockIdx.x + gridDim.x * blockIdx.z;
    if (xcd) {
        const int nb = gridDim.x * gridDim.z, q = nb >> 3, r = nb & 7, x8 = id & 7, idx = id >> 3;
        id = (x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q) + idx;
    }
    const int z = id / (int)gridDim.x, tile = id - z * (int)gridDim.x;
    const int bn = tile / gk, bk = tile - bn * gk;
    const int n0 = bn * TN_, k0 = bk * TK;
    const int m_lo = z * L, m_hi = min(M, m_lo + L);
    out += (int64_t)z * slice_stride;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging assignment: X chunk = 32 rows x TK/4 float4 (4 or 6 per thread), dY chunk = 32 rows x 16 float4 (2 per thread)
    int xr[XU], xc[XU], yr[YU], yc[YU];
#pragma unroll
    for (int u = 0; u < XU; ++u) { const int i = tid + NTHR * u; xr[u] = i / X4; xc[u] = (i - xr[u] * X4) * 4; }
#pragma unroll
    for (int u = 0; u < YU; ++u) { const int i = tid + NTHR * u; yr[u] = i / Y4; yc[u] = (i - yr[u] * Y4) * 4; }
    f32x4 xv[XU], yv[YU];
    // buffer loads with 32-bit byte offsets where the operands allow it (both blocks below 4 GiB: always, for the shapes of
    // this model): a global load's 64-bit addresses cost the SIMD as much matrix-pipe time as its data (mdt_tiles.h: WStream)
    const bool small = (int64_t)M * ldx < ((int64_t)1 << 30) && (int64_t)M * ldy < ((int64_t)1 << 30);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, 0xffffffffu, 0x00020000);
    auto fetch = [&](int mb) {
        if (small) {
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const unsigned m = (unsigned)min(mb + xr[u], m_hi - 1);
                xv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (m * (unsigned)ldx + (unsigned)min(k0 + xc[u], K - 4)) << 2, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < YU; ++u) {
                const unsigned m = (unsigned)min(mb + yr[u], m_hi - 1);
                yv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, (m * (unsigned)ldy + (unsigned)min(n0 + yc[u], N - 4)) << 2, 0, 0));
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int64_t m = min(mb + xr[u], m_hi - 1);
            xv[u] = ldg4(X + m * ldx + min(k0 + xc[u], K - 4));
        }
#pragma unroll
        for (int u = 0; u < YU; ++u) {
            const int64_t m = min(mb + yr[u], m_hi - 1);
            yv[u] = ldg4(dY + m * ldy + min(n0 + yc[u], N - 4));
        }
    };
    auto stash = [&](int buf, int mb) {  // rows past the slice and columns past the matrix contribute zeros
#pragma unroll
        for (int u = 0; u < XU; ++u)
            *(f32x4*)(&xs[buf][xr[u] * SX + xc[u]]) = (mb + xr[u] < m_hi && k0 + xc[u] < K) ? xv[u] : zero4;
#pragma unroll
        for (int u = 0; u < YU; ++u)
            *(f32x4*)(&ys[buf][yr[u] * SY + yc[u]]) = (mb + yr[u] < m_hi && n0 + yc[u] < N) ? yv[u] : zero4;
    };
    f32x4 acc[4][KT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = zero4;
    float bsum = 0.f;  // threads 0..63 of k-tile-0 workgroups: column sum of dY[:, n0 + tid]
    const bool do_bias = bpart != nullptr && bk == 0 && tid < TN_;
    const int l16 = lane & 15, h = lane >> 4;
    const int nchunks = (m_hi - m_lo + CM - 1) / CM;
    if (nchunks > 0) {
        fetch(m_lo);
        stash(0, m_lo);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) fetch(m_lo + (c + 1) * CM);
        const float* xb = xs[buf];
        const float* yb = ys[buf];
        // 8 steps of 4 KT MFMAs per chunk; step t + 1's operands (KT + 4 LDS words per lane) are requested BEFORE step t's MFMAs and
        // the order is pinned (round 5): as written before -- read, then multiply -- the compiler issued each step's reads in front of
        // their first use and waited lgkmcnt(0) there, an LDS round trip per 12 MFMAs that three waves per SIMD only partly covered
        auto operands = [&](int t, float (&a)[KT], float (&b)[4]) __attribute__((always_inline)) {
            const int row = 16 * (t >> 2) + 4 * h + (t & 3);
#pragma unroll
            for (int j = 0; j < KT; ++j) a[j] = xb[row * SX + (KT * wk + j) * 16 + l16];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = yb[row * SY + (4 * wn + i) * 16 + l16];
        };
        float a[KT], b[4];
        operands(0, a, b);
#pragma unroll
        for (int t = 0; t < CM / 4; ++t) {
            float an[KT], bn[4];
            if (t + 1 < CM / 4) operands(t + 1, an, bn);
            __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x6);
            if (t + 1 < CM / 4) {
#pragma unroll
                for (int j = 0; j < KT; ++j) a[j] = an[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = bn[i];
            }
        }
        if (do_bias) {
#pragma unroll
            for (int r = 0; r < CM; ++r) bsum += yb[r * SY + tid];
        }
        if (c + 1 < nchunks) stash(buf ^ 1, m_lo + (c + 1) * CM);
        __syncthreads();
    }
    // lane holds out[n0 + 16 (4 wn + i) + l16][k0 + 16 (KT wk + j) + 4 h .. + 3]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + 16 * (4 * wn + i) + l16;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const int k = k0 + 16 * (KT * wk + j) + 4 * h;
            if (n < N && k < K) {
                float* p = out + (int64_t)n * K + k;
                f32x4 v = acc[i][j];
                if (accumulate) v += *(const f32x4*)p;
                *(f32x4*)p = v;
            }
        }
    }
    if (do_bias && n0 + tid < N) bpart[(int64_t)z * N + n0 + tid] = bsum;
}

// ------------------------------------------------------------------------------------------------
// k_gemm_tn_split: the same product with every operand as a THREE-WAY bf16 SPLIT (round 6; mdt_ws.h):
// six v_mfma_f32_16x16x32_bf16 products per 32 rows of m instead of eight v_mfma_f32_16x16x4_f32 per 4 -- fp32 product accuracy,
// a third of the matrix-pipe time.  The reduction index m is the slow index of both operands, and a bf16 MFMA wants EIGHT
// consecutive m of one column per lane: the staging threads transpose on the way into LDS.  Thread = (column quad q, four rows
// 4 h .. 4 h + 3 of the 32-row chunk): four 16-byte loads (rows of one 128-byte line per eight lanes), per column the four m-values
// split into three bf16x4 and stored as 8 bytes at [part][column][8 h]; a fragment read is one ds_read_b128 at [part][column][16 g].
// Column stride 80 bytes: the 8-byte stores of a 16-lane group (two column quads x 8 h) are conflict-free, the reads two-way.
//   workgroup = 8 waves, tile = 128 (n) x TK (k: 128 or 192); wave (wn, wk) = (n half of 64, k quarter of TK / 4): 4 x TK / 64
//   accumulators; chunks of 32 rows double-buffered through registers (next chunk's loads in flight during the MFMAs).
//   MFMA roles: A <- X (i = k column), B <- dY (j = n column): a lane ends with 4 consecutive k of one n, 16-byte stores.
//   Slices, partial outputs and the bias gradient's per-slice column sums as in k_gemm_tn.
// ------------------------------------------------------------------------------------------------
// TNW = 64-column n-groups of the tile (2: 128 wide, 8 waves; 3: 192 wide, 6 waves -- N = 192 / 576 ...), KTW = 16-wide k-tiles per wave:
// waves = (n group, k group of KTW tiles); TK = 16 KTW x (waves / TNW): 128 x 128, 128 x 192 (KTW = 2, 3 with TNW = 2), 192 x 128 (4 with 3)
template <int KTW, int TNW = 2>
__global__ __launch_bounds__(TNW == 2 ? 512 : 384) void k_gemm_tn_split(const float* __restrict__ dY, int64_t ldy, const float* __restrict__ X, int64_t ldx,
                                                       float* __restrict__ out, int64_t slice_stride, int M, int N, int K, int L,
                                                       int accumulate, float* __restrict__ bpart, int xcd) {
    constexpr int NWV = TNW == 2 ? 8 : 6, NTH = 64 * NWV, WKG = NWV / TNW;   // waves, threads, k groups
    constexpr int CM = 32, TN_ = 64 * TNW, TK = 16 * KTW * WKG, NC = TN_ + TK, S = 80, PART = NC * S, BUF = 3 * PART;
    constexpr int NQ = NC / 4, NI = NQ * 8, NU = (NI + NTH - 1) / NTH;   // column quads, staging items (quad, 4-row group), items per thread
    extern __shared__ __attribute__((aligned(16))) char tns_lds[];      // 2 buffers x 3 parts x NC columns x 80 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wk = wave % WKG, wn = wave / WKG;
    const int gk = (K + TK - 1) / TK;
    int id = blockIdx.x + gridDim.x * blockIdx.z;
    if (xcd) {   // whole slices behind one L2 (k_gemm_tn)
        const int nb = gridDim.x * gridDim.z, q = nb >> 3, r = nb & 7, x8 = id & 7, idx = id >> 3;
        id = (x8 < r ? x8 * (q + 1) : r * (q + 1) + (x8 - r) * q) + idx;
    }
    const int z = id / (int)gridDim.x, tile = id - z * (int)gridDim.x;
    const int bn = tile / gk, bk = tile - bn * gk;
    const int n0 = bn * TN_, k0 = bk * TK;
    const int m_lo = z * L, m_hi = min(M, m_lo + L);
    out += (int64_t)z * slice_stride;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- staging items of this thread: item t = tid + NTH u -> rows 4 (t % 8) .. + 3 of the chunk, column quad t / 8 (dY quads first) ----
    int ih[NU], icol[NU];          // first row of the group; LDS column of the quad's first column (-1: no item)
    const float* isrc[NU];         // global address of (row 0 of the chunk's group, quad) at chunk 0
    int64_t ild[NU];
    bool iy[NU], ivalid[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int t = tid + NTH * u, h = t & 7, q = t >> 3;
        ivalid[u] = t < NI;
        ih[u] = 4 * h;
        const bool y = q < TN_ / 4;
        iy[u] = y;
        const int c = y ? 4 * q : 4 * (q - TN_ / 4);            // column inside the tile's dY / X part
        icol[u] = y ? c : TN_ + c;
        const int gc = y ? min(n0 + c, N - 4) : min(k0 + c, K - 4);   // clamped global column (zeros are selected at the store)
        ild[u] = y ? ldy : ldx;
        isrc[u] = (y ? dY : X) + gc;
        ivalid[u] = ivalid[u] && (y ? n0 + c < N : k0 + c < K);
    }
    f32x4 stg[NU][4];
    auto fetch = [&](int mb) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = min(mb + ih[u] + r, m_hi - 1);
                stg[u][r] = ldg4(isrc[u] + m * ild[u]);
            }
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};   // dY items of k-tile-0 workgroups: column sums of this thread's rows (bias gradient)
    const bool do_bias = bpart != nullptr && bk == 0;
    auto stash = [&](char* buf, int mb) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (tid + NTH * u >= NI) continue;
            f32x4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (ivalid[u] && mb + ih[u] + r < m_hi) ? stg[u][r] : zero4;
            if (u == 0 && do_bias && iy[0]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[e] += (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mdt_bf16x4 p1, p2, p3;
                split3_bf16((f32x4){v[0][e], v[1][e], v[2][e], v[3][e]}, p1, p2, p3);
                char* q = buf + (icol[u] + e) * S + 2 * ih[u];
                *(mdt_bf16x4*)q = p1;
                *(mdt_bf16x4*)(q + PART) = p2;
                *(mdt_bf16x4*)(q + 2 * PART) = p3;
            }
        }
    };
    f32x4 acc[KTW][4];
#pragma unroll
    for (int i = 0; i < KTW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = zero4;
    const int l16 = lane & 15, g = lane >> 4;
    const int nchunks = (m_hi - m_lo + CM - 1) / CM;
    if (nchunks > 0) {
        fetch(m_lo);
        stash(tns_lds, m_lo);
    }
    __syncthreads();
    const int xoff = (TN_ + 16 * (KTW * wk) + l16) * S + 16 * g;   // + 16 it * S
    const int yoff = (16 * (4 * wn) + l16) * S + 16 * g;           // + 16 jt * S
    for (int c = 0; c < nchunks; ++c) {
        const char* cur = tns_lds + (c & 1) * BUF;
        if (c + 1 < nchunks) fetch(m_lo + (c + 1) * CM);
        mdt_bf16x8 xf[3][KTW], yf[3][4];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < KTW; ++i) xf[p][i] = *(const mdt_bf16x8*)(cur + p * PART + xoff + 16 * i * S);
#pragma unroll
            for (int j = 0; j < 4; ++j) yf[p][j] = *(const mdt_bf16x8*)(cur + p * PART + yoff + 16 * j * S);
        }
        // the six products, smallest first: x3 y1, x2 y2, x2 y1, x1 y3, x1 y2, x1 y1
#define MDT_TNS_PROD(PX, PY)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < KTW; ++i)                                                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                  \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[PX][i], yf[PY][j], acc[i][j], 0, 0, 0);
        MDT_TNS_PROD(2, 0)
        MDT_TNS_PROD(1, 1)
        MDT_TNS_PROD(1, 0)
        MDT_TNS_PROD(0, 2)
        MDT_TNS_PROD(0, 1)
        MDT_TNS_PROD(0, 0)
#undef MDT_TNS_PROD
        if (c + 1 < nchunks) stash(tns_lds + ((c + 1) & 1) * BUF, m_lo + (c + 1) * CM);
        __syncthreads();
    }
    // lane holds out[n0 + 16 (4 wn + j) + l16][k0 + 16 (KTW wk + i) + 4 g .. + 3]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 16 * (4 * wn + j) + l16;
#pragma unroll
        for (int i = 0; i < KTW; ++i) {
            const int k = k0 + 16 * (KTW * wk + i) + 4 * g;
            if (n < N && k < K) {
                float* p = out + (int64_t)n * K + k;
                f32x4 v = acc[i][j];
                if (accumulate) v += *(const f32x4*)p;
                *(f32x4*)p = v;
            }
        }
    }
    if (do_bias && iy[0]) {   // the eight row groups of a column quad are eight consecutive lanes: add them in a fixed order
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = bs[e];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            bs[e] = v;
        }
        const int q = tid >> 3, n = n0 + 4 * q;
        if ((tid & 7) == 0 && tid + 0 < NI) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < N) bpart[(int64_t)z * N + n + e] = bs[e];
        }
    }
}
static int g_tn_split = -1;   // MDT_HIP_TN_SPLIT / mdt_op_set_tn_split: 0 = the fp32 MFMA kernel everywhere
bool mdt_gemm_tn_split_on() {
    if (g_tn_split < 0) { const char* e = getenv("MDT_HIP_TN_SPLIT"); g_tn_split = e ? atoi(e) : 1; }
    return g_tn_split != 0;
}
extern "C" void mdt_op_set_tn_split(int32_t on) { g_tn_split = on < 0 ? -1 : (on != 0); }
// k-tile of the split kernel for a K-column product: 192 where that pads K less
int mdt_gemm_tn_split_ktile(int K) { return (K + 191) / 192 * 192 < (K + 127) / 128 * 128 ? 192 : 128; }
// tile of the split kernel for an (N, K) product: (n, k) = 192 x 128 where 128-wide n-tiles would pad N by more than an eighth and
// 192-wide ones do not (N = 192, 576 ...), else 128 x (192 where that pads K less, else 128)
void mdt_gemm_tn_split_tile(int N, int K, int* tn, int* tk) {
    const bool n192 = 8 * ((N + 127) / 128 * 128) > 9 * N && 8 * ((N + 191) / 192 * 192) <= 9 * N;
    *tn = n192 ? 192 : 128;
    *tk = n192 ? 128 : mdt_gemm_tn_split_ktile(K);
}
template <int KTW, int TNW>
static hipError_t launch_gemm_tn_split_t(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* out, int64_t slice_stride, int M,
                                         int N, int K, int S, int L, int accumulate, float* bpart, hipStream_t s) {
    constexpr int NWV = TNW == 2 ? 8 : 6, TN_ = 64 * TNW, TK = 16 * KTW * (NWV / TNW);
    constexpr size_t lds = (size_t)2 * 3 * (TN_ + TK) * 80;
    static bool attr_dev[32] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (!attr_dev[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_tn_split<KTW, TNW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_dev[dev] = true;
    }
    hipLaunchKernelGGL((k_gemm_tn_split<KTW, TNW>), dim3(((N + TN_ - 1) / TN_) * ((K + TK - 1) / TK), 1, S), dim3(64 * NWV), lds, s, dY, ldy, X,
                       ldx, out, slice_stride, M, N, K, L, accumulate, bpart, S > 1 ? 1 : 0);
    return hipGetLastError();
}
hipError_t mdt_launch_gemm_tn_split(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* out, int64_t slice_stride, int M, int N,
                                    int K, int S, int L, int accumulate, float* bpart, hipStream_t s) {
    if (M < 1 || N < 4 || K < 4 || (N & 3) || (K & 3) || (ldy & 3) || (ldx & 3) || S < 1) return hipErrorInvalidValue;
    int tn, tk;
    mdt_gemm_tn_split_tile(N, K, &tn, &tk);
#define TNS_ARGS dY, ldy, X, ldx, out, slice_stride, M, N, K, S, L, accumulate, bpart, s
    if (tn == 192) return launch_gemm_tn_split_t<4, 3>(TNS_ARGS);
    return tk == 192 ? launch_gemm_tn_split_t<3, 2>(TNS_ARGS) : launch_gemm_tn_split_t<2, 2>(TNS_ARGS);
#undef TNS_ARGS
}

// dW partials of S row slices of L rows: out + z * slice_stride is slice z's (N, K) product; N, K multiples of 16
// (16-byte aligned rows).  bpart: nullptr or (S, N) per-slice column sums of dY.
// (KT, WN) instantiation behind a shape.  n-tile: 192 wide (12 waves, needs N and K multiples of 192: every Linear of this model)
// from 8192 reduction rows on for matrices of at least 576 x 192, else 128 wide where N is a multiple of 128, else 64; k-tile:
// 192 with the 192-wide n-tile, else the one that pads K less.  MDT_HIP_TN_WIDE = 0 / 1 / 2 forces 64 / 128 / 192 where the shape allows (A/B runs).
void mdt_gemm_tn_tile(int64_t M, int N, int K, int* tn, int* tk) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("MDT_HIP_TN_WIDE"); force = e ? atoi(e) + 1 : 0; }
    const bool ok192 = N % 192 == 0 && K % 192 == 0, ok128 = N % 128 == 0;
    // measured, product + sum of its slices (tools/dw_bench.py, us at 64 / 128 / 192): M = 104448: 1536 x 192 712 / 689 / 628,
    // 192 x 768 357 / - / 307, 576 x 192 288 / - / 285, 192 x 192 115 / - / 139;  M = 10240: 1536 x 384 137 / 130 / 130,
    // 384 x 1536 134 / 128 / 128, 1152 x 384 108 / 98 / 108, 384 x 384 51 / 48 / 63;  M = 4096: 64 wide or a tie
    // (192 wide with ONE round of workgroups, split_rows_tn: 1536 x 192 597, 192 x 768 280, 576 x 192 263)
    // M = 10240 with that slicing: 1536 x 384 119, 384 x 1536 118, 1152 x 384 97, 384 x 384 45 -- 192 wide everywhere it fits
    int w = ok192 && M >= 8192 && (int64_t)N * K >= 576 * 192 ? 192 : (ok128 && M >= 8192 ? 128 : 64);
    if (force == 1) w = 64;
    if (force == 2) w = ok128 ? 128 : 64;
    if (force == 3) w = ok192 ? 192 : (ok128 ? 128 : 64);
    *tn = w;
    *tk = w == 192 ? 192 : mdt_gemm_tn_ktile(K);
}
template <int KT, int WN>
static hipError_t launch_gemm_tn_t(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* out, int64_t slice_stride, int M,
                                   int N, int K, int S, int L, int accumulate, float* bpart, hipStream_t s) {
    constexpr int TN_ = 64 * WN, TK = 64 * KT;
    constexpr size_t lds = (size_t)2 * 32 * ((TN_ + 4) + (TK + 4)) * sizeof(float);
    static bool attr_dev[32] = {false};  // per instantiation AND per device: function attributes are per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (!attr_dev[dev] && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_tn<KT, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_dev[dev] = true;
    }
    static int xcd = -1;  // MDT_HIP_TN_XCD=0: workgroups take (slice, tile) in launch order (A/B runs)
    if (xcd < 0) { const char* e = getenv("MDT_HIP_TN_XCD"); xcd = e ? atoi(e) : 1; }
    hipLaunchKernelGGL((k_gemm_tn<KT, WN>), dim3(((N + TN_ - 1) / TN_) * ((K + TK - 1) / TK), 1, S), dim3(256 * WN), lds, s, dY, ldy, X, ldx,
                       out, slice_stride, M, N, K, L, accumulate, bpart, S > 1 ? xcd : 0);
    return hipGetLastError();
}
hipError_t mdt_launch_gemm_tn(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* out, int64_t slice_stride, int M, int N,
                              int K, int S, int L, int accumulate, float* bpart, hipStream_t s) {
    if (M < 1 || N < 4 || K < 4 || (N & 3) || (K & 3) || (ldy & 3) || (ldx & 3) || S < 1) return hipErrorInvalidValue;
    int tn, tk;
    mdt_gemm_tn_tile(M, N, K, &tn, &tk);
#define TN_ARGS dY, ldy, X, ldx, out, slice_stride, M, N, K, S, L, accumulate, bpart, s
    if (tn == 192) return launch_gemm_tn_t<3, 3>(TN_ARGS);
    if (tk == 192) return tn == 128 ? launch_gemm_tn_t<3, 2>(TN_ARGS) : launch_gemm_tn_t<3, 1>(TN_ARGS);
    return tn == 128 ? launch_gemm_tn_t<2, 2>(TN_ARGS) : launch_gemm_tn_t<2, 1>(TN_ARGS);
#undef TN_ARGS
}
// k-tile width k_gemm_tn uses for a K-column product: the one that pads K less (128 on a tie)
int mdt_gemm_tn_ktile(int K) {
    static int force = -1;  // MDT_HIP_TN_KTILE=128|192: A/B runs
    if (force < 0) { const char* e = getenv("MDT_HIP_TN_KTILE"); force = e ? atoi(e) : 0; }
    if (force == 128 || force == 192) return force;
    const int p128 = (K + 127) / 128 * 128, p192 = (K + 191) / 192 * 192;
    return p192 < p128 ? 192 : 128;
}
