// mdt_mae.hip -- op-level kernels of the masked generative foresight head (include/mdt_mae.h): unmasked multi-head
// self-attention over ~100 tokens with its backward, and the exported RMSNorm / SwishGLU row ops.
//
// The attention of the shipped decoder is 102 tokens x 8 heads of 24 (masked_transformer_decoder.py:68-121 builds
// voltron Blocks of d = 192): 2 % of the head's FLOPs next to its Linears, and the f32-input MFMA runs at the f32 vector
// rate on gfx950, so these are register-blocked VALU kernels over LDS-resident q / k / v: one workgroup per (sample, head),
// 4 x 4 score blocks per thread (8 x 16-byte LDS reads per 64 FMAs), scores kept in LDS (rows padded to 16-byte multiples).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdt_internal.h"
#include "mdt_device.h"

namespace {

constexpr int TMAX = 128;

__device__ __forceinline__ int score_stride(int T) { return ((T + 3) & ~3) + 4; }  // floats; rows 16-byte aligned

// rows [0, T) of NB (T, HD) head slices: global (row stride ld[b]) -> LDS (row stride HD + 4, buffers T4 * (HD + 4) floats
// apart); rows T .. T4-1 are zeroed.  All loads of a sweep (4 per buffer and thread) are requested from clamped addresses
// before the first one is consumed -- a load behind a branch costs one memory round trip each.
template <int HD, int NT, int NB>
__device__ __forceinline__ void load_rows(const float* const (&src)[NB], const int64_t (&ld)[NB], float* dst, int T, int tid) {
    constexpr int H4 = HD / 4, ST = HD + 4, U = 4;
    const int T4 = (T + 3) & ~3, n = T4 * H4;
    for (int i0 = tid; i0 < n; i0 += NT * U) {
        f32x4 v[NB][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * NT, n - 1), t = i / H4, c = i - t * H4;
#pragma unroll
            for (int b = 0; b < NB; ++b) v[b][u] = ldg4(src[b] + (int64_t)min(t, T - 1) * ld[b] + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT, t = i / H4, c = i - t * H4;
            if (i < n) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    *(f32x4*)(dst + b * T4 * ST + t * ST + 4 * c) = t < T ? v[b][u] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// S[i][j] = alpha * a_i . b_j for i, j < T (rows of a / b in LDS, stride HD + 4); 4 x 4 blocks per thread, 64 FMAs per
// 8 x 16-byte LDS reads.  The 16 lanes of an LDS access group form a 4 x 4 patch of blocks: 4 distinct a-rows and 4
// distinct b-rows per access, whose start banks differ (row stride 4 * (HD + 4) floats) -- conflict free.
// MODE 0: store;  MODE 1: S[i][j] = S[i][j] * (value - rowdot[i]) * alpha   (dS from P and dP, in place)
template <int HD, int NT, int MODE>
__device__ __forceinline__ void outer_blocks(const float* a, const float* b, float* S, int sstride, int T, float alpha,
                                             const float* rowdot, int tid) {
    constexpr int H4 = HD / 4, ST = HD + 4;
    const int nb = (T + 3) >> 2, nt = (nb + 3) >> 2;
    for (int it = tid; it < nt * nt * 16; it += NT) {
        const int tile = it >> 4, l = it & 15;
        const int ti = tile / nt, tj = tile - ti * nt;
        const int bi = 4 * ti + (l >> 2), bj = 4 * tj + (l & 3);
        if (bi >= nb || bj >= nb) continue;
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
        const float* ap = a + 4 * bi * ST;  // rows up to T4 - 1 exist (zero padded)
        const float* bp = b + 4 * bj * ST;
#pragma unroll
        for (int c4 = 0; c4 < H4; ++c4) {
            f32x4 av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = *(const f32x4*)(ap + r * ST + 4 * c4); bv[r] = *(const f32x4*)(bp + r * ST + 4 * c4); }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[r][c] = fmaf(av[r].x, bv[c].x, acc[r][c]); acc[r][c] = fmaf(av[r].y, bv[c].y, acc[r][c]);
                    acc[r][c] = fmaf(av[r].z, bv[c].z, acc[r][c]); acc[r][c] = fmaf(av[r].w, bv[c].w, acc[r][c]);
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * bi + r;
            if (i < T) {
                float* p = S + i * sstride + 4 * bj;
                f32x4 o;
                if (MODE == 0) {
                    o = (f32x4){acc[r][0], acc[r][1], acc[r][2], acc[r][3]} * alpha;
                } else {
                    const f32x4 pv = *(const f32x4*)p;
                    const float rd = rowdot[i];
                    o = pv * ((f32x4){acc[r][0], acc[r][1], acc[r][2], acc[r][3]} - rd) * alpha;
                }
                // columns >= T stay zero (the padding is part of later reductions)
                if (4 * bj + 1 > T - 1) o.y = 0.f;
                if (4 * bj + 2 > T - 1) o.z = 0.f;
                if (4 * bj + 3 > T - 1) o.w = 0.f;
                *(f32x4*)p = o;
            }
        }
    }
}

// rows of S (T x T, stride sstride) -> softmax in place; 16 lanes per row (4 rows per wave at a time), 8 columns per lane
template <int NT>
__device__ __forceinline__ void softmax_rows(float* S, int sstride, int T, int tid) {
    const int l16 = tid & 15, grp = tid >> 4;
    for (int i = grp; i < T; i += NT / 16) {
        float* row = S + i * sstride;
        float x[8];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = l16 + 16 * u;
            x[u] = j < T ? row[j] : -INFINITY;
            mx = fmaxf(mx, x[u]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = expf(x[u] - mx); sum += x[u]; }  // exp(-inf) = 0 for the padding
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = l16 + 16 * u;
            if (j < T) row[j] = x[u] * inv;
        }
    }
}

// out[r][:] = sum_k W(r, k) * x[k][:]   with W(r, k) = S[r][k] (TRANS = false) or S[k][r] (TRANS = true); 4 output rows x
// one float4 column per thread, the reduction walked 4 at a time: 8 x 16-byte LDS reads per 64 FMAs.  S and x are zero
// padded to a multiple of 4 rows / columns.  dst: LDS (stride HD + 4) or global (stride ldd).
template <int HD, int NT, bool TRANS>
__device__ __forceinline__ void weighted_rows(const float* S, int sstride, const float* x, float* dst, int64_t ldd, int T, int tid) {
    constexpr int H4 = HD / 4, ST = HD + 4;
    const int nb = (T + 3) >> 2;
    for (int it = tid; it < nb * H4; it += NT) {
        const int rb = it / H4, c4 = it - rb * H4;
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nb; ++kb) {
            f32x4 w[4], xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // !TRANS: w[q] = S[4 rb + q][4 kb .. +3] (row of the output block);  TRANS: w[q] = S[4 kb + q][4 rb .. +3]
                w[q] = *(const f32x4*)(S + (TRANS ? (4 * kb + q) * sstride + 4 * rb : (4 * rb + q) * sstride + 4 * kb));
                xv[q] = *(const f32x4*)(x + (4 * kb + q) * ST + 4 * c4);
            }
            if (TRANS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0] += w[q].x * xv[q]; acc[1] += w[q].y * xv[q]; acc[2] += w[q].z * xv[q]; acc[3] += w[q].w * xv[q];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += w[r].x * xv[0] + w[r].y * xv[1] + w[r].z * xv[2] + w[r].w * xv[3];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * rb + r < T) *(f32x4*)(dst + (int64_t)(4 * rb + r) * ldd + 4 * c4) = acc[r];
    }
}

template <int NT>
__device__ __forceinline__ void zero_lds(float* p, int n, int tid) {
    for (int i = tid; i < (n >> 2); i += NT) *(f32x4*)(p + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_mid_fwd(const float* __restrict__ qkv, int64_t ld, float* __restrict__ out, int64_t ldo,
                                                      int H, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 256, ST = HD + 4;
    const int tid = threadIdx.x, b = blockIdx.x, h = blockIdx.y, D = H * HD;
    const int T4 = (T + 3) & ~3, ss = score_stride(T);
    float* qs = lds;
    float* ks = qs + T4 * ST;
    float* vs = ks + T4 * ST;
    float* S = vs + T4 * ST;
    const float* base = qkv + (int64_t)b * T * ld + h * HD;
    zero_lds<NT>(S, T4 * ss, tid);
    {
        const float* const src[3] = {base, base + D, base + 2 * D};
        const int64_t lds_[3] = {ld, ld, ld};
        load_rows<HD, NT, 3>(src, lds_, qs, T, tid);  // q | k | v are consecutive LDS buffers
    }
    __syncthreads();
    outer_blocks<HD, NT, 0>(qs, ks, S, ss, T, scale, nullptr, tid);
    __syncthreads();
    softmax_rows<NT>(S, ss, T, tid);
    __syncthreads();
    weighted_rows<HD, NT, false>(S, ss, vs, out + (int64_t)b * T * ldo + h * HD, ldo, T, tid);
}

template <int HD>
__global__ __launch_bounds__(512) void k_attn_mid_bwd(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ d_out,
                                                      int64_t ldd, float* __restrict__ d_qkv, int64_t ldg, int H, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512, ST = HD + 4;
    const int tid = threadIdx.x, b = blockIdx.x, h = blockIdx.y, D = H * HD;
    const int T4 = (T + 3) & ~3, ss = score_stride(T);
    float* qs = lds;
    float* ks = qs + T4 * ST;
    float* vs = ks + T4 * ST;
    float* dos = vs + T4 * ST;
    float* os = dos + T4 * ST;
    float* S = os + T4 * ST;
    float* rowdot = S + T4 * ss;
    const float* base = qkv + (int64_t)b * T * ld + h * HD;
    zero_lds<NT>(S, T4 * ss, tid);
    {
        const float* const src[4] = {base, base + D, base + 2 * D, d_out + (int64_t)b * T * ldd + h * HD};
        const int64_t lds_[4] = {ld, ld, ld, ldd};
        load_rows<HD, NT, 4>(src, lds_, qs, T, tid);  // q | k | v | dO are consecutive LDS buffers
    }
    __syncthreads();
    outer_blocks<HD, NT, 0>(qs, ks, S, ss, T, scale, nullptr, tid);
    __syncthreads();
    softmax_rows<NT>(S, ss, T, tid);  // S = P
    __syncthreads();
    // the row products have (T / 4) * (HD / 4) items (156 for T = 102, HD = 24): the two that share an input run side by
    // side on the two halves of the workgroup
    float* g = d_qkv + (int64_t)b * T * ldg + h * HD;
    if (tid < 256) weighted_rows<HD, 256, false>(S, ss, vs, os, ST, T, tid);            // O = P V (the forward output, recomputed)
    else weighted_rows<HD, 256, true>(S, ss, dos, g + 2 * D, ldg, T, tid - 256);        // dV = P^T dO (before P is overwritten)
    __syncthreads();
    // sum_j P_ij dP_ij = dO_i . O_i
    for (int i = tid; i < T; i += NT) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc = fmaf(dos[i * ST + c], os[i * ST + c], acc);
        rowdot[i] = acc;
    }
    __syncthreads();
    outer_blocks<HD, NT, 1>(dos, vs, S, ss, T, scale, rowdot, tid);  // S = scale * P * (dO v^T - rowdot)
    __syncthreads();
    if (tid < 256) weighted_rows<HD, 256, false>(S, ss, ks, g, ldg, T, tid);            // dQ = dS K
    else weighted_rows<HD, 256, true>(S, ss, qs, g + D, ldg, T, tid - 256);             // dK = dS^T Q
}

size_t attn_mid_lds(int hd, int T, bool bwd) {
    const int T4 = (T + 3) & ~3, ss = T4 + 4;
    return ((size_t)(bwd ? 5 : 3) * T4 * (hd + 4) + (size_t)T4 * ss + (bwd ? TMAX : 0)) * sizeof(float);
}

template <int HD>
hipError_t launch_fwd(const float* qkv, int64_t ld, float* out, int64_t ldo, int64_t B, int H, int T, float scale, hipStream_t s) {
    const size_t lds = attn_mid_lds(HD, T, false);
    hipError_t e = hipFuncSetAttribute((const void*)k_attn_mid_fwd<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_attn_mid_fwd<HD>), dim3((unsigned)B, H), dim3(256), lds, s, qkv, ld, out, ldo, H, T, scale);
    return hipGetLastError();
}
template <int HD>
hipError_t launch_bwd(const float* qkv, int64_t ld, const float* d_out, int64_t ldd, float* d_qkv, int64_t ldg, int64_t B, int H, int T,
                      float scale, hipStream_t s) {
    const size_t lds = attn_mid_lds(HD, T, true);
    hipError_t e = hipFuncSetAttribute((const void*)k_attn_mid_bwd<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_attn_mid_bwd<HD>), dim3((unsigned)B, H), dim3(512), lds, s, qkv, ld, d_out, ldd, d_qkv, ldg, H, T, scale);
    return hipGetLastError();
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" mdt_status mdt_op_attn_mid_fwd(const float* qkv, int64_t ld_qkv, float* out, int64_t ld_out, int64_t B, int32_t H, int32_t hd,
                                          int32_t T, float scale, void* stream) {
    if (!qkv || !out || B < 1 || H < 1 || T < 1) return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_mid_fwd: bad argument");
    if (T > TMAX || B > 65535 * 32768ll) return mdt_fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_mid_fwd: T must be <= %d", TMAX);
    if (!aligned16(qkv) || !aligned16(out) || ld_qkv % 4 || ld_out % 4)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_mid_fwd: pointers 16-byte aligned, strides multiples of 4");
    if (attn_mid_lds(hd, T, false) > 160 * 1024) return mdt_fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_mid_fwd: (T, hd) does not fit LDS");
    hipStream_t s = (hipStream_t)stream;
    switch (hd) {
        case 16: LAUNCH(launch_fwd<16>(qkv, ld_qkv, out, ld_out, B, H, T, scale, s)); break;
        case 24: LAUNCH(launch_fwd<24>(qkv, ld_qkv, out, ld_out, B, H, T, scale, s)); break;
        case 32: LAUNCH(launch_fwd<32>(qkv, ld_qkv, out, ld_out, B, H, T, scale, s)); break;
        case 48: LAUNCH(launch_fwd<48>(qkv, ld_qkv, out, ld_out, B, H, T, scale, s)); break;
        case 64: LAUNCH(launch_fwd<64>(qkv, ld_qkv, out, ld_out, B, H, T, scale, s)); break;
        default: return mdt_fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_mid_fwd: head dim %d (supported 16/24/32/48/64)", hd);
    }
    return MDT_OK;
}

extern "C" mdt_status mdt_op_attn_mid_bwd(const float* qkv, int64_t ld_qkv, const float* d_out, int64_t ld_do, float* d_qkv,
                                          int64_t ld_dqkv, int64_t B, int32_t H, int32_t hd, int32_t T, float scale, void* stream) {
    if (!qkv || !d_out || !d_qkv || B < 1 || H < 1 || T < 1) return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_mid_bwd: bad argument");
    if (T > TMAX) return mdt_fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_mid_bwd: T must be <= %d", TMAX);
    if (!aligned16(qkv) || !aligned16(d_out) || !aligned16(d_qkv) || ld_qkv % 4 || ld_do % 4 || ld_dqkv % 4)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_mid_bwd: pointers 16-byte aligned, strides multiples of 4");
    if (attn_mid_lds(hd, T, true) > 160 * 1024) return mdt_fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_mid_bwd: (T, hd) does not fit LDS");
    hipStream_t s = (hipStream_t)stream;
    switch (hd) {
        case 16: LAUNCH(launch_bwd<16>(qkv, ld_qkv, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        case 24: LAUNCH(launch_bwd<24>(qkv, ld_qkv, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        case 32: LAUNCH(launch_bwd<32>(qkv, ld_qkv, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        case 48: LAUNCH(launch_bwd<48>(qkv, ld_qkv, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        case 64: LAUNCH(launch_bwd<64>(qkv, ld_qkv, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        default: return mdt_fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_mid_bwd: head dim %d (supported 16/24/32/48/64)", hd);
    }
    return MDT_OK;
}

extern "C" mdt_status mdt_op_rms_fwd(const float* x, const float* g, float* out, int64_t M, int32_t D, float eps, void* stream) {
    if (!x || !g || !out || M < 1 || D < 1 || D > 512) return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_rms_fwd: bad argument (D <= 512)");
    LAUNCH(mdt_launch_rms_fwd(x, g, out, M, D, eps, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" int64_t mdt_op_rms_bwd_scratch(int64_t M, int32_t D) { return ((M + 3) / 4) * (int64_t)D; }

extern "C" mdt_status mdt_op_rms_bwd(const float* x, const float* g, const float* dy, float* dx, int32_t accumulate_dx, float* dg,
                                     int32_t accumulate_dg, int64_t M, int32_t D, float eps, float* scratch, void* stream) {
    if (!x || !g || !dy || !dx || !scratch || M < 1 || D < 1 || D > 512)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_rms_bwd: bad argument (D <= 512)");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(mdt_launch_rms_bwd(x, g, dy, dx, accumulate_dx, scratch, M, D, eps, s));
    if (dg) LAUNCH(mdt_launch_colsum(scratch, D, (int)((M + 3) / 4), D, dg, accumulate_dg, s));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_swiglu_fwd(const float* u, float* out, int64_t M, int32_t H, void* stream) {
    if (!u || !out || M < 1 || H < 1) return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_swiglu_fwd: bad argument");
    LAUNCH(mdt_launch_swiglu_fwd(u, out, M, H, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_swiglu_bwd(const float* u, const float* d_out, float* du, int64_t M, int32_t H, void* stream) {
    if (!u || !d_out || !du || M < 1 || H < 1) return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_swiglu_bwd: bad argument");
    LAUNCH(mdt_launch_swiglu_bwd(u, d_out, du, M, H, (hipStream_t)stream));
    return MDT_OK;
}
