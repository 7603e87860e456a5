// mdt_mae.hip -- op-level kernels of the masked generative foresight head (include/mdt_mae.h): unmasked multi-head
// self-attention over ~100 tokens with its backward, and the exported RMSNorm / SwishGLU row ops.
//
// The attention of the shipped decoder is 102 tokens x 8 heads of 24 (masked_transformer_decoder.py:68-121 builds
// voltron Blocks of d = 192): 2 % of the head's FLOPs next to its Linears, and the f32-input MFMA runs at the f32 vector
// rate on gfx950, so these are register-blocked VALU kernels over LDS-resident q / k / v: one workgroup per (sample, head),
// 4 x 4 score blocks per thread (8 x 16-byte LDS reads per 64 FMAs), scores kept in LDS with an odd row stride.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdt_internal.h"
#include "mdt_device.h"

namespace {

constexpr int TMAX = 128;

// rows [0, T) of a (T, HD) head slice: global (row stride ld) -> LDS (row stride HD + 4)
template <int HD, int NT>
__device__ __forceinline__ void load_rows(const float* __restrict__ src, int64_t ld, float* dst, int T, int tid) {
    constexpr int H4 = HD / 4, ST = HD + 4;
    for (int i = tid; i < T * H4; i += NT) {
        const int t = i / H4, c = i - t * H4;
        *(f32x4*)(dst + t * ST + 4 * c) = ldg4(src + (int64_t)t * ld + 4 * c);
    }
}

// S[i][j] = alpha * a_i . b_j for i, j < T (rows of a / b in LDS, stride HD + 4); 4 x 4 blocks per thread.
// MODE 0: store;  MODE 1: S[i][j] = S[i][j] * (value - rowdot[i]) * alpha2   (dS from P and dP, in place)
template <int HD, int NT, int MODE>
__device__ __forceinline__ void outer_blocks(const float* a, const float* b, float* S, int sstride, int T, float alpha,
                                             const float* rowdot, int tid) {
    constexpr int H4 = HD / 4, ST = HD + 4;
    const int nb = (T + 3) >> 2;
    for (int it = tid; it < nb * nb; it += NT) {
        const int bi = it / nb, bj = it - bi * nb;
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
        const float* ap[4];
        const float* bp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ap[r] = a + min(4 * bi + r, T - 1) * ST;
            bp[r] = b + min(4 * bj + r, T - 1) * ST;
        }
#pragma unroll
        for (int c4 = 0; c4 < H4; ++c4) {
            f32x4 av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = *(const f32x4*)(ap[r] + 4 * c4); bv[r] = *(const f32x4*)(bp[r] + 4 * c4); }
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
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 4 * bj + c;
                    if (j < T) {
                        float* p = S + i * sstride + j;
                        if (MODE == 0) *p = acc[r][c] * alpha;
                        else *p = *p * (acc[r][c] - rowdot[i]) * alpha;
                    }
                }
            }
        }
    }
}

// rows of S (T x T, stride sstride) -> softmax in place; one wave per row, 2 columns per lane (T <= 128)
template <int NT>
__device__ __forceinline__ void softmax_rows(float* S, int sstride, int T, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < T; i += NT / 64) {
        float* row = S + i * sstride;
        const float x0 = lane < T ? row[lane] : -INFINITY, x1 = lane + 64 < T ? row[lane + 64] : -INFINITY;
        float mx = fmaxf(x0, x1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float e0 = lane < T ? expf(x0 - mx) : 0.f, e1 = lane + 64 < T ? expf(x1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        if (lane < T) row[lane] = e0 * inv;
        if (lane + 64 < T) row[lane + 64] = e1 * inv;
    }
}

// out[i][:] = sum_j W(i, j) * v[j][:]   with W(i, j) = S[i][j] (TRANS = false) or S[j][i] (TRANS = true)
// dst: LDS (stride HD + 4) or global (stride ldd); item = (row, float4 column)
template <int HD, int NT, bool TRANS>
__device__ __forceinline__ void weighted_rows(const float* S, int sstride, const float* v, float* dst, int64_t ldd, int T, float alpha,
                                              int tid) {
    constexpr int H4 = HD / 4, ST = HD + 4;
    for (int it = tid; it < T * H4; it += NT) {
        const int i = it / H4, c4 = it - i * H4;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < T; ++j) {
            const float w = TRANS ? S[j * sstride + i] : S[i * sstride + j];
            acc += w * *(const f32x4*)(v + j * ST + 4 * c4);
        }
        *(f32x4*)(dst + (int64_t)i * ldd + 4 * c4) = acc * alpha;
    }
}

template <int HD>
__global__ __launch_bounds__(256) void k_attn_mid_fwd(const float* __restrict__ qkv, int64_t ld, float* __restrict__ out, int64_t ldo,
                                                      int H, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 256, ST = HD + 4;
    const int tid = threadIdx.x, b = blockIdx.x, h = blockIdx.y, D = H * HD;
    const int ss = T + 1 + (T & 1);  // odd stride (floats)
    float* qs = lds;
    float* ks = qs + T * ST;
    float* vs = ks + T * ST;
    float* S = vs + T * ST;
    const float* base = qkv + (int64_t)b * T * ld + h * HD;
    load_rows<HD, NT>(base, ld, qs, T, tid);
    load_rows<HD, NT>(base + D, ld, ks, T, tid);
    load_rows<HD, NT>(base + 2 * D, ld, vs, T, tid);
    __syncthreads();
    outer_blocks<HD, NT, 0>(qs, ks, S, ss, T, scale, nullptr, tid);
    __syncthreads();
    softmax_rows<NT>(S, ss, T, tid);
    __syncthreads();
    weighted_rows<HD, NT, false>(S, ss, vs, out + (int64_t)b * T * ldo + h * HD, ldo, T, 1.0f, tid);
}

template <int HD>
__global__ __launch_bounds__(512) void k_attn_mid_bwd(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ d_out,
                                                      int64_t ldd, float* __restrict__ d_qkv, int64_t ldg, int H, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512, ST = HD + 4;
    const int tid = threadIdx.x, b = blockIdx.x, h = blockIdx.y, D = H * HD;
    const int ss = T + 1 + (T & 1);
    float* qs = lds;
    float* ks = qs + T * ST;
    float* vs = ks + T * ST;
    float* dos = vs + T * ST;
    float* os = dos + T * ST;
    float* S = os + T * ST;
    float* rowdot = S + T * ss;
    const float* base = qkv + (int64_t)b * T * ld + h * HD;
    load_rows<HD, NT>(base, ld, qs, T, tid);
    load_rows<HD, NT>(base + D, ld, ks, T, tid);
    load_rows<HD, NT>(base + 2 * D, ld, vs, T, tid);
    load_rows<HD, NT>(d_out + (int64_t)b * T * ldd + h * HD, ldd, dos, T, tid);
    __syncthreads();
    outer_blocks<HD, NT, 0>(qs, ks, S, ss, T, scale, nullptr, tid);
    __syncthreads();
    softmax_rows<NT>(S, ss, T, tid);  // S = P
    __syncthreads();
    weighted_rows<HD, NT, false>(S, ss, vs, os, ST, T, 1.0f, tid);  // O = P V (the forward output, recomputed)
    __syncthreads();
    // sum_j P_ij dP_ij = dO_i . O_i
    for (int i = tid; i < T; i += NT) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc = fmaf(dos[i * ST + c], os[i * ST + c], acc);
        rowdot[i] = acc;
    }
    float* g = d_qkv + (int64_t)b * T * ldg + h * HD;
    weighted_rows<HD, NT, true>(S, ss, dos, g + 2 * D, ldg, T, 1.0f, tid);  // dV = P^T dO (before P is overwritten)
    __syncthreads();
    outer_blocks<HD, NT, 1>(dos, vs, S, ss, T, scale, rowdot, tid);  // S = scale * P * (dO v^T - rowdot)
    __syncthreads();
    weighted_rows<HD, NT, false>(S, ss, ks, g, ldg, T, 1.0f, tid);      // dQ = dS K
    weighted_rows<HD, NT, true>(S, ss, qs, g + D, ldg, T, 1.0f, tid);   // dK = dS^T Q
}

size_t attn_mid_lds(int hd, int T, bool bwd) {
    const int ss = T + 1 + (T & 1);
    return ((size_t)(bwd ? 5 : 3) * T * (hd + 4) + (size_t)T * ss + (bwd ? TMAX : 0)) * sizeof(float);
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
