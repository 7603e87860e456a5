// mdt_device.h -- device helpers shared by the kernel translation units (mdt_kernels.hip: inference path,
// mdt_train_kernels.hip: training forward/backward pieces).
#pragma once
#include <hip/hip_runtime.h>

#include "mdt_hip_ops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const f32x4*)p; }

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// Wave-wide sums without LDS-crossbar round trips (ds_bpermute, ~100 cycles per butterfly step): DPP lane swaps inside a
// 16-lane row (xor 1, xor 2 as quad permutes, then the half-row and full-row mirrors leave the row's sum in every lane), then
// row_bcast:15 / row_bcast:31 carry the running total across the four rows into lane 63, which v_readlane hands to everyone.
// MDT_SHFL_REDUCE (tuning build) restores the butterfly.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_get(float v) {  // lanes of rows outside ROWS receive 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef MDT_SHFL_REDUCE
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
#else
    v += dpp_get<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_get<0x140, 0xf>(v);  // row_mirror: every lane holds its row's sum
    v += dpp_get<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_get<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
// the same for each 32-lane half of the wave (two rows of activations normalised side by side): every lane receives the sum
// of its own half
__device__ __forceinline__ float half_wave_sum(float v) {
#ifdef MDT_SHFL_REDUCE
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
#else
    v += dpp_get<0xB1, 0xf>(v);
    v += dpp_get<0x4E, 0xf>(v);
    v += dpp_get<0x141, 0xf>(v);
    v += dpp_get<0x140, 0xf>(v);
    v += dpp_get<0x142, 0xa>(v);  // lanes of rows 1 / 3 now hold the sums of lanes 0..31 / 32..63
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return (threadIdx.x & 32) ? hi : lo;
#endif
}
// wave-wide maximum, same route (rows outside a step's mask keep their own value)
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_keep(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef MDT_SHFL_REDUCE
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, WAVE));
    return v;
#else
    v = fmaxf(v, dpp_keep<0xB1, 0xf>(v));
    v = fmaxf(v, dpp_keep<0x4E, 0xf>(v));
    v = fmaxf(v, dpp_keep<0x141, 0xf>(v));
    v = fmaxf(v, dpp_keep<0x140, 0xf>(v));
    v = fmaxf(v, dpp_keep<0x142, 0xa>(v));
    v = fmaxf(v, dpp_keep<0x143, 0xc>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
// N independent sums advancing together (the steps of different sums interleave: no step waits for its own result)
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#ifdef MDT_SHFL_REDUCE
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float t[N];
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = __shfl_xor(v[i], off, 64);
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] += t[i];
    }
#else
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_get<0xB1, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_get<0x4E, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_get<0x141, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_get<0x140, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_get<0x142, 0xa>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_get<0x143, 0xc>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 63));
#endif
}


// exact-erf GELU (nn.GELU default).  erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 rounding
// level): one rcp, one exp and a degree-5 Horner chain instead of libm erff's ~40 instructions -- the c_fc
// epilogue applies it to 32 values per lane and was a quarter of that kernel's time with erff.
__device__ __forceinline__ float act_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = p * t * __expf(-z * z);          // erfc(z), z >= 0
    const float cdf = x >= 0.f ? 1.0f - 0.5f * q : 0.5f * q;  // Phi(x) without cancellation on the negative side
    return x * cdf;
}
__device__ __forceinline__ float act_mish(float x) {
    float sp = x > 20.0f ? x : log1pf(expf(x));  // torch softplus threshold 20
    return x * tanhf(sp);
}
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + expf(-x)); }

// the same GELU on the GEMM epilogues' vectors of four: written on pairs so that the Horner chain, the squares and the
// final combination become packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32), -|x| once per value instead of |x| and a
// select, the clamp as one v_med3: 9 instructions per value instead of 18 (the fused MLP launch applies it to 32 values per
// lane while its SIMD partner wants the issue slots for MFMAs).  0.5 erfc(|x| / sqrt 2) = h;  x Phi(x) = max(x, 0) - |x| h.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 act_gelu2(f32x2 x) {
#ifdef MDT_GELU_SCALAR  // A/B build: the scalar routine per value
    return (f32x2){act_gelu(x.x), act_gelu(x.y)};
#else
    // (not __builtin_bit_cast on x.x / x.y: on an ext-vector ELEMENT lvalue clang reads element 0 for both)
    const f32x2 nax = -__builtin_elementwise_abs(x);
    const f32x2 d = __builtin_elementwise_fma(nax, (f32x2)(-0.3275911f * 0.70710678118654752440f), (f32x2)(1.0f));
    f32x2 t;
    t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
    f32x2 p = __builtin_elementwise_fma((f32x2)(0.5f * 1.061405429f), t, (f32x2)(0.5f * -1.453152027f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0.5f * 1.421413741f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0.5f * -0.284496736f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0.5f * 0.254829592f));
    const f32x2 q = p * t;
    const f32x2 s = (x * x) * (f32x2)(-0.5f * 1.44269504088896340736f);  // exp(-z^2) = 2^(-x^2 log2(e) / 2)
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(s.x); e.y = __builtin_amdgcn_exp2f(s.y);
    const f32x2 h = q * e;
    const float x0 = x.x, x1 = x.y;
    f32x2 m;  // max(x, 0) as a clamp to [0, inf]
    m.x = __builtin_amdgcn_fmed3f(x0, 0.f, __builtin_inff()); m.y = __builtin_amdgcn_fmed3f(x1, 0.f, __builtin_inff());
    return __builtin_elementwise_fma(nax, h, m);
#endif
}

__device__ __forceinline__ f32x4 apply_act(f32x4 v, int act) {
    if (act == MDT_ACT_GELU) {
        const f32x2 a = act_gelu2(v.xy), b = act_gelu2(v.zw);
        v = (f32x4){a.x, a.y, b.x, b.y};
    } else if (act == MDT_ACT_MISH) {
        v.x = act_mish(v.x); v.y = act_mish(v.y); v.z = act_mish(v.z); v.w = act_mish(v.w);
    } else if (act == MDT_ACT_SILU) {
        v.x = act_silu(v.x); v.y = act_silu(v.y); v.z = act_silu(v.z); v.w = act_silu(v.w);
    }
    return v;
}

// derivatives of the three activations (training backward)
__device__ __forceinline__ float act_gelu_grad(float x) {
    // d/dx [x Phi(x)] = Phi(x) + x phi(x); Phi through the same erfc approximation as act_gelu
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-z * z);
    const float q = p * t * e;
    const float cdf = x >= 0.f ? 1.0f - 0.5f * q : 0.5f * q;
    return fmaf(x * 0.39894228040143267794f, e, cdf);  // phi(x) = exp(-x^2/2) / sqrt(2 pi), and z^2 = x^2/2
}
__device__ __forceinline__ float act_silu_grad(float x) {
    const float s = 1.0f / (1.0f + expf(-x));
    return s * fmaf(x, 1.0f - s, 1.0f);
}
// SiLU / its derivative for the SwishGLU epilogues of the GEMM bodies (gemm_tile GLU 3 / 4, gemm_ws_tile): hardware exp2 and
// reciprocal, 5 / 8 instructions per value (act_silu: libm expf + an IEEE division, ~30 -- 240 vector instructions per lane and
// 32-row tile of the weight-stationary body, whose MFMA loop is 192 instructions long).  |relative error| < 4e-7.
__device__ __forceinline__ float glu_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
__device__ __forceinline__ float glu_silu(float x) { return x * glu_sigmoid(x); }
__device__ __forceinline__ float glu_silu_grad(float x) {
    const float s = glu_sigmoid(x);
    return s * fmaf(x, 1.0f - s, 1.0f);
}
__device__ __forceinline__ float act_mish_grad(float x) {
    // mish = x tanh(sp(x)); d = tanh(sp) + x (1 - tanh(sp)^2) sigmoid(x)
    const float sp = x > 20.0f ? x : log1pf(expf(x));
    const float th = tanhf(sp);
    const float sg = 1.0f / (1.0f + expf(-x));
    return fmaf(x * (1.0f - th * th), sg, th);
}
__device__ __forceinline__ float apply_act1(float v, int act) {
    return act == MDT_ACT_GELU ? act_gelu(v) : act == MDT_ACT_MISH ? act_mish(v) : act == MDT_ACT_SILU ? act_silu(v) : v;
}
__device__ __forceinline__ float apply_act_grad1(float v, int act) {
    return act == MDT_ACT_GELU   ? act_gelu_grad(v)
           : act == MDT_ACT_MISH ? act_mish_grad(v)
           : act == MDT_ACT_SILU ? act_silu_grad(v)
                                 : 1.0f;
}

// reductions over the four 16-lane rows of a wave, same column (lane % 16), through gfx950's v_permlane16_swap /
// v_permlane32_swap (VALU, no LDS round trip): every lane ends with the result
__device__ __forceinline__ float xrow_max(float v) {   // over the four 16-lane rows of the wave (same column m)
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned w = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
    const auto c = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}
__device__ __forceinline__ float xrow_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned w = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
    const auto c = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}

// ------------------------------------------------------------------------------------------------
// counter-based dropout masks (training): Philox4x32-10 keyed by the call's 64-bit seed, counter =
// (element index, site id).  The backward regenerates the mask from the same (seed, site, index), so no mask is
// ever stored.  keep(element) <=> uniform >= p; kept values are scaled by 1 / (1 - p) (nn.Dropout / SDPA dropout_p).
// ------------------------------------------------------------------------------------------------
// One Philox4x32-10 block: the four 32-bit words of counter `ctr` under (seed, site).  Element idx of a site takes word
// idx & 3 of block idx >> 2 (round 6; rounds 1-5 ran a whole block per element and kept one word: ~100 integer instructions per
// dropped-out value, as much as the branch-merge kernels' memory time), so 16-byte accesses pay one block per four elements.
struct philox4_t { uint32_t w[4]; };
__device__ __forceinline__ philox4_t philox4(uint64_t seed, uint32_t site, uint64_t ctr) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = site, c3 = 0x9e3779b9u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return philox4_t{{c0, c1, c2, c3}};
}
__device__ __forceinline__ float dropout_keep(uint32_t word, float p) {   // [0, 1) on a 24-bit grid against p
    return (float)(word >> 8) * (1.0f / 16777216.0f) >= p ? 1.0f / (1.0f - p) : 0.f;
}
// multiplier of element idx: 0 (dropped) or 1 / (1 - p); p == 0 or seed == 0 -> 1
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint32_t site, uint64_t idx, float p) {
    if (p <= 0.f || seed == 0) return 1.f;
    const philox4_t r = philox4(seed, site, idx >> 2);
    const unsigned e = (unsigned)idx & 3u;
    return dropout_keep(e == 0 ? r.w[0] : (e == 1 ? r.w[1] : (e == 2 ? r.w[2] : r.w[3])), p);
}
// the multipliers of elements idx .. idx + 3, idx a multiple of 4: one block
__device__ __forceinline__ f32x4 dropout_scale4(uint64_t seed, uint32_t site, uint64_t idx, float p) {
    if (p <= 0.f || seed == 0) return (f32x4){1.f, 1.f, 1.f, 1.f};
    const philox4_t r = philox4(seed, site, idx >> 2);
    return (f32x4){dropout_keep(r.w[0], p), dropout_keep(r.w[1], p), dropout_keep(r.w[2], p), dropout_keep(r.w[3], p)};
}

// ---- three-way bf16 split of fp32 operands (round 6: mdt_ws.h, mdt_mlp_split.h) ----
// x = p1 + p2 + p3 exactly: each part is the bf16 rounding (v_cvt_pk_bf16_f32) of what the parts before it left.
typedef __bf16 mdt_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mdt_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3_bf16(const f32x4& x, mdt_bf16x4& p1, mdt_bf16x4& p2, mdt_bf16x4& p3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 a = (__bf16)x[e];
        const float r = x[e] - (float)a;
        const __bf16 b = (__bf16)r;
        const float r2 = r - (float)b;
        p1[e] = a; p2[e] = b; p3[e] = (__bf16)r2;
    }
}

// byte offset of the four values at columns c .. c + 3 of a row inside one part of a split tile (row stride rowb bytes): slots
// (half, 0 .. 3) = ((c % 32) / 16, ..) of lane group (c % 16) / 4 in k32 step c / 32 -- the order v_mfma_f32_16x16x32_bf16 reads
// with one ds_read_b128 per part, row tile and k32 step
__device__ __forceinline__ int split_slot(int row, int c, int rowb) {
    return row * rowb + (c >> 5) * 64 + ((c & 15) >> 2) * 16 + ((c & 31) >> 4) * 8;
}
