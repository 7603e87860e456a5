// mdt_ws.h -- the WEIGHT-STATIONARY GEMM body (round 5): shallow products (K <= 256) over very many rows.
//
// Why a third body.  The masked-image head at B = 1024 multiplies 104 448 rows by K = 192 weights (qkv, c_proj, the SwishGLU
// project product and mlp.1's input gradient): a tile's MFMA loop is 12 k-steps long, so a tile is mostly its operands'
// arrival.  Counters of the two existing bodies on those products (profiles/r05_mae_sq_counters.txt): waves WAIT on an
// `s_waitcnt` for two thirds of their life, two waves per SIMD are resident on average although registers and LDS would admit
// four, the MFMA pipe is busy for 0.41 - 0.64 of the launch -- while HBM-side traffic is 1.5 - 2.9 TB/s of 8: the products are
// neither fabric- nor issue-bound, they wait for weight fragments that every one of 13 - 20 thousand workgroups pulls from L2
// again (gemm_tile: 143 bytes through the CU's vector memory path per MFMA at K = 192; the tall body's 128 x 64 tile: 96).
//
// Here a workgroup keeps its weights for its whole life: wave w owns NTW column tiles over the WHOLE of K as NTW * K/16
// fragment quads in registers (24 quads = 96 VGPRs for NTW = 2, K = 192), requested once; the workgroup then walks `tiles`
// consecutive 32-row tiles of A: tile t + 1 travels global -> registers while tile t is multiplied out of LDS (two buffers, ONE
// barrier per tile), and tile t's epilogue stores drain under tile t + 1's MFMAs.  Per MFMA: 16 bytes of A through LDS, nothing
// else.  (The k-steps are pinned like gemm_tile's, and the SwishGLU epilogues use the hardware exp2 / reciprocal: 688 -> 601 us forward,
// 479 -> 348 us backward per block of the head at B = 1024, 28.65 -> 27.5 ms per head step; profiles/r05_ws_ab.txt.)
// Shapes: 12 waves (384-column panels: three waves per SIMD, 256 workgroups) where N allows and the epilogue has no prefetched operands,
// else 8 waves (256-column panels, 240 workgroups).
// Grid = column panels x row chunks, sized to one round of one workgroup per CU; the panels of one row chunk run on the
// SAME XCD (block b -> XCD b % 8), so an A tile comes from HBM once and from that XCD's L2 for the other panels.
//
// Same transposed MFMA form and the same K order as gemm_tile / gemm_tall_tile: results are BIT-IDENTICAL to both.
//   out = epilogue(A @ W^T + bias): GLU = 0 plain rows (aux_mode 0, no activation, no residual / row remap),
//   GLU = 1 the activation + the training hooks of gemm_tile's plain kernels (mdt_gemm_args.act; aux_mode 1: the pre-activation
//   kept beside the activated value; aux_mode 2: the product times act'(u) of the layer below),
//   GLU = 3 / 4 the SwishGLU forward / backward epilogues of gemm_tile (mdt_gemm_args.aux_mode 3 / 4).
//
// Round 6: K = 384 (K16 = 24, ONE column tile per wave: the same 24 fragment quads = 96 VGPRs) for the denoiser's training step
// at B = 1024 (M = 10240 rows): every d x d product of its forward and input-gradient pass (24 launches a step), qkv (N = 3d)
// and c_fc / c_proj's input gradient with their GELU hooks (N = 4d).  The 32 x 128 row tiles pulled the whole 196 KB column
// panel from L2 for every 32 rows (960 workgroups x 196 KB = 188 MB per launch of a 3 GFLOP product, MFMA-busy 0.58); here a
// workgroup pulls it once for its 4 - 10 tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdt_device.h"
#include "mdt_internal.h"
#include "mdt_tiles.h"

// K16 = K / 16 (compile time: the fragments live in registers); NTW column tiles per wave; NWAVES waves.  `tiles` = 32-row
// tiles per workgroup, `panels` = N / (NWAVES * NTW * 16).  lds: 2 * 32 * (K + 4) floats.
// The per-tile barrier of both bodies: the LDS stores of tile t + 1 have landed (lgkmcnt) and everybody is done reading tile t.
// NOT __syncthreads(): its fence also drains vmcnt, i.e. parks every wave until the epilogue stores of tile t have reached memory
// (SQ_WAIT_ANY was 22-34 % of the wave cycles of the split body, profiles/r06_ws_split.txt) -- nothing here reads them back.
__device__ __forceinline__ void ws_tile_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int K16, int NTW, int NWAVES, int GLU>
__device__ __forceinline__ void gemm_ws_tile(const mdt_gemm_args& a, int panel, int chunk, int tiles, float* lds,
                                             const float* __restrict__ zeros, int tid) {
    constexpr int K = K16 * 16, STRIDE = K + 4, TILE = 32 * STRIDE, K4 = K / 4, NT = 64 * NWAVES;
    constexpr int NLD = (32 * K4 + NT - 1) / NT;          // float4 items of an A tile per thread (3 at K = 192, 512 threads)
    static_assert(GLU != 3 || NTW % 2 == 0, "SwishGLU forward pairs the column tiles of a wave");
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const int nt0 = (panel * NWAVES + wave) * NTW;
    const int ntiles = (a.M + 31) >> 5;
    const int t0 = chunk * tiles, t1 = min(t0 + tiles, ntiles);
    if (t0 >= t1) return;                                  // (whole workgroup: no barrier is skipped by a part of it)

    // ---- this wave's weights, once: NTW x K16 fragments (1 KiB each, lane order) ----
    f32x4 w[NTW][K16];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int kc = 0; kc < K16; ++kc) w[j][kc] = ldg4(a.Wp + (((int64_t)(nt0 + j) * K16 + kc) * 64 + lane) * 4);
    const int nq = 4 * (lane >> 4);
    const float* biasp = a.bias != nullptr ? a.bias : zeros;
    int ncol[NTW];
    f32x4 bias_v[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int T = nt0 + j;
        // SwishGLU forward: the weight image interleaves the projected / gate halves tile by tile (mdt_op_pack_weight_glu); bias and
        // output columns are the NATURAL ones: tile T -> half T & 1, columns 16 (T >> 1)
        ncol[j] = (GLU == 3 ? (T & 1) * (a.N >> 1) + (T >> 1) * 16 : T * 16) + nq;
        bias_v[j] = ldg4(biasp + ncol[j]);
    }

    // ---- A tile t: item idx = tid + NT * u -> row idx / K4, float4 column idx % K4 (rows past M re-read the last row) ----
    int lrow[NLD], lcol[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int idx = min(tid + NT * u, 32 * K4 - 1);
        lrow[u] = idx / K4;
        lcol[u] = 4 * (idx - lrow[u] * K4);
    }
    f32x4 stage[NLD];
    auto request = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int64_t m = min(32 * t + lrow[u], a.M - 1);
            stage[u] = ldg4(a.A + m * a.lda + lcol[u]);
        }
    };
    auto commit = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            if (tid + NT * u < 32 * K4) *(f32x4*)(buf + lrow[u] * STRIDE + lcol[u]) = stage[u];
    };
    request(t0);
    commit(lds);
    __syncthreads();

    const int aoff = (lane & 15) * STRIDE + 4 * (lane >> 4);
    for (int t = t0; t < t1; ++t) {
        const float* cur = lds + ((t - t0) & 1) * TILE;
        float* nxt = lds + ((t - t0 + 1) & 1) * TILE;
        const bool more = t + 1 < t1;
        if (more) request(t + 1);                          // travels under this tile's MFMAs
        // rows of this lane in the two 16-row halves of the tile, and (SwishGLU backward) the u operands beside them
        int64_t oo[2];
        bool okr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mr = 32 * t + 16 * i + (lane & 15);
            okr[i] = mr < a.M;
            oo[i] = (int64_t)min(mr, a.M - 1) * a.ldo;
        }
        // (a twelve-wave instantiation would have to do without: 168 VGPRs have no room for the prefetch -- measured 448 us against 348,
        //  so the dispatcher keeps the backward product on the 8-wave shape)
        constexpr bool AUX_EARLY = GLU == 4 && NWAVES < 12;
        f32x4 pj[AUX_EARLY ? 2 : 1][AUX_EARLY ? NTW : 1], gt[AUX_EARLY ? 2 : 1][AUX_EARLY ? NTW : 1];
        if constexpr (AUX_EARLY) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    pj[i][j] = ldg4(a.aux + oo[i] + ncol[j]);
                    gt[i][j] = ldg4(a.aux + oo[i] + a.N + ncol[j]);
                }
        }
        f32x4 acc[2][NTW];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = zero4;
        f32x4 af[2], an[2];
        af[0] = *(const f32x4*)(cur + aoff);
        af[1] = *(const f32x4*)(cur + aoff + 16 * STRIDE);
        // The schedule of a k-step is PINNED (MDT_SCHED_PIN, mdt_tiles.h): left alone, hipcc sinks the next step's two LDS reads to
        // three MFMAs in front of their first use and waits for them there -- an exposed LDS round trip per 16 MFMAs (the ISA of the
        // first version).  Written and kept: first quarter of the step's MFMAs, the next step's reads, the other three quarters.
#pragma unroll
        for (int kc = 0; kc < K16; ++kc) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][kc][0], af[i][0], acc[i][j], 0, 0, 0);
            MDT_SCHED_PIN
            if (kc + 1 < K16) {
                an[0] = *(const f32x4*)(cur + aoff + 16 * (kc + 1));
                an[1] = *(const f32x4*)(cur + aoff + 16 * STRIDE + 16 * (kc + 1));
            }
            MDT_SCHED_PIN
#pragma unroll
            for (int e = 1; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][kc][e], af[i][e], acc[i][j], 0, 0, 0);
            MDT_SCHED_PIN
            af[0] = an[0];
            af[1] = an[1];
        }
        // ---- epilogue of tile t: lane holds out[32 t + 16 i + lane % 16][ncol[j] .. + 3] ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (GLU == 3) {
                float* up = const_cast<float*>(a.aux) + 2 * oo[i];
#pragma unroll
                for (int j = 0; j < NTW; j += 2) {
                    f32x4 v = acc[i][j] + bias_v[j];
                    const f32x4 g = acc[i][j + 1] + bias_v[j + 1];
#ifndef MDT_DIAG_WS_NOSTORE_U   // timing experiment only (WRONG results): the product without its 642 MB of u stores
                    if (okr[i]) {
                        *(f32x4*)(up + ncol[j]) = v;
                        *(f32x4*)(up + ncol[j + 1]) = g;
                    }
#endif
                    v.x *= glu_silu(g.x); v.y *= glu_silu(g.y); v.z *= glu_silu(g.z); v.w *= glu_silu(g.w);
                    if (okr[i]) *(f32x4*)(a.out + oo[i] + ncol[j]) = v;
                }
            } else if constexpr (GLU == 4) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    f32x4 v = acc[i][j] + bias_v[j];
                    f32x4 g, pv;
                    if constexpr (AUX_EARLY) { g = gt[i][j]; pv = pj[i][j]; }
                    else { pv = ldg4(a.aux + oo[i] + ncol[j]); g = ldg4(a.aux + oo[i] + a.N + ncol[j]); }
                    f32x4 dg;
                    dg.x = v.x * pv.x * glu_silu_grad(g.x); dg.y = v.y * pv.y * glu_silu_grad(g.y);
                    dg.z = v.z * pv.z * glu_silu_grad(g.z); dg.w = v.w * pv.w * glu_silu_grad(g.w);
                    v.x *= glu_silu(g.x); v.y *= glu_silu(g.y); v.z *= glu_silu(g.z); v.w *= glu_silu(g.w);
                    if (okr[i]) {
                        *(f32x4*)(a.out + oo[i] + a.N + ncol[j]) = dg;
                        *(f32x4*)(a.out + oo[i] + ncol[j]) = v;
                    }
                }
            } else if constexpr (GLU == 1) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    f32x4 v = acc[i][j] + bias_v[j];
                    if (a.aux_mode == 2) {
                        const f32x4 u = ldg4(a.aux + oo[i] + ncol[j]);   // rows clamped above: in bounds
                        v.x *= apply_act_grad1(u.x, a.act); v.y *= apply_act_grad1(u.y, a.act);
                        v.z *= apply_act_grad1(u.z, a.act); v.w *= apply_act_grad1(u.w, a.act);
                    } else {
                        if (a.aux_mode == 1 && okr[i]) *(f32x4*)(const_cast<float*>(a.aux) + oo[i] + ncol[j]) = v;
                        v = apply_act(v, a.act);
                    }
                    if (okr[i]) *(f32x4*)(a.out + oo[i] + ncol[j]) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    if (okr[i]) *(f32x4*)(a.out + oo[i] + ncol[j]) = acc[i][j] + bias_v[j];
            }
        }
        if (more) commit(nxt);
        ws_tile_barrier();                                 // tile t + 1 is in LDS; everybody is done reading tile t
    }
}

// ------------------------------------------------------------------------------------------------
// The same body with every operand SPLIT THREE WAYS into bf16 (round 6, second half): x = x1 + x2 + x3 with 8 mantissa bits each,
// and a k32 step = six v_mfma_f32_16x16x32_bf16 products (x1 w1, x1 w2, x2 w1, x1 w3, x2 w2, x3 w1; fp32 accumulation) instead of
// eight v_mfma_f32_16x16x4_f32: fp32's 24 bits of product accuracy (profiles/r06_bf16_split_probe.txt: error 7e-7 against the
// fp32 product's own 1.1e-6 at K = 1536) at 6 x 16 = 96 matrix-pipe clocks per k32 instead of 8 x 32 = 256.
// Why this body and not the row tiles: the split form wants 2.7x the operand bytes per clock; here the weights are split ONCE per
// workgroup into registers (the fp32 fragments of the existing packed image: lane l of fragment kc holds k = 16 kc + 4 (l / 16) + e,
// and the eight bf16 a lane feeds to one MFMA are its quads of fragments 2 kk and 2 kk + 1 -- the MFMA's k order is a free
// permutation as long as both operands agree), and the A tile is split by the threads that stage it, on its way into LDS
// (three 8-byte stores per 16-byte load, in the slot order the MFMA reads: one ds_read_b128 per part, row tile and k32 step).
// K16 even, ONE column tile per wave, 8 waves.  LDS: 2 buffers x 3 parts x 32 rows x (2 K + 32) bytes (150 KB at K = 384).
// NOT bit-identical to the fp32 bodies (other products, other order): the tests hold it to float64 with the fp32 bodies' tolerance.
// (mdt_bf16x8 / mdt_bf16x4, split3_bf16, split_slot: mdt_device.h)
template <int K16, int NTW, int NWAVES, int GLU>
__device__ __forceinline__ void gemm_ws_split_tile(const mdt_gemm_args& a, int panel, int chunk, int tiles, char* lds,
                                                   const float* __restrict__ zeros, int tid) {
    static_assert(K16 % 2 == 0, "a k32 step is two fragments of the fp32 image");
    static_assert(K16 * NTW == 24 || (K16 == 12 && NTW == 1 && NWAVES == 12), "144 registers of weights: K = 384 with one column tile per wave, or K = 192 with two (72 with one: twelve waves)");
    static_assert(GLU != 3 || NTW % 2 == 0, "SwishGLU forward pairs tiles");
    static_assert(GLU != 1 || NTW == 1, "the activation hooks are written for one column tile");
    constexpr int K = K16 * 16, K32 = K16 / 2, K4 = K / 4, NT = 64 * NWAVES;
    constexpr int ROWB = 2 * K + 32, PART = 32 * ROWB, TILEB = 3 * PART;       // bytes; + 32: ds_read_b128's 16-lane groups hit 64 banks
    constexpr int NLD = (32 * K4 + NT - 1) / NT;
    static_assert(NLD <= K32, "one staged item per k32 step");
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, m16 = lane & 15;
    const int nt0 = (panel * NWAVES + wave) * NTW;
    const int ntiles = (a.M + 31) >> 5;
    const int t0 = chunk * tiles, t1 = min(t0 + tiles, ntiles);
    if (t0 >= t1) return;

    const int nq = 4 * g;
    const float* biasp = a.bias != nullptr ? a.bias : zeros;
    int ncol[NTW];
    f32x4 bias_v[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int T = nt0 + j;
        // (SwishGLU forward: the image interleaves the projected / gate halves tile by tile; bias and output columns are the natural ones)
        ncol[j] = (GLU == 3 ? (T & 1) * (a.N >> 1) + (T >> 1) * 16 : T * 16) + nq;
        bias_v[j] = ldg4(biasp + ncol[j]);
    }

    // ---- A tile t.  Eight waves: wave w stages rows 4 w .. 4 w + 3 (4 x K4 float4 items, NLD = K4 / 16 per lane): item u of lane l is
    //      element j = l + 64 u of those rows, row j / K4, float4 column j % K4.  With K4 = 96, items u and u + 3 are the same column two
    //      rows apart (K4 = 48 has three items), so three (global, LDS) offset pairs per lane + constants address all of them -- no
    //      index arithmetic between the MFMAs.  Other wave counts (twelve: the 192-column panels of N = 192 / 576): item u of thread i is
    //      element i + NT u of the tile, one offset pair per item ----
    constexpr bool WAVE_LOCAL = NWAVES == 8;
    static_assert(!WAVE_LOCAL || (K4 == 96 && NLD == 6) || (K4 == 48 && NLD == 3), "the wave-local staging pattern is written for K = 384 / 192");
    static_assert(WAVE_LOCAL || 32 * K4 % NT == 0, "the tile's items divide over the threads");
    constexpr int NP = WAVE_LOCAL ? 3 : NLD;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int goff[NP], loff[NP];                                // floats from the first row (of the wave / of the tile), bytes from the buffer
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int j = WAVE_LOCAL ? lane + 64 * u : tid + NT * u, r = j / K4, c = 4 * (j - r * K4);
        goff[u] = r * (int)a.lda + c;
        // the four values at columns c .. c + 3 of a row are slots (half, 0 .. 3) = ((c % 32) / 16, ..) of lane group (c % 16) / 4 in
        // k32 step c / 32: eight bytes per part at  row * ROWB + (c / 32) * 64 + ((c % 16) / 4) * 16 + ((c % 32) / 16) * 8
        loff[u] = ((WAVE_LOCAL ? 4 * wave_u : 0) + r) * ROWB + (c >> 5) * 64 + ((c & 15) >> 2) * 16 + ((c & 31) >> 4) * 8;
    }
    f32x4 stage[NLD];
    auto request = [&](int t) __attribute__((always_inline)) {
        const int r0 = 32 * t + (WAVE_LOCAL ? 4 * wave_u : 0), rows = WAVE_LOCAL ? 4 : 32;   // wave-uniform
        if (r0 + rows <= a.M) {
            const float* pt = a.A + (int64_t)r0 * a.lda;
#pragma unroll
            for (int u = 0; u < NLD; ++u) stage[u] = WAVE_LOCAL ? ldg4(pt + (u / 3) * 2 * a.lda + goff[u % 3]) : ldg4(pt + goff[u % NP]);
        } else {                                           // the ragged end: rows past M re-read the last row (never stored)
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int j = WAVE_LOCAL ? lane + 64 * u : tid + NT * u, r = j / K4, c = 4 * (j - r * K4);
                stage[u] = ldg4(a.A + (int64_t)min(r0 + r, a.M - 1) * a.lda + c);
            }
        }
    };
    auto commit = [&](char* buf, int u) __attribute__((always_inline)) {
        mdt_bf16x4 p1, p2, p3;
        split3_bf16(stage[u], p1, p2, p3);
        char* q = WAVE_LOCAL ? buf + loff[u % 3] + (u / 3) * 2 * ROWB : buf + loff[u % NP];
        *(mdt_bf16x4*)q = p1;
        *(mdt_bf16x4*)(q + PART) = p2;
        *(mdt_bf16x4*)(q + 2 * PART) = p3;
    };
    request(t0);                                           // travels under the weights' split
    // ---- this wave's weights, once: NTW x K16 fp32 fragments -> three bf16 parts of NTW x K32 eight-value operands ----
    mdt_bf16x8 w1[NTW][K32], w2[NTW][K32], w3[NTW][K32];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const float* wp = a.Wp + ((int64_t)(nt0 + j) * K16 * 64 + lane) * 4;
#pragma unroll
        for (int kk = 0; kk < K32; ++kk) {
            const f32x4 lo = ldg4(wp + (int64_t)(2 * kk) * 256), hi = ldg4(wp + (int64_t)(2 * kk + 1) * 256);
            mdt_bf16x4 l1, l2, l3, h1, h2, h3;
            split3_bf16(lo, l1, l2, l3);
            split3_bf16(hi, h1, h2, h3);
            w1[j][kk] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            w2[j][kk] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
            w3[j][kk] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) commit(lds, u);
    __syncthreads();

    const int aoff = m16 * ROWB + g * 16;
    for (int t = t0; t < t1; ++t) {
        const char* cur = lds + ((t - t0) & 1) * TILEB;
        char* nxt = lds + ((t - t0 + 1) & 1) * TILEB;
        const bool more = t + 1 < t1;
        if (more) request(t + 1);                          // travels under this tile's MFMAs
        int64_t oo[2];
        bool okr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mr = 32 * t + 16 * i + m16;
            okr[i] = mr < a.M;
            oo[i] = (int64_t)min(mr, a.M - 1) * a.ldo;
        }
        f32x4 acc[2][NTW];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = zero4;
        // the epilogue's second operands (backward hooks), asked for inside the last k32 step: the registers of the next step's
        // x3 / x2 operands are free from their last products on (earlier the body spills), and the epilogue does not start with a
        // memory round trip.  GLU == 1: act'(aux); GLU == 4: projected value and gate of the SwishGLU forward, row tile 0.
        constexpr int NAUX = GLU == 1 ? 2 : (GLU == 4 ? 2 * NTW : 1);
        f32x4 auxv[NAUX];
#pragma unroll
        for (int q = 0; q < NAUX; ++q) auxv[q] = zero4;
        mdt_bf16x8 x1[2], x2[2], x3[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* p = cur + aoff + i * 16 * ROWB;
            x1[i] = *(const mdt_bf16x8*)p; x2[i] = *(const mdt_bf16x8*)(p + PART); x3[i] = *(const mdt_bf16x8*)(p + 2 * PART);
        }
#pragma unroll
        for (int kk = 0; kk < K32; ++kk) {
            // the six products of this k32 step, smallest first, row and column tiles alternating (a dependent MFMA sits at least two
            // issue slots behind its producer); each part of the next step is read into the registers of this step's right behind its
            // last use, so no second operand set is live (256 registers hold the 144 of the weights, the staged tile and this)
            const bool nx = kk + 1 < K32;
            const char* p0 = cur + aoff + (kk + 1) * 64;
            const char* p1 = p0 + 16 * ROWB;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j][kk], x3[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j][kk], x3[1], acc[1][j], 0, 0, 0);
            }
            MDT_SCHED_PIN
            if (nx) { x3[0] = *(const mdt_bf16x8*)(p0 + 2 * PART); x3[1] = *(const mdt_bf16x8*)(p1 + 2 * PART); }
            if (kk == K32 - 1) {
                if constexpr (GLU == 1) {
                    if (a.aux_mode == 2) {
                        auxv[0] = ldg4(a.aux + oo[0] + ncol[0]);       // rows clamped above: in bounds
                        auxv[1] = ldg4(a.aux + oo[1] + ncol[0]);
                    }
                } else if constexpr (GLU == 4) {
                    auxv[0] = ldg4(a.aux + oo[0] + ncol[0]);
                    auxv[1] = ldg4(a.aux + oo[0] + a.N + ncol[0]);
                }
            }
            MDT_SCHED_PIN
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[j][kk], x2[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[j][kk], x2[1], acc[1][j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j][kk], x2[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j][kk], x2[1], acc[1][j], 0, 0, 0);
            }
            MDT_SCHED_PIN
            if (nx) { x2[0] = *(const mdt_bf16x8*)(p0 + PART); x2[1] = *(const mdt_bf16x8*)(p1 + PART); }
            if constexpr (GLU == 4 && NTW > 1) {
                if (kk == K32 - 1) {
                    auxv[2] = ldg4(a.aux + oo[0] + ncol[1]);
                    auxv[3] = ldg4(a.aux + oo[0] + a.N + ncol[1]);
                }
            }
            MDT_SCHED_PIN
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[j][kk], x1[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[j][kk], x1[1], acc[1][j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[j][kk], x1[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[j][kk], x1[1], acc[1][j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j][kk], x1[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j][kk], x1[1], acc[1][j], 0, 0, 0);
            }
            MDT_SCHED_PIN
            if (nx) { x1[0] = *(const mdt_bf16x8*)p0; x1[1] = *(const mdt_bf16x8*)p1; }
            // one staged item of tile t + 1 per step of the last ones: its split and its three LDS stores issue between this tile's
            // MFMAs instead of in a phase of their own (all eight waves would sit in that phase together, matrix pipe idle)
            if (more && kk >= K32 - NLD) commit(nxt, kk - (K32 - NLD));
            MDT_SCHED_PIN
        }
        // ---- epilogue of tile t: lane holds out[32 t + 16 i + lane % 16][ncol[j] .. + 3] ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (GLU == 3) {
                float* up = const_cast<float*>(a.aux) + 2 * oo[i];
#pragma unroll
                for (int j = 0; j < NTW; j += 2) {
                    f32x4 v = acc[i][j] + bias_v[j];
                    const f32x4 gv = acc[i][j + 1] + bias_v[j + 1];
                    if (okr[i]) {
                        *(f32x4*)(up + ncol[j]) = v;
                        *(f32x4*)(up + ncol[j + 1]) = gv;
                    }
                    v.x *= glu_silu(gv.x); v.y *= glu_silu(gv.y); v.z *= glu_silu(gv.z); v.w *= glu_silu(gv.w);
                    if (okr[i]) *(f32x4*)(a.out + oo[i] + ncol[j]) = v;
                }
            } else if constexpr (GLU == 4) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    f32x4 v = acc[i][j] + bias_v[j];
                    f32x4 gv, pv;
                    if (i == 0) { pv = auxv[2 * j]; gv = auxv[2 * j + 1]; }
                    else { pv = ldg4(a.aux + oo[i] + ncol[j]); gv = ldg4(a.aux + oo[i] + a.N + ncol[j]); }
                    f32x4 dg;
                    dg.x = v.x * pv.x * glu_silu_grad(gv.x); dg.y = v.y * pv.y * glu_silu_grad(gv.y);
                    dg.z = v.z * pv.z * glu_silu_grad(gv.z); dg.w = v.w * pv.w * glu_silu_grad(gv.w);
                    v.x *= glu_silu(gv.x); v.y *= glu_silu(gv.y); v.z *= glu_silu(gv.z); v.w *= glu_silu(gv.w);
                    if (okr[i]) {
                        *(f32x4*)(a.out + oo[i] + a.N + ncol[j]) = dg;
                        *(f32x4*)(a.out + oo[i] + ncol[j]) = v;
                    }
                }
            } else if constexpr (GLU == 1) {
                f32x4 v = acc[i][0] + bias_v[0];
                if (a.aux_mode == 2) {
                    const f32x4 u = auxv[i];
                    v.x *= apply_act_grad1(u.x, a.act); v.y *= apply_act_grad1(u.y, a.act);
                    v.z *= apply_act_grad1(u.z, a.act); v.w *= apply_act_grad1(u.w, a.act);
                } else {
                    if (a.aux_mode == 1 && okr[i]) *(f32x4*)(const_cast<float*>(a.aux) + oo[i] + ncol[0]) = v;
                    v = apply_act(v, a.act);
                }
                if (okr[i]) *(f32x4*)(a.out + oo[i] + ncol[0]) = v;
            } else {
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    if (okr[i]) *(f32x4*)(a.out + oo[i] + ncol[j]) = acc[i][j] + bias_v[j];
            }
        }
        ws_tile_barrier();                                 // tile t + 1 is in LDS; everybody is done reading tile t
    }
}
