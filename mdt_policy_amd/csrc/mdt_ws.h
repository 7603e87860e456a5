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
        __syncthreads();                                   // tile t + 1 is in LDS; everybody is done reading tile t
    }
}
