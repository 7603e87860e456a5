// mdt_tall.h -- the TALL GEMM body (round 4): 128-row tiles, BOTH operands staged through LDS by LDS-DMA
// (global_load_lds_dwordx4), every staged element reused by 2-4 waves.
//
// Why a second body (tools/micro/wstream_probe.hip, profiles/r04_probes.txt (1)): gemm_tile (mdt_tiles.h) hands every wave
// its own weight fragments straight from L2 into VGPRs -- ideal for ONE 32-row tile per CU (the B = 256 sampler: 80 row tiles),
// where no operand is shared between waves anyway -- but it pulls 128 B through the CU's vector-memory path per MFMA, and a
// wave that issues those requests is a wave that is not issuing MFMAs.  Where there are THOUSANDS of tiles (the training step:
// M = 10 240 ... 104 448 rows) a 128 x 128 tile moves 4x fewer bytes per MFMA, no MFMA wave ever waits on `vmcnt` for an
// operand register (the DMA lands in LDS; the only waits are one counted `vmcnt` + one barrier per 32-deep K block), and two
// workgroups per CU cover each other's epilogues.
//
//   out = epilogue(A @ W^T), plain prologue (no LayerNorm: those products keep gemm_tile), K % 32 == 0, N % 16 == 0.
//   tile: 128 rows x (WN * NT * 16) columns, K in blocks of 32; WM x WN waves, wave = 64 x (NT * 16) (MT = 4 row tiles).
//   same transposed MFMA form as gemm_tile: D[n][m] = sum_k W[n][k] X[m][k] -- the lane ends with 4 consecutive output
//   columns of one row, epilogue operands / stores are 16 bytes wide.
//
// LDS image of one stage (BK = 32):
//   B: the tile's column fragments exactly as k_pack_weight stores them -- fragment (column tile t, k16 step s) is 1 KiB,
//      lane l's 16 bytes at l * 16: ONE global_load_lds_dwordx4 per fragment (the DMA writes lane-linear, which IS the
//      fragment order), read back with one conflict-free ds_read_b128 per lane.
//   A: [128 rows][8 chunks of 16 B]; a DMA instruction moves 8 rows x 128 B (full cache lines: lane i -> row i / 8, LDS chunk
//      i % 8).  A fragment read takes row l % 16, chunk 4 s + l / 16 of 16 different rows: at a plain 128-byte row stride
//      the 16 lanes one ds_read_b128 cycle serves would hit 2 of the 16 bank groups.  The image is therefore XOR-swizzled --
//      LDS chunk p of row r holds GLOBAL chunk p ^ ((r >> 1) & 7) -- and since the DMA cannot scatter its LDS side, the
//      swizzle is applied to the per-lane GLOBAL address (same 128-byte line: coalescing is untouched).  With it the lane
//      groups {0-3, 12-15, 20-27}, ... of a ds_read_b128 fall into 16 different bank groups.
// Pipeline: NS stages; stage t + NS - 1 is requested while stage t is multiplied; a wave waits for ITS OWN requests of
// stage t with a counted `s_waitcnt vmcnt`, then one raw s_barrier makes everybody's part visible (and says that stage
// t - 1 has been read by all: its buffer is the one the new requests overwrite).  Never __syncthreads() here: its fence
// drains the DMA queue (vmcnt(0)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "mdt_device.h"
#include "mdt_internal.h"
#include "mdt_tiles.h"   // MDT_TS stamps (tuning builds), xcd_remap

#define MDT_TALL_BK 32

typedef __attribute__((address_space(3))) void mdt_lds_void;
typedef __attribute__((address_space(1))) const void mdt_glb_cvoid;

// 64 lanes x 16 bytes: global (per lane: resource base + 32-bit byte offset) -> LDS (wave-uniform base + lane * 16).  The
// BUFFER form: a global_load_lds sends 64 x 8 bytes of address through the SIMD's register read path (tools/micro/
// wstream_probe.hip: 40.7 cycles of matrix-pipe time per global_load_lds_dwordx4; the VGPR loads went 33 -> 17.7 with 32-bit
// offsets).  Operand images stay below 4 GiB from their base (checked by mdt_gemm_tall_supported).
__device__ __forceinline__ void tall_dma16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (mdt_lds_void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void tall_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WM x WN compute waves; NT column tiles per wave; NS pipeline stages; RES: out = out + gate * value; LW = 1: one extra LOADER
// wave issues every DMA request of the workgroup (a DMA costs the issuing wave 60-180 cycles in which it issues no MFMA: with
// the requests spread over the compute waves a 128 x 128 tile alone on a CU ran at 58 % of the pipe rate), LW = 0: each compute
// wave requests its share of every stage itself
template <int WM, int WN, int NT, int NS, bool RES, int LW>
__device__ __forceinline__ void gemm_tall_tile(const mdt_gemm_args& a, int by, int bx, float* lds, const float* __restrict__ zeros,
                                               int tid) {
    constexpr int MT = 4, BM = WM * MT * 16, BN = WN * NT * 16, BK = MDT_TALL_BK, NWAVES = WM * WN;
    constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
    constexpr int A_DMA = BM / 8, B_DMA = (BN / 16) * 2;          // 1-KiB DMA instructions per stage
    constexpr int NLOAD = LW ? 1 : NWAVES;                        // waves that issue them
    constexpr int DMA_PER_WAVE = (A_DMA + B_DMA) / NLOAD;
    static_assert((A_DMA + B_DMA) % NLOAD == 0, "stage does not split evenly over the loading waves");
    static_assert((NS - 2) * DMA_PER_WAVE <= 63, "vmcnt is a 6-bit counter");
    static_assert(BM == 128, "the A image is written for 128-row tiles");
    MDT_TS(0)
    MDT_TS_HWID()
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const bool loader = LW ? wave == NWAVES : true, compute = wave < NWAVES;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int m0 = by * BM, n0t = bx * (BN / 16);                // first row / first column TILE of the workgroup
    const int N16 = a.N >> 4, K16 = a.K >> 4, KT = a.K / BK;

    // ---- DMA assignments of a loading wave (the same for every stage): instruction q = lw + NLOAD * u ----
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.Wp, 0, 0xffffffffu, 0x00020000);
    unsigned dsrc[DMA_PER_WAVE];       // per-lane byte offset (from a.A / a.Wp) of the next stage to request
    int ddst[DMA_PER_WAVE];            // wave-uniform LDS float offset inside a stage
    int dstep[DMA_PER_WAVE];           // bytes per stage
    bool disA[DMA_PER_WAVE];
    if (loader) {
        const int lw = LW ? 0 : wave;
#pragma unroll
        for (int u = 0; u < DMA_PER_WAVE; ++u) {
            const int q = lw + NLOAD * u;
            if (q < A_DMA) {               // rows 8 q .. 8 q + 7, the stage's 128 bytes of each
                const int r = 8 * q + (lane >> 3), p = lane & 7, c = p ^ ((r >> 1) & 7);
                const int64_t m = min(m0 + r, a.M - 1);   // rows past M re-read the last row (masked in the epilogue)
                dsrc[u] = (unsigned)((m * a.lda + 4 * c) << 2);
                ddst[u] = 256 * q;
                dstep[u] = BK * 4;
                disA[u] = true;
            } else {                       // fragment (column tile t, k16 step s) of the packed weight image
                const int f = q - A_DMA, t = f >> 1, s = f & 1;
                const int nt = min(n0t + t, N16 - 1);     // a partial last tile re-reads a valid fragment
                dsrc[u] = (unsigned)((((int64_t)nt * K16 + s) * 256 + lane * 4) << 2);
                ddst[u] = A_FLOATS + 256 * f;
                dstep[u] = 2 * 1024;
                disA[u] = false;
            }
        }
    }
    auto request = [&](int kt) {       // the next stage (kt) -> buffer kt % NS
        float* base = lds + (kt % NS) * STAGE;
#pragma unroll
        for (int u = 0; u < DMA_PER_WAVE; ++u) {
            tall_dma16(disA[u] ? rsA : rsW, dsrc[u], base + ddst[u]);
            dsrc[u] += dstep[u];
        }
    };
    // ---- fill the pipeline ----
    if (loader) {
#pragma unroll
        for (int t = 0; t < NS - 1; ++t)
            if (t < KT) request(t);
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    if (LW && !compute) {
        // ---- the loader wave: stage kt landed -> barrier kt (says so, and learns that stage kt - 1 has been read) -> request
        //      stage kt + NS - 1 into the buffer stage kt - 1 lived in ----
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + NS - 1 <= KT) tall_wait_vm<(NS - 2) * DMA_PER_WAVE>();
            else tall_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < KT) request(kt + NS - 1);
        }
        return;
    }

    // fragment addresses inside a stage (floats)
    int aoff[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int s = 0; s < 2; ++s) aoff[i][s] = r * BK + 4 * ((4 * s + (lane >> 4)) ^ ((r >> 1) & 7));
    }
    const int boff = A_FLOATS + (wn * NT * 2) * 256 + lane * 4;   // + (2 j + s) * 256

#if defined(MDT_DEBUG_TIMING) && defined(MDT_TILES_TIMING_OWNER)
    unsigned long long t_sync = 0;   // cycles between arriving at a stage's wait and leaving its barrier
#endif
    for (int kt = 0; kt < KT; ++kt) {
#if defined(MDT_DEBUG_TIMING) && defined(MDT_TILES_TIMING_OWNER)
        const unsigned long long ts_a = __builtin_readcyclecounter();
#endif
        if constexpr (!LW) {
            // my requests of stage kt have landed (at most NS - 2 younger stages of mine stay in flight) ...
            if (kt + NS - 1 <= KT) tall_wait_vm<(NS - 2) * DMA_PER_WAVE>();
            else tall_wait_vm<0>();   // the tail: fewer stages behind this one than the pipeline holds
        }
        // ... and so have everybody else's; stage kt - 1 has been read by all
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if defined(MDT_DEBUG_TIMING) && defined(MDT_TILES_TIMING_OWNER)
        t_sync += __builtin_readcyclecounter() - ts_a;
        if (kt == 0) { MDT_TS(1) }
#endif
        if constexpr (!LW) {
            if (kt + NS - 1 < KT) request(kt + NS - 1);
        }
        const float* st = lds + (kt % NS) * STAGE;
        f32x4 af[2][MT], bf[2][NT];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < MT; ++i) af[s][i] = *(const f32x4*)(st + aoff[i][s]);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[s][j] = *(const f32x4*)(st + boff + (2 * j + s) * 256);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[s][j][e], af[s][i][e], acc[i][j], 0, 0, 0);
    }

    MDT_TS(3)
#if defined(MDT_DEBUG_TIMING) && defined(MDT_TILES_TIMING_OWNER)
    if (threadIdx.x == 0 && g_dbg_ts != nullptr) g_dbg_ts[(size_t)blockIdx.x * 8 + 5] = t_sync;
#endif
    // ---- epilogue: lane holds out[m0 + wm*64 + i*16 + lane%16][(n0t + wn*NT + j)*16 + 4*(lane/16) .. +3] ----
    // The common cases are their own straight-line instantiations (ACT / AUX fixed, FULL: no row or column of the tile is
    // outside the matrix, so no store carries a predicate).  With everything decided per value at run time -- activation
    // switch, training hooks, `ok` masks as exec-mask branches -- the epilogue was ~140 instructions per 16 x 16 tile: 9.3 k
    // cycles for a 128 x 128 tile alone on its CU against 49 k of MFMA issue (profiles/r04_gemm_train_shapes.txt, "Where the time of a 128 x 128 tile goes").
    const int nq = 4 * (lane >> 4);
    const float* biasp = a.bias != nullptr ? a.bias : zeros;
    const float* rvp = a.rowvec != nullptr ? a.rowvec : zeros;
    const bool gated = RES && a.gate_off >= 0;
    auto epilogue = [&](auto ACT_, auto AUX_, auto FULL_) {
        constexpr int ACT = decltype(ACT_)::value, AUX = decltype(AUX_)::value;   // -1: read a.act / a.aux_mode at run time
        constexpr bool FULL = decltype(FULL_)::value;
        const int act = ACT >= 0 ? ACT : a.act, aux_mode = AUX >= 0 ? AUX : a.aux_mode;
        int ncol[NT];
        f32x4 bias_v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int T = FULL ? n0t + wn * NT + j : min(n0t + wn * NT + j, N16 - 1);
            ncol[j] = T * 16 + nq;
            bias_v[j] = ldg4(biasp + ncol[j]) + ldg4(rvp + ncol[j]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mr = m0 + wm * 64 + i * 16 + (lane & 15);
            const int m = FULL ? mr : min(mr, a.M - 1);
            const int64_t orow = a.gin == 1 ? (int64_t)m * a.gout + a.goff : (int64_t)(m / a.gin) * a.gout + (m % a.gin) + a.goff;
            float* orp = a.out + orow * a.ldo;
            const float* axp = a.aux + orow * a.ldo;
            f32x4 gate_v[NT], res_v[NT], aux_v[NT];
            if constexpr (RES) {
                const float* gp = gated ? a.mod + a.gate_off + (a.mod_stride == 0 ? 0 : (int64_t)(m / a.rows_per_sample) * a.mod_stride)
                                        : zeros;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    gate_v[j] = ldg4(gp + ncol[j]);
                    res_v[j] = ldg4(orp + ncol[j]);
                }
            } else if (aux_mode == 2) {
#pragma unroll
                for (int j = 0; j < NT; ++j) aux_v[j] = ldg4(axp + ncol[j]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bool ok = FULL || (mr < a.M && n0t + wn * NT + j < N16);
                f32x4 v = acc[i][j] + bias_v[j];
                if constexpr (RES) {
                    v = apply_act(v, act);
                    v = res_v[j] + (gated ? gate_v[j] * v : v);
                } else if (aux_mode == 2) {
                    const f32x4 u = aux_v[j];
                    v.x *= apply_act_grad1(u.x, act); v.y *= apply_act_grad1(u.y, act);
                    v.z *= apply_act_grad1(u.z, act); v.w *= apply_act_grad1(u.w, act);
                } else {
                    if (aux_mode == 1 && ok) *(f32x4*)(const_cast<float*>(axp) + ncol[j]) = v;
                    v = apply_act(v, act);
                }
                if (ok) *(f32x4*)(orp + ncol[j]) = v;
            }
        }
    };
    using std::integral_constant;
    const bool full = m0 + BM <= a.M && n0t + BN / 16 <= N16;
    typedef integral_constant<bool, true> T_;
    typedef integral_constant<bool, false> F_;
#define MDT_TALL_EPI(ACT, AUX)                                                \
    {                                                                         \
        if (full) epilogue(integral_constant<int, ACT>(), integral_constant<int, AUX>(), T_()); \
        else epilogue(integral_constant<int, ACT>(), integral_constant<int, AUX>(), F_());      \
    }
    if constexpr (RES) {
        if (a.act == MDT_ACT_NONE) MDT_TALL_EPI(MDT_ACT_NONE, 0)
        else epilogue(integral_constant<int, -1>(), integral_constant<int, 0>(), F_());
    } else {
        if (a.aux_mode == 0 && a.act == MDT_ACT_NONE) MDT_TALL_EPI(MDT_ACT_NONE, 0)
        else if (a.aux_mode == 0 && a.act == MDT_ACT_GELU) MDT_TALL_EPI(MDT_ACT_GELU, 0)
        else if (a.aux_mode == 1 && a.act == MDT_ACT_GELU) MDT_TALL_EPI(MDT_ACT_GELU, 1)
        else if (a.aux_mode == 2 && a.act == MDT_ACT_GELU) MDT_TALL_EPI(MDT_ACT_GELU, 2)
        else epilogue(integral_constant<int, -1>(), integral_constant<int, -1>(), F_());
    }
#undef MDT_TALL_EPI
    MDT_TS(4)
}
