// mdt_tiles.h -- the per-workgroup TILE BODIES of the decoder's kernels; mdt_kernels.hip wraps each in a one-launch-per-
// operation kernel (grid = tiles).
// A body is written once; the template flag COH selects how ACTIVATIONS are read:
//   COH = false : plain global loads (between launches the kernel boundary makes everything visible) -- every shipped kernel
//   COH = true  : `buffer_load_dwordx4 ... sc1` -- bypasses the reading CU's vector L1, which is never refreshed by
//                 another CU's stores; served by the XCD's L2, where the producer's plain stores live once its
//                 `s_waitcnt vmcnt(0)` has retired (MI355X_MICROARCH.md "inter-workgroup visibility";
//                 tools/micro/persist_probe.hip: 0 stale words in 1500 phases x 32 peers x 16 KiB under uneven load).
//                 What a body needs when it reads rows another workgroup of the SAME launch wrote.  The persistent decoder
//                 kernel of rounds 2-4 (one launch per sampler call, per-XCD barriers between phases) was its one user; it
//                 never beat the launch sequence (profiles/r02_persist_*, docs/history/) and was removed in round 6.
// Weights, LayerNorm vectors and the modulation table are written before the launch and always read with plain loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdt_internal.h"
#include "mdt_device.h"

// XCD-aware block id: the dispatcher places block b on XCD b % 8 (speed-only assumption); give every XCD a
// contiguous range of logical tiles so the row tiles it touches stay in its private L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
#ifdef MDT_NO_XCD_REMAP  // A/B build: dispatch order = logical order (a row tile's column tiles land on different XCDs)
    return bid;
#endif
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// the per-sample folded cross-attention matrices (98 KB per sample and block) are read once per launch by one workgroup,
// but again at each of the 10 steps: plain loads (measured: non-temporal loads, -DMDT_NT_STREAM, cost 1.5 % at B = 256)
#ifdef MDT_NT_STREAM
#define MDT_LD_STREAM(p) __builtin_nontemporal_load((const f32x4*)(p))
#else
#define MDT_LD_STREAM(p) ldg4(p)
#endif

#ifdef MDT_NO_SAMPLE_REMAP  // A/B build: per-sample kernels keep sample b on XCD b % 8
#define MDT_SAMPLE_REMAP(bid, n) (bid)
#else
#define MDT_SAMPLE_REMAP(bid, n) xcd_remap(bid, n)
#endif

// ---- activation loader ----
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <bool COH>
struct ActLd;
template <>
struct ActLd<false> {
    const float* base;
    __device__ __forceinline__ explicit ActLd(const float* p) : base(p) {}
    __device__ __forceinline__ f32x4 ld4(int64_t off) const { return ldg4(base + off); }
    __device__ __forceinline__ float ld1(int64_t off) const { return base[off]; }
};
template <>
struct ActLd<true> {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ explicit ActLd(const float* p)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xffffffffu, 0x00020000)) {}
    // 16 bytes at float offset `off`; off * 4 must stay below 4 GiB
    __device__ __forceinline__ f32x4 ld4(int64_t off) const {
        const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(off << 2), 0, 16 /* sc1 */);
        return __builtin_bit_cast(f32x4, r);
    }
    __device__ __forceinline__ float ld1(int64_t off) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(off << 2), 0, 16));
    }
};

// ---- weight-fragment stream ----
// A wave's handle on one column tile of a packed weight image: fragment k16 sits 256 floats behind fragment k16 - 1, lane l's
// 16 bytes at l * 4 floats.  Fetched with BUFFER loads (resource = the image, 32-bit per-lane byte offset): a global_load
// sends 64 lanes x 8 bytes of address through the SIMD's register read path before its 1 KiB of data comes back, and while
// either moves the SIMD issues no MFMA -- tools/micro/wstream_probe.hip: 33 cycles of matrix-pipe time per global_load_dwordx4
// against 17.7 per buffer_load_dwordx4 ... offen (17.3 with no address register at all: what is left is the data's way into the
// registers), i.e. 113 -> 101 us for the k-loop of the fused MLP's first product alone.  (Round 3 measured buffer loads 7 %
// SLOWER in mlp_tile: that form kept the tile base in the scalar offset behind a v_readfirstlane per request.)
// -DMDT_W_GLOBAL: A/B build with the 64-bit global loads.  Offsets are bytes: an image (and every batched slice of one,
// mdt_gemm_args.bs_w) stays below 4 GiB.
#ifdef MDT_W_GLOBAL
struct WStream {
    const float* p;
    __device__ __forceinline__ WStream operator+(int floats) const { return WStream{p + floats}; }
};
__device__ __forceinline__ WStream wstream(const float* image, int64_t float_off) { return WStream{image + float_off}; }
__device__ __forceinline__ f32x4 wld4(const WStream& w) { return ldg4(w.p); }
#else
struct WStream {
    __amdgpu_buffer_rsrc_t rs;
    unsigned off;  // bytes
    __device__ __forceinline__ WStream operator+(int floats) const { return WStream{rs, off + 4u * (unsigned)floats}; }
};
__device__ __forceinline__ WStream wstream(const float* image, int64_t float_off) {
    return WStream{__builtin_amdgcn_make_buffer_rsrc((void*)image, 0, 0xffffffffu, 0x00020000), (unsigned)(float_off << 2)};
}
__device__ __forceinline__ f32x4 wld4(const WStream& w) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.off, 0, 0));
}
#endif

// prologue kinds: plain copy | LayerNorm | LayerNorm + modulate with ONE broadcast row (sampler: one sigma per
// step) | LayerNorm + modulate with a per-sample row (GCDenoiser.forward / loss with per-sample sigma)
enum { PRO_PLAIN = 0, PRO_LN = 1, PRO_LN_MOD_BCAST = 2, PRO_LN_MOD_ROWS = 3,
       PRO_ATTN = 4 /* the activation tile is the causal self-attention output of the tile's rows, computed here (attn_stage_tile) */ };

// what the attention prologue (PRO_ATTN) needs beside the projection's own mdt_gemm_args
struct mdt_attn_pro {
    const float* qkv;   // (M, 3 D) rows: q | k | v column blocks of D = H * hd each
    int64_t ldq;
    int T;              // rows per sample (<= 16); key j visible to query i of the same sample iff j <= i
    float scale;
};

// output store of the epilogues: a WRITE-THROUGH (sc0 sc1) store, so that the tile does not stay dirty in L2 until the
// end-of-kernel write-back (a launch that leaves B dirty bytes pays B / 6 TB/s at its boundary: 2 us behind the 11.8 MB of a
// fused MLP's slabs or of q | k | v; with one tile per CU nothing else hides it).  With three fat launches per block it pays
// 1.9 % per sampler call at B = 256 (round 1, six thin launches: +1 %, inside the noise; B = 1024, the training step and the
// MGF head do not care: several tiles per CU hide the write-back).  -DMDT_ST_PLAIN: A/B build (plain stores).
__device__ __forceinline__ void st4(float* p, f32x4 v) {
#if defined(MDT_ST_PLAIN)
    *(f32x4*)p = v;
#elif defined(MDT_ST_NT)
    __builtin_nontemporal_store(v, (f32x4*)p);
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ f32x4 sel4(bool c, f32x4 a, f32x4 b) { return c ? a : b; }
__device__ __forceinline__ float hsum4(f32x4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float hsq4(f32x4 v) { return (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }

#if defined(MDT_DEBUG_TIMING) && defined(MDT_TILES_TIMING_OWNER)
// tuning-only build (-DMDT_DEBUG_TIMING): thread 0 of every workgroup records shader-clock stamps of its phases
__device__ unsigned long long* g_dbg_ts = nullptr;
#define MDT_TS(i)                                                                         \
    if (threadIdx.x == 0 && g_dbg_ts != nullptr) {                                        \
        g_dbg_ts[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter();            \
    }
// mlp_tile: thread 256 -- wave 4, the SIMD partner of wave 0 -- stamps too, into a second block of rows behind the grid's
#define MDT_TS2(i)                                                                        \
    if ((threadIdx.x & 255) == 0 && g_dbg_ts != nullptr) {                               \
        g_dbg_ts[((size_t)blockIdx.x + (threadIdx.x >> 8) * gridDim.x) * 8 + (i)] = __builtin_readcyclecounter(); \
    }
// a body that runs BEHIND another stamped body in one kernel (k_xattn_gemm_smallm) stamps into the rows behind the grid's
#define MDT_TSO(off, i)                                                                   \
    if (threadIdx.x == 0 && g_dbg_ts != nullptr) {                                        \
        g_dbg_ts[((size_t)blockIdx.x + (off)) * 8 + (i)] = __builtin_readcyclecounter();  \
    }
#define MDT_TS_HWID()                                                                     \
    if (threadIdx.x == 0 && g_dbg_ts != nullptr) {                                        \
        unsigned hw, xcc;                                                                 \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                  \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                \
        g_dbg_ts[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;      \
    }
#else
#define MDT_TS(i)
#define MDT_TS2(i)
#define MDT_TSO(off, i)
#define MDT_TS_HWID()
#endif

// The enclosing kernel says with KSTEP_PRIO whether a k-step's MFMA block runs at raised issue priority: it pays where
// loader waves share the SIMDs with the MFMA waves (k_gemm_pipe: mlp.c_proj 37.6 -> 36.1 us) and costs 1-3 % where
// every wave does both (k_gemm).  -DMDT_NO_KSTEP_PRIO switches it off for A/B runs.
#ifdef MDT_NO_KSTEP_PRIO
#define MDT_PRIO(x)
#else
#define MDT_PRIO(x) if constexpr (KSTEP_PRIO) __builtin_amdgcn_s_setprio(x);
#endif
#ifndef MDT_RING_ADD
#define MDT_RING_ADD 0  // tuning builds: deeper weight-fragment rings (tools/gpu_ring_ab.sh)
#endif
// The prefetches of a k-step are written at its top and must be ISSUED there: left alone, the machine scheduler sinks the
// loads of one ring slot out of three down to their first use (one exposed L2 round trip per three k-steps in every GEMM
// loop of this file).  A scheduling barrier that only scalar / vector ALU instructions may cross pins memory instructions
// and MFMAs to the k-step they were written in.  -DMDT_NO_SCHED_PIN: A/B build.
#ifdef MDT_NO_SCHED_PIN
#define MDT_SCHED_PIN
#else
#define MDT_SCHED_PIN __builtin_amdgcn_sched_barrier(0x6);
#endif
// timing experiments only (WRONG results; tools/gpu_alone.sh, profiles/r03_mlp_alone_probe.txt): k-steps without their weight
// loads (-DMDT_EXP_NOLOAD) / without their LDS reads (-DMDT_EXP_NOLDS)
#ifdef MDT_EXP_NOLOAD
#define MDT_EXP_LDG(p, old) (old)
#else
#define MDT_EXP_LDG(p, old) wld4(p)
#endif
#ifdef MDT_EXP_NOLDS
#define MDT_EXP_LDS(x, old) (old)
#else
#define MDT_EXP_LDS(x, old) (x)
#endif
// one k-step of the MFMA main loop (uses the enclosing kernel's ring / wp / ap / stride / acc / kg / K16): prefetch the fragment R-1 steps ahead (clamped, never branches), then 4 MFMAs per tile pair
#ifndef MDT_KSTEP_SPREAD
#define MDT_KSTEP(U, KC)                                                                                  \
    {                                                                                                     \
        const int kpf = min(kg + (KC) + R - 1, K16 - 1);                                                  \
        _Pragma("unroll") for (int j = 0; j < NTW; ++j) ring[((U) + R - 1) % R][j] =                      \
            MDT_EXP_LDG(wp[j] + kpf * 256, ring[((U) + R - 1) % R][j]);                                   \
        MDT_SCHED_PIN                                                                                     \
        MDT_PRIO(1)                                                                                       \
        MDT_KSTEP_MFMAS(U, 0)                                                                             \
        MDT_SCHED_PIN                                                                                     \
        f32x4 avn[MTILES]; /* activation fragments of the NEXT k-step, requested behind the first quarter of the MFMAs: \
                              whatever LDS wait the compiler puts at the top of a step then finds them long landed */ \
        _Pragma("unroll") for (int i = 0; i < MTILES; ++i) avn[i] =                                       \
            MDT_EXP_LDS(*(const f32x4*)(ap + i * 16 * stride + min((KC) + 1, nk - 1) * 16), av[i]);       \
        MDT_SCHED_PIN                                                                                     \
        MDT_KSTEP_MFMAS(U, 1)                                                                             \
        MDT_KSTEP_MFMAS(U, 2)                                                                             \
        MDT_KSTEP_MFMAS(U, 3)                                                                             \
        MDT_PRIO(0)                                                                                       \
        MDT_SCHED_PIN                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MTILES; ++i) av[i] = avn[i];                                \
    }
#else
/* A/B form: the step's NTW fragment requests are SPREAD over the step, one in front of each quarter of its MFMAs, instead of
   going out back to back at its top (tools/micro/wstream_probe.hip: 8 waves x 4 requests in one burst queue at the CU's one
   address path and keep the issuing waves from their MFMAs: 117 -> 109 us for k_mlp's first product alone) */
#define MDT_KSTEP_LD1(U, J)                                                                               \
    if constexpr ((J) < NTW) {                                                                            \
        ring[((U) + R - 1) % R][(J) < NTW ? (J) : 0] =                                                    \
            MDT_EXP_LDG(wp[(J) < NTW ? (J) : 0] + kpf * 256, ring[((U) + R - 1) % R][(J) < NTW ? (J) : 0]); \
        MDT_SCHED_PIN                                                                                     \
    }
#define MDT_KSTEP(U, KC)                                                                                  \
    {                                                                                                     \
        const int kpf = min(kg + (KC) + R - 1, K16 - 1);                                                  \
        MDT_KSTEP_LD1(U, 0)                                                                               \
        MDT_PRIO(1)                                                                                       \
        MDT_KSTEP_MFMAS(U, 0)                                                                             \
        MDT_SCHED_PIN                                                                                     \
        f32x4 avn[MTILES];                                                                                \
        _Pragma("unroll") for (int i = 0; i < MTILES; ++i) avn[i] =                                       \
            MDT_EXP_LDS(*(const f32x4*)(ap + i * 16 * stride + min((KC) + 1, nk - 1) * 16), av[i]);       \
        MDT_SCHED_PIN                                                                                     \
        MDT_KSTEP_LD1(U, 1)                                                                               \
        MDT_KSTEP_MFMAS(U, 1)                                                                             \
        MDT_SCHED_PIN                                                                                     \
        MDT_KSTEP_LD1(U, 2)                                                                               \
        MDT_KSTEP_MFMAS(U, 2)                                                                             \
        MDT_SCHED_PIN                                                                                     \
        MDT_KSTEP_LD1(U, 3)                                                                               \
        MDT_KSTEP_MFMAS(U, 3)                                                                             \
        MDT_PRIO(0)                                                                                       \
        MDT_SCHED_PIN                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MTILES; ++i) av[i] = avn[i];                                \
    }
#endif
#define MDT_KSTEP_MFMAS(U, E)                                                                             \
    _Pragma("unroll") for (int i = 0; i < MTILES; ++i) {                                                  \
        _Pragma("unroll") for (int j = 0; j < NTW; ++j) acc[i][j] =                                       \
            __builtin_amdgcn_mfma_f32_16x16x4f32(ring[(U)][j][(E)], av[i][(E)], acc[i][j], 0, 0, 0);      \
    }

// Geometry: NWAVES waves (4 or 8); tile = (MTILES*16 rows) x (NWAVES * NTW * 16 columns), full K.  Wave w owns NTW
// column tiles and ALL row tiles of the workgroup tile, so a weight fragment is fetched once per workgroup and
// reused from registers across the row tiles.  Wide tiles (8 waves x NTW 3..4) keep the number of workgroups that
// re-read / re-normalise the same activation rows at N / (128*NTW) instead of N / 64.
// Activation tile -> LDS (shared by the GEMM kernels): plain copy of the (MT x klen) chunk at column k0, or LayerNorm
// (+ adaLN modulate) of whole rows (k0 = 0, klen = K <= 512).  All global loads of the phase are in flight together.
// XP > 1 (LayerNorm prologues only): the rows are the sum of XP slabs a.A + x * a.a_part_stride (the partial outputs of
// the fused MLP launch, mlp_tile below), added in slab order; `merge_out` (the column-0 workgroup of a row tile passes
// a.a_merged, the others nullptr) also receives the summed rows.
// XB (XP > 1): row batches the other slabs are requested in -- 1 where the registers allow it, else 2 (the caller knows its own load)
// SPLIT (round 6, LayerNorm prologues only): the rows are stored as their three-way bf16 split in MFMA slot order (mdt_device.h
// split3_bf16 / split_slot: part q at (char*)lds + q * MT * stride, `stride` = the split tile's row stride in BYTES) instead of fp32
// rows of `stride` floats -- the staged tile of mdt_mlp_split.h without an fp32 copy in LDS and a pass over it.
template <int MTILES, int NWAVES, int PRO, bool COH, int XP = 1, int XB = 2, bool SPLIT = false>
__device__ __forceinline__ void gemm_stage_tile(const mdt_gemm_args& a, float* lds, int stride, int m0, int k0, int klen,
                                                const float* __restrict__ zeros, int tid, int lane, int wave,
                                                float* merge_out = nullptr) {
    static_assert(!SPLIT || PRO != PRO_PLAIN, "the split store belongs to the LayerNorm prologues");
    constexpr int MT = MTILES * 16;
    constexpr int NT = 64 * NWAVES;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n4 = klen >> 2;
    const ActLd<COH> LA(a.A);
    if constexpr (PRO != PRO_PLAIN) {
        // ---- LayerNorm (+ adaLN modulate) prologue: each wave owns a slab of RPW consecutive rows, whole
        //      rows live in registers (K <= 512 -> two float4 per lane); single chunk by construction ----
        constexpr int RPW = MT / NWAVES;
        const int r0 = wave * RPW;
        int cc[2];
        bool cv[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            cv[p] = lane + 64 * p < n4;
            cc[p] = 4 * min(lane + 64 * p, n4 - 1);
        }
        f32x4 v[RPW][2], w[2], bb[2];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int64_t m = min(m0 + r0 + r, a.M - 1);
#pragma unroll
            for (int p = 0; p < 2; ++p) v[r][p] = LA.ld4(m * a.lda + cc[p]);
        }
        if constexpr (XP > 1) {
            // the other slabs, summed in slab order; the column-0 workgroup also leaves the sum in a.a_merged for the residual GEMM
            // that follows.  ALL of a wave's slab rows are requested at once (round 5: 236 VGPRs, no spill).  Round 3 split them into
            // two batches of rows because the kernel spilled then; the second batch was requested only after the first had
            // arrived and been stored -- two serial round trips to the Infinity Cache in the prologue of every qkv launch
            // (B = 256 sampler call 4.397 -> 4.377 ms in alternating runs).  The instantiations that would still spill -- four slabs
            // (d = 512), four column tiles per wave -- keep the two batches (XB = 2).
            constexpr int HB = XB == 1 || RPW < 2 ? RPW : RPW / 2;
#pragma unroll
            for (int r0b = 0; r0b < RPW; r0b += HB) {
                f32x4 vx[XP - 1][HB][2];
#pragma unroll
                for (int x = 1; x < XP; ++x)
#pragma unroll
                    for (int r = 0; r < HB; ++r) {
                        const int64_t m = min(m0 + r0 + r0b + r, a.M - 1);
#pragma unroll
                        for (int p = 0; p < 2; ++p) vx[x - 1][r][p] = LA.ld4((int64_t)x * a.a_part_stride + m * a.lda + cc[p]);
                    }
#pragma unroll
                for (int r = 0; r < HB; ++r)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int x = 1; x < XP; ++x) v[r0b + r][p] = v[r0b + r][p] + vx[x - 1][r][p];
                        if (merge_out != nullptr && cv[p] && m0 + r0 + r0b + r < a.M)
                            st4(merge_out + (int64_t)(m0 + r0 + r0b + r) * a.lda + cc[p], v[r0b + r][p]);
                    }
            }
        }
        const float* lnb = a.ln_b != nullptr ? a.ln_b : zeros;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            w[p] = ldg4(a.ln_w + cc[p]);
            bb[p] = ldg4(lnb + cc[p]);
        }
        // modulation vectors, fetched up front.  BCAST: one row for the whole batch.  ROWS: a slab of RPW
        // consecutive rows touches at most 2 samples when rows_per_sample >= RPW -> two candidates.
        constexpr int NC = PRO == PRO_LN_MOD_ROWS ? 2 : 1;
        f32x4 sh[NC][2], sc[NC][2];
        int s_lo = 0;
        bool slow_mod = false;
        if constexpr (PRO == PRO_LN_MOD_BCAST) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                sh[0][p] = ldg4(a.mod + a.shift_off + cc[p]);
                sc[0][p] = ldg4(a.mod + a.scale_off + cc[p]);
            }
        }
        if constexpr (PRO == PRO_LN_MOD_ROWS) {
            s_lo = min(m0 + r0, a.M - 1) / a.rows_per_sample;
            const int s_hi = min(m0 + r0 + RPW - 1, a.M - 1) / a.rows_per_sample;
            slow_mod = s_hi > s_lo + 1;
            const float* mlo = a.mod + (int64_t)s_lo * a.mod_stride;
            const float* mhi = a.mod + (int64_t)min(s_lo + 1, s_hi) * a.mod_stride;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                sh[0][p] = ldg4(mlo + a.shift_off + cc[p]);
                sc[0][p] = ldg4(mlo + a.scale_off + cc[p]);
                sh[NC - 1][p] = ldg4(mhi + a.shift_off + cc[p]);
                sc[NC - 1][p] = ldg4(mhi + a.scale_off + cc[p]);
            }
        }
        // (Round 5 measured pinning the prologue's requests to the order they are written in -- weight ring and epilogue operands in
        //  front of the rows, the LayerNorm vectors in front of the reductions: 4.405 against 4.378 ms per B = 256 call.  The machine
        //  scheduler's own order -- the rows, whose Infinity-Cache round trip is the longest, FIRST, everything L2-resident behind
        //  them -- is the better one; nothing is pinned here.)
        const float inv_k = 1.0f / (float)klen;
        float red[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            red[r] = 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                v[r][p] = sel4(cv[p], v[r][p], zero4);
                red[r] += hsum4(v[r][p]);
            }
        }
#ifndef MDT_EXP_NOLN   // timing experiment only (WRONG results): the prologue without its two wave reductions
        wave_sum_n<RPW>(red);
#endif
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float mean = red[r] * inv_k;
            red[r] = 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                v[r][p] = sel4(cv[p], v[r][p] - mean, zero4);
                red[r] += hsq4(v[r][p]);
            }
        }
#ifndef MDT_EXP_NOLN
        wave_sum_n<RPW>(red);
#endif
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int m = m0 + r0 + r;
            const float rstd = 1.0f / sqrtf(red[r] * inv_k + 1e-5f);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                f32x4 y = v[r][p] * rstd * w[p] + bb[p];
                if constexpr (PRO == PRO_LN_MOD_BCAST) y = sh[0][p] + y * sc[0][p];
                if constexpr (PRO == PRO_LN_MOD_ROWS) {
                    const int smp = min(m, a.M - 1) / a.rows_per_sample;
                    f32x4 shv = sel4(smp == s_lo, sh[0][p], sh[NC - 1][p]);
                    f32x4 scv = sel4(smp == s_lo, sc[0][p], sc[NC - 1][p]);
                    if (slow_mod) {  // rows_per_sample < RPW: rare generic path, one round trip per row
                        const float* mr = a.mod + (int64_t)smp * a.mod_stride;
                        shv = ldg4(mr + a.shift_off + cc[p]);
                        scv = ldg4(mr + a.scale_off + cc[p]);
                    }
                    y = shv + y * scv;
                }
                y = sel4(m < a.M, y, zero4);
                if constexpr (SPLIT) {
                    if (cv[p]) {
                        mdt_bf16x4 p1, p2, p3;
                        split3_bf16(y, p1, p2, p3);
                        char* q = (char*)lds + split_slot(r0 + r, cc[p], stride);
                        *(mdt_bf16x4*)q = p1;
                        *(mdt_bf16x4*)(q + MT * stride) = p2;
                        *(mdt_bf16x4*)(q + 2 * MT * stride) = p3;
                    }
                } else {
                    if (cv[p]) *(f32x4*)(lds + (r0 + r) * stride + cc[p]) = y;
                }
            }
        }
    } else {
        // ---- plain staging of the (MT x klen) activation chunk: 32 lanes sweep a row in 512-byte pieces,
        //      NT/32 rows at a time; all loads of the chunk are in flight together (no divisions) ----
        constexpr int RG = NT / 32;      // rows covered per sweep
        constexpr int U = MT / RG;       // sweeps
        const int rg = tid >> 5, l32 = tid & 31;
        const int nv = (n4 + 31) >> 5;   // 512-byte pieces per row (<= 6 for kchunk <= 768)
        for (int v0 = 0; v0 < nv; v0 += 3) {
            f32x4 st[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t m = min(m0 + rg + RG * u, a.M - 1);
#pragma unroll
                for (int v = 0; v < 3; ++v)
                    st[u][v] = LA.ld4(m * a.lda + k0 + 4 * min(l32 + 32 * (v0 + v), n4 - 1));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = rg + RG * u;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int c4 = l32 + 32 * (v0 + v);
                    if (c4 < n4) *(f32x4*)(lds + row * stride + 4 * c4) = sel4(m0 + row < a.M, st[u][v], zero4);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Attention prologue of the self-attention output projection (PRO_ATTN; large batches): the activation tile of the
// projection -- softmax_causal(q k^T / sqrt(hd)) v for the tile's 32 rows, all 8 heads -- is computed HERE instead of by a
// launch of its own (k_attn: 8 us of every decoder block at B = 256, plus a kernel boundary and a round trip of the attention
// output).  Causal attention only looks BACK inside a sample, so a 32-row tile needs the q rows of the tile and the k / v rows
// from the first row of its first sample on: at most 32 + T - 1 <= 47 rows.  The 3 column workgroups of a row tile repeat it
// (0.2 % of the FLOPs).  Heads are done in two halves of 4 (LDS: 49.7 KB activation tile + 100 KB of q / k / v half rows);
// all global loads of both halves are requested up front.  Thread = (head of the half, row, quarter of the head dimension);
// the 4 quarters of a dot product meet through DPP quad swaps (no LDS round trip).  512 threads, H = 8.
//   xa: activation tile [32][stride]; scratch: (32 + 2 * 48) * (4 * HD + 16) floats behind it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_get<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    return v;
}
template <int HD, int TKC>
__device__ __forceinline__ void attn_stage_tile(const mdt_gemm_args& a, const mdt_attn_pro& ap, float* xa, int stride, int m0,
                                                float* scratch, int tid) {
    // row stride = 16 (mod 64) floats: the 16 lanes one ds_read_b128 cycle serves -- 4 (row, head) pairs x 4 quarters, 12 floats
    // apart -- then fall into 16 different 16-byte bank groups (with the usual + 4 padding they collide 2-3 ways and the
    // attention, all LDS reads, took 3.5 us per half instead of ~1.5)
    constexpr int HH = 4, HHD = HH * HD, ST = HHD + 16, H4 = HHD / 4, KVR = 48, DS = HD / 4, NL = 12;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int T = ap.T, D = 2 * HHD;
    const int kv0 = (m0 / T) * T;                       // first key / value row the tile can see
    const int nkv = min(m0 + 32, a.M) - kv0;            // <= 32 + T - 1 <= 47
    float* qs = scratch;                                // [32][ST]
    float* ks = qs + 32 * ST;                           // [KVR][ST]
    float* vs = ks + KVR * ST;                          // [KVR][ST]
    const int nq4 = 32 * H4, nk4 = nkv * H4, total = nq4 + 2 * nk4;   // float4 items of one half
    // ---- every global load of both halves, now ----
    f32x4 t[2][NL];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int idx = min(tid + 512 * u, total - 1);
            int64_t src;
            if (idx < nq4) {
                const int r = idx / H4, c = idx - r * H4;
                src = (int64_t)min(m0 + r, a.M - 1) * ap.ldq + hh * HHD + 4 * c;
            } else {
                const int i2 = idx - nq4, which = i2 >= nk4, i3 = i2 - which * nk4, r = i3 / H4, c = i3 - r * H4;
                src = (int64_t)(kv0 + r) * ap.ldq + (1 + which) * D + hh * HHD + 4 * c;
            }
            t[hh][u] = ldg4(ap.qkv + src);
        }
    // ---- this thread's (head, row, quarter) ----
    const int lp = tid & 3, pr = tid >> 2, r = pr & 31, hl = pr >> 5;   // 128 pairs = 4 heads x 32 rows
    const int m = m0 + r;
    const bool valid = m < a.M;
    const int mc = min(m, a.M - 1), smp = mc / T, tq = mc - smp * T;   // position inside the sample: keys 0 .. tq
    const int kb = smp * T - kv0;                                       // the sample's first row in ks / vs
    const int d0 = hl * HD + lp * DS;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if (hh) __syncthreads();                        // everyone is done reading the first half's rows
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int idx = tid + 512 * u;
            if (idx < total) {
                float* dst;
                if (idx < nq4) { const int rr = idx / H4; dst = qs + rr * ST + 4 * (idx - rr * H4); }
                else {
                    const int i2 = idx - nq4, which = i2 >= nk4, i3 = i2 - which * nk4, rr = i3 / H4;
                    dst = (which ? vs : ks) + rr * ST + 4 * (i3 - rr * H4);
                }
                *(f32x4*)dst = t[hh][u];
            }
        }
        __syncthreads();
        float q[DS];
#pragma unroll
        for (int d = 0; d < DS; d += 4) {
            const f32x4 x = *(const f32x4*)(qs + r * ST + d0 + d);
            q[d] = x.x; q[d + 1] = x.y; q[d + 2] = x.z; q[d + 3] = x.w;
        }
        float sc[TKC];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TKC; ++j) {
            const float* kp = ks + min(kb + min(j, tq), KVR - 1) * ST + d0;   // clamped: masked keys re-read a visible one
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < DS; d += 4) {
                const f32x4 x = *(const f32x4*)(kp + d);
                dot = fmaf(q[d], x.x, dot); dot = fmaf(q[d + 1], x.y, dot); dot = fmaf(q[d + 2], x.z, dot); dot = fmaf(q[d + 3], x.w, dot);
            }
            dot = quad_sum(dot);                        // the four quarters of the head dimension
            sc[j] = j <= tq ? dot * ap.scale : -INFINITY;
            mx = fmaxf(mx, sc[j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < TKC; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }   // key 0 is always visible: mx is finite
        const float inv = 1.0f / sum;
        float o[DS];
#pragma unroll
        for (int d = 0; d < DS; ++d) o[d] = 0.f;
#pragma unroll
        for (int j = 0; j < TKC; ++j) {
            const float pj = sc[j] * inv;
            const float* vp = vs + min(kb + min(j, tq), KVR - 1) * ST + d0;
#pragma unroll
            for (int d = 0; d < DS; d += 4) {
                const f32x4 x = *(const f32x4*)(vp + d);
                o[d] = fmaf(pj, x.x, o[d]); o[d + 1] = fmaf(pj, x.y, o[d + 1]); o[d + 2] = fmaf(pj, x.z, o[d + 2]); o[d + 3] = fmaf(pj, x.w, o[d + 3]);
            }
        }
#pragma unroll
        for (int d = 0; d < DS; d += 4)
            *(f32x4*)(xa + r * stride + hh * HHD + d0 + d) = valid ? (f32x4){o[d], o[d + 1], o[d + 2], o[d + 3]} : zero4;
    }
}

// ------------------------------------------------------------------------------------------------
// ONE sample's causal self-attention (8 heads of HD, T <= TKC <= 16 rows) -> xa (rows >= T zero): the attention stage of
// attn_xattn_tile.  q | k | v of a row are adjacent (ldq == 3 D), so the sample's rows are ONE contiguous block of T * 3 D
// floats: thread t requests float4 items t, t + 512, ... of it and puts them at the same (row, column) in LDS -- no division,
// no 64-bit arithmetic per request (the general stage above spends ~1100 integer instructions per thread on its 9 + 28
// addresses: 9.5 k cycles from entry until the rows are in LDS even on an idle chip, all VALU issue).  Row stride 3 D + 16 =
// 16 (mod 64) floats.  Wave = head: both attention products on the MFMA pipe (below).  512 threads = 8 heads.
//   scratch: 16 * (3 D + 16) floats.  `after_loads`: see attn_stage_tile.
// ------------------------------------------------------------------------------------------------
template <int HD, int TKC, class F>
__device__ __forceinline__ void attn_sample_tile(const mdt_attn_pro& ap, float* xa, int stride, int m0, float* scratch, int tid,
                                                 F after_loads) {
    constexpr int D = 8 * HD, R4 = 3 * D / 4, ST3 = 3 * D + 16;
    constexpr int NL = (TKC * R4 + 511) / 512;          // 6 float4 per thread for 10 rows of d = 384, 9 for 16
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int T = ap.T, total = T * R4;
    const f32x4* src = (const f32x4*)(ap.qkv + (int64_t)m0 * (3 * D));
    f32x4 t[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) t[u] = src[min(tid + 512 * u, total - 1)];
    after_loads();
    {
        int rr = tid / R4, c = tid - rr * R4;            // item tid + 512 u sits at (row rr, float4 column c)
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            if (tid + 512 * u < total) *(f32x4*)(scratch + rr * ST3 + 4 * c) = t[u];
            c += 512 % R4; rr += 512 / R4;
            if (c >= R4) { c -= R4; ++rr; }
        }
    }
    __syncthreads();
#ifdef MDT_TS_ROWS_LANDED
    MDT_TS(7)
#endif
    // ---- wave h = head h, both products on the MFMA pipe (round 5; rounds 3-4: thread = (head, row, quarter of the head
    //      dimension), ~60 LDS reads and ~250 FMAs per thread with 6 of the 16 row slots idle: 4 k of the kernel's 42 k clocks).
    //      Scores with TRANSPOSED operand roles: A = the head's key rows, B = its query rows -> the lane ends with
    //      S[query lane % 16][keys 4 g .. 4 g + 3], g = lane / 16; the masked softmax of a query runs over the lane's four values
    //      and the four 16-lane rows of the wave (xrow_max / xrow_sum: v_permlane swaps, no LDS); the probabilities, still in
    //      those registers, are the B operand of O^T = V^T P^T (k = key 4 g + e), whose A operand is read from the v rows
    //      as scalars; the lane ends with 4 consecutive features of its query's output: one 16-byte store into xa.
    //      Rows past T are never read (indices clamped to T - 1: a masked key's probability is an exact 0, times a finite v). ----
    constexpr int KH = HD / 16;
    const int lane = tid & 63, h = tid >> 6, m = lane & 15, g = lane >> 4;
    const int rc = min(m, T - 1);
    f32x4 sc = zero4;
    {
        const float* qp = scratch + rc * ST3 + h * HD + 4 * g;
        const float* kp = qp + D;
        f32x4 kf[KH], qf[KH];
#pragma unroll
        for (int kc = 0; kc < KH; ++kc) { kf[kc] = *(const f32x4*)(kp + 16 * kc); qf[kc] = *(const f32x4*)(qp + 16 * kc); }
#pragma unroll
        for (int kc = 0; kc < KH; ++kc)
#pragma unroll
            for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kc][e], qf[kc][e], sc, 0, 0, 0);
    }
    // V^T fragments (scalars: feature 16 nt + m of key row 4 g + e) are requested before the softmax arithmetic
    float vt[KH][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float* vp = scratch + min(4 * g + e, T - 1) * ST3 + 2 * D + h * HD + m;
#pragma unroll
        for (int nt = 0; nt < KH; ++nt) vt[nt][e] = vp[16 * nt];
    }
    f32x4 pr;
    {
        float mx = -INFINITY;
        bool vis[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vis[e] = 4 * g + e <= rc;                    // causal inside the sample; rc < T
            sc[e] = vis[e] ? sc[e] * ap.scale : -INFINITY;
            mx = fmaxf(mx, sc[e]);
        }
        mx = xrow_max(mx);                               // key 0 is always visible: finite
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pr[e] = vis[e] ? expf(sc[e] - mx) : 0.f; sum += pr[e]; }
        const float inv = 1.0f / xrow_sum(sum);
        pr = pr * inv;
    }
    const bool valid = m < T;
#pragma unroll
    for (int nt = 0; nt < KH; ++nt) {
        f32x4 o = zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[nt][e], pr[e], o, 0, 0, 0);
        *(f32x4*)(xa + m * stride + h * HD + 16 * nt + 4 * g) = valid ? o : zero4;
    }
}

// ------------------------------------------------------------------------------------------------
// fused GEMM tile:  out = epilogue( prologue(A) @ W^T ) for row tile `by` and column tile `bx`
//   64 * NWAVES threads; tile = (MTILES*16 rows) x (NWAVES * NTW * 16 columns); full K.
//   wave w owns NTW column tiles and ALL row tiles of the workgroup tile.
//   Memory-level parallelism rule for every phase: all global loads of a phase are issued back to back from
//   clamped (always valid) addresses, consumed afterwards; validity is applied by selects / masked stores.  A
//   branch around a load makes hipcc wait vmcnt(0) right behind it -- one full L2 round trip per load.
// No thread leaves early (a caller may follow the tile with a barrier).
// ------------------------------------------------------------------------------------------------
// GLU (compile time; 0 in every kernel but k_gemm_glu): 3 = SwishGLU forward on this product's epilogue, 4 = SwishGLU backward
// (mdt_gemm_args.aux_mode 3 / 4) -- their own instantiations, so that the plain kernels carry none of their code
template <int MTILES, int NTW, int NWAVES, int PRO, bool RES, bool COH, int XP = 1, int GLU = 0, int AHD = 48, int ATKC = 16>
__device__ __forceinline__ void gemm_tile(const mdt_gemm_args& a, int kchunk, int by, int bx, float* lds,
                                          const float* __restrict__ zeros, int tid, const mdt_attn_pro* ap = nullptr) {
    MDT_TS(0)
    MDT_TS_HWID()
    constexpr int MT = MTILES * 16;
    constexpr bool KSTEP_PRIO = false;
    // weight-fragment ring: R-1 k-steps of 1-KiB loads in flight per column tile.  A k-step is only 8 MFMAs
    // (256 pipe cycles) with one column tile per wave, so the narrow variants need the deeper ring to cover L2 latency.
    constexpr int R = (NTW == 1 ? 6 : (NTW == 2 ? 4 : 3)) + MDT_RING_ADD;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const int m0 = by * MT;
    const int N16 = a.N >> 4, K16 = a.K >> 4;
    const int nt0 = (bx * NWAVES + wave) * NTW;
    const bool active = nt0 < N16;  // wave has at least one real column tile
    const int stride = kchunk + 4;  // floats; 16-byte aligned rows, breaks the power-of-two bank stride
    const ActLd<COH> LO(a.out);

    // ---- weight stream: one continuous k16 index over the whole K, independent of the LDS chunking ----
    WStream wp[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = min(nt0 + j, N16 - 1);  // clamp: a partial last wave re-reads a valid tile
        wp[j] = wstream(a.Wp, (int64_t)nt * K16 * 256 + lane * 4);
    }
    f32x4 ring[R][NTW];
#pragma unroll
    for (int u = 0; u < R - 1; ++u)
#pragma unroll
        for (int j = 0; j < NTW; ++j) ring[u][j] = wld4(wp[j] + min(u, K16 - 1) * 256);

    f32x4 acc[MTILES][NTW];
#pragma unroll
    for (int i = 0; i < MTILES; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = zero4;

    // ---- epilogue operands (bias / gate / residual) are requested NOW, in the same latency window as the
    //      activation tile: vmcnt retires loads in order, so a load issued later would stall the weight ring ----
    const int nq = 4 * (lane >> 4);
    const bool gated = RES && a.gate_off >= 0;
    int ncol[NTW];
    constexpr int NRES = RES ? NTW : 1;  // residual GEMMs (out += gate * value) also prefetch gate and old value
    f32x4 bias_v[NTW], gate_v[MTILES][NRES], res_v[MTILES][NRES];
    int64_t ooff[MTILES];
    {
        const float* biasp = a.bias != nullptr ? a.bias : zeros;
        const float* rvp = a.rowvec != nullptr ? a.rowvec : zeros;
        // SwishGLU forward (aux_mode 3): the weight image interleaves the projected / gate halves tile by tile
        // (mdt_op_pack_weight_glu); bias and output columns are the NATURAL ones: tile T -> half T & 1, columns 16 (T >> 1)
        constexpr bool glu_fwd = GLU == 3 && (NTW % 2 == 0);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int T = min(nt0 + j, N16 - 1);
            ncol[j] = (glu_fwd ? (T & 1) * (a.N >> 1) + (T >> 1) * 16 : T * 16) + nq;
            bias_v[j] = ldg4(biasp + ncol[j]) + ldg4(rvp + ncol[j]);
        }
#pragma unroll
        for (int i = 0; i < MTILES; ++i) {
            const int m = min(m0 + i * 16 + (lane & 15), a.M - 1);
            const int64_t orow =
                a.gin == 1 ? (int64_t)m * a.gout + a.goff : (int64_t)(m / a.gin) * a.gout + (m % a.gin) + a.goff;
            ooff[i] = orow * a.ldo;
            if constexpr (RES) {
                const float* gp = zeros;
                if (gated)
                    gp = a.mod + a.gate_off +
                         (a.mod_stride == 0 ? 0 : (int64_t)(m / a.rows_per_sample) * a.mod_stride);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    gate_v[i][j] = ldg4(gp + ncol[j]);
                    res_v[i][j] = LO.ld4(ooff[i] + ncol[j]);
                }
            }
        }
    }

    int kg = 0;  // global k16 index of the weight stream
    // A LayerNorm / attention prologue stages whole rows: ONE chunk, and the compiler is told so (a loop with compile-time bounds
    // 0 .. 1): as a run-time loop its header made the row requests wait for the weight ring and the epilogue operands requested
    // above (`vmcnt(4)` in front of the first row load: an L2 round trip before the rows' Infinity-Cache one in every LayerNorm GEMM)
    constexpr bool ONE_CHUNK = PRO != PRO_PLAIN;
    const int k_end = ONE_CHUNK ? 1 : a.K, k_step = ONE_CHUNK ? 1 : kchunk;
    for (int k0 = 0; k0 < k_end; k0 += k_step) {
        const int klen = ONE_CHUNK ? a.K : min(kchunk, a.K - k0);
        if (k0 > 0) __syncthreads();  // everyone is done reading the previous chunk
        if constexpr (PRO == PRO_ATTN)
            attn_stage_tile<AHD, ATKC>(a, *ap, lds, stride, m0, lds + MT * stride, tid);
        else
            gemm_stage_tile<MTILES, NWAVES, PRO, COH, XP, (NTW <= 3 && XP <= 3) ? 1 : 2>(a, lds, stride, m0, k0, klen, zeros, tid, lane, wave,
                                                          bx == 0 ? a.a_merged : nullptr);
        MDT_TS(1)
        __syncthreads();
        MDT_TS(2)

        if (active) {
            const int nk = klen >> 4;
            const float* ap = lds + (lane & 15) * stride + 4 * (lane >> 4);
            f32x4 av[MTILES];
#pragma unroll
            for (int i = 0; i < MTILES; ++i) av[i] = *(const f32x4*)(ap + i * 16 * stride);
            int kc = 0;
            for (; kc + R <= nk; kc += R) {
#pragma unroll
                for (int u = 0; u < R; ++u) MDT_KSTEP(u, kc + u)
            }
            if (kc < nk) {  // tail: nk % R steps, then rotate the ring so that slot 0 is the next k-step again
                const int rem = nk - kc;
#pragma unroll
                for (int u = 0; u < R - 1; ++u)
                    if (u < rem) MDT_KSTEP(u, kc + u)
                for (int r = 0; r < rem; ++r) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const f32x4 first = ring[0][j];
#pragma unroll
                        for (int u = 0; u + 1 < R; ++u) ring[u][j] = ring[u + 1][j];
                        ring[R - 1][j] = first;
                    }
                }
            }
            kg += nk;
        }
    }

    // ---- epilogue: lane holds out[m0 + i*16 + lane%16][n .. n+3], n = tile*16 + 4*(lane/16) ----
    MDT_TS(3)
    if (active) {
#pragma unroll
        for (int i = 0; i < MTILES; ++i) {
            const bool mok = m0 + i * 16 + (lane & 15) < a.M;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const bool ok = mok && nt0 + j < N16;
                f32x4 v = acc[i][j] + bias_v[j];
                if constexpr (GLU == 4) {
                    // SwishGLU backward: v = d_out[m][c..c+3]; u row = [projected | gate] beside it; two stores
                    const f32x4 pj = *(const f32x4*)(a.aux + ooff[i] + ncol[j]);  // rows clamped above: in bounds
                    const f32x4 gt = *(const f32x4*)(a.aux + ooff[i] + a.N + ncol[j]);
                    f32x4 dg;
                    dg.x = v.x * pj.x * glu_silu_grad(gt.x); dg.y = v.y * pj.y * glu_silu_grad(gt.y);
                    dg.z = v.z * pj.z * glu_silu_grad(gt.z); dg.w = v.w * pj.w * glu_silu_grad(gt.w);
                    v.x *= glu_silu(gt.x); v.y *= glu_silu(gt.y); v.z *= glu_silu(gt.z); v.w *= glu_silu(gt.w);
                    if (ok) st4(a.out + ooff[i] + a.N + ncol[j], dg);
                } else if constexpr (GLU == 3 && NTW % 2 == 0) {
                    // even tile: projected, odd tile: its gate (same lane, same columns): u -> aux (row stride 2 ldo),
                    // projected * silu(gate) -> out, stored by the even tile
                    if (ok) st4(const_cast<float*>(a.aux) + 2 * ooff[i] + ncol[j], v);
                    if ((j & 1) == 0) {
                        const f32x4 gt = acc[i][(j + 1) % NTW] + bias_v[(j + 1) % NTW];
                        v.x *= glu_silu(gt.x); v.y *= glu_silu(gt.y); v.z *= glu_silu(gt.z); v.w *= glu_silu(gt.w);
                    } else {
                        continue;  // the gate tile has no column of its own in `out`
                    }
                } else if constexpr (PRO == PRO_PLAIN && !RES && !COH) {
                    // training hooks (mdt_gemm_args.aux): keep the pre-activation beside the activated value, or turn the
                    // product into the gradient of the activation below it
                    if (a.aux_mode == 2) {
                        const f32x4 u = *(const f32x4*)(a.aux + ooff[i] + ncol[j]);  // rows clamped above: in bounds
                        v.x *= apply_act_grad1(u.x, a.act); v.y *= apply_act_grad1(u.y, a.act);
                        v.z *= apply_act_grad1(u.z, a.act); v.w *= apply_act_grad1(u.w, a.act);
                    } else {
                        if (a.aux_mode == 1 && ok) st4(const_cast<float*>(a.aux) + ooff[i] + ncol[j], v);
                        v = apply_act(v, a.act);
                    }
                } else {
                    v = apply_act(v, a.act);
                }
                if constexpr (RES) v = res_v[i][j] + (gated ? gate_v[i][j] * v : v);
#if defined(MDT_GEMM_OUT_PLAIN)   // A/B build: the GEMM's output tile stays in the XCD's L2 (plain store) for a same-XCD reader -- round 4: 4.50 vs 4.445 ms per B = 256 call, write-through stays
                if (ok) *(f32x4*)(a.out + ooff[i] + ncol[j]) = v;
#else
                if (ok) st4(a.out + ooff[i] + ncol[j], v);
#endif
            }
        }
    }
    MDT_TS(4)
}

// wave-level LDS flags of mlp_tile (no fences: LDS serves a CU's requests in order, a wave's flag store is issued behind its
// data stores and their lgkmcnt has retired; vmcnt -- the weight ring in flight -- is never waited on here).  The accesses
// are volatile, and address-space inference leaves volatile accesses alone: without the explicit LDS pointer type they
// become FLAT instructions, which count on vmcnt as well and would drain the weight ring at every poll.
typedef __attribute__((address_space(3))) int mdt_lds_int;
typedef int mdt_i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) mdt_i32x4 mdt_lds_i32x4;
__device__ __forceinline__ void mlp_set_flag(int* f, int lane) {
    if (lane == 0) *(volatile mdt_lds_int*)(mdt_lds_int*)f = 1;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void mlp_wait_flag(const int* f) {
    while (__builtin_amdgcn_readfirstlane(*(const volatile mdt_lds_int*)(const mdt_lds_int*)f) == 0) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
}
// chunk c of the hidden slice = the 64 columns wave c produced; `ready` caches what this wave has already seen
__device__ __forceinline__ void mlp_wait_chunk(const int* flg, int c, unsigned& ready) {
    while (!((ready >> c) & 1u)) {
        const volatile mdt_lds_i32x4* q = (const volatile mdt_lds_i32x4*)(const mdt_lds_int*)flg;
        const mdt_i32x4 a = q[0], b = q[1];
        const unsigned m = (a.x != 0) | ((a.y != 0) << 1) | ((a.z != 0) << 2) | ((a.w != 0) << 3) | ((b.x != 0) << 4) |
                           ((b.y != 0) << 5) | ((b.z != 0) << 6) | ((b.w != 0) << 7);
        ready = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
        if (!((ready >> c) & 1u)) __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// fused MLP tile:  the whole MLP sublayer  x + gate * (act(prologue(x) W1^T + b1) W2^T + b2)  for row tile `by` and hidden
// slice `s` (512 of the 4D hidden columns).  Phase 1 is gemm_tile<2, 4, 8> on W1's rows [512 s, 512 s + 512) with the
// activated (32 x 512) result kept in LDS; phase 2 multiplies it by the matching K-slice of W2 (all D output columns: wave w
// owns NTW2 = D / 128 column tiles) and stores the PARTIAL product as slab s (slab 0 also carries b2 and the residual x).
// The hidden layer never reaches memory, the second GEMM has no prologue and no launch; the S = 4D / 512 slabs are added in
// slab order by whoever reads them next (gemm_stage_tile / head_rows with XP = S).  512 threads;
// lds: 32 * (D + 4) + 32 * 516 floats.  `f` = the first Linear's arguments (A = x), `p` = the second's.
// ------------------------------------------------------------------------------------------------
// SKEW (`skew` > 0): the two waves that share a SIMD (w and w + 4) run `skew` k-steps apart instead of in lockstep, and the
// workgroup barrier between the two products is replaced by one "my 64 hidden columns are in LDS" flag per wave that the
// second product's K walk checks at its 64-column chunk boundaries (natural K order: the sums are bit-identical).  While
// the early wave of a SIMD runs its GELU epilogue (VALU) the late one still has MFMAs of the first product to issue, and
// while the late one runs its GELU the early one is already multiplying the chunks that are ready: the activation block
// and the early wave's final epilogue move under the partner's MFMAs.  lds: + 16 ints behind the hidden slice.
template <int NTW2, int PRO>
__device__ __forceinline__ void mlp_tile(const mdt_gemm_args& f, const mdt_gemm_args& p, float* __restrict__ parts,
                                         int64_t part_stride, int by, int s, float* lds, const float* __restrict__ zeros,
                                         int tid, int skew_arg = 0) {
    const int skew = skew_arg & 0xff;
    const bool prio = (skew_arg & 0x100) != 0;  // MFMA loops at raised issue priority (the partner's VALU epilogue fills the gaps)
    MDT_TS2(0)
    MDT_TS_HWID()
    constexpr int MTILES = 2, NWAVES = 8, MT = 32, HS = 512, HSTR = HS + 4, NTW1 = 4;
    constexpr bool KSTEP_PRIO = false;
#ifdef MDT_MLP_RING  // tuning builds: depth of both weight-fragment rings
    constexpr int R1 = MDT_MLP_RING, R2 = MDT_MLP_RING;
#else
    constexpr int R1 = 3, R2 = NTW2 == 1 ? 6 : (NTW2 == 2 ? 4 : 3);
#endif
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const int m0 = by * MT;
    const int D = f.K, stride1 = D + 4;
    float* xn = lds;                  // [32][D + 4]   normalised (+ modulated) rows
    float* hs = lds + MT * stride1;   // [32][516]     this slice of the activated hidden layer
    int* flg = (int*)(hs + MT * HSTR);  // [0..7] wave w's hidden columns are in LDS; [8..11] early wave w passed k-step `skew`
    const int nq = 4 * (lane >> 4);
    const bool late = skew > 0 && wave >= NWAVES / 2;  // the second wave of its SIMD (waves go to SIMDs round-robin)
    if (tid < 16) flg[tid] = 0;

    // ---- phase 1 operands: weight ring of W1 rows [512 s + 64 wave, + 64), bias ----
    const int K16a = D >> 4;
    const int nt1 = (s * NWAVES + wave) * NTW1;
    WStream wp1[NTW1];
    f32x4 ring1[R1][NTW1], acc1[MTILES][NTW1], b1[NTW1];
#pragma unroll
    for (int j = 0; j < NTW1; ++j) wp1[j] = wstream(f.Wp, (int64_t)(nt1 + j) * K16a * 256 + lane * 4);
#pragma unroll
    for (int u = 0; u < R1 - 1; ++u)
#pragma unroll
        for (int j = 0; j < NTW1; ++j) ring1[u][j] = wld4(wp1[j] + min(u, K16a - 1) * 256);
    {
        const float* bp = f.bias != nullptr ? f.bias : zeros;
#pragma unroll
        for (int j = 0; j < NTW1; ++j) b1[j] = ldg4(bp + (nt1 + j) * 16 + nq);
    }
#pragma unroll
    for (int i = 0; i < MTILES; ++i)
#pragma unroll
        for (int j = 0; j < NTW1; ++j) acc1[i][j] = zero4;
    gemm_stage_tile<MTILES, NWAVES, PRO, false>(f, xn, stride1, m0, 0, D, zeros, tid, lane, wave);
    MDT_TS2(1)
    __syncthreads();
    MDT_TS2(2)
    {
        constexpr int NTW = NTW1, R = R1;
        const int K16 = K16a, nk = K16a, stride = stride1, kg = 0;
        WStream (&wp)[NTW] = wp1;
        f32x4 (&ring)[R][NTW] = ring1;
        f32x4 (&acc)[MTILES][NTW] = acc1;
        const float* ap = xn + (lane & 15) * stride + 4 * (lane >> 4);
        f32x4 av[MTILES];
#pragma unroll
        for (int i = 0; i < MTILES; ++i) av[i] = *(const f32x4*)(ap + i * 16 * stride);
        if (late) mlp_wait_flag(flg + 8 + (wave & 3));
        if (prio) __builtin_amdgcn_s_setprio(2);
        // the early wave walks `skew` k-steps (whole ring rounds) before it lets its partner start: two loops, so that no
        // control flow inside a loop body makes the compiler's LDS-counter bookkeeping pessimistic
        const int ksk = late ? 0 : min(skew, nk) / R * R;
        int kc = 0;
        for (; kc < ksk; kc += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) MDT_KSTEP(u, kc + u)
        }
        if (!late) mlp_set_flag(flg + 8 + (wave & 3), lane);
        for (; kc + R <= nk; kc += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) MDT_KSTEP(u, kc + u)
        }
        if (kc < nk) {
            const int rem = nk - kc;
#pragma unroll
            for (int u = 0; u < R - 1; ++u)
                if (u < rem) MDT_KSTEP(u, kc + u)
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
    }
    MDT_TS2(3)
    // ---- phase 2 operands, requested before the activation epilogue so that they travel while it runs: the first
    //      fragments of W2's K-slice [512 s, 512 s + 512), then bias / gate / residual rows of the output tile ----
    const int K16b = HS >> 4;  // k16 steps of the slice
    WStream wp2[NTW2];
    f32x4 ring2[R2][NTW2], acc2[MTILES][NTW2], b2[NTW2], gate_v[MTILES][NTW2], res_v[MTILES][NTW2];
    int ncol[NTW2];
#pragma unroll
    for (int j = 0; j < NTW2; ++j) {
        const int nt = wave * NTW2 + j;
        wp2[j] = wstream(p.Wp, ((int64_t)nt * (p.K >> 4) + (int64_t)s * K16b) * 256 + lane * 4);
        ncol[j] = nt * 16 + nq;
    }
#pragma unroll
    for (int u = 0; u < R2 - 1; ++u)
#pragma unroll
        for (int j = 0; j < NTW2; ++j) ring2[u][j] = wld4(wp2[j] + u * 256);
    const bool gated = p.mod != nullptr && p.gate_off >= 0;
    {
        const float* bp = (s == 0 && p.bias != nullptr) ? p.bias : zeros;
#pragma unroll
        for (int j = 0; j < NTW2; ++j) b2[j] = ldg4(bp + ncol[j]);
#pragma unroll
        for (int i = 0; i < MTILES; ++i) {
            const int64_t m = min(m0 + i * 16 + (lane & 15), f.M - 1);
            const float* gp = gated ? p.mod + p.gate_off + (p.mod_stride == 0 ? 0 : (m / p.rows_per_sample) * p.mod_stride)
                                    : zeros;
            const float* rp = s == 0 ? f.A + m * f.lda : zeros;  // slab 0 carries the residual stream
#pragma unroll
            for (int j = 0; j < NTW2; ++j) {
                gate_v[i][j] = ldg4(gp + ncol[j]);
                res_v[i][j] = ldg4(rp + ncol[j]);
                acc2[i][j] = zero4;
            }
        }
    }
    // ---- activation epilogue of phase 1 -> LDS (lane holds hidden[i*16 + lane%16][tile*16 + 4*(lane/16) .. +3]) ----
#pragma unroll
    for (int i = 0; i < MTILES; ++i)
#pragma unroll
        for (int j = 0; j < NTW1; ++j)
            *(f32x4*)(hs + (i * 16 + (lane & 15)) * HSTR + (wave * NTW1 + j) * 16 + nq) = apply_act(acc1[i][j] + b1[j], f.act);
    MDT_TS2(4)
    unsigned ready = 0xffu;  // bit c: the 64 hidden columns of chunk c (wave c's) are known to be in LDS
    if (skew > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's hidden columns have landed
        mlp_set_flag(flg + wave, lane);
        ready = 0;
    } else {
        __syncthreads();
    }
    {
        constexpr int NTW = NTW2, R = R2;
        const int K16 = K16b, nk = K16b, stride = HSTR, kg = 0;
        WStream (&wp)[NTW] = wp2;
        f32x4 (&ring)[R][NTW] = ring2;
        f32x4 (&acc)[MTILES][NTW] = acc2;
        const float* ap = hs + (lane & 15) * stride + 4 * (lane >> 4);
        f32x4 av[MTILES];
        mlp_wait_chunk(flg, 0, ready);
        if (prio) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < MTILES; ++i) av[i] = *(const f32x4*)(ap + i * 16 * stride);
        // a k-step prefetches the activation fragments of the NEXT one: the chunk that step belongs to must be there
#define MDT_KSTEP_C(U, KC)                                                         \
    {                                                                              \
        if ((((KC) + 1) & 3) == 0 && (KC) + 1 < nk) mlp_wait_chunk(flg, ((KC) + 1) >> 2, ready); \
        MDT_KSTEP(U, KC)                                                           \
    }
        int kc = 0;
        for (; kc + R <= nk; kc += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) MDT_KSTEP_C(u, kc + u)
        }
        if (kc < nk) {
            const int rem = nk - kc;
#pragma unroll
            for (int u = 0; u < R - 1; ++u)
                if (u < rem) MDT_KSTEP_C(u, kc + u)
        }
#undef MDT_KSTEP_C
        if (prio) __builtin_amdgcn_s_setprio(0);
    }
    MDT_TS2(5)
    float* out = parts + (int64_t)s * part_stride;
#pragma unroll
    for (int i = 0; i < MTILES; ++i) {
        const int m = m0 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NTW2; ++j) {
            f32x4 v = acc2[i][j] + b2[j];
            v = res_v[i][j] + (gated ? gate_v[i][j] * v : v);
            if (m < f.M) st4(out + (int64_t)m * p.ldo + ncol[j], v);
        }
    }
    MDT_TS2(6)
}

// ------------------------------------------------------------------------------------------------
// small-M GEMM tile: 16 rows (m0 ..) x 16 columns (n_tile), K split over the workgroup's 8 waves (interleaved k16
// steps), every wave keeps 4 weight fragments + 4 activation fragments in flight, partial 16x16 tiles meet in LDS
// and wave 0 runs the epilogue.  Activations come straight from global memory in MFMA-fragment order (lane: row l%16,
// 4 consecutive k), LayerNorm / modulate applied in registers from per-row statistics the workgroup computes first.
// 512 threads.  s_stat: 32 floats, red: 8*64*4 floats of LDS.
// ------------------------------------------------------------------------------------------------
// XL: the tile's rows are read from LDS (`xl`, row stride `xls` floats, row 0 = row m0; k_xattn_gemm_smallm leaves the
// cross-attention's output there) instead of a.A; `rows` (<= 16) of the tile are real.
template <bool COH, bool XL = false>
__device__ __forceinline__ void gemm_smallm_tile(const mdt_gemm_args& a, int n_tile, int m0, float* s_stat, float* red,
                                                 const float* __restrict__ zeros, int tid, const float* xl = nullptr, int xls = 0,
                                                 int rows = 16) {
    MDT_TSO(XL ? gridDim.x : 0, 0)
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const int K16 = a.K >> 4;
    const ActLd<COH> LA(a.A), LO(a.out);
    float* s_mean = s_stat;
    float* s_rstd = s_stat + 16;
    // epilogue operands (bias, row vector, gate, old output) are requested NOW by every wave -- wave 0 alone uses them, the
    // others' copies are L1 hits: no load sits behind a branch, and the epilogue does not start with a memory round trip (a
    // rollout-sized call is a chain of ~250 such kernels)
    const int e_row = m0 + (lane & 15);
    const int64_t e_mc = min(e_row, a.M - 1);
    const int e_ncol = n_tile * 16 + 4 * (lane >> 4);
    const int64_t e_orow = a.gin == 1 ? e_mc * a.gout + a.goff : (e_mc / a.gin) * a.gout + (e_mc % a.gin) + a.goff;
    const int64_t e_oo = e_orow * a.ldo + e_ncol;
    const f32x4 e_bias = ldg4((a.bias != nullptr ? a.bias : zeros) + e_ncol) + ldg4((a.rowvec != nullptr ? a.rowvec : zeros) + e_ncol);
    const bool e_gated = a.residual && a.gate_off >= 0;
    const f32x4 e_gate = ldg4(e_gated ? a.mod + a.gate_off + (a.mod_stride == 0 ? 0 : (e_mc / a.rows_per_sample) * a.mod_stride) + e_ncol
                                      : zeros + e_ncol);
    const f32x4 e_res = LO.ld4(a.residual ? e_oo : 0);
    const float* wbase = a.Wp + (int64_t)n_tile * K16 * 256 + lane * 4;
    // rollout batches (at most two samples' rows): the wave's first round of weight fragments is requested in front of the row
    // statistics (round 5: B = 1 1.295 -> 1.286 ms, B = 2 1.339 -> 1.331; from B = 8 on it LOSES 1 %, the fragments of sixteen row
    // tiles' workgroups then queue in front of each other's rows).  -DMDT_SMALLM_WF_LATE: A/B build
#ifdef MDT_SMALLM_WF_LATE
    const bool WF_EARLY = false;
#else
    const bool WF_EARLY = a.M <= 24;
#endif
    f32x4 wf0[4];
    if (a.ln) {  // row statistics: 32 threads per row, whole row in registers (K <= 512)
        const int r = tid >> 5, l32 = tid & 31;
        const int64_t m = min(m0 + r, a.M - 1);
        const int n4 = a.K >> 2;
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = l32 + 32 * i;
            if constexpr (XL) v[i] = *(const f32x4*)(xl + min(r, rows - 1) * xls + 4 * min(c4, n4 - 1));
            else v[i] = LA.ld4(m * a.lda + 4 * min(c4, n4 - 1));
        }
        if (WF_EARLY) {
            // the wave's first round of weight fragments (all of them for K <= 512), requested BEHIND the rows and in front of the
            // statistics: the rows retire first (vmcnt is in order), the fragments travel under the statistics and the barrier
#pragma unroll
            for (int u = 0; u < 4; ++u) wf0[u] = ldg4(wbase + min(wave + 8 * u, K16 - 1) * 256);
            __builtin_amdgcn_sched_barrier(0);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = l32 + 32 * i;
            v[i] = sel4(c4 < n4, v[i], zero4);
            sum += hsum4(v[i]);
        }
        sum = half_wave_sum(sum);
        const float mean = sum / (float)a.K;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = l32 + 32 * i;
            if (c4 < n4) sq += hsq4(v[i] - mean);
        }
        sq = half_wave_sum(sq);
        if (l32 == 0) { s_mean[r] = mean; s_rstd[r] = 1.0f / sqrtf(sq / (float)a.K + 1e-5f); }
        MDT_TSO(XL ? gridDim.x : 0, 1)
        __syncthreads();
        MDT_TSO(XL ? gridDim.x : 0, 2)
    }
    const int mrow = m0 + (lane & 15);
    const int64_t mc = min(mrow, a.M - 1);
    const bool mok = mrow < a.M && (lane & 15) < rows;
    const int kq = 4 * (lane >> 4);
    const int64_t xoff = mc * a.lda + kq;
    const float* xlp = XL ? xl + min(lane & 15, rows - 1) * xls + kq : nullptr;
    const bool modded = a.ln && a.mod != nullptr && a.shift_off >= 0;
    const float* mrw = modded ? a.mod + (a.mod_stride == 0 ? 0 : (mc / a.rows_per_sample) * a.mod_stride) : zeros;
    const float mean = a.ln ? s_mean[lane & 15] : 0.f, rstd = a.ln ? s_rstd[lane & 15] : 1.f;
    f32x4 acc = zero4;
    for (int ks0 = wave; ks0 < K16; ks0 += 32) {  // 4 of this wave's k16 steps per round, all loads issued first
        f32x4 wf[4], xv[4], lw[4], lb[4], sh[4], sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = min(ks0 + 8 * u, K16 - 1);
            if (WF_EARLY && a.ln && ks0 == wave) wf[u] = wf0[u];
            else wf[u] = ldg4(wbase + ks * 256);
            if constexpr (XL) xv[u] = *(const f32x4*)(xlp + ks * 16);
            else xv[u] = LA.ld4(xoff + ks * 16);
            if (a.ln) {
                lw[u] = ldg4(a.ln_w + ks * 16 + kq);
                lb[u] = a.ln_b ? ldg4(a.ln_b + ks * 16 + kq) : zero4;
                if (modded) { sh[u] = ldg4(mrw + a.shift_off + ks * 16 + kq); sc[u] = ldg4(mrw + a.scale_off + ks * 16 + kq); }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks0 + 8 * u < K16) {
                f32x4 x = xv[u];
                if (a.ln) {
                    x = (x - mean) * rstd * lw[u] + lb[u];
                    if (modded) x = sh[u] + x * sc[u];
                }
                x = sel4(mok, x, zero4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][e], x[e], acc, 0, 0, 0);
            }
        }
    }
    MDT_TSO(XL ? gridDim.x : 0, 3)
    *(f32x4*)(red + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    MDT_TSO(XL ? gridDim.x : 0, 4)
    if (wave == 0) {
        f32x4 v = *(const f32x4*)(red + lane * 4);
#pragma unroll
        for (int w = 1; w < 8; ++w) v = v + *(const f32x4*)(red + (w * 64 + lane) * 4);
        // ---- epilogue (as gemm_tile): lane holds out[mrow][ncol .. ncol+3]; operands fetched at entry
        v = apply_act(v + e_bias, a.act);
        if (a.residual) v = e_res + (e_gated ? e_gate * v : v);
        if (mok) *(f32x4*)(a.out + e_oo) = v;  // (rollout-sized launches: plain stores, write-through costs them 0.8 %)
    }
    MDT_TSO(XL ? gridDim.x : 0, 5)
}

// ------------------------------------------------------------------------------------------------
// attn_proj tile: sample b's self-attention fused into 16 columns (n_tile) of its output projection.
//   out (+)= gate * (attn(q, k, v) @ Wp^T + bias)            rows T <= 16, 8 heads
// The split-K small-M GEMM gives wave w the k-range of head w (8 waves = 8 heads, head_dim = K / 8), so each wave first
// computes ITS head's attention output for the T rows (q / k / v of the head staged in the wave's own LDS region, scores
// and softmax by the wave alone, no workgroup barrier) and then feeds it to the MFMAs as the activation fragment.  Every
// workgroup (16 output columns) repeats the T x T attention -- 19 kFLOP per head -- in exchange for one launch and one
// round trip of the attention output per block.  The weight fragments are requested before the attention starts.
// lds: 8 * 3 * T * (HD+4) floats (round 5: only the T real rows, no probability tile -- 50 KB at T = 10, d = 384, so that two
// workgroups share a CU); red: 8*64*4 floats.  512 threads.
// ------------------------------------------------------------------------------------------------
template <int HD, bool COH>
__device__ __forceinline__ void attn_proj_tile(const mdt_gemm_args& a, const float* __restrict__ qkv_base, int64_t ldq, int T,
                                               int causal, float scale, int n_tile, int b, float* lds, float* red,
                                               const float* __restrict__ zeros, int tid) {
    constexpr int ST = HD + 4, H4 = HD / 4, KS = HD / 16;  // padded row stride, float4 per row, k16 steps per head
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, h = tid >> 6;
    const int K16 = a.K >> 4, D = a.K;
    const ActLd<COH> LQ(qkv_base), LO(a.out);
    const int64_t qoff = (int64_t)b * T * ldq;  // rows [b * T, b * T + T) of qkv and of the output
    // weight fragments of this head's k-range (in flight while the attention runs)
    const float* wbase = a.Wp + (int64_t)n_tile * K16 * 256 + lane * 4;
    f32x4 wf[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) wf[u] = ldg4(wbase + (h * KS + u) * 256);
    // epilogue operands requested now by every wave (wave 0 uses them): no round trip behind the reduction
    const int e_ncol = n_tile * 16 + 4 * (lane >> 4);
    const int64_t e_mc = min(lane & 15, T - 1);
    const int64_t e_oo = ((int64_t)b * T + e_mc) * a.ldo + e_ncol;
    const f32x4 e_bias = ldg4((a.bias != nullptr ? a.bias : zeros) + e_ncol);
    const bool e_gated = a.residual && a.gate_off >= 0;
    const f32x4 e_gate = ldg4(e_gated ? a.mod + a.gate_off + (int64_t)b * a.mod_stride + e_ncol : zeros + e_ncol);
    const f32x4 e_res = LO.ld4(a.residual ? e_oo : 0);
    float* qs = lds + h * (3 * T * ST);   // [T][ST] this head's q rows (the wave's own region of 3 T rows: no workgroup barrier)
    float* ks = qs + T * ST;
    float* vs = ks + T * ST;
    // the head's q | k | v rows: ALL requests first, then the LDS stores (round 5: as a run-time loop of load -> store the second
    // trip's requests went out only when the first trip's data had arrived -- a second memory round trip, ~1.4 us of a 6.6 us
    // launch, in every decoder block of a rollout-sized call)
    constexpr int NIT = (16 * H4 + 63) / 64;
    const int nitems = T * H4;
    f32x4 tq[NIT], tk[NIT], tv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(lane + 64 * it, nitems - 1);
        const int t = i / H4, c = i - t * H4;
        const int64_t row = qoff + (int64_t)t * ldq + h * HD + 4 * c;
        tq[it] = LQ.ld4(row);
        tk[it] = LQ.ld4(row + D);
        tv[it] = LQ.ld4(row + 2 * D);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = lane + 64 * it;
        if (i < nitems) {
            const int t = i / H4, c = i - t * H4;
            *(f32x4*)(qs + t * ST + 4 * c) = tq[it];
            *(f32x4*)(ks + t * ST + 4 * c) = tk[it];
            *(f32x4*)(vs + t * ST + 4 * c) = tv[it];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- both attention products on the MFMA pipe, as attn_sample_tile runs them (round 5; before: ~100 scores by vector FMAs,
    //      a serial softmax on T lanes, the weighted values through LDS).  Scores with transposed operand roles (A = key rows, B =
    //      query rows): the lane ends with S[query m = lane % 16][keys 4 g .. 4 g + 3], g = lane / 16; softmax over the lane's four
    //      values and the four 16-lane rows of the wave (xrow_max / xrow_sum); the probabilities are the B operand of O^T = V^T P^T,
    //      and the lane ends with O[query m][features 16 u + 4 g .. + 3] -- exactly the activation fragment of the projection's
    //      k-step u: the attention output never leaves the registers.  Rows past T are never read (indices clamped to T - 1). ----
    const int m = lane & 15, g = lane >> 4;
    const int rc = min(m, T - 1);
    f32x4 sc = zero4;
    {
        const float* qp = qs + rc * ST + 4 * g;
        const float* kp = ks + rc * ST + 4 * g;
        f32x4 kf[KS], qf[KS];
#pragma unroll
        for (int kc = 0; kc < KS; ++kc) { kf[kc] = *(const f32x4*)(kp + 16 * kc); qf[kc] = *(const f32x4*)(qp + 16 * kc); }
#pragma unroll
        for (int kc = 0; kc < KS; ++kc)
#pragma unroll
            for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kc][e], qf[kc][e], sc, 0, 0, 0);
    }
    float vt[KS][4];   // V^T fragments: feature 16 nt + m of key row 4 g + e
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float* vp = vs + min(4 * g + e, T - 1) * ST + m;
#pragma unroll
        for (int nt = 0; nt < KS; ++nt) vt[nt][e] = vp[16 * nt];
    }
    f32x4 pr;
    {
        float mx = -INFINITY;
        bool vis[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vis[e] = 4 * g + e < T && (!causal || 4 * g + e <= rc);   // key 0 is always visible
            sc[e] = vis[e] ? sc[e] * scale : -INFINITY;
            mx = fmaxf(mx, sc[e]);
        }
        mx = xrow_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pr[e] = vis[e] ? expf(sc[e] - mx) : 0.f; sum += pr[e]; }
        const float inv = 1.0f / xrow_sum(sum);
        pr = pr * inv;
    }
    // ---- this head's slice of the projection: lane holds row m, k = 16 u + 4 g .. + 3 of the head's range ----
    const int mrow = m;
    const bool rok = m < T;
    f32x4 acc = zero4;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        f32x4 o = zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[u][e], pr[e], o, 0, 0, 0);
        const f32x4 x = rok ? o : zero4;   // rows T .. 15 of the activation tile are zero
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][e], x[e], acc, 0, 0, 0);
    }
    *(f32x4*)(red + (h * 64 + lane) * 4) = acc;
    __syncthreads();
    if (h == 0) {
        f32x4 v = *(const f32x4*)(red + lane * 4);
#pragma unroll
        for (int w = 1; w < 8; ++w) v = v + *(const f32x4*)(red + (w * 64 + lane) * 4);
        const bool mok = mrow < T;
        v = v + e_bias;
        if (a.residual) v = e_res + (e_gated ? e_gate * v : v);
        if (mok) *(f32x4*)(a.out + e_oo) = v;  // (rollout-sized launches: plain stores, write-through costs them 0.8 %)
    }
}

// ------------------------------------------------------------------------------------------------
// small attention tile: sample b, head group hg of nhg (Hl = H / nhg heads, a contiguous column range of q/k/v).  The
// rows (all Hl heads) are staged in LDS with batched 16-byte loads, then thread (head, query row) runs
// softmax(q k^T) v out of LDS with the scores in registers.  10x10 / 10x4 / 4x4 score matrices are 0.1 % of the FLOPs,
// so this stays on the VALU.  256 threads (tid 0..255); `lds` is this group's own region; `sync` is the barrier the 256
// threads share with whoever else runs the same code (a workgroup barrier).
// TKC: compile-time bound on the number of keys (4 / 10 / 16, the smallest >= Tk), ROPE: rotary embedding on q/k.
// With both fixed the score / softmax / PV loops are straight-line code over clamped rows with select masks, so the
// compiler can batch the LDS reads instead of waiting on each one behind a branch.
// ------------------------------------------------------------------------------------------------
template <int HD, int TKC, bool ROPE, bool COH>
__device__ __forceinline__ void attn_tile(const mdt_attn_args& a, const float* __restrict__ rope_cos,
                                          const float* __restrict__ rope_sin, float scale, int b, int hg, int nhg, float* lds,
                                          int tid) {
    MDT_TS(0)
    MDT_TS_HWID()
    constexpr int ROT = 32;  // rotary dims (position_embeddings.py / transformer_blocks.py:108)
    const int Hl = a.H / nhg;              // heads of this group
    const int coff = hg * Hl * HD;         // first column
    const int D = Hl * HD, d4 = D >> 2;    // row length staged in LDS
    // q / k / v live in one allocation (the handle's workspace): one loader at the lowest of the three pointers
    const float* lo = a.q < a.k ? a.q : a.k;
    lo = lo < a.v ? lo : a.v;
    const ActLd<COH> LQ(lo);
    const int64_t oq = a.q - lo, ok = a.k - lo, ov = a.v - lo;
    float* qs = lds;                  // [Tq][D]
    float* ks = qs + a.Tq * D;        // [Tk][D]
    float* vs = ks + a.Tk * D;        // [Tk][D]
    const int nq = a.Tq * d4, nkv = a.Tk * d4, total = nq + 2 * nkv;
    constexpr int BATCH = 12;
    for (int base = 0; base < total; base += 256 * BATCH) {
        f32x4 t[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = min(base + u * 256 + tid, total - 1);
            int64_t src;
            if (idx < nq) {
                const int r = idx / d4;
                src = oq + (int64_t)(b * a.Tq + r) * a.ldq + coff + 4 * (idx - r * d4);
            } else if (idx < nq + nkv) {
                const int i2 = idx - nq, r = i2 / d4;
                src = ok + (int64_t)(b * a.Tk + r) * a.ldkv + coff + 4 * (i2 - r * d4);
            } else {
                const int i2 = idx - nq - nkv, r = i2 / d4;
                src = ov + (int64_t)(b * a.Tk + r) * a.ldkv + coff + 4 * (i2 - r * d4);
            }
            t[u] = LQ.ld4(src);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < total) *(f32x4*)(lds + 4 * idx) = t[u];  // q | k | v are laid out back to back
        }
    }
    MDT_TS(1)
    __syncthreads();
    MDT_TS(2)
    // ---- compute: LP lanes share one (head, query row); each owns a DS-wide slice of the head dimension ----
    constexpr int LP = HD == 48 ? 3 : (HD >= 32 ? 2 : 1);
    constexpr int DS = HD / LP;  // 16 or 32 dims per lane
    float* part = vs + a.Tk * D;  // [pairs][16 keys][LP] partial scores
    const int npairs = Hl * a.Tq;
    const int pr = min(tid % npairs, npairs - 1), ps = min(tid / npairs, LP - 1);  // pair, slice (clamped: idle lanes recompute)
    const bool live = tid < npairs * LP;
    const int t = pr % a.Tq, h = pr / a.Tq;
    const int nk = a.causal ? min(a.Tk, t + 1) : a.Tk;
    const int d0 = h * HD + ps * DS;
    float q[DS];
#pragma unroll
    for (int d = 0; d < DS; d += 4) {
        const f32x4 x = *(const f32x4*)(qs + t * D + d0 + d);
        q[d] = x.x; q[d + 1] = x.y; q[d + 2] = x.z; q[d + 3] = x.w;
    }
    if constexpr (ROPE) {
#pragma unroll
        for (int i = 0; i < DS / 2; ++i) {
            const int gi = (ps * DS) / 2 + i;  // rotary pair index inside the head
            if (gi < ROT / 2) {
                const float c = rope_cos[t * 16 + gi], sn = rope_sin[t * 16 + gi];
                const float x1 = q[2 * i], x2 = q[2 * i + 1];
                q[2 * i] = x1 * c - x2 * sn;
                q[2 * i + 1] = x2 * c + x1 * sn;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TKC; ++j) {
        const int jc = min(j, a.Tk - 1);
        const float* kp = ks + jc * D + d0;
        float kr[DS];
#pragma unroll
        for (int d = 0; d < DS; d += 4) {
            const f32x4 x = *(const f32x4*)(kp + d);
            kr[d] = x.x; kr[d + 1] = x.y; kr[d + 2] = x.z; kr[d + 3] = x.w;
        }
        if constexpr (ROPE) {
#pragma unroll
            for (int i = 0; i < DS / 2; ++i) {
                const int gi = (ps * DS) / 2 + i;
                if (gi < ROT / 2) {
                    const float c = rope_cos[jc * 16 + gi], sn = rope_sin[jc * 16 + gi];
                    const float x1 = kr[2 * i], x2 = kr[2 * i + 1];
                    kr[2 * i] = x1 * c - x2 * sn;
                    kr[2 * i + 1] = x2 * c + x1 * sn;
                }
            }
        }
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < DS; ++d) dot = fmaf(q[d], kr[d], dot);
        if (live) part[(pr * 16 + j) * LP + ps] = dot;
    }
    MDT_TS(3)
    __syncthreads();
    float sc[TKC];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < TKC; ++j) {
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < LP; ++u) dot += part[(pr * 16 + j) * LP + u];  // fixed order: deterministic
        sc[j] = j < nk ? dot * scale : -INFINITY;
        mx = fmaxf(mx, sc[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < TKC; ++j) {
        sc[j] = expf(sc[j] - mx);  // exp(-inf) = 0 for masked keys; key 0 is always visible so mx is finite
        sum += sc[j];
    }
    const float inv = 1.0f / sum;
    float o[DS];
#pragma unroll
    for (int d = 0; d < DS; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < TKC; ++j) {
        const float p = sc[j] * inv;
        const float* vp = vs + min(j, a.Tk - 1) * D + d0;
#pragma unroll
        for (int d = 0; d < DS; d += 4) {
            const f32x4 x = *(const f32x4*)(vp + d);
            o[d] = fmaf(p, x.x, o[d]); o[d + 1] = fmaf(p, x.y, o[d + 1]);
            o[d + 2] = fmaf(p, x.z, o[d + 2]); o[d + 3] = fmaf(p, x.w, o[d + 3]);
        }
    }
    if (live) {
        float* op = a.out + (int64_t)(b * a.Tq + t) * a.ldo + coff + d0;
#pragma unroll
        for (int d = 0; d < DS; d += 4) *(f32x4*)(op + d) = (f32x4){o[d], o[d + 1], o[d + 2], o[d + 3]};
    }
    MDT_TS(4)
}

__device__ __forceinline__ float edm_c_in(float sigma, float sd) { return 1.0f / sqrtf(sigma * sigma + sd * sd); }

// ------------------------------------------------------------------------------------------------
// action head rows: decoder LN -> action_pred -> EDM combine -> (DDIM update) -> (next step's embedding) for rows
// base .. base + RW - 1 (RW = 2: the action_pred / action_emb weight fragments a wave fetches are used
// twice); AMAX (8 or 16) bounds the action dimension at compile time so no load sits behind a branch.
// The caller guarantees base < a.M (wave-uniform).
// ------------------------------------------------------------------------------------------------
// RW = rows per wave: 2 when the fragments of action_pred / action_emb a wave fetches should be used twice, 1 in
// k_head (twice the waves, half the dot products per wave: the launch is a latency chain -- B = 256 sampler call 4.712 -> 4.686 ms,
// four rows per wave 4.705).
template <int AMAX, bool COH, int XP = 1, int RW = 2>
__device__ __forceinline__ void head_rows(const mdt_head_args& a, int base, int lane, const float* __restrict__ zeros) {
    const int n4 = a.D >> 2;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const ActLd<COH> LY(a.y), LX(a.x);
    int cc[2];
    bool cv[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        cv[p] = lane + 64 * p < n4;
        cc[p] = 4 * min(lane + 64 * p, n4 - 1);
    }
    int64_t row[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) row[r] = min(base + r, a.M - 1);
    // ---- every global operand is requested up front (clamped addresses, no load behind a branch) ----
    f32x4 v[RW][2], w[2], bb[2], wp[AMAX][2];
    float xin[RW][AMAX], bpv[AMAX], sigma[RW];
    const float* lnb = a.ln_b != nullptr ? a.ln_b : zeros;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int r = 0; r < RW; ++r) v[r][p] = LY.ld4(row[r] * a.D + cc[p]);
        w[p] = ldg4(a.ln_w + cc[p]);
        bb[p] = ldg4(lnb + cc[p]);
    }
    f32x4 vx[XP > 1 ? XP - 1 : 1][RW][2];  // the other slabs of a fused MLP's output (mlp_tile): summed in slab order
    if constexpr (XP > 1) {
#pragma unroll
        for (int x = 1; x < XP; ++x)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < RW; ++r) vx[x - 1][r][p] = LY.ld4((int64_t)x * a.y_part_stride + row[r] * a.D + cc[p]);
    }
#pragma unroll
    for (int c = 0; c < AMAX; ++c) {
        const int ce = min(c, a.A - 1);
#pragma unroll
        for (int p = 0; p < 2; ++p) wp[c][p] = ldg4(a.Wp + (int64_t)ce * a.D + cc[p]);
#pragma unroll
        for (int r = 0; r < RW; ++r) xin[r][c] = LX.ld1(row[r] * a.A + ce);
        bpv[c] = a.bp[ce];
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) sigma[r] = a.sigma[(row[r] / a.rows_per_sample) * a.sigma_stride];
    // operands of the fused next-step embedding travel in the same latency window (Wa is the (A, D) image)
    const float* Wa = a.y_next != nullptr ? a.Wa : a.Wp;
    const float* bap = a.y_next != nullptr ? a.ba : zeros;
    f32x4 wa[AMAX][2], ba4[2];
#pragma unroll
    for (int c = 0; c < AMAX; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) wa[c][p] = ldg4(Wa + (int64_t)min(c, a.A - 1) * a.D + cc[p]);
#pragma unroll
    for (int p = 0; p < 2; ++p) ba4[p] = ldg4(bap + cc[p]);
    float ratio = 0.f, coef = 0.f, sig_next = 1.f;
    if (a.mode == MDT_HEAD_DDIM) {
        ratio = a.step[0];
        coef = a.step[1];
        sig_next = a.step[2];
    }
    // ---- LayerNorm of the rows ----
    const float inv_d = 1.0f / (float)a.D;
    if constexpr (XP > 1) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int x = 1; x < XP; ++x) v[r][p] = v[r][p] + vx[x - 1][r][p];
    }
    float red[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        red[r] = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) { v[r][p] = sel4(cv[p], v[r][p], zero4); red[r] += hsum4(v[r][p]); }
    }
    wave_sum_n<RW>(red);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const float mean = a.no_ln ? 0.f : red[r] * inv_d;
        red[r] = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) { v[r][p] = sel4(cv[p], v[r][p] - mean, zero4); red[r] += hsq4(v[r][p]); }
    }
    wave_sum_n<RW>(red);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const float rstd = 1.0f / sqrtf(red[r] * inv_d + 1e-5f);
#pragma unroll
        for (int p = 0; p < 2; ++p) v[r][p] = sel4(cv[p] && !a.no_ln, v[r][p] * rstd * w[p] + bb[p], v[r][p]);
    }
    // ---- action_pred: RW * AMAX dot products reduced together ----
    float res[RW * AMAX];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < AMAX; ++c) {
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                s += (v[r][p].x * wp[c][p].x + v[r][p].y * wp[c][p].y) + (v[r][p].z * wp[c][p].z + v[r][p].w * wp[c][p].w);
            res[r * AMAX + c] = s;
        }
    wave_sum_n<RW * AMAX>(res);
    const float sd = a.sigma_data;
    const float cin_next = edm_c_in(sig_next, sd);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const float den2 = sigma[r] * sigma[r] + sd * sd;
        const float c_skip = sd * sd / den2;
        const float c_out = sigma[r] * sd / sqrtf(den2);
#pragma unroll
        for (int c = 0; c < AMAX; ++c) {
            const float F = res[r * AMAX + c] + bpv[c];
            float o = F;
            if (a.mode != MDT_HEAD_RAW) {
                const float den = F * c_out + xin[r][c] * c_skip;
                o = a.mode == MDT_HEAD_DDIM ? ratio * xin[r][c] + coef * den : den;
            }
            res[r * AMAX + c] = o;
        }
    }
    // all lanes hold all results (xor-butterfly sums); lanes 0..A-1 store one each
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < AMAX; ++c)
            if (c < a.A && lane == c && base + r < a.M) a.out[row[r] * a.A + c] = res[r * AMAX + c];
    if (a.y_next != nullptr) {
        f32x4 acc[RW][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r][p] = ba4[p];
#pragma unroll
        for (int c = 0; c < AMAX; ++c)
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const float xv = c < a.A ? res[r * AMAX + c] * cin_next : 0.f;
#pragma unroll
                for (int p = 0; p < 2; ++p) acc[r][p] += xv * wa[c][p];
            }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (cv[p] && base + r < a.M) *(f32x4*)(a.y_next + row[r] * a.D + cc[p]) = acc[r][p];
    }
}

// ------------------------------------------------------------------------------------------------
// collapsed cross-attention of sample b (see k_xattn_fold / k_xattn_apply in mdt_kernels.hip) on the MFMA pipe.  512 threads.
//   ln3 of the sample's Ta <= 16 rows -> LDS (one 16-row activation tile, rows >= Ta zero)
//   scores  S[t][p] = xn[t] . U[p]          : U (NPP x D, NPP = 4 H: head h's Te <= 4 context tokens at p = 4 h + j) is a
//             WEIGHT image in fragment order (k_xattn_fold writes it so): wave w takes score tile w % (NPP/16) and the
//             k-slice w / (NPP/16) of D; the slices' partial tiles meet in LDS and are added in slice order
//   softmax per (row, head) over the keys j <= min(t, Te - 1) (top-left causal), probabilities of absent keys 0
//   out[t][n] = sum_p P[t][p] Wf[p][n]      : Wf^T (D x NPP) in fragment order too; wave w owns columns [w D/8, (w+1) D/8)
//   y[t] += out[t] + bo
// Every global operand -- 6 + 6 weight fragments per lane at d = 384, the rows, the LayerNorm vectors -- is requested at
// entry.  (The first form walked the rows with scalar FMAs from 40 VGPR quads of operands per thread: 12.1 us per launch at
// B = 256 and 7.2 us at B = 1 with the rows spread over ten workgroups.)
//   lds: 16 (D + 4) + (8 / (NPP/16) + 1) * 16 * (NPP + 4) floats.  D % 128 == 0, D <= 512.
// YL: the sample's rows are read from LDS (`yl`, row stride `yls` floats: attn_xattn_tile leaves the projection's output
// there); the new rows still go to a.y.
// ------------------------------------------------------------------------------------------------
// the sample's operands that do not depend on its rows, in two groups: the fragments of its folded matrices (cold: 98 KB
// that another workgroup used one step ago) and the shared vectors.  attn_xattn_tile requests the first group at ITS entry, a
// whole attention + projection ahead of their use, and the second before the projection's epilogue; xattn_tile consumes them.
// DMAX: compile-time bound of D (sizes the register arrays).
template <int NPP, int DMAX = 512>
struct mdt_xattn_req {
    static constexpr int NTU = NPP / 16, KS = 8 / NTU, KP16 = NPP / 16, KLMAX = DMAX / 16 / KS, NTWMAX = DMAX / 128;
    f32x4 u[KLMAX];                // score fragments: tile wave % NTU, k-blocks of slice wave / NTU
    f32x4 wf[NTWMAX][KP16];        // combination fragments: column tiles wave * NTW .. + NTW - 1
    f32x4 cb;                      // score constants of the softmax thread's head
    f32x4 lw[2], lb[2];            // ln3 weight / bias at the lane's two float4 columns
    f32x4 bo[NTWMAX];              // output bias at the lane's output columns
};
// DX: 0 = D is a.D (run time); otherwise D == DX == DMAX is known at compile time (attn_xattn_tile: all offsets fold)
template <int NPP, int DMAX, int DX = 0>
__device__ __forceinline__ void xattn_request_u(const mdt_xapply_args& a, int b, int tid, mdt_xattn_req<NPP, DMAX>& q) {
    constexpr int NTU = NPP / 16, KS = 8 / NTU, KLMAX = DMAX / 16 / KS;
    const int lane = tid & 63, wave = tid >> 6;
    const int D = DX ? DX : a.D, K16 = D >> 4, KL = K16 / KS;
    const float* Ub = a.U + (int64_t)b * NPP * D;
    const int ntu = wave % NTU, ks = wave / NTU;
#pragma unroll
    for (int kk = 0; kk < KLMAX; ++kk)
        q.u[kk] = MDT_LD_STREAM(Ub + ((int64_t)(ntu * K16 + min(ks * KL + kk, K16 - 1)) * 64 + lane) * 4);
    q.cb = ldg4(a.c + (int64_t)b * NPP + 4 * min(tid >> 4, NPP / 4 - 1));   // softmax thread (t, h) = (tid % 16, tid / 16)
}
template <int NPP, int DMAX, int DX = 0>
__device__ __forceinline__ void xattn_request_wf(const mdt_xapply_args& a, int b, int tid, mdt_xattn_req<NPP, DMAX>& q) {
    constexpr int KP16 = NPP / 16, NTWMAX = DMAX / 128;
    const int lane = tid & 63, wave = tid >> 6;
    const int D = DX ? DX : a.D, N16 = D >> 4, NTW = D >> 7;
    const float* Wb = a.Wf + (int64_t)b * NPP * D;
#pragma unroll
    for (int j = 0; j < NTWMAX; ++j)
#pragma unroll
        for (int kc = 0; kc < KP16; ++kc)
            q.wf[j][kc] = MDT_LD_STREAM(Wb + ((int64_t)(min(wave * NTW + min(j, NTW - 1), N16 - 1) * KP16 + kc) * 64 + lane) * 4);
}
template <int NPP, int DMAX, int DX = 0>
__device__ __forceinline__ void xattn_request_vec(const mdt_xapply_args& a, const float* __restrict__ zeros, int tid,
                                                  mdt_xattn_req<NPP, DMAX>& q) {
    constexpr int NTWMAX = DMAX / 128;
    const int lane = tid & 63, wave = tid >> 6;
    const int D = DX ? DX : a.D, n4 = D >> 2, NTW = D >> 7;
    const float* lnb = a.ln_b != nullptr ? a.ln_b : zeros;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = 4 * min(lane + 64 * p, n4 - 1);
        q.lw[p] = ldg4(a.ln_w + c);
        q.lb[p] = ldg4(lnb + c);
    }
    const float* bop = a.bo != nullptr ? a.bo : zeros;
#pragma unroll
    for (int j = 0; j < NTWMAX; ++j) q.bo[j] = ldg4(bop + (wave * NTW + min(j, NTW - 1)) * 16 + 4 * (lane >> 4));
}

// OL: the new rows ALSO go to LDS (`yo`, row stride `yos`: k_xattn_gemm_smallm multiplies them on the spot) and reach a.y only
// if `wr` (one workgroup of those that repeat the sample's cross-attention writes it).
template <int NPP, bool COH, bool YL = false, int DMAX = 512, int DX = 0, bool OL = false>
__device__ __forceinline__ void xattn_tile(const mdt_xapply_args& a, int b, float* lds, const float* __restrict__ zeros, int tid,
                                           const float* yl = nullptr, int yls = 0, const mdt_xattn_req<NPP, DMAX>* pre = nullptr,
                                           float* yo = nullptr, int yos = 0, bool wr = true) {
    constexpr int NTU = NPP / 16, KS = 8 / NTU, KP16 = NPP / 16, PS = NPP + 4;
    constexpr int KLMAX = DMAX / 16 / KS, NTWMAX = DMAX / 128;   // D <= DMAX
    if constexpr (!YL) {
        MDT_TS(0)
        MDT_TS_HWID()
    }
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const int D = DX ? DX : a.D, Ta = a.Ta, Te = a.Te;
    const int n4 = D >> 2, K16 = D >> 4;
    const int KL = K16 / KS, NTW = D >> 7;
    const int xs = D + 4;
    float* xn = lds;                                     // [16][xs] normalised rows
    float* part = xn + 16 * xs;                          // [KS][16][PS] partial score tiles
    float* prob = part + KS * 16 * PS;                   // [16][PS] probabilities
    const ActLd<COH> LY(a.y);
    const int64_t yoff = (int64_t)b * Ta * D;

    // ---- requests, in the order they are consumed (vmcnt retires in order) ----
    // LayerNorm: wave w takes rows w and w + 8
    int cc[2];
    bool cv[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        cv[p] = lane + 64 * p < n4;
        cc[p] = 4 * min(lane + 64 * p, n4 - 1);
    }
    f32x4 v[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = min(wave + 8 * r, Ta - 1);
            if constexpr (YL) v[r][p] = *(const f32x4*)(yl + row * yls + cc[p]);
            else v[r][p] = LY.ld4(yoff + (int64_t)row * D + cc[p]);
        }
    // LayerNorm vectors, this wave's score fragments (tile ntu, k-blocks [ks KL, ks KL + KL)) and combination fragments
    // (column tiles wave * NTW .. + NTW - 1, both k-blocks of the NPP probabilities), bias, score constants
    mdt_xattn_req<NPP, DMAX> q;
    if constexpr (YL) q = *pre;
    else {
        xattn_request_vec<NPP, DMAX, DX>(a, zeros, tid, q);
        xattn_request_u<NPP, DMAX, DX>(a, b, tid, q);
        xattn_request_wf<NPP, DMAX, DX>(a, b, tid, q);
    }
    const int ntu = wave % NTU, ks = wave / NTU;
    // old rows at this lane's output position: row lane % 16, columns tile * 16 + 4 (lane / 16) .. + 3
    const int orow = min(lane & 15, Ta - 1), nq = 4 * (lane >> 4);
    f32x4 yold[NTWMAX];
#pragma unroll
    for (int j = 0; j < NTWMAX; ++j) {
        const int nc = (wave * NTW + min(j, NTW - 1)) * 16 + nq;
        if constexpr (YL) yold[j] = *(const f32x4*)(yl + orow * yls + nc);
        else yold[j] = LY.ld4(yoff + (int64_t)orow * D + nc);
    }

    // ---- LayerNorm (ln3: weight + bias) -> xn; rows >= Ta of the tile are zero ----
    const float inv_d = 1.0f / (float)D;
    float red[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        red[r] = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) { v[r][p] = sel4(cv[p], v[r][p], zero4); red[r] += hsum4(v[r][p]); }
    }
    wave_sum_n<2>(red);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float mean = red[r] * inv_d;
        red[r] = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) { v[r][p] = sel4(cv[p], v[r][p] - mean, zero4); red[r] += hsq4(v[r][p]); }
    }
    wave_sum_n<2>(red);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int t = wave + 8 * r;
        const float rstd = 1.0f / sqrtf(red[r] * inv_d + 1e-5f);
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (cv[p]) *(f32x4*)(xn + t * xs + cc[p]) = t < Ta ? v[r][p] * rstd * q.lw[p] + q.lb[p] : zero4;
    }
    if constexpr (YL) { MDT_TS(5) } else { MDT_TS(1) }
    __syncthreads();
    if constexpr (!YL) { MDT_TS(2) }
    // ---- partial scores of this wave's k-slice: lane ends with S[t = lane % 16][p = ntu * 16 + 4 (lane / 16) .. + 3] ----
    {
        const float* xp = xn + (lane & 15) * xs + 4 * (lane >> 4) + ks * KL * 16;
        f32x4 acc = zero4;
#pragma unroll
        for (int kk = 0; kk < KLMAX; ++kk)
            if (kk < KL) {
                const f32x4 av = *(const f32x4*)(xp + kk * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q.u[kk][e], av[e], acc, 0, 0, 0);
            }
        *(f32x4*)(part + (ks * 16 + (lane & 15)) * PS + ntu * 16 + 4 * (lane >> 4)) = acc;
    }
    if constexpr (!YL) { MDT_TS(3) }
    __syncthreads();
    // ---- masked softmax per (row, head): key j visible iff j <= t (top-left causal) and j < Te ----
    if (tid < 4 * NPP) {
        const int t = tid & 15, h = tid >> 4;
        f32x4 sc = q.cb;
#pragma unroll
        for (int sl = 0; sl < KS; ++sl) sc += *(const f32x4*)(part + (sl * 16 + t) * PS + 4 * h);   // fixed order
        const int nk = min(Te, t + 1);
        float mx = sc.x;
        if (nk > 1) mx = fmaxf(mx, sc.y);
        if (nk > 2) mx = fmaxf(mx, sc.z);
        if (nk > 3) mx = fmaxf(mx, sc.w);
        f32x4 e;
        e.x = expf(sc.x - mx);
        e.y = nk > 1 ? expf(sc.y - mx) : 0.f;
        e.z = nk > 2 ? expf(sc.z - mx) : 0.f;
        e.w = nk > 3 ? expf(sc.w - mx) : 0.f;
        const float inv = 1.0f / ((e.x + e.y) + (e.z + e.w));
        *(f32x4*)(prob + t * PS + 4 * h) = t < Ta ? e * inv : zero4;
    }
    __syncthreads();
    // ---- combination + residual: y[t][n .. n + 3] = old + bo + sum_p P[t][p] Wf[p][n ..] ----
    {
        const float* pp = prob + (lane & 15) * PS + 4 * (lane >> 4);
        f32x4 pv[KP16];
#pragma unroll
        for (int kc = 0; kc < KP16; ++kc) pv[kc] = *(const f32x4*)(pp + kc * 16);
        float* yb = (a.y_out != nullptr ? a.y_out : a.y) + yoff + (int64_t)(lane & 15) * D;
#pragma unroll
        for (int j = 0; j < NTWMAX; ++j)
            if (j < NTW) {
                f32x4 acc = zero4;
#pragma unroll
                for (int kc = 0; kc < KP16; ++kc)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q.wf[j][kc][e], pv[kc][e], acc, 0, 0, 0);
                const f32x4 yn = yold[j] + (q.bo[j] + acc);
                if constexpr (OL) *(f32x4*)(yo + (lane & 15) * yos + (wave * NTW + j) * 16 + nq) = yn;
                if ((lane & 15) < Ta && wr) st4(yb + (wave * NTW + j) * 16 + nq, yn);
            }
    }
    if constexpr (YL) { MDT_TS(6) } else { MDT_TS(4) }
}

// ------------------------------------------------------------------------------------------------
// attn_xattn_tile: ONE SAMPLE per workgroup through the middle of a decoder block (k_attn_xattn; batches that give every CU
// one sample): causal self-attention of the sample's T rows (attn_stage_tile, all 8 heads in one pass, no halo) -> output
// projection on the MFMA pipe (one 16-row tile, wave w owns columns [w D/8, (w+1) D/8)) + gate + residual -> the rows stay in
// LDS -> ln3 and the collapsed cross-attention on them (xattn_tile<.., YL>) -> the residual stream is written ONCE.
// Replaces k_attn_proj_wide + k_xattn_apply: one launch, no round trip of the rows between them, and the sample's folded
// operands (98 KB, cold: another workgroup used them one step ago) travel while the projection and its epilogue run.
//   a = the projection's arguments (A unused: the activation tile is computed here), at = q | k | v rows, x = the cross-
//   attention's arguments (x.y == a.out).  512 threads; D = 8 * HD.
//   the q | k | v rows of the sample are one contiguous block (at.ldq == 3 D).
//   lds: 16 * (D + 4) + 48 * (D + 16) floats (the attention's rows; afterwards the projected rows + xattn_tile's scratch).
// ------------------------------------------------------------------------------------------------
template <int HD, int TKC, int NPP>
__device__ __forceinline__ void attn_xattn_tile(const mdt_gemm_args& a, const mdt_attn_pro& at, const mdt_xapply_args& x, int b,
                                                float* lds, const float* __restrict__ zeros, int tid) {
    constexpr int MTILES = 1, NTW = HD / 16, D = 8 * HD, K16 = D / 16;
    constexpr bool KSTEP_PRIO = false;
    constexpr int R = NTW == 1 ? 6 : (NTW == 2 ? 4 : 3);   // deeper rings change nothing here (measured: R = 4, 6)
    MDT_TS(0)
    MDT_TS_HWID()
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const int T = at.T, m0 = b * T;
    constexpr int stride = D + 4;
    float* xa = lds;                       // [16][stride]: the attention output = the projection's activation tile
    float* scr = xa + 16 * stride;         // 16 x (3 D + 16): q | k | v rows; then yl [16][stride] and xattn_tile's scratch
    float* yl = scr;
    float* xscr = scr + 16 * stride;

    const int nt0 = wave * NTW;
    WStream wp[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) wp[j] = wstream(a.Wp, (int64_t)(nt0 + j) * K16 * 256 + lane * 4);
    f32x4 ring[R][NTW];
    f32x4 acc[MTILES][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[0][j] = zero4;
    const int nq = 4 * (lane >> 4);
    const bool gated = a.gate_off >= 0;
    int ncol[NTW];
    f32x4 bias_v[NTW], gate_v[NTW], res_v[NTW];
    mdt_xattn_req<NPP, D> xq;

    // ---- the sample's causal self-attention -> xa (rows >= T zero).  Behind its q / k / v requests: the projection's first
    //      weight fragments and its epilogue operands ----
    attn_sample_tile<HD, TKC>(at, xa, stride, m0, scr, tid, [&]() {
#pragma unroll
        for (int u = 0; u < R - 1; ++u)
#pragma unroll
            for (int j = 0; j < NTW; ++j) ring[u][j] = wld4(wp[j] + min(u, K16 - 1) * 256);
        const float* biasp = a.bias != nullptr ? a.bias : zeros;
        const int m = min(m0 + min(lane & 15, T - 1), a.M - 1);
        const float* gp = zeros;
        if (gated) gp = a.mod + a.gate_off + (a.mod_stride == 0 ? 0 : (int64_t)(m / a.rows_per_sample) * a.mod_stride);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            ncol[j] = (nt0 + j) * 16 + nq;
            bias_v[j] = ldg4(biasp + ncol[j]);
            gate_v[j] = ldg4(gp + ncol[j]);
            res_v[j] = ldg4(a.out + (int64_t)m * a.ldo + ncol[j]);
        }
    });
    MDT_TS(1)
    __syncthreads();
    MDT_TS(2)
    // The cross-attention's fragments of the sample's folded operands are cold (98 KB per sample that another workgroup used
    // one step ago: 25 MB per launch).  Requested at entry they held the q / k / v rows back (rows in LDS after 13.7 k cycles
    // instead of 7.6 k: the whole chip asks for 41 MB at once); requested here, the score half travels under the projection
    // (vmcnt retires in order: the k loop's first fragment waits ~1 k cycles behind it) ...
    xattn_request_u<NPP, D, D>(x, b, tid, xq);

    // ---- projection: 16 x D x D, transposed-form MFMA k-steps on the ring ----
    {
        int kg = 0;
        const int nk = K16;
        const float* ap = xa + (lane & 15) * stride + 4 * (lane >> 4);
        f32x4 av[MTILES];
        av[0] = *(const f32x4*)ap;
        int kc = 0;
        for (; kc + R <= nk; kc += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) MDT_KSTEP(u, kc + u)
        }
        if (kc < nk) {
            const int rem = nk - kc;
#pragma unroll
            for (int u = 0; u < R - 1; ++u)
                if (u < rem) MDT_KSTEP(u, kc + u)
        }
        (void)kg;
    }
    MDT_TS(3)
    // ... and the combination half + the ln3 / bias vectors behind the k loop's last weight fragment: nothing queues behind
    // them, and they have the epilogue, ln3, the scores and the softmax to arrive
    xattn_request_wf<NPP, D, D>(x, b, tid, xq);
    xattn_request_vec<NPP, D, D>(x, zeros, tid, xq);
    // ---- epilogue: out = old + gate * (acc + bias) -> LDS rows (lane holds row lane % 16, columns ncol .. ncol + 3) ----
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        f32x4 v = acc[0][j] + bias_v[j];
        v = res_v[j] + (gated ? gate_v[j] * v : v);
        *(f32x4*)(yl + (lane & 15) * stride + ncol[j]) = v;
    }
    __syncthreads();
    MDT_TS(4)
    // ---- ln3 -> scores against U -> masked softmax -> Wf combination -> residual -> x.y ----
    xattn_tile<NPP, false, true, D, D>(x, b, xscr, zeros, tid, yl, stride, &xq);
}
