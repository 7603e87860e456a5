// mdt_mlp_split.h -- the fused MLP launch (mdt_tiles.h mlp_tile) in the THREE-WAY bf16 SPLIT form (round 6).
//
// Same work per workgroup as mlp_tile -- row tile `by` (32 rows) x hidden slice `s` (512 of the 4 D hidden columns):
//   x + gate * (act(prologue(x) W1^T + b1) W2^T + b2),  partial slab s of the second product to parts + s * part_stride --
// with every contraction as six v_mfma_f32_16x16x32_bf16 products per k32 step (x = x1 + x2 + x3, w = w1 + w2 + w3 in bf16,
// products x3 w1, x2 w2, x2 w1, x1 w3, x1 w2, x1 w1, fp32 accumulation: fp32's product accuracy, mdt_ws.h) instead of eight
// v_mfma_f32_16x16x4_f32: 96 matrix-pipe clocks per k32 step and accumulator instead of 256.
//
// Where the operands come from:
//   weights     PRE-SPLIT images (k_pack_weight_split: made wherever the fp32 fragment image is made -- mdt_load_param(s), i.e. also
//               after every optimizer step): fragment (column tile nt, k32 step kk, part p) = 1 KiB at ((nt K32 + kk) 3 + p) 1024,
//               lane l's 16 bytes = the part's eight bf16 of W[16 nt + l % 16][32 kk + 16 h + 4 (l / 16) + e], slot 4 h + e.  A wave
//               streams its fragments L2 -> registers as mlp_tile does (ring of two k32 steps): 1.5x the bytes of the fp32 image in
//               a third of the matrix-pipe time -- 49 B per clock and CU of the vector memory path's 64 at full rate (fp32: 14);
//               tools/micro/mlp_split_probe.hip measured the two phases with this traffic at 31 us against k_mlp's 52-58.
//   activations the LayerNorm (+ modulate) prologue of mlp_tile (gemm_stage_tile) stores its rows as their split, in the MFMA slot
//               order of mdt_ws.h (one ds_read_b128 per part, row tile and k32 step);
//   hidden      the activation epilogue of the first product splits its values on the way into LDS, same slot order.
// LDS: split x tile 3 x 32 x (2 D + 32) B, overlaid after the first product by the split hidden slice 3 x 32 x 1056 B = 101 KB
// (D <= 512: 101 KB either way).
// No wave skew here (mlp_tile's flags): the hidden slice overlays the x tile, so the two products are separated by barriers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdt_device.h"
#include "mdt_internal.h"
#include "mdt_tiles.h"
#include "mdt_ws.h"   // mdt_bf16x8 / mdt_bf16x4, split3_bf16

// ring of R k32 steps x NT column tiles x 3 parts of weight fragments.  Fragment (j, kk, p) of the wave sits at byte
// base + j * tile_stride + (kk * 3 + p) * 1024 of the image, lane l's 16 bytes at l * 16.  BUFFER loads (resource = the image, one
// 32-bit per-lane offset per column tile, the step / part offset in the scalar operand): a global_load sends 64 x 8 bytes of address
// through the SIMD's register read path, 33 clocks of matrix-pipe time against 17.7 (mdt_tiles.h WStream) -- and this form requests
// 1.5x the fragments of the fp32 one per step.
template <int NT, int R>
struct SplitRing {
    mdt_bf16x8 w[R][NT][3];
    __amdgpu_buffer_rsrc_t rs;
    const char* gbase;
    unsigned voff[NT];
    __device__ __forceinline__ void open(const void* image, int64_t first_byte, int64_t tile_stride, int lane) {
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)image, 0, 0xffffffffu, 0x00020000);
        gbase = (const char*)image;
#pragma unroll
        for (int j = 0; j < NT; ++j) voff[j] = (unsigned)(first_byte + j * tile_stride) + 16u * (unsigned)lane;
    }
    __device__ __forceinline__ void request(int slot, int kk) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#ifdef MDT_SPLIT_W_GLOBAL   // A/B build: 64-bit global loads
                w[slot][j][p] = __builtin_bit_cast(mdt_bf16x8, *(const f32x4*)(gbase + voff[j] + (kk * 3 + p) * 1024));
#else
                w[slot][j][p] = __builtin_bit_cast(mdt_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], (kk * 3 + p) * 1024, 0));
#endif
    }
};

// one product phase: acc[i][j] += sum over K32 k32 steps; ring slots 0 .. R - 2 hold steps 0 .. R - 2 on entry (requested by the
// caller, early); activations from the split tile at `xa` (row stride rowb, part
// stride part)
template <int NT, int K32, int R>
__device__ __forceinline__ void split_phase(SplitRing<NT, R>& ring, const char* xa, int rowb, int part, int lane, f32x4 (&acc)[2][NT]) {
    const int aoff = (lane & 15) * rowb + (lane >> 4) * 16;
    mdt_bf16x8 x1[2], x2[2], x3[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const char* q = xa + aoff + i * 16 * rowb;
        x1[i] = *(const mdt_bf16x8*)q; x2[i] = *(const mdt_bf16x8*)(q + part); x3[i] = *(const mdt_bf16x8*)(q + 2 * part);
    }
#pragma unroll
    for (int kk = 0; kk < K32; ++kk) {
        const int u = kk % R, un = (kk + R - 1) % R;
        const bool nx = kk + 1 < K32;
        if (kk + R - 1 < K32) ring.request(un, kk + R - 1);
        const char* p0 = xa + aoff + (kk + 1) * 64;
        const char* p1 = p0 + 16 * rowb;
        MDT_SCHED_PIN
        // the six products of this k32 step, smallest first; each activation part of the NEXT step is read into this step's registers
        // right behind its last use (mdt_ws.h)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][0], x3[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][0], x3[1], acc[1][j], 0, 0, 0);
        }
        MDT_SCHED_PIN
        if (nx) { x3[0] = *(const mdt_bf16x8*)(p0 + 2 * part); x3[1] = *(const mdt_bf16x8*)(p1 + 2 * part); }
        MDT_SCHED_PIN
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][1], x2[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][1], x2[1], acc[1][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][0], x2[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][0], x2[1], acc[1][j], 0, 0, 0);
        }
        MDT_SCHED_PIN
        if (nx) { x2[0] = *(const mdt_bf16x8*)(p0 + part); x2[1] = *(const mdt_bf16x8*)(p1 + part); }
        MDT_SCHED_PIN
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][2], x1[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][2], x1[1], acc[1][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][1], x1[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][1], x1[1], acc[1][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][0], x1[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.w[u][j][0], x1[1], acc[1][j], 0, 0, 0);
        }
        MDT_SCHED_PIN
        if (nx) { x1[0] = *(const mdt_bf16x8*)p0; x1[1] = *(const mdt_bf16x8*)p1; }
        MDT_SCHED_PIN
    }
}

// LDS bytes of mlp_split_tile at model width D
__host__ __device__ constexpr int mlp_split_lds_bytes(int D) {
    const int xs = 3 * 32 * (2 * D + 32), hs = 3 * 32 * (2 * 512 + 32);
    return xs > hs ? xs : hs;
}

template <int NTW2, int PRO>
__device__ __forceinline__ void mlp_split_tile(const mdt_gemm_args& f, const mdt_gemm_args& p, const char* __restrict__ w1s,
                                               const char* __restrict__ w2s, float* __restrict__ parts, int64_t part_stride, int by,
                                               int s, char* lds, const float* __restrict__ zeros, int tid) {
    constexpr int MTILES = 2, NWAVES = 8, MT = 32, HS = 512, NTW1 = 4, R = 2;
    constexpr int D = 128 * NTW2, K32a = D / 32, K32b = HS / 32;
    constexpr int ROWB1 = 2 * D + 32, PART1 = MT * ROWB1, ROWB2 = 2 * HS + 32, PART2 = MT * ROWB2;
    static_assert(NTW2 >= 1 && NTW2 <= 4, "D <= 512");
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const int m0 = by * MT;
    const int nq = 4 * (lane >> 4);
    char* xs = lds;                                   // split x tile: 3 parts x [32][ROWB1]
    char* hs = lds;                                   // split hidden slice: 3 parts x [32][ROWB2] (after the first product)

    // ---- first product's operands: the ring's first step and the bias travel under the prologue ----
    const int nt1 = (s * NWAVES + wave) * NTW1;       // hidden column tiles of this wave (global)
    SplitRing<NTW1, R> ring1;
    ring1.open(w1s, (int64_t)nt1 * K32a * 3072, (int64_t)K32a * 3072, lane);
#pragma unroll
    for (int u = 0; u < R - 1; ++u) ring1.request(u, u);
    f32x4 b1[NTW1];
    {
        const float* bp = f.bias != nullptr ? f.bias : zeros;
#pragma unroll
        for (int j = 0; j < NTW1; ++j) b1[j] = ldg4(bp + (nt1 + j) * 16 + nq);
    }
    // LayerNorm (+ modulate) of the tile's rows, stored as their split in MFMA slot order
    gemm_stage_tile<MTILES, NWAVES, PRO, false, 1, 2, true>(f, (float*)xs, ROWB1, m0, 0, D, zeros, tid, lane, wave);
    __syncthreads();
    f32x4 acc1[MTILES][NTW1];
#pragma unroll
    for (int i = 0; i < MTILES; ++i)
#pragma unroll
        for (int j = 0; j < NTW1; ++j) acc1[i][j] = zero4;
    split_phase<NTW1, K32a, R>(ring1, xs, ROWB1, PART1, lane, acc1);

    // ---- second product's operands, requested before the activation epilogue so that they travel while it runs: the first
    //      fragments of W2's K slice [512 s, 512 s + 512), then bias / gate / residual rows of the output tile ----
    const int K32p = p.K >> 5;                        // k32 steps of the whole second product (4 D / 32)
    SplitRing<NTW2, R> ring2;
    ring2.open(w2s, ((int64_t)(wave * NTW2) * K32p + (int64_t)s * K32b) * 3072, (int64_t)K32p * 3072, lane);
#pragma unroll
    for (int u = 0; u < R - 1; ++u) ring2.request(u, u);
    // (d = 512, four column tiles per wave: gate and residual rows are requested behind the second product instead -- 64 registers
    //  the product's ring and accumulators need, 116 bytes of scratch otherwise)
    constexpr bool EARLY = NTW2 <= 3;
    f32x4 acc2[MTILES][NTW2], b2[NTW2], gate_v[MTILES][NTW2], res_v[MTILES][NTW2];
    int ncol[NTW2];
#pragma unroll
    for (int j = 0; j < NTW2; ++j) ncol[j] = (wave * NTW2 + j) * 16 + nq;
    const bool gated = p.mod != nullptr && p.gate_off >= 0;
    auto load_gate_res = [&]() __attribute__((always_inline)) {
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
            }
        }
    };
    {
        const float* bp = (s == 0 && p.bias != nullptr) ? p.bias : zeros;
#pragma unroll
        for (int j = 0; j < NTW2; ++j) b2[j] = ldg4(bp + ncol[j]);
#pragma unroll
        for (int i = 0; i < MTILES; ++i)
#pragma unroll
            for (int j = 0; j < NTW2; ++j) acc2[i][j] = zero4;
        if constexpr (EARLY) load_gate_res();
    }
    __syncthreads();                                  // everybody has read the x tile: the hidden slice overwrites it
    // ---- activation epilogue of the first product -> split hidden slice (lane holds hidden[16 i + lane % 16][tile * 16 + nq .. + 3]) ----
#pragma unroll
    for (int i = 0; i < MTILES; ++i)
#pragma unroll
        for (int j = 0; j < NTW1; ++j) {
            mdt_bf16x4 p1, p2, p3;
            split3_bf16(apply_act(acc1[i][j] + b1[j], f.act), p1, p2, p3);
            char* q = hs + split_slot(i * 16 + (lane & 15), (wave * NTW1 + j) * 16 + nq, ROWB2);
            *(mdt_bf16x4*)q = p1;
            *(mdt_bf16x4*)(q + PART2) = p2;
            *(mdt_bf16x4*)(q + 2 * PART2) = p3;
        }
    __syncthreads();
    split_phase<NTW2, K32b, R>(ring2, hs, ROWB2, PART2, lane, acc2);

    if constexpr (!EARLY) load_gate_res();
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
}

// ------------------------------------------------------------------------------------------------
// The LayerNorm-prologue product on the wide tiles (gemm_tile<2, NTW, 8, PRO, ...>: 32 rows x 8 waves x NTW column tiles, whole K
// staged once) in the split form -- the decoder's qkv products (K = d = 384 -> N = 1152: 25 us a launch, 40 launches a sampler call),
// whose rows are the sum of the fused MLP's slabs (XP > 1: gemm_stage_tile adds them on read, the column-0 tile also leaves the sum
// in a.a_merged).  The first half of mlp_split_tile with a plain epilogue: out = act(product + bias), plain output rows.
// LDS: the split x tile (76.8 KB at D = 384).
// ------------------------------------------------------------------------------------------------
template <int ND, int NTW, int PRO, int XP>
__device__ __forceinline__ void gemm_ln_split_tile(const mdt_gemm_args& a, int by, int bx, char* lds, const float* __restrict__ zeros, int tid) {
    constexpr int MTILES = 2, NWAVES = 8, MT = 32, R = 2;
    constexpr int D = 128 * ND, K32 = D / 32, ROWB = 2 * D + 32, PART = MT * ROWB;
    static_assert(ND >= 1 && ND <= 4, "D <= 512");
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    const int m0 = by * MT, nq = 4 * (lane >> 4);
    char* xs = lds;
    const int nt0 = (bx * NWAVES + wave) * NTW;       // N is a multiple of the panel width: every wave has NTW real column tiles
    SplitRing<NTW, R> ring;
    ring.open(a.Wp_split, (int64_t)nt0 * K32 * 3072, (int64_t)K32 * 3072, lane);
#pragma unroll
    for (int u = 0; u < R - 1; ++u) ring.request(u, u);
    f32x4 bias_v[NTW];
    {
        const float* bp = a.bias != nullptr ? a.bias : zeros;
#pragma unroll
        for (int j = 0; j < NTW; ++j) bias_v[j] = ldg4(bp + (nt0 + j) * 16 + nq);
    }
    gemm_stage_tile<MTILES, NWAVES, PRO, false, XP, (NTW <= 3 && XP <= 3) ? 1 : 2, true>(a, (float*)xs, ROWB, m0, 0, D, zeros, tid, lane, wave,
                                                                                         bx == 0 ? a.a_merged : nullptr);
    __syncthreads();
    f32x4 acc[MTILES][NTW];
#pragma unroll
    for (int i = 0; i < MTILES; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = zero4;
    split_phase<NTW, K32, R>(ring, xs, ROWB, PART, lane, acc);
#pragma unroll
    for (int i = 0; i < MTILES; ++i) {
        const int m = m0 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            if (m < a.M) st4(a.out + (int64_t)m * a.ldo + (nt0 + j) * 16 + nq, apply_act(acc[i][j] + bias_v[j], a.act));
    }
}
