// mdt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the MDT action-denoising hot path.
//
// Design:
//   * every dense contraction runs on v_mfma_f32_16x16x4_f32 (exact fp32 MFMA; bf16/fp16 operands fail
//     the 1e-3 parity gate) in the TRANSPOSED form  D[n][m] = sum_k W[n][k] * X[m][k]:
//     the weight fragment is the MFMA "A" operand and the activation fragment the "B" operand, so that a
//     lane ends up with 4 CONSECUTIVE output columns of one row -> 16-byte epilogue loads/stores.
//   * weights are pre-packed once (k_pack_weight) into fragment-major order: one 16(n) x 16(k) block is
//     1 KiB, lane l holds W[n0 + l%16][k0 + 4*(l/16) .. +3].  A wave streams its weight fragments with
//     one fully coalesced global_load_dwordx4 per block straight into VGPRs (each weight element is used
//     by exactly one wave of a workgroup, so an LDS round trip would be pure overhead) and reuses them
//     across the workgroup's row tiles from registers.
//   * the MFMA k-index is a free permutation as long as both operands agree: lane group h = l/16 feeds
//     k = 4h + j to the j-th of 4 back-to-back MFMAs, so both operands are fetched as float4.
//   * activations (the small operand: B*10 rows) are staged per workgroup in LDS (row stride K+4 floats ->
//     ds_read_b128 fragments, at most a 2-way bank conflict on one lane group), with LayerNorm + adaLN
//     modulation fused into the staging pass and bias / GELU / Mish / SiLU / gate * residual fused into
//     the epilogue.
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "mdt_internal.h"

#include "mdt_device.h"

#define MDT_TILES_TIMING_OWNER  // this translation unit owns the -DMDT_DEBUG_TIMING stamp buffer
#include "mdt_tiles.h"  // the tile bodies
#include "mdt_tall.h"  // the tall LDS-staged GEMM body (round 4)
#include "mdt_ws.h"    // the weight-stationary GEMM body (round 5)
#include "mdt_mlp_split.h"  // the fused MLP launch in the three-way bf16 split form (round 6)

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_weight(const float* __restrict__ w, int n_rows, int K, float* __restrict__ packed,
                              int n_off, int K16) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_rows * K) return;
    const int n = (int)(idx / K), k = (int)(idx % K);
    const int nn = n + n_off;
    const int nt = nn >> 4, ni = nn & 15, kc = k >> 4, h = (k & 15) >> 2, j = k & 3;
    packed[(((int64_t)nt * K16 + kc) * 64 + (ni + 16 * h)) * 4 + j] = w[idx];
}

// SwishGLU project weight (2H, K): 16-row tile 2t of the image = projected rows [16t, +16), tile 2t+1 = gate rows [H + 16t, +16)
__global__ void k_pack_weight_glu(const float* __restrict__ w, int H, int K, float* __restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K4 = K >> 2;
    if (idx >= (int64_t)2 * H * K4) return;
    const int r = (int)(idx / K4), c = 4 * (int)(idx - (int64_t)r * K4);
    const int half = r >= H, rr = r - half * H;
    const int nt = 2 * (rr >> 4) + half, ni = rr & 15, kc = c >> 4, h = (c & 15) >> 2;
    *(f32x4*)(packed + (((int64_t)nt * (K >> 4) + kc) * 64 + (ni + 16 * h)) * 4) = *(const f32x4*)(w + (int64_t)r * K + c);
}
hipError_t mdt_launch_pack_weight_glu(const float* w, int H, int K, float* packed, hipStream_t s) {
    const int64_t n = (int64_t)2 * H * (K >> 2);
    hipLaunchKernelGGL(k_pack_weight_glu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, H, K, packed);
    return hipGetLastError();
}

// three-way bf16 split fragment image (mdt_mlp_split.h): thread = 4 consecutive k of one row -> eight bytes in each of the three
// parts of fragment (row tile r / 16, k32 step c / 32): lane (r % 16) + 16 ((c % 16) / 4), half (c % 32) / 16
__device__ __forceinline__ void pack_split_quad(const float* __restrict__ src, char* __restrict__ image, int rs, int c, int K, int n_off) {
    mdt_bf16x4 p1, p2, p3;
    split3_bf16(*(const f32x4*)(src + (int64_t)rs * K + c), p1, p2, p3);
    const int r = rs + n_off;   // row of the image
    char* q = image + (((int64_t)(r >> 4) * (K >> 5) + (c >> 5)) * 3) * 1024 + ((r & 15) + 16 * ((c & 15) >> 2)) * 16 + ((c & 31) >> 4) * 8;
    *(mdt_bf16x4*)q = p1;
    *(mdt_bf16x4*)(q + 1024) = p2;
    *(mdt_bf16x4*)(q + 2048) = p3;
}
__global__ void k_pack_weight_split(const float* __restrict__ w, int n_rows, int K, char* __restrict__ image, int n_off) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K4 = K >> 2;
    if (idx >= (int64_t)n_rows * K4) return;
    const int r = (int)(idx / K4), c = 4 * (int)(idx - (int64_t)r * K4);
    pack_split_quad(w, image, r, c, K, n_off);
}
hipError_t mdt_launch_pack_weight_split(const float* w, int n_rows, int K, void* image, hipStream_t s, int n_off) {
    if (K % 32) return hipErrorInvalidValue;   // (row counts / offsets that are not multiples of 16 leave the other rows of a tile alone)
    const int64_t n = (int64_t)n_rows * (K >> 2);
    hipLaunchKernelGGL(k_pack_weight_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, n_rows, K, (char*)image, n_off);
    return hipGetLastError();
}

// the split image straight from the fp32 fragment image of the same weight (no raw weight needed): lane l of split fragment
// (nt, kk, part) = the part of lane l's quads of fp32 fragments (nt, 2 kk) and (nt, 2 kk + 1)
__global__ __launch_bounds__(256) void k_split_from_packed(const float* __restrict__ wp, int n_frag, int K32, char* __restrict__ image) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (nt * K32 + kk) * 64 + lane
    if (idx >= (int64_t)n_frag * 64) return;
    const int lane = (int)(idx & 63);
    const int64_t f = idx >> 6;                                        // nt * K32 + kk
    const f32x4 lo = *(const f32x4*)(wp + ((2 * f) * 64 + lane) * 4), hi = *(const f32x4*)(wp + ((2 * f + 1) * 64 + lane) * 4);
    mdt_bf16x4 l1, l2, l3, h1, h2, h3;
    split3_bf16(lo, l1, l2, l3);
    split3_bf16(hi, h1, h2, h3);
    char* q = image + f * 3072 + lane * 16;
    *(mdt_bf16x8*)q = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *(mdt_bf16x8*)(q + 1024) = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
    *(mdt_bf16x8*)(q + 2048) = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
}
hipError_t mdt_launch_split_from_packed(const float* wp, int N, int K, void* image, hipStream_t s) {
    if (N % 16 || K % 32) return hipErrorInvalidValue;
    const int n_frag = (N / 16) * (K / 32);
    hipLaunchKernelGGL(k_split_from_packed, dim3((unsigned)((n_frag * 64 + 255) / 256)), dim3(256), 0, s, wp, n_frag, K / 32, (char*)image);
    return hipGetLastError();
}

// Every parameter image of one load_state_dict / optimizer step in ONE launch (mdt_load_params): a table of moves --
// raw copies, fragment packs, transposed fragment packs (training), transposes, column pads -- and a (move, chunk) list,
// one workgroup per 1024 source elements.  The per-parameter launches this replaces (~230 of 3-6 us for MDT-V) were
// ~0.9 ms of every training step.
__global__ __launch_bounds__(256) void k_multi_load(const mdt_load_entry* __restrict__ tab, const int2* __restrict__ blocks) {
    const int2 bk = blocks[blockIdx.x];
    const mdt_load_entry e = tab[bk.x];
    const int64_t n = (int64_t)e.rows * e.K;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t idx = (int64_t)bk.y * 1024 + threadIdx.x + 256 * u;
        if (e.kind == MDT_LOAD_PACK_T) {
            // thread = (4 consecutive source rows, column): one 16-byte fragment slot of the W^T image (k_pack_weight_t4)
            const int rows4 = (e.rows + 3) >> 2;
            if (idx >= (int64_t)rows4 * e.K) continue;
            const int r4 = (int)(idx / e.K), c = (int)(idx - (int64_t)r4 * e.K);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * r4 + j;
                v[j] = r < e.rows ? e.src[(int64_t)r * e.K + c] : 0.f;
            }
            const int k = e.p0 + 4 * r4;
            const int nt = c >> 4, ni = c & 15, kc = k >> 4, h = (k & 15) >> 2;
            *(f32x4*)(e.dst + (((int64_t)nt * e.p1 + kc) * 64 + (ni + 16 * h)) * 4) = v;
            continue;
        }
        if (e.kind == MDT_LOAD_PACK_SPLIT) {
            const int K4 = e.K >> 2;
            if (idx >= (int64_t)e.rows * K4) continue;
            const int r = (int)(idx / K4), c = 4 * (int)(idx - (int64_t)r * K4);
            pack_split_quad(e.src, (char*)e.dst, r, c, e.K, e.p0);
            continue;
        }
        if (e.kind == MDT_LOAD_PACK) {
            // thread = 4 consecutive k of one row: a 16-byte read and one 16-byte fragment slot (K is a multiple of 16)
            const int K4 = e.K >> 2;
            if (idx >= (int64_t)e.rows * K4) continue;
            const int r = (int)(idx / K4), c = 4 * (int)(idx - (int64_t)r * K4);
            const int nn = r + e.p0;
            const int nt = nn >> 4, ni = nn & 15, kc = c >> 4, h = (c & 15) >> 2;
            *(f32x4*)(e.dst + (((int64_t)nt * (e.K >> 4) + kc) * 64 + (ni + 16 * h)) * 4) = *(const f32x4*)(e.src + (int64_t)r * e.K + c);
            continue;
        }
        if (idx >= n) continue;
        const float v = e.src[idx];
        if (e.kind == MDT_LOAD_RAW) {
            e.dst[idx] = v;
        } else {
            const int r = (int)(idx / e.K), c = (int)(idx - (int64_t)r * e.K);
            if (e.kind == MDT_LOAD_TRANSPOSE) {
                e.dst[(int64_t)c * e.rows + r] = v;
            } else {  // MDT_LOAD_PAD_COLS: row pitch p0
                e.dst[(int64_t)r * e.p0 + c] = v;
            }
        }
    }
}
hipError_t mdt_launch_multi_load(const mdt_load_entry* tab, const int2* blocks, int n_blocks, hipStream_t s) {
    if (n_blocks < 1) return hipSuccess;
    hipLaunchKernelGGL(k_multi_load, dim3(n_blocks), dim3(256), 0, s, tab, blocks);
    return hipGetLastError();
}

#ifdef MDT_DEBUG_TIMING
extern "C" int mdt_debug_set_timing_buffer(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_ts), &p, sizeof(p));
}
#endif

// ------------------------------------------------------------------------------------------------
// fused GEMM:  out = epilogue( prologue(A) @ W^T ); one workgroup per tile, body in mdt_tiles.h (gemm_tile)
// ------------------------------------------------------------------------------------------------
template <int MTILES, int NTW, int NWAVES, int PRO, bool RES>
__global__ __launch_bounds__(64 * NWAVES) void k_gemm(mdt_gemm_args a, int kchunk, int grid_n,
                                                      const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (blockIdx.z) {  // batched launch (split-K partial products): every operand advances by its batch stride
        a.A += (int64_t)blockIdx.z * a.bs_a; a.Wp += (int64_t)blockIdx.z * a.bs_w; a.out += (int64_t)blockIdx.z * a.bs_out;
    }
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical / grid_n, bx = logical - by * grid_n;
    gemm_tile<MTILES, NTW, NWAVES, PRO, RES, false>(a, kchunk, by, bx, lds, zeros, threadIdx.x);
}

// SwishGLU riding on a plain-prologue product's epilogue (mdt_gemm_args.aux_mode 3 forward / 4 backward): own instantiations
template <int MTILES, int NTW, int NWAVES, int GLU>
__global__ __launch_bounds__(64 * NWAVES) void k_gemm_glu(mdt_gemm_args a, int kchunk, int grid_n, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical / grid_n, bx = logical - by * grid_n;
    gemm_tile<MTILES, NTW, NWAVES, PRO_PLAIN, false, false, 1, GLU>(a, kchunk, by, bx, lds, zeros, threadIdx.x);
}

// the same wide tile reading its rows as the sum of XP partial slabs (the output of a fused MLP launch, k_mlp below)
template <int NTW, int PRO, int XP>
__global__ __launch_bounds__(512) void k_gemm_merge(mdt_gemm_args a, int kchunk, int grid_n, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical / grid_n, bx = logical - by * grid_n;
    gemm_tile<2, NTW, 8, PRO, false, false, XP>(a, kchunk, by, bx, lds, zeros, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// k_mlp: the MLP sublayer as one launch; workgroup = (row tile of 32, 512-wide slice of the hidden layer); body in
// mdt_tiles.h (mlp_tile).  The S slices of a row tile are neighbours in the logical order, so they share an XCD's L2.
// ------------------------------------------------------------------------------------------------
template <int NTW2, int PRO>
__global__ __launch_bounds__(512) void k_mlp(mdt_gemm_args f, mdt_gemm_args p, float* __restrict__ parts, int64_t part_stride,
                                             int n_slices, const float* __restrict__ zeros, int skew) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical / n_slices, s = logical - by * n_slices;
    mlp_tile<NTW2, PRO>(f, p, parts, part_stride, by, s, lds, zeros, threadIdx.x, skew);
}

// k_mlp_split: the same launch in the three-way bf16 split form (mdt_mlp_split.h).  Workgroups in SLICE-major order, XCD x taking a
// contiguous range of it (block b runs on XCD b % 8): an XCD's L2 then holds one or two slices of the split weight images (2.4 MB
// each at d = 384) instead of all of them.  grid = 8 * ceil(tiles * slices / 8).
template <int NTW2, int PRO>
__global__ __launch_bounds__(512) void k_mlp_split(mdt_gemm_args f, mdt_gemm_args p, const char* __restrict__ w1s, const char* __restrict__ w2s,
                                                   float* __restrict__ parts, int64_t part_stride, int n_slices, int gm,
                                                   const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) char lds_c[];
    const int total = gm * n_slices, per = (total + 7) >> 3;
    const int i = blockIdx.x >> 3, w = (blockIdx.x & 7) * per + i;
    if (i >= per || w >= total) return;
    const int s = w / gm, by = w - s * gm;
    mlp_split_tile<NTW2, PRO>(f, p, w1s, w2s, parts, part_stride, by, s, lds_c, zeros, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// k_gemm_pipe: the plain-prologue GEMM for K that does not fit LDS in one piece (mlp.c_proj, K = 4d), with
// LW extra LOADER waves.  vmcnt retires a wave's loads in order, so an MFMA wave that also fetched the next
// activation chunk would stall its weight ring behind those loads; here the NWAVES compute waves only ever
// wait on weight fragments while the loader waves copy chunk c+1 into the other LDS buffer.  One workgroup
// barrier per chunk.
// ------------------------------------------------------------------------------------------------
template <int MTILES, int NTW, int NWAVES, int LW, bool RES>
__global__ __launch_bounds__(64 * (NWAVES + LW)) void k_gemm_pipe(mdt_gemm_args a, int kchunk, int grid_n,
                                                                 const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int MT = MTILES * 16;
    constexpr bool KSTEP_PRIO = true;  // (a priority raised once for the waves' whole life measured 1.5 % slower)
    constexpr int R = (NTW == 1 ? 6 : (NTW == 2 ? 4 : 3)) + MDT_RING_ADD;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= NWAVES;
    if (blockIdx.z) {  // batched launch (split-K partial products): every operand advances by its batch stride
        a.A += (int64_t)blockIdx.z * a.bs_a; a.Wp += (int64_t)blockIdx.z * a.bs_w; a.out += (int64_t)blockIdx.z * a.bs_out;
    }
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical / grid_n, bx = logical - by * grid_n;
    const int m0 = by * MT;
    const int N16 = a.N >> 4, K16 = a.K >> 4;
    const int nt0 = (bx * NWAVES + min(wave, NWAVES - 1)) * NTW;
    const bool active = !loader && nt0 < N16;
    const int stride = kchunk + 4;
    const int bufsz = MT * stride;       // floats per LDS buffer
    const int nchunks = (a.K + kchunk - 1) / kchunk;

    // loader: 32 lanes sweep a row in 512-byte pieces, 2*LW rows per sweep (kchunk <= 384 -> 3 pieces)
    auto stage = [&](int c) {
        constexpr int RG = 2 * LW, U = MT / RG;
        const int lt = tid - 64 * NWAVES, rg = lt >> 5, l32 = lt & 31;
        const int k0 = c * kchunk, n4 = min(kchunk, a.K - k0) >> 2;
        float* dst = lds + (c & 1) * bufsz;
        f32x4 st[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = min(m0 + rg + RG * u, a.M - 1);
#pragma unroll
            for (int v = 0; v < 3; ++v) st[u][v] = ldg4(a.A + m * a.lda + k0 + 4 * min(l32 + 32 * v, n4 - 1));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = rg + RG * u;
#pragma unroll
            for (int v = 0; v < 3; ++v)
                if (l32 + 32 * v < n4) *(f32x4*)(dst + row * stride + 4 * (l32 + 32 * v)) = sel4(m0 + row < a.M, st[u][v], zero4);
        }
    };

    WStream wp[NTW];
    f32x4 ring[R][NTW];
    f32x4 acc[MTILES][NTW];
    const int nq = 4 * (lane >> 4);
    const bool gated = RES && a.gate_off >= 0;
    int ncol[NTW];
    constexpr int NRES = RES ? NTW : 1;
    f32x4 bias_v[NTW], gate_v[MTILES][NRES], res_v[MTILES][NRES];
    float* optr[MTILES];
    if (loader) {
        stage(0);
    } else {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int nt = min(nt0 + j, N16 - 1);
            wp[j] = wstream(a.Wp, (int64_t)nt * K16 * 256 + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < R - 1; ++u)
#pragma unroll
            for (int j = 0; j < NTW; ++j) ring[u][j] = wld4(wp[j] + min(u, K16 - 1) * 256);
#pragma unroll
        for (int i = 0; i < MTILES; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = zero4;
        const float* biasp = a.bias != nullptr ? a.bias : zeros;
        const float* rvp = a.rowvec != nullptr ? a.rowvec : zeros;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            ncol[j] = min(nt0 + j, N16 - 1) * 16 + nq;
            bias_v[j] = ldg4(biasp + ncol[j]) + ldg4(rvp + ncol[j]);
        }
#pragma unroll
        for (int i = 0; i < MTILES; ++i) {
            const int m = min(m0 + i * 16 + (lane & 15), a.M - 1);
            const int64_t orow =
                a.gin == 1 ? (int64_t)m * a.gout + a.goff : (int64_t)(m / a.gin) * a.gout + (m % a.gin) + a.goff;
            optr[i] = a.out + orow * a.ldo;
            if constexpr (RES) {
                const float* gp = zeros;
                if (gated)
                    gp = a.mod + a.gate_off + (a.mod_stride == 0 ? 0 : (int64_t)(m / a.rows_per_sample) * a.mod_stride);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    gate_v[i][j] = ldg4(gp + ncol[j]);
                    res_v[i][j] = ldg4(optr[i] + ncol[j]);
                }
            }
        }
    }
    __syncthreads();
    int kg = 0;
    for (int c = 0; c < nchunks; ++c) {
        if (loader) {
            if (c + 1 < nchunks) stage(c + 1);
        } else if (active) {
            const int nk = min(kchunk, a.K - c * kchunk) >> 4;
            const float* ap = lds + (c & 1) * bufsz + (lane & 15) * stride + 4 * (lane >> 4);
            f32x4 av[MTILES];
#pragma unroll
            for (int i = 0; i < MTILES; ++i) av[i] = *(const f32x4*)(ap + i * 16 * stride);
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
        __syncthreads();
    }
    if (!active) return;
#pragma unroll
    for (int i = 0; i < MTILES; ++i) {
        const bool mok = m0 + i * 16 + (lane & 15) < a.M;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            f32x4 v = apply_act(acc[i][j] + bias_v[j], a.act);
            if constexpr (RES) v = res_v[i][j] + (gated ? gate_v[i][j] * v : v);
            if (mok && nt0 + j < N16) st4(optr[i] + ncol[j], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_attn_proj_smallm: a sample's self-attention fused into its output projection (rollout batches: B = 1 and a few more;
// one workgroup per 16 output columns and sample); body in mdt_tiles.h (attn_proj_tile)
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(512) void k_attn_proj_smallm(mdt_gemm_args a, const float* __restrict__ qkv, int64_t ldq, int T,
                                                         int causal, float scale, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    attn_proj_tile<HD, false>(a, qkv, ldq, T, causal, scale, blockIdx.x, blockIdx.y, lds, red, zeros, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// k_attn_proj_wide: the self-attention output projection of a LARGE batch with the (causal) attention of the tile's rows
// computed in its prologue (mdt_tiles.h: attn_stage_tile) -- replaces k_attn + the projection GEMM (20 us -> one launch per
// decoder block at B = 256).  32 x 128 tiles, 8 waves, residual epilogue (out += gate * value).
// ------------------------------------------------------------------------------------------------
template <int HD, int TKC>
__global__ __launch_bounds__(512) void k_attn_proj_wide(mdt_gemm_args a, mdt_attn_pro ap, int grid_n, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical / grid_n, bx = logical - by * grid_n;
    gemm_tile<2, 1, 8, PRO_ATTN, true, false, 1, 0, HD, TKC>(a, a.K, by, bx, lds, zeros, threadIdx.x, &ap);
}

// ------------------------------------------------------------------------------------------------
// k_attn_xattn: one workgroup per SAMPLE through self-attention, its output projection and the collapsed cross-attention
// (mdt_tiles.h: attn_xattn_tile) -- replaces k_attn_proj_wide + k_xattn_apply for batches of at most one sample per CU.
// ------------------------------------------------------------------------------------------------
template <int HD, int TKC>
__global__ __launch_bounds__(512) void k_attn_xattn(mdt_gemm_args a, mdt_attn_pro at, mdt_xapply_args x, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    attn_xattn_tile<HD, TKC, 32>(a, at, x, MDT_SAMPLE_REMAP(blockIdx.x, gridDim.x), lds, zeros, threadIdx.x);
}

// 256 KiB of zeros per device: stands in for absent bias / rowvec / LayerNorm-bias vectors.  ensure_zeros() points
// g_zeros at the CURRENT device's buffer (a process normally drives one GPU; a second one gets its own buffer).
static const int ZEROS_FLOATS = 65536;
static const int MAX_DEVICES = 32;
static float* g_zeros_dev[MAX_DEVICES] = {nullptr};
static thread_local float* g_zeros = nullptr;

static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 0;
    return dev;
}

static hipError_t ensure_zeros() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (g_zeros_dev[dev] == nullptr) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (g_zeros_dev[dev] == nullptr) {
            float* p = nullptr;
            e = hipMalloc((void**)&p, ZEROS_FLOATS * sizeof(float));
            if (e != hipSuccess) return e;
            e = hipMemset(p, 0, ZEROS_FLOATS * sizeof(float));
            if (e != hipSuccess) return e;
            g_zeros_dev[dev] = p;
        }
    }
    g_zeros = g_zeros_dev[dev];
    return hipSuccess;
}

const float* mdt_zeros() { return ensure_zeros() == hipSuccess ? g_zeros : nullptr; }

template <int MTILES, int NTW, int NWAVES, int PRO, bool RES>
static hipError_t launch_gemm_r(const mdt_gemm_args& a, int kchunk, hipStream_t s) {
    const int MT = MTILES * 16, NTC = NWAVES * NTW * 16;
    const int gn = (a.N + NTC - 1) / NTC, gm = (a.M + MT - 1) / MT;
    const size_t lds = (size_t)MT * (kchunk + 4) * sizeof(float);
    static size_t lds_attr_dev[MAX_DEVICES] = {0};  // per instantiation and per device (function attributes are per device)
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm<MTILES, NTW, NWAVES, PRO, RES>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_gemm<MTILES, NTW, NWAVES, PRO, RES>), dim3(gn * gm, 1, a.batch > 1 ? a.batch : 1), dim3(64 * NWAVES),
                       lds, s, a, kchunk, gn, g_zeros);
    return hipGetLastError();
}

template <int MTILES, int NTW, int NWAVES, int GLU>
static hipError_t launch_gemm_glu(const mdt_gemm_args& a, int kchunk, hipStream_t s) {
    const int MT = MTILES * 16, NTC = NWAVES * NTW * 16;
    const int gn = (a.N + NTC - 1) / NTC, gm = (a.M + MT - 1) / MT;
    const size_t lds = (size_t)MT * (kchunk + 4) * sizeof(float);
    static size_t lds_attr_dev[MAX_DEVICES] = {0};
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_glu<MTILES, NTW, NWAVES, GLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_gemm_glu<MTILES, NTW, NWAVES, GLU>), dim3(gn * gm), dim3(64 * NWAVES), lds, s, a, kchunk, gn, g_zeros);
    return hipGetLastError();
}

template <int MTILES, int NTW, int NWAVES, int LW, bool RES>
static hipError_t launch_gemm_pipe_r(const mdt_gemm_args& a, int kchunk, hipStream_t s) {
    const int MT = MTILES * 16, NTC = NWAVES * NTW * 16;
    const int gn = (a.N + NTC - 1) / NTC, gm = (a.M + MT - 1) / MT;
    const size_t lds = (size_t)2 * MT * (kchunk + 4) * sizeof(float);  // double-buffered activation chunk
    static size_t lds_attr_dev[MAX_DEVICES] = {0};
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_pipe<MTILES, NTW, NWAVES, LW, RES>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_gemm_pipe<MTILES, NTW, NWAVES, LW, RES>), dim3(gn * gm, 1, a.batch > 1 ? a.batch : 1),
                       dim3(64 * (NWAVES + LW)), lds, s, a, kchunk, gn, g_zeros);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_gemm_tall: 128-row tiles, both operands through LDS by LDS-DMA (body: mdt_tall.h).  Plain prologue, K % 32 == 0.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int NS, bool RES, int LW>
__global__ __launch_bounds__(64 * (WM * WN + LW)) void k_gemm_tall(mdt_gemm_args a, int grid_n, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical / grid_n, bx = logical - by * grid_n;
    // (staggering the start of the workgroups that share a CU -- b and b + 256 by the dispatcher's round-robin -- by fractions of
    //  a stage changed nothing: profiles/r04_gemm_train_shapes.txt)
    gemm_tall_tile<WM, WN, NT, NS, RES, LW>(a, by, bx, lds, zeros, threadIdx.x);
}
bool mdt_gemm_tall_supported(const mdt_gemm_args& a) {
    // (32-bit byte offsets from the operand bases: the activation block and the weight image each below 4 GiB)
    if ((int64_t)a.M * a.lda >= ((int64_t)1 << 30) || (int64_t)a.N * a.K >= ((int64_t)1 << 30)) return false;
    return !a.ln && a.a_parts <= 1 && a.batch <= 1 && a.K % MDT_TALL_BK == 0 && (a.N & 15) == 0 && a.M >= 1 && (a.lda & 3) == 0 &&
           (a.aux_mode == 0 || ((a.aux_mode == 1 || a.aux_mode == 2) && !a.residual));
}
template <int WM, int WN, int NT, int NS, bool RES, int LW>
static hipError_t launch_gemm_tall_r(const mdt_gemm_args& a, hipStream_t s) {
    constexpr int BM = WM * 64, BN = WN * NT * 16;
    const int gn = (a.N + BN - 1) / BN, gm = (a.M + BM - 1) / BM;
    const size_t lds = (size_t)NS * (BM + BN) * MDT_TALL_BK * sizeof(float);
    static bool attr_dev[MAX_DEVICES] = {false};
    bool& done = attr_dev[current_device()];
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_tall<WM, WN, NT, NS, RES, LW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL((k_gemm_tall<WM, WN, NT, NS, RES, LW>), dim3(gn * gm), dim3(64 * (WM * WN + LW)), lds, s, a, gn, g_zeros);
    return hipGetLastError();
}
template <int WM, int WN, int NT, int NS, int LW = 0>
static hipError_t launch_gemm_tall(const mdt_gemm_args& a, hipStream_t s) {
    return a.residual ? launch_gemm_tall_r<WM, WN, NT, NS, true, LW>(a, s) : launch_gemm_tall_r<WM, WN, NT, NS, false, LW>(a, s);
}

// ------------------------------------------------------------------------------------------------
// k_gemm_ws: shallow products over very many rows with the weights held in registers for the workgroup's whole life
// (body: mdt_ws.h).  Block b runs on XCD b % 8: the `panels` column panels of one row chunk are consecutive blocks of ONE XCD.
// ------------------------------------------------------------------------------------------------
template <int K16, int NTW, int NWAVES, int GLU>
__global__ __launch_bounds__(64 * NWAVES) void k_gemm_ws(mdt_gemm_args a, int tiles, int panels, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
    const int panel = i % panels, chunk = (i / panels) * 8 + xcd;
    gemm_ws_tile<K16, NTW, NWAVES, GLU>(a, panel, chunk, tiles, lds, zeros, threadIdx.x);
}
template <int K16, int NTW, int NWAVES, int GLU>
__global__ __launch_bounds__(64 * NWAVES) void k_gemm_ws_split(mdt_gemm_args a, int tiles, int panels, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) char lds_b[];
    // 256 blocks, one per CU: block b runs on XCD b % 8, and XCD x takes the 32 consecutive (chunk, panel) pairs 32 x .. 32 x + 31 of
    // the chunk-major order, so the panels of a chunk (the readers of one A tile) share an L2 -- or two, where a chunk straddles --
    // whatever the panel count divides (12 panels: 21 chunks of 19 tiles on 252 CUs, not 16 chunks of 24 on 192)
    const int b = blockIdx.x, w = (b & 7) * 32 + (b >> 3);
    const int chunk = w / panels, panel = w - chunk * panels;
    if (chunk * tiles >= ((a.M + 31) >> 5)) return;
    gemm_ws_split_tile<K16, NTW, NWAVES, GLU>(a, panel, chunk, tiles, lds_b, zeros, threadIdx.x);
}
// Workgroup = 8 waves x 2 column tiles (256-column panels), one per CU.  Measured and dropped (profiles/r05_ws_ab.txt): 4 waves x 2
// tiles as two independent workgroups per CU (27.6 vs 27.5 ms per head step), 4 waves x 3 tiles for the N = 576 / 192 products
// (272 VGPRs: qkv 225 -> 215 us against the tall body, c_proj 81.6 -> 81.0 against the row tiles: not worth a shape).
static int ws_shape(const mdt_gemm_args& a) {
    static int w12 = -1;  // MDT_HIP_WS_WAVES=8: the 8-wave shape everywhere (A/B runs)
    if (w12 < 0) { const char* e = getenv("MDT_HIP_WS_WAVES"); w12 = e && atoi(e) == 8 ? 0 : 1; }
    if (a.K == 384) {
        // one column tile per wave: 128- (8 waves) or 192-column (12 waves) panels.  One round of one workgroup per CU either way;
        // what differs is how evenly the row tiles divide: cost ~ tiles per workgroup x waves (a SIMD's waves share its matrix pipe).
        // M = 10240: N = 384 -> 8 waves (3 panels x 80 chunks x 4 tiles), N = 1152 / 1536 -> 12 waves (6 x 40 x 8 / 8 x 32 x 10)
        const int ntiles = (a.M + 31) / 32;
        int best = 0, best_cost = 0;
        for (int nw : {12, 8}) {
            if ((nw == 12 && !w12) || a.N % (nw * 16)) continue;
            const int panels = a.N / (nw * 16), groups = std::max(1, std::min(256 / (8 * panels), (ntiles + 7) / 8));
            if (8 * panels > 256) continue;
            const int tiles = (ntiles + 8 * groups - 1) / (8 * groups), cost = tiles * nw;
            if (!best || cost < best_cost) { best = nw; best_cost = cost; }
        }
        return best;
    }
    // 384-column panels, three waves per SIMD, 4 / 2 panels x 64 / 128 row chunks = 256 workgroups: the per-tile epilogue + barrier is
    // amortised over 18.4 k instead of 12.3 k clocks of MFMA issue and all 256 CUs work (SwishGLU forward 601 -> 544 us, plain 272 -> 248).
    // Not for the SwishGLU backward epilogue: at 168 VGPRs its u operands cannot be requested a tile ahead (348 -> 448 us)
    if (w12 && a.N % 384 == 0 && a.aux_mode != 4) return 12;
    return a.N % 256 == 0 ? 8 : 0;
}
// the three-way bf16 split of the weight-stationary body (mdt_ws.h): K = 384, 128-column panels (8 waves, one column tile each)
static int g_ws_split = -1;
static bool ws_split_on() {
    if (g_ws_split < 0) { const char* e = getenv("MDT_HIP_WS_SPLIT"); g_ws_split = e ? atoi(e) : 1; }
    return g_ws_split != 0;
}
extern "C" void mdt_op_set_ws_split(int32_t on) { g_ws_split = on < 0 ? -1 : (on != 0); }
// split form only: plain K = 192 products whose N is a multiple of 192 but not of 256 (the masked-image head's qkv and output
// projections, N = 576 / 192) -- twelve waves (three per SIMD) x one column tile.  (Their fp32 form was measured and dropped, above: no gain over the
// tall body; the split form is 1.4x.)
static bool ws_split_192(const mdt_gemm_args& a) {
    return ws_split_on() && a.K == 192 && a.N % 192 == 0 && a.N % 256 != 0 && a.N / 192 <= 32 && a.aux_mode == 0 && a.act == MDT_ACT_NONE;
}
bool mdt_gemm_ws_supported(const mdt_gemm_args& a) {
    static int k384 = -1;  // MDT_HIP_WS384=0: the K = 384 products stay on the row tiles / the tall body (A/B runs)
    if (k384 < 0) { const char* e = getenv("MDT_HIP_WS384"); k384 = e ? atoi(e) : 1; }
    const bool common = !a.ln && a.a_parts <= 1 && a.batch <= 1 && a.M >= 32 && (a.lda & 3) == 0 && (a.ldo & 3) == 0 && !a.residual &&
                        a.gin == 1 && a.gout == 1 && a.goff == 0 && a.rowvec == nullptr && (int64_t)a.N * a.K < ((int64_t)1 << 30);
    if (a.K == 384)   // plain rows, or an activation / the training hooks on the epilogue (GLU = 1 instantiation)
        return k384 && common && (a.aux_mode == 0 || ((a.aux_mode == 1 || a.aux_mode == 2) && a.aux != nullptr)) && ws_shape(a) != 0;
    const bool mode_ok = a.aux_mode == 0 || ((a.aux_mode == 3 || a.aux_mode == 4) && a.aux != nullptr);
    return common && a.K == 192 && (ws_shape(a) != 0 || ws_split_192(a)) && a.act == MDT_ACT_NONE && mode_ok;
}
template <int K16, int GLU, int NW, int NTW>
static hipError_t launch_gemm_ws_t(const mdt_gemm_args& a, hipStream_t s) {
    const int panels = a.N / (NW * NTW * 16), ntiles = (a.M + 31) / 32;
    // one round of one workgroup per CU: 8 x panels x groups blocks
    const int groups = std::max(1, std::min(256 / (8 * panels), (ntiles + 7) / 8));
    const int chunks = 8 * groups, tiles = (ntiles + chunks - 1) / chunks;
    const size_t lds = (size_t)2 * 32 * (K16 * 16 + 4) * sizeof(float);
    static bool attr_dev[MAX_DEVICES] = {false};
    bool& done = attr_dev[current_device()];
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_ws<K16, NTW, NW, GLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL((k_gemm_ws<K16, NTW, NW, GLU>), dim3(chunks * panels), dim3(64 * NW), lds, s, a, tiles, panels, g_zeros);
    return hipGetLastError();
}
template <int K16, int NTW, int GLU, int NW = 8>
static hipError_t launch_gemm_ws_split(const mdt_gemm_args& a, hipStream_t s) {
    const int panels = a.N / (NW * NTW * 16), ntiles = (a.M + 31) / 32;
    const int chunks = std::max(1, std::min(256 / panels, ntiles)), tiles = (ntiles + chunks - 1) / chunks;
    const size_t lds = (size_t)2 * 3 * 32 * (2 * K16 * 16 + 32);
    static bool attr_dev[MAX_DEVICES] = {false};
    bool& done = attr_dev[current_device()];
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_ws_split<K16, NTW, NW, GLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL((k_gemm_ws_split<K16, NTW, NW, GLU>), dim3(256), dim3(64 * NW), lds, s, a, tiles, panels, g_zeros);
    return hipGetLastError();
}
static hipError_t launch_gemm_ws(const mdt_gemm_args& a, hipStream_t s) {
    if (ws_split_on() && a.K == 384 && a.N % 128 == 0 && a.N / 128 <= 32) {
        const bool hooks = a.aux_mode != 0 || a.act != MDT_ACT_NONE;
        return hooks ? launch_gemm_ws_split<24, 1, 1>(a, s) : launch_gemm_ws_split<24, 1, 0>(a, s);
    }
    if (ws_split_on() && a.K == 192 && a.N % 256 == 0 && a.N / 256 <= 32 && a.act == MDT_ACT_NONE &&
        (a.aux_mode == 0 || a.aux_mode == 3 || a.aux_mode == 4)) {
        if (a.aux_mode == 3) return launch_gemm_ws_split<12, 2, 3>(a, s);
        if (a.aux_mode == 4) return launch_gemm_ws_split<12, 2, 4>(a, s);
        return launch_gemm_ws_split<12, 2, 0>(a, s);
    }
    if (ws_split_192(a)) return launch_gemm_ws_split<12, 1, 0, 12>(a, s);   // N = 192 / 576 ...: 192-column panels, twelve waves x one column tile
    if (a.K == 384) {
        const bool hooks = a.aux_mode != 0 || a.act != MDT_ACT_NONE;
        if (ws_shape(a) == 12) return hooks ? launch_gemm_ws_t<24, 1, 12, 1>(a, s) : launch_gemm_ws_t<24, 0, 12, 1>(a, s);
        return hooks ? launch_gemm_ws_t<24, 1, 8, 1>(a, s) : launch_gemm_ws_t<24, 0, 8, 1>(a, s);
    }
    if (ws_shape(a) == 12) return a.aux_mode == 3 ? launch_gemm_ws_t<12, 3, 12, 2>(a, s) : launch_gemm_ws_t<12, 0, 12, 2>(a, s);
    if (a.aux_mode == 3) return launch_gemm_ws_t<12, 3, 8, 2>(a, s);
    if (a.aux_mode == 4) return launch_gemm_ws_t<12, 4, 8, 2>(a, s);
    return launch_gemm_ws_t<12, 0, 8, 2>(a, s);
}

template <int MTILES, int NTW, int NWAVES, int PRO>
static hipError_t launch_gemm_t(const mdt_gemm_args& a, int kchunk, hipStream_t s) {
    return a.residual ? launch_gemm_r<MTILES, NTW, NWAVES, PRO, true>(a, kchunk, s)
                      : launch_gemm_r<MTILES, NTW, NWAVES, PRO, false>(a, kchunk, s);
}

// activation chunk length: whole K when it fits (always for the LayerNorm prologue), else the largest divisor of K
// that is a multiple of 16 and <= cap (cap 768 = 97 KiB of LDS for the one-workgroup-per-CU wide tiles, else 384)
// ------------------------------------------------------------------------------------------------
// k_gemm_smallm: the same fused GEMM for FEW rows (rollout batches: B = 1 is M = 10 decoder / 4 encoder rows).
// There the tiled kernel above has N/64 = 6 workgroups, each walking the whole K behind a 6 KB load window:
// 11-17 us of pure latency.  Here the parallelism comes from the reduction instead: one workgroup per 16 columns
// (and per 16 rows), its 8 waves split K between them (interleaved k16 steps), every wave keeps 4 weight
// fragments + 4 activation fragments in flight, partial 16x16 tiles meet in LDS and wave 0 runs the epilogue.
// Activations come straight from global memory in MFMA-fragment order (lane: row l%16, 4 consecutive k), with
// LayerNorm / modulate applied in registers from per-row statistics the workgroup computes first.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_gemm_smallm(mdt_gemm_args a, const float* __restrict__ zeros) {
    __shared__ float s_stat[32];
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    if (blockIdx.z) {
        a.A += (int64_t)blockIdx.z * a.bs_a; a.Wp += (int64_t)blockIdx.z * a.bs_w; a.out += (int64_t)blockIdx.z * a.bs_out;
    }
    gemm_smallm_tile<false>(a, blockIdx.x, blockIdx.y * 16, s_stat, red, zeros, threadIdx.x);
}

// Two INDEPENDENT small products in one launch (round 5): blocks [0, nxa) x row tiles of `a`, then those of `b`.  A rollout-sized
// sampler call is a chain of ~200 such ~5 us launches, and a handful of them do not depend on their neighbours (the sigma-MLP /
// adaLN table chain beside the encoder chain, the token embedding beside the goal embedding): riding in a neighbour's launch they cost
// nothing (mdt_gemm_side_*, below).
__global__ __launch_bounds__(512) void k_gemm_smallm2(mdt_gemm_args a, mdt_gemm_args b, int nxa, int nya, const float* __restrict__ zeros) {
    __shared__ float s_stat[32];
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    const int na = nxa * nya;
    if ((int)blockIdx.x < na) gemm_smallm_tile<false>(a, blockIdx.x % nxa, (blockIdx.x / nxa) * 16, s_stat, red, zeros, threadIdx.x);
    else {
        const int i = blockIdx.x - na, nxb = b.N >> 4;
        gemm_smallm_tile<false>(b, i % nxb, (i / nxb) * 16, s_stat, red, zeros, threadIdx.x);
    }
}

static int g_mdt_mid_max = 1400;    // rows up to which the 16 x 64 tiled geometry is used (env MDT_HIP_MID_MAX)
// k_xattn_gemm_smallm: the collapsed cross-attention of sample b and, on its output rows, 16 columns (blockIdx.x) of the
// LayerNorm + modulate -> Linear that follows (mlp.c_fc) -- rollout batches: the cross-attention launch (one workgroup, 6.5 us
// of latency per decoder block at B = 1) disappears into the c_fc launch, whose 96 workgroups each repeat it on the MFMA pipe
// (48 MFMAs per wave, the sample's 98 KB of folded operands from L2); workgroup 0 writes the residual stream.
template <int NPP>
__global__ __launch_bounds__(512) void k_xattn_gemm_smallm(mdt_xapply_args x, mdt_gemm_args a, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float s_stat[32];
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    const int b = blockIdx.y, ys = x.D + 4;
    float* yo = lds;                       // [16][D + 4]: the sample's rows after the cross-attention
    xattn_tile<NPP, false, false, 512, 0, true>(x, b, lds + 16 * ys, zeros, threadIdx.x, nullptr, 0, nullptr, yo, ys, blockIdx.x == 0);
    __syncthreads();
    gemm_smallm_tile<false, true>(a, blockIdx.x, b * x.Ta, s_stat, red, zeros, threadIdx.x, yo, ys, x.Ta);
}

static int g_mdt_smallm_rows = 512;   // ... up to this many rows (env MDT_HIP_SMALLM_ROWS)
static int g_mdt_smallm_tiles = 0;    // 0: the measured rule below; > 0 (env MDT_HIP_SMALLM_TILES): one threshold for every product (A/B runs): fewer 16 x 64 tiles than this -> the split-K kernel (env MDT_HIP_SMALLM_TILES)
static int g_mdt_smallm_max = -1;  // rows up to which k_gemm_smallm is used (env MDT_HIP_SMALLM_MAX, default below)

int mdt_gemm_kchunk(int K, int ln, int cap) {
    if (ln || K <= 512) return K;
    for (int c = cap; c >= 16; c -= 16)
        if (K % c == 0) return c;
    return 16;
}

template <int MTILES, int NTW, int NWAVES>
static hipError_t launch_gemm_pro(const mdt_gemm_args& a, int kchunk, hipStream_t s) {
    if (!a.ln) return launch_gemm_t<MTILES, NTW, NWAVES, PRO_PLAIN>(a, kchunk, s);
    if (a.mod != nullptr && a.shift_off >= 0)
        return a.mod_stride == 0 ? launch_gemm_t<MTILES, NTW, NWAVES, PRO_LN_MOD_BCAST>(a, kchunk, s)
                                 : launch_gemm_t<MTILES, NTW, NWAVES, PRO_LN_MOD_ROWS>(a, kchunk, s);
    return launch_gemm_t<MTILES, NTW, NWAVES, PRO_LN>(a, kchunk, s);
}

int g_mdt_gemm_force = 0;  // tuning hook: 0 = heuristic, 1.. selects a geometry below, -1 = the split-K small-M kernel
static const bool g_mdt_gemm_nopipe = getenv("MDT_HIP_NOPIPE") != nullptr;  // A/B switch for k_gemm_pipe

// Self-attention of ONE sample fused into its output projection (k_attn_proj_smallm).  `p` is the projection's GEMM
// (A ignored: the attention output never reaches memory; M = the sample's T rows); q / k / v are the three column
// blocks of the (T, 3 K) qkv rows.  Supported: 8 heads of 16 / 32 / 48 / 64, T <= 16, no RoPE, plain output rows.
static bool attn_proj_disabled() {  // MDT_HIP_NO_ATTN_PROJ=1: the separate attention + projection launches (A/B runs)
    static int off = -1;
    if (off < 0) off = getenv("MDT_HIP_NO_ATTN_PROJ") != nullptr;
    return off != 0;
}

// WStream (mdt_tiles.h) addresses a weight image with 32-bit byte offsets from its base: every launcher that feeds one checks
// the image size (mdt_launch_gemm does for the GEMMs; the fused MLP and attn_xattn tiles bound N and K by their shape rules)
static bool w_image_ok(int64_t N, int64_t K) { return N * K < ((int64_t)1 << 30); }

bool mdt_attn_proj_supported(const mdt_gemm_args& p, int H, int hd, int T, int rope) {
    return w_image_ok(p.N, p.K) && H == 8 && (hd == 16 || hd == 32 || hd == 48 || hd == 64) && p.K == H * hd && T >= 1 && T <= 16 && p.M >= T &&
           p.M % T == 0 && p.M / T <= 64 && (p.rows_per_sample == T || p.M == T) && !rope &&
           !(p.N & 15) && p.gin == 1 && p.gout == 1 && p.goff == 0 && !p.ln && p.act == MDT_ACT_NONE && p.rowvec == nullptr &&
           p.batch <= 1 && !attn_proj_disabled();
}

template <int HD>
static hipError_t launch_attn_proj_t(const mdt_gemm_args& p, const float* qkv, int64_t ldq, int T, int causal, hipStream_t s) {
    const size_t lds = (size_t)8 * 3 * T * (HD + 4) * sizeof(float);
    static size_t lds_attr_dev[MAX_DEVICES] = {0};
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > 48 * 1024 && lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_proj_smallm<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_attn_proj_smallm<HD>), dim3(p.N >> 4, p.M / T), dim3(512), lds, s, p, qkv, ldq, T, causal,
                       1.0f / sqrtf((float)HD), g_zeros);
    return hipGetLastError();
}

hipError_t mdt_launch_attn_proj(const mdt_gemm_args& p, const float* qkv, int64_t ldq, int H, int hd, int T, int causal,
                                hipStream_t s) {
    if (!mdt_attn_proj_supported(p, H, hd, T, 0)) return hipErrorInvalidValue;
    hipError_t ze = ensure_zeros();
    if (ze != hipSuccess) return ze;
    switch (hd) {
        case 16: return launch_attn_proj_t<16>(p, qkv, ldq, T, causal, s);
        case 32: return launch_attn_proj_t<32>(p, qkv, ldq, T, causal, s);
        case 48: return launch_attn_proj_t<48>(p, qkv, ldq, T, causal, s);
        default: return launch_attn_proj_t<64>(p, qkv, ldq, T, causal, s);
    }
}

static hipError_t launch_gemm_merge(const mdt_gemm_args& a, hipStream_t s);
static bool gemm_ln_split_applies(const mdt_gemm_args& a);
static hipError_t launch_gemm_ln_split(const mdt_gemm_args& a, hipStream_t s);

bool mdt_attn_proj_wide_supported(const mdt_gemm_args& p, int H, int hd, int T, int causal, int rope) {
    return w_image_ok(p.N, p.K) && H == 8 && (hd == 16 || hd == 32 || hd == 48) && p.K == H * hd && T >= 1 && T <= 16 && causal && !rope && p.residual &&
           p.M >= T && p.M % T == 0 && p.rows_per_sample == T && !(p.N & 15) && p.gin == 1 && p.gout == 1 && p.goff == 0 && !p.ln &&
           p.act == MDT_ACT_NONE && p.rowvec == nullptr && p.batch <= 1 && !p.aux_mode;
}

template <int HD, int TKC>
static hipError_t launch_attn_proj_wide_t(const mdt_gemm_args& p, const mdt_attn_pro& ap, hipStream_t s) {
    const int gn = (p.N + 127) / 128, gm = (p.M + 31) / 32;
    const size_t lds = ((size_t)32 * (p.K + 4) + (size_t)(32 + 2 * 48) * (4 * HD + 16)) * sizeof(float);
    static size_t lds_attr_dev[MAX_DEVICES] = {0};
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_proj_wide<HD, TKC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_attn_proj_wide<HD, TKC>), dim3(gn * gm), dim3(512), lds, s, p, ap, gn, g_zeros);
    return hipGetLastError();
}

hipError_t mdt_launch_attn_proj_wide(const mdt_gemm_args& p, const float* qkv, int64_t ldq, int H, int hd, int T, hipStream_t s) {
    if (!mdt_attn_proj_wide_supported(p, H, hd, T, 1, 0)) return hipErrorInvalidValue;
    hipError_t ze = ensure_zeros();
    if (ze != hipSuccess) return ze;
    mdt_attn_pro ap;
    ap.qkv = qkv; ap.ldq = ldq; ap.T = T; ap.scale = 1.0f / sqrtf((float)hd);
    const bool t10 = T <= 10;
    switch (hd) {
        case 16: return t10 ? launch_attn_proj_wide_t<16, 10>(p, ap, s) : launch_attn_proj_wide_t<16, 16>(p, ap, s);
        case 32: return t10 ? launch_attn_proj_wide_t<32, 10>(p, ap, s) : launch_attn_proj_wide_t<32, 16>(p, ap, s);
        default: return t10 ? launch_attn_proj_wide_t<48, 10>(p, ap, s) : launch_attn_proj_wide_t<48, 16>(p, ap, s);
    }
}

bool mdt_xattn_apply_supported(int D, int H, int Te, int Ta);

// Self-attention + projection + collapsed cross-attention of one sample per workgroup: `p` = the projection's arguments
// (as for mdt_launch_attn_proj_wide), `x` = the cross-attention's (x.y == p.out).  Head dimension 48 (d = 384).
bool mdt_attn_xattn_supported(const mdt_gemm_args& p, const mdt_xapply_args& x, int H, int hd, int T, int causal, int rope) {
    return mdt_attn_proj_wide_supported(p, H, hd, T, causal, rope) && hd == 48 && p.N == p.K && p.ldo == p.N && x.y == p.out &&
           x.D == p.N && x.H == H && H == 8 && x.Ta == T && (int64_t)x.B * T == p.M &&
           mdt_xattn_apply_supported(x.D, x.H, x.Te, x.Ta);
}

template <int HD, int TKC>
static hipError_t launch_attn_xattn_t(const mdt_gemm_args& p, const mdt_attn_pro& at, const mdt_xapply_args& x, hipStream_t s) {
    const int D = 8 * HD;
    const size_t lds = ((size_t)16 * (D + 4) + (size_t)48 * (D + 16)) * sizeof(float);
    static bool attr_dev[MAX_DEVICES] = {false};
    bool& attr = attr_dev[current_device()];
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_xattn<HD, TKC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL((k_attn_xattn<HD, TKC>), dim3(x.B), dim3(512), lds, s, p, at, x, g_zeros);
    return hipGetLastError();
}

hipError_t mdt_launch_attn_xattn(const mdt_gemm_args& p, const float* qkv, int64_t ldq, const mdt_xapply_args& x, int H, int hd,
                                 int T, hipStream_t s) {
    if (!mdt_attn_xattn_supported(p, x, H, hd, T, 1, 0) || ldq != 3 * (int64_t)p.K) return hipErrorInvalidValue;
    hipError_t ze = ensure_zeros();
    if (ze != hipSuccess) return ze;
    mdt_attn_pro at;
    at.qkv = qkv; at.ldq = ldq; at.T = T; at.scale = 1.0f / sqrtf((float)hd);
    return T <= 10 ? launch_attn_xattn_t<48, 10>(p, at, x, s) : launch_attn_xattn_t<48, 16>(p, at, x, s);
}

// Side jobs: products that do not depend on the launches they are queued beside (and that nothing launched before
// mdt_gemm_side_flush reads).  Each one rides in the launch of the NEXT product that goes to the split-K small-M kernel, in queue
// order, one per launch (so a side job may depend on the side job before it); what is left is launched by mdt_gemm_side_flush.
// Only small-M products are accepted (M <= 16 rows, the small-M kernel's own shape rules); anything else is launched at once.
static thread_local std::vector<mdt_gemm_args> g_side_jobs;
static thread_local size_t g_side_next = 0;
static bool smallm_shape_ok(const mdt_gemm_args& a) {
    return a.M >= 1 && a.M <= 15 && (!a.ln || a.K <= 512) && a.batch <= 1 && a.K <= 4096 && !a.aux_mode && a.a_parts <= 1 &&
           a.N <= ZEROS_FLOATS && a.K <= ZEROS_FLOATS && !(a.N & 15) && (int64_t)a.N * a.K < ((int64_t)1 << 30);
}
// process-wide (the queue itself is per host thread): atomics, host threads may drive different handles side by side
static std::atomic<int> g_side_override{-1};     // mdt_op_set_side_jobs (tests / A-B runs): 0 = off, 1 = on, -1 = the environment's choice
static std::atomic<int64_t> g_side_paired{0};    // launches that took a side job along (mdt_op_side_jobs_paired)
static bool side_enabled() {
    static int v = -1;  // MDT_HIP_SIDE_JOBS=0: every product its own launch (A/B runs)
    if (v < 0) { const char* e = getenv("MDT_HIP_SIDE_JOBS"); v = e ? atoi(e) : 1; }
    const int ov = g_side_override.load(std::memory_order_relaxed);
    return (ov >= 0 ? ov != 0 : v != 0) && g_mdt_gemm_force == 0;
}
extern "C" void mdt_op_set_side_jobs(int32_t on) { g_side_override.store(on < 0 ? -1 : (on != 0)); }
extern "C" int64_t mdt_op_side_jobs_paired(void) { return g_side_paired.load(); }
hipError_t mdt_gemm_side_push(const mdt_gemm_args& a, hipStream_t s) {
    if (!side_enabled() || !smallm_shape_ok(a)) return mdt_launch_gemm(a, s);
    g_side_jobs.push_back(a);
    return hipSuccess;
}
hipError_t mdt_gemm_side_push_front(const mdt_gemm_args& a, hipStream_t s) {   // rides in the NEXT small-M launch, ahead of the queue
    if (!side_enabled() || !smallm_shape_ok(a)) return mdt_launch_gemm(a, s);
    g_side_jobs.insert(g_side_jobs.begin() + g_side_next, a);
    return hipSuccess;
}
void mdt_gemm_side_drop() { g_side_jobs.clear(); g_side_next = 0; }   // error paths: forget what was queued
size_t mdt_gemm_side_pending() { return g_side_jobs.size() - g_side_next; }
hipError_t mdt_gemm_side_launch_front(hipStream_t s) {   // the head of the queue as a launch of its own, the rest stays queued
    if (g_side_next >= g_side_jobs.size()) return hipSuccess;
    const mdt_gemm_args a = g_side_jobs[g_side_next];
    g_side_jobs.erase(g_side_jobs.begin() + g_side_next);
    std::vector<mdt_gemm_args> keep;   // hidden while the job is launched: it must not take its successor along
    keep.swap(g_side_jobs);
    const size_t next = g_side_next;
    g_side_next = 0;
    const hipError_t e = mdt_launch_gemm(a, s);
    g_side_jobs.swap(keep);
    g_side_next = next;
    return e;
}
hipError_t mdt_gemm_side_flush(hipStream_t s) {
    // the rest of the queue, each job as a launch of its own: the queue is emptied FIRST (a job launched from here must not take
    // its successor -- which may depend on it -- along)
    std::vector<mdt_gemm_args> rest(g_side_jobs.begin() + g_side_next, g_side_jobs.end());
    mdt_gemm_side_drop();
    for (const mdt_gemm_args& a : rest) {
        hipError_t e = mdt_launch_gemm(a, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t mdt_launch_gemm(const mdt_gemm_args& a, hipStream_t s) {
    if (a.N > ZEROS_FLOATS || a.K > ZEROS_FLOATS) return hipErrorInvalidValue;
    // the weight stream is addressed with 32-bit byte offsets from the image's base (buffer loads, mdt_tiles.h: WStream)
    if ((int64_t)a.N * a.K >= ((int64_t)1 << 30)) return hipErrorInvalidValue;
    hipError_t ze = ensure_zeros();
    if (ze != hipSuccess) return ze;
    if (g_mdt_smallm_max < 0) {
        const char* e = getenv("MDT_HIP_SMALLM_MAX");
        g_mdt_smallm_max = e ? atoi(e) : 15;  // one row tile; beyond it the tile-count rule below decides (round 1 had 192 rows here, before the half-height tiles)
        const char* f = getenv("MDT_HIP_MID_MAX");
        if (f) g_mdt_mid_max = atoi(f);
        const char* t = getenv("MDT_HIP_SMALLM_TILES");
        if (t) g_mdt_smallm_tiles = atoi(t);
        const char* r = getenv("MDT_HIP_SMALLM_ROWS");
        if (r) g_mdt_smallm_rows = atoi(r);
    }
    // (not for the batched split-K products of the weight gradients: few output rows there come with a DEEP reduction --
    // N = 192 layers of the masked-image decoder: 595 us as 16-column split-K tiles vs ~300 us tiled)
    // the training hooks of the epilogue (aux) exist in the plain-prologue, non-residual tiled kernels only
    if (a.aux_mode && (!a.aux || a.ln || a.residual || a.batch > 1 || a.K > 512)) return hipErrorInvalidValue;
    if (a.aux_mode == 3 || a.aux_mode == 4) {  // SwishGLU on the epilogue: own kernels (4 waves; forward: pairs of column tiles per wave)
        if ((a.N & (a.aux_mode == 3 ? 31 : 15)) || a.gin != 1 || a.gout != 1 || a.goff != 0 || a.act != MDT_ACT_NONE)
            return hipErrorInvalidValue;
        // from 8192 rows on, K = 192: the weight-stationary body (mdt_ws.h) with the same epilogues -- round 5; MDT_HIP_WS=0: the 32-row
        // tiles (A/B runs).  The masked-image head's two SwishGLU products at B = 1024 (104448 rows).
        static int ws = -1;
        if (ws < 0) { const char* e = getenv("MDT_HIP_WS"); ws = e ? atoi(e) : 1; }
        if (ws && a.M >= 8192 && g_mdt_gemm_force <= 0 && mdt_gemm_ws_supported(a)) return launch_gemm_ws(a, s);
        const int kc = mdt_gemm_kchunk(a.K, 0, 384);
        if (a.aux_mode == 3)
            return (a.N % 256 == 0) ? launch_gemm_glu<2, 4, 4, 3>(a, kc, s) : launch_gemm_glu<2, 2, 4, 3>(a, kc, s);
        return (a.N % 192 == 0) ? launch_gemm_glu<2, 3, 4, 4>(a, kc, s) : launch_gemm_glu<2, 2, 4, 4>(a, kc, s);
    }
    if (gemm_ln_split_applies(a)) return launch_gemm_ln_split(a, s);   // the wide LayerNorm-prologue products in the bf16 split form
    if (a.a_parts > 1) return launch_gemm_merge(a, s);
    // ... and beyond one row tile, the products whose half-height (16 x 64) tiling would leave most of the chip empty keep the
    // split-K kernel, which has 4x the workgroups: fewer than 60 tiles behind a LayerNorm prologue (every 16-column workgroup
    // repeats the row statistics), fewer than 100 / 160 (up to 192 / 512 rows) behind a plain one -- the two N = 384 residual
    // projections of a block at B = 12 ... 32.  Measured per sampler call (tools/latency.py, profiles/r04_lowbatch.txt):
    // B = 8 1.79 -> 1.69 ms, 10 2.06 -> 1.72, 16 2.43 -> 1.74, 19 2.93 -> 2.07, 24 2.36 -> 2.17, 32 2.39 -> 2.19.
    const int64_t tiles6 = (int64_t)((a.M + 15) / 16) * ((a.N + 63) / 64);
    const bool few_tiles = a.M <= g_mdt_smallm_rows &&
                           tiles6 < (g_mdt_smallm_tiles > 0 ? g_mdt_smallm_tiles : (a.ln ? 60 : (a.M <= 192 ? 100 : 160)));
    // (geometry hook -1: the split-K kernel wherever it applies -- tests that pin it against its fused variants)
    if ((a.M <= g_mdt_smallm_max || few_tiles || g_mdt_gemm_force < 0) && g_mdt_gemm_force <= 0 && (!a.ln || a.K <= 512) && a.batch <= 1 && a.K <= 4096 && !a.aux_mode) {
        if (g_side_next < g_side_jobs.size() && a.batch <= 1 && a.M <= 16 * 64) {  // a queued side job rides in this launch
            const mdt_gemm_args b = g_side_jobs[g_side_next++];
            ++g_side_paired;
            const int nxa = a.N >> 4, nya = (a.M + 15) >> 4;
            hipLaunchKernelGGL(k_gemm_smallm2, dim3(nxa * nya + (b.N >> 4) * ((b.M + 15) >> 4)), dim3(512), 0, s, a, b, nxa, nya, g_zeros);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(k_gemm_smallm, dim3(a.N >> 4, (a.M + 15) >> 4, a.batch > 1 ? a.batch : 1), dim3(512), 0, s, a, g_zeros);
        return hipGetLastError();
    }
    // Geometry selection (rows are in tiles of 32).  The decoder at B = 256 has 80 row tiles for 256 CUs:
    //   wide  (8 waves, 32 x 512 / 32 x 384): N >= 1024 -> 240 workgroups, each activation tile re-read 3x
    //   mid   (8 waves, 32 x 128)           : N  < 1024 -> 240 workgroups for N = 384
    //   small (4 waves, 32 x 64)            : few row tiles (small batches): more, smaller workgroups
    const int gm = (a.M + 31) / 32;
    // Pick the geometry with the best (wave fill) x (tile efficiency): workgroups run in waves of ~256 (one per CU), so
    // a count just above a multiple of 256 wastes most of its last wave (M = 10240, N = 384 on 32x384 tiles: 320
    // workgroups = 1.25 waves); wide tiles amortise the per-workgroup prologue / epilogue better than narrow ones.
    const int tile_n[5] = {0, 64, 128, 384, 512};
    const float eff[5] = {0.f, 0.85f, 0.92f, 1.0f, 1.0f};
    int geo = 1;
    float best = -1.f;
    for (int g = 4; g >= 1; --g) {
        const int cols = (a.N + tile_n[g] - 1) / tile_n[g];
        const int cnt = gm * cols;
        const float fill = (float)cnt / (float)(((cnt + 255) / 256) * 256);
        const float used = (float)a.N / (float)(cols * tile_n[g]);   // columns of the last tile that are real work
        const float score = fill * eff[g] * used;
        if (score > best) { best = score; geo = g; }
    }
    // deep K in several LDS chunks (mlp.c_proj and its kin, K = 4d) at large row counts: co-resident 4-wave 32 x 128
    // workgroups beat the 8-wave loader-wave kernel (tools/gemm_micro.py, N = 384, K = 1536: M = 10240 116 vs 131 us,
    // M = 5120 60 vs 66 us; at M = 2560 the loader-wave kernel still wins, 37.7 vs 42.5 us)
    if (geo == 2 && !a.ln && a.K > 512 && a.M >= 4096) geo = 5;
    // mid-size row counts (batches of ~20..140 chunks, and the training path's 384..1536-row dW products): half-height
    // tiles double the workgroup count; measured 4-17 % faster per sampler call up to M ~ 1400, slower beyond 2000
    // ... unless the output is so wide that they would be thousands (the stacked adaLN projection of a training batch:
    // 1024 x 9216 -> 9216 workgroups that each re-read their weight tile): then the scored choice stands
    if (a.M <= g_mdt_mid_max && (int64_t)((a.M + 15) / 16) * ((a.N + 63) / 64) <= 4096) {
        geo = 6;
        // ... except where 4-wave 32 x 192 tiles come out as (nearly) whole rounds of one workgroup per CU: the encoder's
        // and the B ~ 120..140 decoder's wide products (tools/gemm_shapes.py: 1024 x 1536 22.0 -> 18.3 us, 1024 x 3072 33.7 ->
        // 28.0, 1280 x 1152 21.6 -> 17.8; 1280 x 1536 -- 320 tiles -- 27.8 vs 28.5: stays; 1024 x 1152 -- 192 tiles -- wins
        // alone, 18.9 -> 17.5, and loses inside the encoder, 18.1 -> 19.5: stays)
        if (a.N % 192 == 0 && a.N >= 1152 && a.K <= 512 && a.batch <= 1) {
            const int cnt9 = gm * (a.N / 192);
            if ((float)cnt9 / (float)(((cnt9 + 255) / 256) * 256) >= 0.9f) geo = 9;
        }
    }
    // training-sized row counts without a LayerNorm prologue (forward / input-gradient products of a B = 1024 step, the
    // masked-image head's 104 k rows): 4 waves x 32 x 192 tiles -- two or three co-resident per CU, each weight fragment reused by
    // one wave only -- beat every 8-wave geometry whenever N is a multiple of 192 (tools/gemm_train_shapes.py: 10240 x 1536 x 384
    // 126 -> 107 us, 104448 x 192 x 768 372 -> 282 us, 104448 x 576 x 192 332 -> 233 us); N = 384 keeps the choices above
    // (36.5 vs 38.4 us at K = 384, the co-resident 32 x 128 tiles at K = 1536)
    {
        static int g9 = -1;
        if (g9 < 0) { const char* e = getenv("MDT_HIP_GEO_TRAIN"); g9 = e ? atoi(e) : 9; }
        if (g9 && !a.ln && a.M >= 4096 && a.N % 192 == 0 && a.N != 384 && a.batch <= 1) geo = g9;
    }
    // ... and from 8192 rows on the TALL body (mdt_tall.h: 128-row tiles, both operands staged in LDS by LDS-DMA, each staged
    // element used by 2-4 waves) where the column count is a multiple of its 64-wide tiles: the forward / input-gradient
    // products of a B = 1024 training step and the masked-image head's 104 k rows (tools/gemm_train_shapes.py,
    // profiles/r04_gemm_train_shapes.txt: 104448 x 576 x 192 249 -> 213-231 us, 104448 x 192 x 768 290 -> 263-270,
    // 10240 x 384 x 1536 124 -> 105-112, 10240 x 1152 x 384 94 -> 88-92; 10240 x 384 x 384 unchanged, 4096 rows lose)
    // ... and the K = 192 products among them whose column count is a multiple of 256 on the weight-stationary body (mdt_ws.h);
    // MDT_HIP_WS=0: A/B runs
    {
        static int ws = -1;
        if (ws < 0) { const char* e = getenv("MDT_HIP_WS"); ws = e ? atoi(e) : 1; }
        if (ws && a.M >= 8192 && (a.aux_mode == 0 || a.K == 384) && g_mdt_gemm_force <= 0 && mdt_gemm_ws_supported(a)) return launch_gemm_ws(a, s);
    }
    {
        static int gt = -1;
        if (gt < 0) { const char* e = getenv("MDT_HIP_GEO_TALL"); gt = e ? atoi(e) : 23; }
        if (gt && a.M >= 8192 && a.N % 64 == 0 && (a.N > 384 || a.K > 384) && a.batch <= 1 && mdt_gemm_tall_supported(a)) geo = gt;
    }
    if (a.batch > 1) {  // split-K partial products (deep reductions): geometry chosen for those, env override for A/B runs
        static int bgeo = -1;
        if (bgeo < 0) { const char* e = getenv("MDT_HIP_BATCH_GEO"); bgeo = e ? atoi(e) : 5; }
        geo = bgeo;
    }
    if (g_mdt_gemm_force > 0) geo = g_mdt_gemm_force;
    {   // tuning hooks (A/B runs): geometry of the wide (N >= 1024) / narrow products of large batches
        static int gw = -1, gn = -1;
        if (gw < 0) { const char* e = getenv("MDT_HIP_GEO_WIDE"); gw = e ? atoi(e) : 0; const char* f = getenv("MDT_HIP_GEO_NARROW"); gn = f ? atoi(f) : 0; }
        if (a.M > g_mdt_mid_max && a.batch <= 1 && g_mdt_gemm_force <= 0) {
            if (a.N >= 1024 && gw) geo = gw;
            if (a.N < 1024 && gn) geo = gn;
        }
    }
    if (geo == 30) {  // forced: the weight-stationary body (tests pin it against the other bodies)
        if (mdt_gemm_ws_supported(a)) return launch_gemm_ws(a, s);
        geo = 0;
    }
    if (geo >= 10 && (geo != 23 || !mdt_gemm_tall_supported(a))) geo = 0;  // (forced geometry on a product the tall body does not take)
    switch (geo) {
        // the tall body (mdt_tall.h): 128-row tiles, operands by LDS-DMA, ONE loader wave issuing every DMA request of the workgroup.
        // (Round 4 also carried 128 x 128 / 128 x 64 / 128 x 96 tiles without the loader wave -- geometries 10 / 12 / 16 -- which no
        //  dispatcher rule reached after tools/gemm_train_shapes.py had measured them: pruned in round 5.)
        case 23: return launch_gemm_tall<2, 2, 2, 3, 1>(a, s);  // 4 + 1 waves 128 x 64,  3 stages (72 KiB: two per CU)
        case 1: return launch_gemm_pro<2, 1, 4>(a, mdt_gemm_kchunk(a.K, a.ln, 384), s);
        case 2:
            if (!a.ln && a.K > 512 && !g_mdt_gemm_nopipe) {  // multi-chunk K: loader waves double-buffer the activation chunk
                const int kc = mdt_gemm_kchunk(a.K, 0, 384);
                return a.residual ? launch_gemm_pipe_r<2, 1, 8, 4, true>(a, kc, s)
                                  : launch_gemm_pipe_r<2, 1, 8, 4, false>(a, kc, s);
            }
            return launch_gemm_pro<2, 1, 8>(a, mdt_gemm_kchunk(a.K, a.ln, 768), s);
        case 3: return launch_gemm_pro<2, 3, 8>(a, mdt_gemm_kchunk(a.K, a.ln, 768), s);
        case 4: return launch_gemm_pro<2, 4, 8>(a, mdt_gemm_kchunk(a.K, a.ln, 768), s);
        case 6: {
            // 4 waves 16 x 64.  Deep K (the encoder's / a mid batch's c_proj, K = 4d in four LDS chunks): with two LOADER waves that
            // double-buffer the activation chunk -- as a single-buffer loop each chunk was a memory round trip of its own in front of
            // its 24 k-steps, and half-height tiles have few workgroups per CU to hide it (MDT_HIP_PIPE6=0: A/B runs)
            static int p6 = -1;
            if (p6 < 0) { const char* e = getenv("MDT_HIP_PIPE6"); p6 = e ? atoi(e) : 1; }
            // (up to one workgroup per CU: beyond that the co-resident workgroups hide each other's round trips -- the encoder's
            //  1024-row c_proj, 384 workgroups, measured +0.1 % with the loader waves)
            if (p6 && !a.ln && a.K > 512 && !a.aux_mode && !g_mdt_gemm_nopipe && (int64_t)((a.M + 15) / 16) * ((a.N + 63) / 64) <= 256) {
                const int kc = mdt_gemm_kchunk(a.K, 0, 384);
                return a.residual ? launch_gemm_pipe_r<1, 1, 4, 2, true>(a, kc, s) : launch_gemm_pipe_r<1, 1, 4, 2, false>(a, kc, s);
            }
            return launch_gemm_pro<1, 1, 4>(a, mdt_gemm_kchunk(a.K, a.ln, 384), s);
        }
        case 7: return launch_gemm_pro<4, 2, 4>(a, mdt_gemm_kchunk(a.K, a.ln, 384), s);  // 4 waves 64 x 128 (deep-K products)
        case 8: return launch_gemm_pro<2, 4, 4>(a, mdt_gemm_kchunk(a.K, a.ln, 384), s);  // 4 waves 32 x 256: two workgroups per CU
        case 9: return launch_gemm_pro<2, 3, 4>(a, mdt_gemm_kchunk(a.K, a.ln, 384), s);  // 4 waves 32 x 192
        default: return launch_gemm_pro<2, 2, 4>(a, mdt_gemm_kchunk(a.K, a.ln, 384), s);
    }
}

extern "C" void mdt_op_set_gemm_geometry(int32_t geo) { g_mdt_gemm_force = geo; }

// ---- fused MLP sublayer (k_mlp) ----
bool mdt_mlp_supported(const mdt_gemm_args& f, const mdt_gemm_args& p) {
    const int D = f.K;
    return D >= 128 && D <= 512 && D % 128 == 0 && f.N == 4 * D && p.N == D && p.K == 4 * D && f.ln && f.batch <= 1 &&
           f.gin == 1 && f.gout == 1 && f.goff == 0 && f.rowvec == nullptr && !f.aux_mode && f.a_parts <= 1 && f.M >= 1 &&
           f.rows_per_sample >= 1 && p.rows_per_sample >= 1;
}
int mdt_mlp_slices(int D) { return 4 * D / 512; }

// mlp_tile's wave schedule: low byte = k-steps the second wave of a SIMD starts behind the first (0: lockstep, workgroup
// barrier between the two products), | 256 = MFMA loops at raised issue priority.  Measured at B = 256 (tools/gpu_skew_ab3.sh,
// profiles/r03_mlp_skew_ab.txt): 0 -> 4.89, 6 -> 4.86, 18 | 256 -> 4.82 ms per sampler call.  MDT_HIP_MLP_SKEW / the hook: A/B runs, tests.
static int g_mlp_skew = -1;
static int mlp_skew() {
    if (g_mlp_skew < 0) { const char* e = getenv("MDT_HIP_MLP_SKEW"); g_mlp_skew = e ? atoi(e) & 0x1ff : (18 | 256); }
    return g_mlp_skew;
}
extern "C" void mdt_op_set_mlp_skew(int32_t v) { g_mlp_skew = v < 0 ? -1 : (v & 0x1ff); }

template <int NTW2, int PRO>
static hipError_t launch_mlp_t(const mdt_gemm_args& f, const mdt_gemm_args& p, float* parts, int64_t part_stride, hipStream_t s) {
    const int S = mdt_mlp_slices(f.K), gm = (f.M + 31) / 32;
    const size_t lds = (size_t)32 * (f.K + 4 + 516) * sizeof(float) + 16 * sizeof(int);  // + the wave flags of mlp_tile
    const int skew = mlp_skew();
    static size_t lds_attr_dev[MAX_DEVICES] = {0};
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp<NTW2, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_mlp<NTW2, PRO>), dim3(gm * S), dim3(512), lds, s, f, p, parts, part_stride, S, g_zeros, skew);
    return hipGetLastError();
}
template <int NTW2>
static hipError_t launch_mlp_pro(const mdt_gemm_args& f, const mdt_gemm_args& p, float* parts, int64_t part_stride, hipStream_t s) {
    if (f.mod != nullptr && f.shift_off >= 0)
        return f.mod_stride == 0 ? launch_mlp_t<NTW2, PRO_LN_MOD_BCAST>(f, p, parts, part_stride, s)
                                 : launch_mlp_t<NTW2, PRO_LN_MOD_ROWS>(f, p, parts, part_stride, s);
    return launch_mlp_t<NTW2, PRO_LN>(f, p, parts, part_stride, s);
}
hipError_t mdt_launch_mlp(const mdt_gemm_args& f, const mdt_gemm_args& p, float* parts, int64_t part_stride, hipStream_t s) {
    if (!mdt_mlp_supported(f, p)) return hipErrorInvalidValue;
    hipError_t ze = ensure_zeros();
    if (ze != hipSuccess) return ze;
    switch (f.K / 128) {
        case 1: return launch_mlp_pro<1>(f, p, parts, part_stride, s);
        case 2: return launch_mlp_pro<2>(f, p, parts, part_stride, s);
        case 3: return launch_mlp_pro<3>(f, p, parts, part_stride, s);
        default: return launch_mlp_pro<4>(f, p, parts, part_stride, s);
    }
}

// ---- the LayerNorm-prologue product on the wide tiles in the split form (gemm_ln_split_tile) ----
template <int ND, int NTW, int PRO, int XP>
__global__ __launch_bounds__(512) void k_gemm_ln_split(mdt_gemm_args a, int grid_n, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) char lds_c[];
    const int logical = xcd_remap(blockIdx.x, gridDim.x);   // the column panels of a row tile are neighbours: they share the rows' L2
    const int by = logical / grid_n, bx = logical - by * grid_n;
    gemm_ln_split_tile<ND, NTW, PRO, XP>(a, by, bx, lds_c, zeros, threadIdx.x);
}
template <int ND, int NTW, int PRO, int XP>
static hipError_t launch_gemm_ln_split_t(const mdt_gemm_args& a, hipStream_t s) {
    const int gn = a.N / (8 * NTW * 16), gm = (a.M + 31) / 32;
    const int lds = 3 * 32 * (2 * 128 * ND + 32);
    static bool attr_dev[MAX_DEVICES] = {false};
    bool& done = attr_dev[current_device()];
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_ln_split<ND, NTW, PRO, XP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL((k_gemm_ln_split<ND, NTW, PRO, XP>), dim3(gn * gm), dim3(512), lds, s, a, gn, g_zeros);
    return hipGetLastError();
}
template <int ND, int NTW, int PRO>
static hipError_t launch_gemm_ln_split_x(const mdt_gemm_args& a, hipStream_t s) {
    switch (a.a_parts) {
        case 2: return launch_gemm_ln_split_t<ND, NTW, PRO, 2>(a, s);
        case 3: return launch_gemm_ln_split_t<ND, NTW, PRO, 3>(a, s);
        case 4: return launch_gemm_ln_split_t<ND, NTW, PRO, 4>(a, s);
        default: return launch_gemm_ln_split_t<ND, NTW, PRO, 1>(a, s);
    }
}
template <int ND, int NTW>
static hipError_t launch_gemm_ln_split_pro(const mdt_gemm_args& a, hipStream_t s) {
    if (a.mod != nullptr && a.shift_off >= 0)
        return a.mod_stride == 0 ? launch_gemm_ln_split_x<ND, NTW, PRO_LN_MOD_BCAST>(a, s) : launch_gemm_ln_split_x<ND, NTW, PRO_LN_MOD_ROWS>(a, s);
    return launch_gemm_ln_split_x<ND, NTW, PRO_LN>(a, s);
}
// which products take it: the split image is there, LayerNorm prologue over whole rows of K = 384, column count a multiple of the
// 384-wide panels, plain output rows, and enough rows that the wide tiles are the choice anyway (the fused MLP's threshold)
static bool gemm_ln_split_applies(const mdt_gemm_args& a) {
    return a.Wp_split != nullptr && mdt_mlp_split_enabled() && a.ln && (a.K == 384 || a.K == 512) && a.N % 384 == 0 &&
           a.M >= mdt_split_min_rows() && a.batch <= 1 && !a.residual && a.gin == 1 && a.gout == 1 && a.goff == 0 && a.rowvec == nullptr && !a.aux_mode &&
           a.a_parts <= 4 && (a.lda & 3) == 0 && (a.ldo & 3) == 0 && g_mdt_gemm_force == 0;
}
static hipError_t launch_gemm_ln_split(const mdt_gemm_args& a, hipStream_t s) {
    // (the tile is written for any D <= 512 and 256-wide panels too; instantiated for the two shipped widths: MDT-V d = 384, MDT d = 512)
    return a.K == 512 ? launch_gemm_ln_split_pro<4, 3>(a, s) : launch_gemm_ln_split_pro<3, 3>(a, s);
}

// ---- the fused MLP sublayer in the three-way bf16 split form (k_mlp_split) ----
static int g_mlp_split = -1;   // MDT_HIP_MLP_SPLIT / mdt_op_set_mlp_split: 0 = the fp32 launch everywhere
bool mdt_mlp_split_enabled() {
    if (g_mlp_split < 0) { const char* e = getenv("MDT_HIP_MLP_SPLIT"); g_mlp_split = e ? atoi(e) : 1; }
    return g_mlp_split != 0;
}
extern "C" void mdt_op_set_mlp_split(int32_t on) { g_mlp_split = on < 0 ? -1 : (on != 0); }
int mdt_split_min_rows() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MDT_HIP_SPLIT_MIN_ROWS"); v = e ? std::max(1, atoi(e)) : 768; }
    return v;
}
bool mdt_mlp_split_supported(const mdt_gemm_args& f, const mdt_gemm_args& p) {
    return mdt_mlp_supported(f, p);
}
template <int NTW2, int PRO>
static hipError_t launch_mlp_split_t(const mdt_gemm_args& f, const mdt_gemm_args& p, const void* w1s, const void* w2s, float* parts,
                                     int64_t part_stride, hipStream_t s) {
    const int S = mdt_mlp_slices(f.K), gm = (f.M + 31) / 32;
    const int lds = mlp_split_lds_bytes(128 * NTW2);
    static bool attr_dev[MAX_DEVICES] = {false};
    bool& done = attr_dev[current_device()];
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_split<NTW2, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    const int per = (gm * S + 7) / 8;
    hipLaunchKernelGGL((k_mlp_split<NTW2, PRO>), dim3(8 * per), dim3(512), lds, s, f, p, (const char*)w1s, (const char*)w2s, parts,
                       part_stride, S, gm, g_zeros);
    return hipGetLastError();
}
template <int NTW2>
static hipError_t launch_mlp_split_pro(const mdt_gemm_args& f, const mdt_gemm_args& p, const void* w1s, const void* w2s, float* parts,
                                       int64_t part_stride, hipStream_t s) {
    if (f.mod != nullptr && f.shift_off >= 0)
        return f.mod_stride == 0 ? launch_mlp_split_t<NTW2, PRO_LN_MOD_BCAST>(f, p, w1s, w2s, parts, part_stride, s)
                                 : launch_mlp_split_t<NTW2, PRO_LN_MOD_ROWS>(f, p, w1s, w2s, parts, part_stride, s);
    return launch_mlp_split_t<NTW2, PRO_LN>(f, p, w1s, w2s, parts, part_stride, s);
}
hipError_t mdt_launch_mlp_split(const mdt_gemm_args& f, const mdt_gemm_args& p, const void* w1s, const void* w2s, float* parts,
                                int64_t part_stride, hipStream_t s) {
    if (!mdt_mlp_split_supported(f, p) || !w1s || !w2s) return hipErrorInvalidValue;
    hipError_t ze = ensure_zeros();
    if (ze != hipSuccess) return ze;
    switch (f.K / 128) {
        case 1: return launch_mlp_split_pro<1>(f, p, w1s, w2s, parts, part_stride, s);
        case 2: return launch_mlp_split_pro<2>(f, p, w1s, w2s, parts, part_stride, s);
        case 3: return launch_mlp_split_pro<3>(f, p, w1s, w2s, parts, part_stride, s);
        default: return launch_mlp_split_pro<4>(f, p, w1s, w2s, parts, part_stride, s);
    }
}

// a LayerNorm-prologue GEMM whose rows are the sum of a.a_parts slabs (2..4): wide tiles only
template <int NTW, int PRO, int XP>
static hipError_t launch_gemm_merge_t(const mdt_gemm_args& a, hipStream_t s) {
    const int NTC = 8 * NTW * 16;
    const int gn = (a.N + NTC - 1) / NTC, gm = (a.M + 31) / 32;
    const size_t lds = (size_t)32 * (a.K + 4) * sizeof(float);
    static size_t lds_attr_dev[MAX_DEVICES] = {0};
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_merge<NTW, PRO, XP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_gemm_merge<NTW, PRO, XP>), dim3(gn * gm), dim3(512), lds, s, a, a.K, gn, g_zeros);
    return hipGetLastError();
}
template <int NTW, int PRO>
static hipError_t launch_gemm_merge_x(const mdt_gemm_args& a, hipStream_t s) {
    switch (a.a_parts) {
        case 2: return launch_gemm_merge_t<NTW, PRO, 2>(a, s);
        case 3: return launch_gemm_merge_t<NTW, PRO, 3>(a, s);
        case 4: return launch_gemm_merge_t<NTW, PRO, 4>(a, s);
        default: return hipErrorInvalidValue;
    }
}
template <int NTW>
static hipError_t launch_gemm_merge_pro(const mdt_gemm_args& a, hipStream_t s) {
    if (a.mod != nullptr && a.shift_off >= 0)
        return a.mod_stride == 0 ? launch_gemm_merge_x<NTW, PRO_LN_MOD_BCAST>(a, s) : launch_gemm_merge_x<NTW, PRO_LN_MOD_ROWS>(a, s);
    return launch_gemm_merge_x<NTW, PRO_LN>(a, s);
}
static hipError_t launch_gemm_merge(const mdt_gemm_args& a, hipStream_t s) {
    if (!a.ln || a.K > 512 || a.residual || a.batch > 1 || a.aux_mode || a.a_parts > 4 || (a.N & 15)) return hipErrorInvalidValue;
    // 32 x 384 tiles when they divide N (qkv of d = 384: 1152), else 32 x 512
    return (a.N % 384 == 0) ? launch_gemm_merge_pro<3>(a, s) : launch_gemm_merge_pro<4>(a, s);
}

// ------------------------------------------------------------------------------------------------
// small attention: one workgroup per sample.  The sample's q / k / v rows (all heads) are staged in LDS with
// batched 16-byte loads, then thread (head, query row) runs softmax(q k^T) v out of LDS with the scores in
// registers.  10x10 / 10x4 / 4x4 score matrices are 0.1 % of the FLOPs, so this stays on the VALU.
// ------------------------------------------------------------------------------------------------
// body in mdt_tiles.h (attn_tile); gridDim.y workgroups share a sample, each taking H / gridDim.y heads
template <int HD, int TKC, bool ROPE>
__global__ __launch_bounds__(256) void k_attn(mdt_attn_args a, const float* __restrict__ rope_cos,
                                              const float* __restrict__ rope_sin, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // sample -> XCD as the GEMMs map row tiles -> XCD (xcd_remap): the q/k/v rows this workgroup reads were written
    // by GEMM tiles of the same XCD and the rows it writes are read there again (speed only)
    const int b = MDT_SAMPLE_REMAP(blockIdx.x, gridDim.x);
    attn_tile<HD, TKC, ROPE, false>(a, rope_cos, rope_sin, scale, b, blockIdx.y, gridDim.y, lds, threadIdx.x);
}

template <int HD, int TKC, bool ROPE>
static hipError_t launch_attn_tt(const mdt_attn_args& a, const float* rc, const float* rs, hipStream_t s) {
    constexpr int LP = HD == 48 ? 3 : (HD >= 32 ? 2 : 1);
    const int hs = (a.H % 2 == 0 && a.B >= 64) ? 2 : 1;  // head split: 2 half-size workgroups per sample
    const int Hl = a.H / hs;
    const size_t lds = ((size_t)(a.Tq + 2 * a.Tk) * Hl * HD + (size_t)Hl * a.Tq * 16 * LP) * sizeof(float);
    static size_t lds_attr_dev[MAX_DEVICES] = {0};
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn<HD, TKC, ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_attn<HD, TKC, ROPE>), dim3(a.B, hs), dim3(256), lds, s, a, rc, rs, 1.0f / sqrtf((float)HD));
    return hipGetLastError();
}

template <int HD>
static hipError_t launch_attn_t(const mdt_attn_args& a, const float* rc, const float* rs, hipStream_t s) {
    if constexpr (HD >= 32) {
        if (a.rope) {
            if (a.Tk <= 4) return launch_attn_tt<HD, 4, true>(a, rc, rs, s);
            if (a.Tk <= 10) return launch_attn_tt<HD, 10, true>(a, rc, rs, s);
            return launch_attn_tt<HD, 16, true>(a, rc, rs, s);
        }
    }
    if (a.Tk <= 4) return launch_attn_tt<HD, 4, false>(a, rc, rs, s);
    if (a.Tk <= 10) return launch_attn_tt<HD, 10, false>(a, rc, rs, s);
    return launch_attn_tt<HD, 16, false>(a, rc, rs, s);
}

hipError_t mdt_launch_attention(const mdt_attn_args& a, const float* rope_cos, const float* rope_sin,
                                hipStream_t s) {
    const int lp = a.hd == 48 ? 3 : (a.hd >= 32 ? 2 : 1);
    const size_t need = ((size_t)(a.Tq + 2 * a.Tk) * a.H * a.hd + (size_t)a.H * a.Tq * 16 * lp) * sizeof(float);
    if (a.H * a.Tq * lp > 256 || need > 160 * 1024) return hipErrorInvalidValue;
    switch (a.hd) {
        case 16: return launch_attn_t<16>(a, rope_cos, rope_sin, s);
        case 32: return launch_attn_t<32>(a, rope_cos, rope_sin, s);
        case 48: return launch_attn_t<48>(a, rope_cos, rope_sin, s);
        case 64: return launch_attn_t<64>(a, rope_cos, rope_sin, s);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// row LayerNorm (encoder final LN -> ctx): one wave per row, D <= 512
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ln_row(const float* __restrict__ in, int n4, int lane, f32x4 (&v)[2], float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c4 = lane + 64 * p;
        v[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c4 < n4) v[p] = *(const f32x4*)(in + 4 * c4);
        s += (v[p].x + v[p].y) + (v[p].z + v[p].w);
    }
    const float inv_d = 1.0f / (float)(n4 * 4);
    const float mean = wave_sum(s) * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c4 = lane + 64 * p;
        if (c4 < n4) {
            v[p] -= mean;
            sq += (v[p].x * v[p].x + v[p].y * v[p].y) + (v[p].z * v[p].z + v[p].w * v[p].w);
        }
    }
    rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + 1e-5f);
}

__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ out, float* __restrict__ out2,
                                                   int M, int D) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int n4 = D >> 2;
    f32x4 v[2];
    float rstd;
    ln_row(in + (int64_t)m * D, n4, lane, v, rstd);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c4 = lane + 64 * p;
        if (c4 < n4) {
            f32x4 y = v[p] * rstd * *(const f32x4*)(w + 4 * c4);
            if (b != nullptr) y += *(const f32x4*)(b + 4 * c4);
            *(f32x4*)(out + (int64_t)m * D + 4 * c4) = y;
            if (out2 != nullptr) *(f32x4*)(out2 + (int64_t)m * D + 4 * c4) = y;
        }
    }
}

// out2 (optional): a second copy of the rows (the encoder's final norm leaves the context in the workspace AND in the caller's
// latent_encoder_emb buffer: no device-to-device copy launch behind it)
hipError_t mdt_launch_layernorm(const float* in, const float* w, const float* b, float* out, int M, int D,
                                hipStream_t s, float* out2) {
    hipLaunchKernelGGL(k_layernorm, dim3((M + 3) / 4), dim3(256), 0, s, in, w, b, out, out2, M, D);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sigma embedding: e[r] = [sin(s f_j) | cos(s f_j)], s = ln(sigma_r)/4     (mdtv_transformer.py:13-25,239)
// ------------------------------------------------------------------------------------------------
__global__ void k_sigma_emb(const float* __restrict__ sigma, int64_t sstride, const float* __restrict__ freqs,
                            float* __restrict__ out, int R, int D) {
    const int half = D >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * half) return;
    const int r = idx / half, j = idx % half;
    const float s = logf(sigma[(int64_t)r * sstride]) / 4.0f;
    const float ang = s * freqs[j];
    out[(int64_t)r * D + j] = sinf(ang);
    out[(int64_t)r * D + half + j] = cosf(ang);
}

// per-step DDIM scalars from a device-resident schedule: steps[i] = {sigma_{i+1}/sigma_i as exp(-t')/exp(-t), -expm1(-h),
// sigma_{i+1}, sigma_i} with t = -ln sigma, h = t' - t        (gc_sampling.py:946-950, fp32 like the reference's tensors)
__global__ void k_ddim_steps(const float* __restrict__ sigmas, int n, float* __restrict__ steps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s0 = sigmas[i], s1 = sigmas[i + 1];
    const float t = -logf(s0), tn = -logf(s1);
    const float h = tn - t;
    steps[4 * i + 0] = expf(-tn) / expf(-t);
    steps[4 * i + 1] = -expm1f(-h);
    steps[4 * i + 2] = s1;
    steps[4 * i + 3] = s0;
}

hipError_t mdt_launch_ddim_steps(const float* sigmas_dev, int n, float* steps, hipStream_t s) {
    hipLaunchKernelGGL(k_ddim_steps, dim3((n + 63) / 64), dim3(64), 0, s, sigmas_dev, n, steps);
    return hipGetLastError();
}

hipError_t mdt_launch_sigma_emb(const float* sigma, int64_t sstride, const float* freqs, float* out, int R, int D,
                                hipStream_t s) {
    const int n = R * (D / 2);
    hipLaunchKernelGGL(k_sigma_emb, dim3((n + 255) / 256), dim3(256), 0, s, sigma, sstride, freqs, out, R, D);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// narrow-input Linear (+ activation): out[m][n] = act(b[n] + sum_a X[m][a] * WT[a][n]),  A <= 16   (VALU: the first layer
// of proprio_emb, Linear(proprio_dim = 8, 2d), mdtv_transformer.py:160-164).  `pre` (optional) keeps the pre-activation
// rows for the training tape.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_narrow_linear(const float* __restrict__ X, const float* __restrict__ WT,
                                                       const float* __restrict__ b, float* __restrict__ pre,
                                                       float* __restrict__ out, int64_t n, int A, int N, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t m = i / N;
    const int c = (int)(i - m * N);
    float acc = 0.f;
    for (int a = 0; a < A; ++a) acc = fmaf(X[m * A + a], WT[(int64_t)a * N + c], acc);
    acc += b[c];
    if (pre) pre[i] = acc;
    out[i] = apply_act1(acc, act);
}
hipError_t mdt_launch_narrow_linear(const float* X, const float* WT, const float* b, float* pre, float* out, int M, int A,
                                    int N, int act, hipStream_t s) {
    if (A < 1 || A > 16) return hipErrorInvalidValue;
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(k_narrow_linear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, WT, b, pre, out, n, A, N, act);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// action embedding: y[m][:] = (x[m][:] * c_in(sigma_b)) @ Wa^T + ba          (K = action_dim = 7: VALU)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_action_embed(const float* __restrict__ x, const float* __restrict__ sigma,
                                                      int64_t sstride, float sd, const float* __restrict__ WaT,
                                                      const float* __restrict__ ba, float* __restrict__ y, int M,
                                                      int A, int D, int rps) {
    const int n4 = D >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
    const float cin = sigma != nullptr ? edm_c_in(sigma[(int64_t)(m / rps) * sstride], sd) : 1.0f;
    f32x4 acc = *(const f32x4*)(ba + n);
    for (int c = 0; c < A; ++c) {
        const float xv = x[(int64_t)m * A + c] * cin;
        const f32x4 w = *(const f32x4*)(WaT + (int64_t)c * D + n);  // weight stored transposed: (A, D)
        acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y);
        acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
    }
    *(f32x4*)(y + (int64_t)m * D + n) = acc;
}

__global__ void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * Cc) return;
    const int r = idx / Cc, c = idx % Cc;
    dst[(int64_t)c * R + r] = src[idx];
}

hipError_t mdt_launch_transpose(const float* src, float* dst, int R, int Cc, hipStream_t s) {
    hipLaunchKernelGGL(k_transpose, dim3((R * Cc + 255) / 256), dim3(256), 0, s, src, dst, R, Cc);
    return hipGetLastError();
}

hipError_t mdt_launch_action_embed(const float* x, const float* sigma, int64_t sstride, float sd, const float* Wa,
                                   const float* ba, float* y, int M, int A, int D, int rps, hipStream_t s) {
    const int64_t n = (int64_t)M * (D / 4);
    hipLaunchKernelGGL(k_action_embed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, sigma, sstride, sd, Wa,
                       ba, y, M, A, D, rps);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The once-per-call scalar work of the DDIM sampler as ONE launch (round 5; it was a host-to-device copy of the schedule +
// k_ddim_steps + k_sigma_emb + k_action_embed: four dependent ~4.5 us launches in front of every sampler call, a tenth of a
// rollout-sized call).  A host schedule travels in the kernel arguments; a device schedule is read in place.  Blocks
// [0, n_act) embed the first noisy actions (k_action_embed's arithmetic with c_in(sigma_0)), blocks [n_act, n_act + n_emb) write
// the sinusoidal sigma embeddings of all steps (k_sigma_emb's), the last block the per-step DDIM scalars (k_ddim_steps's) --
// each from the schedule itself, so nothing in the launch depends on anything else in it, and every value has the bits the
// separate kernels gave.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_prep(const float* __restrict__ sig_dev, mdt_sched_arg sv, int n_steps,
                                                     float* __restrict__ steps, const float* __restrict__ freqs,
                                                     float* __restrict__ sig_e, int D, const float* __restrict__ x, float sd,
                                                     const float* __restrict__ WaT, const float* __restrict__ ba,
                                                     float* __restrict__ y, int M, int A, int n_act, int n_emb) {
    const int b = blockIdx.x;
    if (b < n_act) {
        const int n4 = D >> 2;
        const int64_t idx = (int64_t)b * 256 + threadIdx.x;
        if (idx >= (int64_t)M * n4) return;
        const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
        const float cin = edm_c_in(sig_dev ? sig_dev[0] : sv.s[0], sd);
        f32x4 acc = *(const f32x4*)(ba + n);
        for (int c = 0; c < A; ++c) {
            const float xv = x[(int64_t)m * A + c] * cin;
            const f32x4 w = *(const f32x4*)(WaT + (int64_t)c * D + n);
            acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y);
            acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
        }
        *(f32x4*)(y + (int64_t)m * D + n) = acc;
    } else if (b < n_act + n_emb) {
        const int half = D >> 1;
        const int idx = (b - n_act) * 256 + threadIdx.x;
        if (idx >= n_steps * half) return;
        const int r = idx / half, j = idx % half;
        const float sg = sig_dev ? sig_dev[r] : sv.s[r];
        const float ang = (logf(sg) / 4.0f) * freqs[j];
        sig_e[(int64_t)r * D + j] = sinf(ang);
        sig_e[(int64_t)r * D + half + j] = cosf(ang);
    } else {
        const int i = threadIdx.x;
        if (i >= n_steps) return;
        const float s0 = sig_dev ? sig_dev[i] : sv.s[i], s1 = sig_dev ? sig_dev[i + 1] : sv.s[i + 1];
        const float t = -logf(s0), tn = -logf(s1);
        const float h = tn - t;
        steps[4 * i + 0] = expf(-tn) / expf(-t);
        steps[4 * i + 1] = -expm1f(-h);
        steps[4 * i + 2] = s1;
        steps[4 * i + 3] = s0;
    }
}
// sigmas_dev or sigmas_host (exactly one non-null): n_steps + 1 levels, n_steps <= MDT_SCHED_MAX; sig_e == nullptr: no embeddings
hipError_t mdt_launch_sample_prep(const float* sigmas_dev, const float* sigmas_host, int n_steps, float* steps, const float* freqs,
                                  float* sig_e, int D, const float* x, float sd, const float* Wa, const float* ba, float* y, int M,
                                  int A, hipStream_t s) {
    if (n_steps < 1 || n_steps > MDT_SCHED_MAX || (!sigmas_dev) == (!sigmas_host) || (D & 3)) return hipErrorInvalidValue;
    mdt_sched_arg sv;
    memset(&sv, 0, sizeof sv);
    if (sigmas_host) memcpy(sv.s, sigmas_host, (size_t)(n_steps + 1) * sizeof(float));
    const int n_act = (int)(((int64_t)M * (D / 4) + 255) / 256);
    const int n_emb = sig_e ? (n_steps * (D / 2) + 255) / 256 : 0;
    hipLaunchKernelGGL(k_sample_prep, dim3(n_act + n_emb + 1), dim3(256), 0, s, sigmas_dev, sv, n_steps, steps, freqs, sig_e, D, x, sd,
                       Wa, ba, y, M, A, n_act, n_emb);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// action head: decoder LN -> action_pred -> EDM combine -> (DDIM update) -> (next step's embedding)
// one wave per action-token row; A <= 16
// ------------------------------------------------------------------------------------------------
// body in mdt_tiles.h (head_rows): each wave handles ONE row
template <int AMAX, int XP>
__global__ __launch_bounds__(256) void k_head(mdt_head_args a, const float* __restrict__ zeros) {
    const int base = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (base >= a.M) return;  // wave-uniform
    head_rows<AMAX, false, XP, 1>(a, base, threadIdx.x & 63, zeros);
}

hipError_t mdt_launch_head(const mdt_head_args& a, hipStream_t s) {
    hipError_t e = ensure_zeros();
    if (e != hipSuccess) return e;
    const int grid = (a.M + 3) / 4;  // 4 waves x 1 row per workgroup
    if (a.y_parts > 1) {  // rows = the sum of a fused MLP's slabs
        if (a.A > 8 || a.y_parts > 4) return hipErrorInvalidValue;
        switch (a.y_parts) {
            case 2: hipLaunchKernelGGL((k_head<8, 2>), dim3(grid), dim3(256), 0, s, a, g_zeros); break;
            case 3: hipLaunchKernelGGL((k_head<8, 3>), dim3(grid), dim3(256), 0, s, a, g_zeros); break;
            default: hipLaunchKernelGGL((k_head<8, 4>), dim3(grid), dim3(256), 0, s, a, g_zeros); break;
        }
        return hipGetLastError();
    }
    if (a.A <= 8) hipLaunchKernelGGL((k_head<8, 1>), dim3(grid), dim3(256), 0, s, a, g_zeros);
    else hipLaunchKernelGGL((k_head<16, 1>), dim3(grid), dim3(256), 0, s, a, g_zeros);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Collapsed cross-attention.  The decoder's cross-attention sublayer
//     y += c_proj( softmax_causal( (ln3(y) Wq^T + bq) K^T / sqrt(hd) ) V )          (transformer_blocks.py:300-302)
// attends to only Te (= 4) context tokens whose K / V do not depend on sigma or on the actions.  Per sample b,
// head h and context token j the two projections fold into the context once per sampler call:
//     U [b][h][j][:] = 1/sqrt(hd) * sum_{d in h} K[b][j][d] * Wq[d][:]        score = ln3(y) . U + c
//     c [b][h][j]    = 1/sqrt(hd) * sum_{d in h} K[b][j][d] * bq[d]
//     Wf[b][h][j][:] = sum_{d in h} V[b][j][d] * Wo[:][d]                    out   = sum_{h,j} P[h][j] * Wf
// (k_xattn_fold), after which every denoising step runs the whole sublayer as ONE kernel per sample
// (k_xattn_apply): LayerNorm, H*Te dot products of length d, H masked softmaxes over Te values, a
// (H*Te) x d combination and the residual add -- 12x fewer FLOPs than the two d x d projections and three
// launches (q GEMM, attention, c_proj GEMM) fewer per block and step.  Exact algebra, fp32 rounding differs.
// Both folded matrices are stored as MFMA WEIGHT IMAGES (the fragment order of k_pack_weight) with every head padded to 4
// context tokens, p = 4 h + j (NPP = 4 H rows; rows of absent tokens are zero):
//     U  image: weight (NPP x D):   block (p / 16, d / 16), lane (p % 16) + 16 ((d % 16) / 4), element d % 4
//     Wf image: weight (D x NPP):   block (n / 16, p / 16), lane (n % 16) + 16 ((p % 16) / 4), element p % 4  ( = j )
// so that k_xattn_apply's two contractions are a handful of v_mfma_f32_16x16x4_f32 per wave (mdt_tiles.h: xattn_tile).
// ------------------------------------------------------------------------------------------------
// the folds of ALL decoder blocks of one sampler call are one launch: blockIdx.z picks the block's argument set
struct mdt_xfold_table { mdt_xfold_args a[8]; };
// One workgroup = (8 samples, a group of 4 heads = one 16-row block of the images, decoder block, feature part): both folds as
// MFMA products with K = hd.  The 32 activation rows are (sample, token) pairs -- 4 rows per sample, rows of absent tokens zero,
// which also zeroes their rows of the images --
//   U : transposed form, A = fragment of the packed image of Wq^T (features x head dims), B = K rows: the lane ends with 4
//       consecutive features of one (sample, token) = one 16-byte slot of the U image
//   Wf: plain form, A = V rows, B = fragment of the packed image of c_proj.weight: the lane ends with the 4 tokens of one
//       (sample, feature) = one 16-byte slot of the Wf image
// wave w owns head 4 hg + w % 4 and the feature tiles part, part + P, ... (part = 2 fs + w / 4 of P = 2 FS parts).
// Round 5: both weight operands are MFMA fragment images, one coalesced 16-byte load per lane and k-block (round 3-4 read query.weight
// and the transposed c_proj.weight element by element: 24 four-byte loads per lane and feature tile, 63 us per call at B = 256 with
// the matrix pipe busy for 14 of them, 15 us at B = 1); the next tile's fragments are requested before the current tile's MFMAs; the
// four heads of a workgroup write whole 256-byte / 1-KiB runs of the images; small batches spread the feature tiles over more
// workgroups (FS).  Same products in the same order: the images have the bits the earlier kernel wrote.
// (The first form of all: one thread per feature, K / V as scalar operands of 1500 FMAs per thread, 93 us.)
template <int TE>
__global__ __launch_bounds__(512) void k_xattn_fold(mdt_xfold_table tab, int FS) {
    constexpr int SB = 8, K16MAX = 4;     // samples per workgroup; hd <= 64
    const mdt_xfold_args& a = tab.a[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hg = blockIdx.y / FS, fs = blockIdx.y - hg * FS;
    const int h = 4 * hg + (wave & 3);
    const int bg = blockIdx.x * SB;       // first sample of the group
    const int HD = a.hd, D = a.D, NPP = 4 * a.H, K16 = D >> 4, KP16 = NPP >> 4, KH = HD >> 4;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float scale = 1.0f / sqrtf((float)HD);
    // ---- activation fragments: row lane % 16 of M-tile mt = (sample bg + 4 mt + (lane % 16) / 4, token (lane % 16) % 4) ----
    const int rs = (lane & 15) >> 2, rt = lane & 3, kq = 4 * (lane >> 4);
    f32x4 kf[2][K16MAX], vf[2][K16MAX];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int smp = bg + 4 * mt + rs;
        const bool live = smp < a.B && rt < TE;
        const float* row = a.kv + (int64_t)(min(smp, a.B - 1) * TE + min(rt, TE - 1)) * a.ldkv + h * HD + kq;  // K at +0, V at +D
#pragma unroll
        for (int kc = 0; kc < K16MAX; ++kc) {
            const int ko = min(kc, KH - 1) * 16;
            kf[mt][kc] = sel4(live, ldg4(row + ko), zero4);
            vf[mt][kc] = sel4(live, ldg4(row + D + ko), zero4);
        }
    }
    // ---- weight fragments: block (feature tile nt, k-block kb0 + kc) of either image, 1 KiB in lane order ----
    const int P = 2 * FS, part = 2 * fs + (wave >> 2);
    const int kb0 = (h * HD) >> 4;
    f32x4 wq[K16MAX], wo[K16MAX];
    auto request = [&](int nt, f32x4* q, f32x4* o) {
        const int ntc = min(nt, K16 - 1);
#pragma unroll
        for (int kc = 0; kc < K16MAX; ++kc) {
            const int64_t off = (((int64_t)ntc * K16 + kb0 + min(kc, KH - 1)) * 64 + lane) * 4;
            q[kc] = ldg4(a.WqT_p + off);
            o[kc] = ldg4(a.Wo_p + off);
        }
    };
    request(part, wq, wo);
    for (int nt = part; nt < K16; nt += P) {
        f32x4 wqn[K16MAX], won[K16MAX];
        request(nt + P, wqn, won);  // (clamped past the end: an L2 hit nobody uses)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 accU = zero4, accW = zero4;
#pragma unroll
            for (int kc = 0; kc < K16MAX; ++kc)
                if (kc < KH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        accU = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[kc][e], kf[mt][kc][e], accU, 0, 0, 0);
                        accW = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[mt][kc][e], wo[kc][e], accW, 0, 0, 0);
                    }
                }
            // U: lane holds (sample bg + 4 mt + rs, token rt), features 16 nt + 4 (lane / 16) .. + 3
            {
                const int smp = bg + 4 * mt + rs, pp = 4 * h + rt;
                if (smp < a.B)
                    *(f32x4*)(a.U + (int64_t)smp * NPP * D + (((int64_t)(pp >> 4) * K16 + nt) * 64 + (pp & 15) + 16 * (lane >> 4)) * 4) = accU * scale;
            }
            // Wf: lane holds (sample bg + 4 mt + lane / 16, tokens 0..3), feature 16 nt + lane % 16
            {
                const int smp = bg + 4 * mt + (lane >> 4);
                if (smp < a.B)
                    *(f32x4*)(a.Wf + (int64_t)smp * NPP * D + (((int64_t)nt * KP16 + (h >> 2)) * 64 + (lane & 15) + 16 * (h & 3)) * 4) = accW;
            }
        }
#pragma unroll
        for (int kc = 0; kc < K16MAX; ++kc) { wq[kc] = wqn[kc]; wo[kc] = won[kc]; }
    }
    if (fs == 0 && threadIdx.x < SB * 16) {   // c[b][h][j] for the group's four heads
        const int hh = 4 * hg + (threadIdx.x >> 5), sb = (threadIdx.x >> 2) & 7, j = threadIdx.x & 3;
        if (bg + sb < a.B) {
            float acc = 0.f;
            if (j < TE) {
                const float* kr = a.kv + (int64_t)((bg + sb) * TE + j) * a.ldkv + hh * HD;
                for (int d = 0; d < HD; ++d) acc = fmaf(a.bq[hh * HD + d], kr[d], acc);
            }
            a.c[(int64_t)(bg + sb) * NPP + hh * 4 + j] = acc * scale;
        }
    }
}

// n argument sets of equal shape (B, H, hd, D, Te) -- the decoder blocks of one model -- folded by one launch (n <= 8 per
// launch; more go in groups)
hipError_t mdt_launch_xattn_fold_n(const mdt_xfold_args* sets, int n, hipStream_t s) {
    constexpr int SB = 8;
    if (n < 1) return hipErrorInvalidValue;
    const mdt_xfold_args& a = sets[0];
    if (a.D > 512 || a.D % 128 || a.Te < 1 || a.Te > 4 || (a.H != 4 && a.H != 8) || a.hd % 16 || a.hd > 64 || (a.ldkv & 3) ||
        a.H * a.hd != a.D)
        return hipErrorInvalidValue;
    for (int i = 1; i < n; ++i)
        if (sets[i].B != a.B || sets[i].H != a.H || sets[i].hd != a.hd || sets[i].D != a.D || sets[i].Te != a.Te)
            return hipErrorInvalidValue;
    for (int i0 = 0; i0 < n; i0 += 8) {
        mdt_xfold_table tab;
        const int cnt = n - i0 < 8 ? n - i0 : 8;
        for (int i = 0; i < 8; ++i) tab.a[i] = sets[i0 + (i < cnt ? i : 0)];
        // feature parts: enough workgroups to fill the chip when the batch alone does not (a rollout call: 1 sample group)
        const int groups = (a.B + SB - 1) / SB, base = groups * (a.H / 4) * cnt;
        const int FS = std::max(1, std::min((256 + base - 1) / base, (a.D >> 4) / 2));
        const dim3 grid(groups, (a.H / 4) * FS, cnt);
        switch (a.Te) {
            case 1: hipLaunchKernelGGL((k_xattn_fold<1>), grid, dim3(512), 0, s, tab, FS); break;
            case 2: hipLaunchKernelGGL((k_xattn_fold<2>), grid, dim3(512), 0, s, tab, FS); break;
            case 3: hipLaunchKernelGGL((k_xattn_fold<3>), grid, dim3(512), 0, s, tab, FS); break;
            default: hipLaunchKernelGGL((k_xattn_fold<4>), grid, dim3(512), 0, s, tab, FS); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t mdt_launch_xattn_fold(const mdt_xfold_args& a, hipStream_t s) { return mdt_launch_xattn_fold_n(&a, 1, s); }

// One workgroup (512 threads) per sample; body in mdt_tiles.h (xattn_tile)
template <int NPP>
__global__ __launch_bounds__(512) void k_xattn_apply(mdt_xapply_args a, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    xattn_tile<NPP, false>(a, MDT_SAMPLE_REMAP(blockIdx.x, gridDim.x), lds, zeros, threadIdx.x);
}

// which configurations the collapsed path covers (others keep the q GEMM + attention + c_proj GEMM sequence)
bool mdt_xattn_apply_supported(int D, int H, int Te, int Ta) {
    return D >= 128 && D <= 512 && D % 128 == 0 && (H == 4 || H == 8) && Te >= 1 && Te <= 4 && Ta >= 1 && Ta <= 16;
}
// floats of LDS xattn_tile needs
size_t mdt_xattn_lds_floats(int D, int H) {
    const int npp = 4 * H, ks = 8 / (npp / 16);
    return (size_t)16 * (D + 4) + (size_t)(ks + 1) * 16 * (npp + 4);
}

// Cross-attention + the Linear behind it in one launch (rollout batches): `x` as for mdt_launch_xattn_apply with a SEPARATE
// output array (x.y_out != x.y: the workgroups that repeat a sample's cross-attention read x.y while one of them writes), `g` =
// the Linear on the same rows (g.A == x.y, LayerNorm prologue, K = x.D, M = x.B * x.Ta, no residual, plain row mapping).
bool mdt_xattn_gemm_supported(const mdt_xapply_args& x, const mdt_gemm_args& g) {
    return w_image_ok(g.N, g.K) && mdt_xattn_apply_supported(x.D, x.H, x.Te, x.Ta) && x.y_out != nullptr && x.y_out != x.y && g.ln && g.K == x.D && g.A == x.y &&
           g.lda == x.D &&
           g.M == x.B * x.Ta && !(g.N & 15) && !g.residual && g.batch <= 1 && !g.aux_mode && g.a_parts <= 1 && g.gin == 1 &&
           g.gout == 1 && g.goff == 0 && g.rows_per_sample == x.Ta;
}
template <int NPP>
static hipError_t launch_xattn_gemm_t(const mdt_xapply_args& x, const mdt_gemm_args& g, hipStream_t s) {
    const size_t lds = ((size_t)16 * (x.D + 4) + mdt_xattn_lds_floats(x.D, x.H)) * sizeof(float);
    static size_t lds_attr_dev[MAX_DEVICES] = {0};
    size_t& lds_attr = lds_attr_dev[current_device()];
    if (lds > lds_attr) {  // up to 78 KB of dynamic LDS at d = 512
        hipError_t e = hipFuncSetAttribute((const void*)k_xattn_gemm_smallm<NPP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_attr = lds;
    }
    hipLaunchKernelGGL((k_xattn_gemm_smallm<NPP>), dim3(g.N >> 4, x.B), dim3(512), lds, s, x, g, g_zeros);
    return hipGetLastError();
}
hipError_t mdt_launch_xattn_gemm(const mdt_xapply_args& x, const mdt_gemm_args& g, hipStream_t s) {
    if (!mdt_xattn_gemm_supported(x, g)) return hipErrorInvalidValue;
    hipError_t e = ensure_zeros();
    if (e != hipSuccess) return e;
    return x.H == 8 ? launch_xattn_gemm_t<32>(x, g, s) : launch_xattn_gemm_t<16>(x, g, s);
}

hipError_t mdt_launch_xattn_apply(const mdt_xapply_args& a, hipStream_t s) {
    if (!mdt_xattn_apply_supported(a.D, a.H, a.Te, a.Ta)) return hipErrorInvalidValue;
    hipError_t e = ensure_zeros();
    if (e != hipSuccess) return e;
    const size_t lds = mdt_xattn_lds_floats(a.D, a.H) * sizeof(float);
    if (a.H == 8) hipLaunchKernelGGL((k_xattn_apply<32>), dim3(a.B), dim3(512), lds, s, a, g_zeros);
    else hipLaunchKernelGGL((k_xattn_apply<16>), dim3(a.B), dim3(512), lds, s, a, g_zeros);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GCDenoiser.loss pieces (score_wrappers.py:59-63)
// ------------------------------------------------------------------------------------------------
__global__ void k_noise_input(const float* __restrict__ act, const float* __restrict__ noise,
                              const float* __restrict__ sigma, float* __restrict__ noised, int64_t n, int per_sample) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    noised[i] = noise ? act[i] + noise[i] * sigma[i / per_sample] : act[i];  // noise == nullptr: the rows are given noisy
}

// deterministic two-stage reduction: loss = mean_i (F_i - (a_i - c_skip*noised_i)/c_out)^2
// Stage 1: up to MDT_LOSS_PARTS workgroups, each over a contiguous slice, element-strided (coalesced) with four elements of a
// thread in flight; stage 2: one workgroup adds the partials in a fixed order.  (One 1024-thread workgroup over everything
// took 51 us at B = 1024 walking elements and 104 us walking whole samples per thread -- one CU's worth of memory requests
// in flight either way -- of a 10 ms training step.)
__global__ __launch_bounds__(256) void k_loss_partial(const float* __restrict__ F, const float* __restrict__ act,
                                                      const float* __restrict__ noised, const float* __restrict__ sigma, float sd,
                                                      int n, int per_sample, int chunk, float* __restrict__ part) {
    __shared__ float red[4];
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    const float sd2 = sd * sd;
    float s = 0.f;
    for (int i0 = lo + threadIdx.x; i0 < hi; i0 += 1024) {
        float f[4], a[4], x[4], sg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + 256 * u, hi - 1);
            f[u] = F[i]; a[u] = act[i]; x[u] = noised[i]; sg[u] = sigma[i / per_sample];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float den2 = sg[u] * sg[u] + sd2;
            const float c_skip = sd2 / den2, inv_c_out = sqrtf(den2) / (sg[u] * sd);
            const float d = f[u] - (a[u] - c_skip * x[u]) * inv_c_out;
            if (i0 + 256 * u < hi) s = fmaf(d, d, s);
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void k_loss_final(const float* __restrict__ part, int nparts, float inv_n, float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}

hipError_t mdt_launch_noise_input(const float* act, const float* noise, const float* sigma, float* noised, int64_t n,
                                  int per_sample, hipStream_t s) {
    hipLaunchKernelGGL(k_noise_input, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, act, noise, sigma, noised, n,
                       per_sample);
    return hipGetLastError();
}

hipError_t mdt_launch_loss_reduce(const float* F, const float* act, const float* noised, const float* sigma, float sd,
                                  int64_t n, int per_sample, float* loss, float* part, hipStream_t s) {
    if (n < 1 || n > 0x7fffffff || !part) return hipErrorInvalidValue;
    const int chunk = (int)std::max<int64_t>(1024, ((n + MDT_LOSS_PARTS - 1) / MDT_LOSS_PARTS + 255) / 256 * 256);
    const int nparts = (int)((n + chunk - 1) / chunk);
    hipLaunchKernelGGL(k_loss_partial, dim3(nparts), dim3(256), 0, s, F, act, noised, sigma, sd, (int)n, per_sample, chunk, part);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(256), 0, s, part, nparts, 1.0f / (float)n, loss);
    return hipGetLastError();
}

hipError_t mdt_launch_pack_weight(const float* w, int n_rows, int K, float* packed, int n_off, hipStream_t s) {
    const int64_t n = (int64_t)n_rows * K;
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, n_rows, K, packed, n_off,
                       K / 16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Perceiver resampler pieces (perceiver_resampler.py)
// ------------------------------------------------------------------------------------------------
// x_f + time_pos_emb[t] * mask[b][t]                                    (perceiver_resampler.py:141-148)
__global__ void k_add_time_emb(const float* __restrict__ media, const float* __restrict__ tpe,
                               const uint8_t* __restrict__ mask, float* __restrict__ out, int64_t n4, int T, int n,
                               int d4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int64_t row = i / d4;
    const int c = (int)(i - row * d4);
    const int64_t frame = row / n;  // b * T + t
    const int t = (int)(frame % T);
    const float mk = mask ? (mask[frame] ? 1.f : 0.f) : 1.f;
    const f32x4 x = ldg4(media + i * 4), e = ldg4(tpe + ((int64_t)t * d4 + c) * 4);
    f32x4 o;
    o.x = x.x + e.x * mk; o.y = x.y + e.y * mk; o.z = x.z + e.z * mk; o.w = x.w + e.w * mk;
    st4(out + i * 4, o);
}

// repeat(latents, 'q d -> b q d')                                       (perceiver_resampler.py:154)
__global__ void k_bcast_rows(const float* __restrict__ src, float* __restrict__ out, int64_t n4, int per4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    st4(out + i * 4, ldg4(src + (i % per4) * 4));
}

hipError_t mdt_launch_add_time_emb(const float* media, const float* tpe, const uint8_t* mask, float* out, int64_t B,
                                   int T, int n, int D, hipStream_t s) {
    const int64_t n4 = B * T * n * (D / 4);
    hipLaunchKernelGGL(k_add_time_emb, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, media, tpe, mask, out, n4, T,
                       n, D / 4);
    return hipGetLastError();
}

hipError_t mdt_launch_bcast_rows(const float* src, float* out, int64_t B, int R, int D, hipStream_t s) {
    const int64_t n4 = B * R * (D / 4);
    hipLaunchKernelGGL(k_bcast_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, src, out, n4, R * (D / 4));
    return hipGetLastError();
}

// Few queries (the learnt latents, <= 16) over many keys (media tokens + latents, ~400): one workgroup per
// (sample, head).                                            (PerceiverAttentionLayer.forward, :51-82)
//   1. thread per key: the key's HD floats in registers, dotted with every (pre-scaled) query held in LDS
//   2. wave per query: max / exp / sum over the keys (scores stay in LDS)
//   3. thread (key group g, feature d): P.V over the group's keys for all queries, V read once and coalesced over
//      d; the G = 256 / HD partial sums meet in LDS.
// 0.3 % of the resampler's FLOPs (the K/V projections of the media tokens are the other 99 %), VALU.
template <int HD>
__global__ __launch_bounds__(256) void k_attn_long(const float* __restrict__ q, int64_t ldq,
                                                   const float* __restrict__ k, const float* __restrict__ v,
                                                   int64_t ldkv, float* __restrict__ out, int64_t ldo, int Tq, int Tk,
                                                   float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int H4 = HD / 4, G = 256 / HD, QMAX = 16;
    const int tid = threadIdx.x, b = blockIdx.x, h = blockIdx.y;
    const int Tkp = (Tk + 3) & ~3;
    float* qs = lds;                    // [Tq][HD]   scaled queries
    float* S = qs + Tq * HD;            // [Tq][Tkp]  scores -> unnormalised probabilities
    float* inv = S + Tq * Tkp;          // [16]       1 / sum
    float* red = inv + QMAX;            // [G][Tq][HD]
    for (int i = tid; i < Tq * HD; i += 256) {
        const int qi = i / HD, d = i - qi * HD;
        qs[i] = q[((int64_t)b * Tq + qi) * ldq + h * HD + d] * scale;
    }
    __syncthreads();
    const float* kb = k + (int64_t)b * Tk * ldkv + h * HD;
    const float* vb = v + (int64_t)b * Tk * ldkv + h * HD;
    for (int f0 = 0; f0 < Tk; f0 += 256) {
        const int f = min(f0 + tid, Tk - 1);
        f32x4 kr[H4];
#pragma unroll
        for (int c = 0; c < H4; ++c) kr[c] = ldg4(kb + (int64_t)f * ldkv + c * 4);
        for (int qi = 0; qi < Tq; ++qi) {
            const f32x4* qv = reinterpret_cast<const f32x4*>(qs + qi * HD);
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < H4; ++c) {
                const f32x4 t = qv[c];
                acc = fmaf(t.x, kr[c].x, acc); acc = fmaf(t.y, kr[c].y, acc);
                acc = fmaf(t.z, kr[c].z, acc); acc = fmaf(t.w, kr[c].w, acc);
            }
            if (f0 + tid < Tk) S[qi * Tkp + f] = acc;
        }
    }
    __syncthreads();
    const int w = tid >> 6, lane = tid & 63;
    for (int qi = w; qi < Tq; qi += 4) {
        float* row = S + qi * Tkp;
        float mx = -INFINITY;
        for (int f = lane; f < Tk; f += 64) mx = fmaxf(mx, row[f]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int f = lane; f < Tk; f += 64) {
            const float e = expf(row[f] - mx);
            row[f] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) inv[qi] = 1.f / sum;
    }
    __syncthreads();
    const int g = tid / HD, d = tid - g * HD;
    float acc[QMAX];
#pragma unroll
    for (int qi = 0; qi < QMAX; ++qi) acc[qi] = 0.f;
    for (int f0 = g; f0 < Tk; f0 += 4 * G) {
        float vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[u] = vb[(int64_t)min(f0 + u * G, Tk - 1) * ldkv + d];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = f0 + u * G;
            if (f < Tk) {
#pragma unroll
                for (int qi = 0; qi < QMAX; ++qi)
                    if (qi < Tq) acc[qi] = fmaf(S[qi * Tkp + f], vv[u], acc[qi]);
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < QMAX; ++qi)
        if (qi < Tq) red[(g * Tq + qi) * HD + d] = acc[qi];
    __syncthreads();
    for (int i = tid; i < Tq * HD; i += 256) {
        const int qi = i / HD, dd = i - qi * HD;
        float t = 0.f;
        for (int gg = 0; gg < G; ++gg) t += red[(gg * Tq + qi) * HD + dd];
        out[((int64_t)b * Tq + qi) * ldo + h * HD + dd] = t * inv[qi];
    }
}

static size_t attn_long_lds(int hd, int Tq, int Tk) {
    return (size_t)(Tq * hd + Tq * ((Tk + 3) & ~3) + 16 + 256 * Tq) * sizeof(float);
}

bool mdt_attention_long_supported(int hd, int Tq, int Tk) {
    return (hd == 16 || hd == 32 || hd == 64) && Tq >= 1 && Tq <= 16 && Tk >= 1 && Tk <= 4096 &&
           attn_long_lds(hd, Tq, Tk) <= 160 * 1024;
}

template <int HD>
static hipError_t launch_attn_long_t(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                     float* out, int64_t ldo, int B, int H, int Tq, int Tk, float scale, hipStream_t s) {
    const size_t lds = attn_long_lds(HD, Tq, Tk);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_long<HD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_attn_long<HD>), dim3(B, H), dim3(256), lds, s, q, ldq, k, v, ldkv, out, ldo, Tq, Tk, scale);
    return hipGetLastError();
}

hipError_t mdt_launch_attention_long(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                     float* out, int64_t ldo, int B, int H, int hd, int Tq, int Tk, float scale,
                                     hipStream_t s) {
    if (!mdt_attention_long_supported(hd, Tq, Tk)) return hipErrorInvalidValue;
    switch (hd) {
        case 16: return launch_attn_long_t<16>(q, ldq, k, v, ldkv, out, ldo, B, H, Tq, Tk, scale, s);
        case 32: return launch_attn_long_t<32>(q, ldq, k, v, ldkv, out, ldo, B, H, Tq, Tk, scale, s);
        default: return launch_attn_long_t<64>(q, ldq, k, v, ldkv, out, ldo, B, H, Tq, Tk, scale, s);
    }
}
