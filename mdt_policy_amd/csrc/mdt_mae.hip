// mdt_mae.hip -- op-level kernels of the masked generative foresight head (include/mdt_mae.h): unmasked multi-head
// self-attention over ~100 tokens with its backward, and the exported RMSNorm / SwishGLU row ops.
//
// The attention of the shipped decoder is 102 tokens x 8 heads of 24 (masked_transformer_decoder.py:68-121 builds
// voltron Blocks of d = 192).  One workgroup of 8 waves per (sample, head) keeps q / k / v (and dO, O) and the T x T scores
// in LDS; every product is 16 x 16 x 4 f32 MFMA tiles over those LDS operands (16-byte fragment reads, see below), the row
// softmax in between is VALU work on 16 lanes per row.  (Round 2 started with 4 x 4 register-blocked VALU products: they
// were LDS-bandwidth bound at 1 flop per byte read; the MFMA tiles need a quarter of that.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdt_internal.h"
#include "mdt_device.h"

namespace {

constexpr int TMAX = 128;

#ifdef MDT_DEBUG_TIMING
// tuning-only build: thread 0 of workgroup (0, 0) stamps the shader clock at its phase boundaries (mdt_mae_debug_ts)
__device__ unsigned long long g_mae_ts[16];
#define MAE_TS(i) if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_mae_ts[i] = __builtin_readcyclecounter();
// ... and thread 0 of EVERY workgroup (the first 8192) stamps the constant-rate 100 MHz clock at its start and end, with the
// hardware id of where it ran (mdt_mae_debug_wg): the residency picture of a launch
__device__ unsigned long long g_mae_wg[3 * 8192];
#define MAE_WG(e)                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.y < 8192) {                            \
        const unsigned w_ = blockIdx.x + gridDim.x * blockIdx.y;                                     \
        g_mae_wg[3 * w_ + e] = __builtin_amdgcn_s_memrealtime();                                     \
        if (e == 0) g_mae_wg[3 * w_ + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
    }
#else
#define MAE_TS(i)
#define MAE_WG(e)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int score_stride(int T) { return ((T + 15) & ~15) + 4; }  // floats: 4 (mod 8), >= ceil16(T)

// rows [0, T) of NB (T, HD) head slices: global (row stride ld[b]) -> LDS (row stride HD + 4, buffers T16 * (HD + 4) floats
// apart, T16 = ceil16(T)); rows T .. T16-1 are zeroed (reduction padding of the MFMA products).  All loads of a sweep (4 per
// buffer and thread) are requested from clamped addresses before the first one is consumed -- a load behind a branch costs
// one memory round trip each.
template <int HD, int NT, int NB>
__device__ __forceinline__ void load_rows(const float* const (&src)[NB], const int64_t (&ld)[NB], float* dst, int T, int tid) {
    constexpr int H4 = HD / 4, ST = HD + 4, U = 4;
    const int T16 = (T + 15) & ~15, n = T16 * H4;
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
                    *(f32x4*)(dst + b * T16 * ST + t * ST + 4 * c) = t < T ? v[b][u] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// --- MFMA products over LDS-resident operands (v_mfma_f32_16x16x4_f32; 64 lanes = 16 rows/cols x 4 k-groups g) -----------
// One instruction multiplies a 16 x 4 by a 4 x 16 panel; WHICH four reduction indices a step covers is free as long as both
// operands agree, so a lane fetches FOUR CONSECUTIVE reduction elements with one 16-byte LDS read (element e of k-group g is
// index 16 c + 4 g + e) and feeds them to four successive MFMAs -- a quarter of the LDS instructions of 4-byte fragment reads.
// Row strides of 4 (mod 8) floats keep those reads conflict free (8 consecutive rows start in 8 different 16-byte bank groups).
// Accumulator layout: acc[r] = C[4 g + r][lane % 16].

// S[i][j] = alpha * a_i . b_j over HD (rows of a / b in LDS, stride HD + 4).  A work item is a PAIR of column tiles of one
// row tile: the a-fragments are read once and feed two independent accumulator chains (a dependent MFMA waits for its
// predecessor; two chains keep the pipe busy).  Items are dealt round-robin to the waves.
// MODE 0: store;  MODE 1: S[i][j] = S[i][j] * (value - rowdot[i]) * alpha   (dS from P and dP, in place).
// Rows i >= T are not written; columns T .. T16-1 of the written rows are set to zero (they are reduction padding later).
template <int HD, int NW, int MODE>
__device__ __forceinline__ void mma_scores(const float* a, const float* b, float* S, int ss, int T, float alpha,
                                           const float* rowdot, int wave, int lane) {
    constexpr int ST = HD + 4, C16 = HD / 16, REM8 = (HD % 16) == 8;
    const int n = (T + 15) >> 4, np = (n + 1) >> 1, m = lane & 15, g = lane >> 4;
    for (int item = wave; item < n * np; item += NW) {
        const int ti = item / np, tj0 = 2 * (item - ti * np);
        const bool two = tj0 + 1 < n;
        const float* ap = a + (16 * ti + m) * ST + 4 * g;
        const float* bp0 = b + (16 * tj0 + m) * ST + 4 * g;
        const float* bp1 = bp0 + (two ? 16 * ST : 0);  // odd tile count: the second chain repeats the first (discarded)
        f32x4 av[C16 + 1], bv0[C16 + 1], bv1[C16 + 1];
#pragma unroll
        for (int c = 0; c < C16; ++c) {
            av[c] = *(const f32x4*)(ap + 16 * c); bv0[c] = *(const f32x4*)(bp0 + 16 * c); bv1[c] = *(const f32x4*)(bp1 + 16 * c);
        }
        f32x2 a2 = {0.f, 0.f}, b20 = {0.f, 0.f}, b21 = {0.f, 0.f};
        if (REM8) {
            a2 = *(const f32x2*)(ap - 4 * g + 16 * C16 + 2 * g);
            b20 = *(const f32x2*)(bp0 - 4 * g + 16 * C16 + 2 * g);
            b21 = *(const f32x2*)(bp1 - 4 * g + 16 * C16 + 2 * g);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C16; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][e], bv0[c][e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][e], bv1[c][e], acc1, 0, 0, 0);
            }
        if (REM8) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b20.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b21.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b20.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b21.y, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            const f32x4 acc = h ? acc1 : acc0;
            const int j = 16 * (tj0 + h) + m;
            float pv[4] = {0.f, 0.f, 0.f, 0.f}, rd[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = min(16 * ti + 4 * g + r, T - 1);
                    pv[r] = S[i * ss + j]; rd[r] = rowdot[i];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + 4 * g + r;
                if (i < T) {
                    const float o = MODE == 1 ? pv[r] * (acc[r] - rd[r]) * alpha : acc[r] * alpha;
                    S[i * ss + j] = j < T ? o : 0.f;
                }
            }
        }
    }
}

// One ROW tile (16 rows, all NC = ceil(HD / 16) column tiles; HD = 24: the second one half empty) of
// C[i][c] = sum_k W(i, k) * x[k][c]  (i < T, c < HD) with W(i, k) = S[i][k] (TRANS = false) or S[k][i] (TRANS = true).
// The reduction runs over T16 = ceil16(T) indices: S is zero there (rows AND columns) and the rows of x are zero padded.
// The W fragment of a 16-index chunk is read once and feeds NC independent accumulator chains; the next chunk's fragments
// are requested before the current chunk's MFMAs issue.  dst: LDS (stride HD + 4) or global (stride ldd).
// NFIX > 0: the number of 16-index chunks is that compile-time constant (the chunk loop unrolls)
template <int HD, bool TRANS, int NFIX = 0>
__device__ __forceinline__ void mma_rows_tile(const float* S, int ss, const float* x, float* dst, int64_t ldd, int T, int ti,
                                              int lane) {
    constexpr int ST = HD + 4, NC = (HD + 15) / 16;
    const int n = NFIX > 0 ? NFIX : (T + 15) >> 4, m = lane & 15, g = lane >> 4;
    f32x4 acc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* sp = TRANS ? S + (4 * g) * ss + 16 * ti + m : S + (16 * ti + m) * ss + 4 * g;
    const float* xp = x + (4 * g) * ST + m;
    auto load_w = [&](int c) -> f32x4 {
        if (TRANS) { const float* q = sp + 16 * c * ss; return (f32x4){q[0], q[ss], q[2 * ss], q[3 * ss]}; }
        return *(const f32x4*)(sp + 16 * c);
    };
    auto load_x = [&](int c, int q) -> f32x4 {
        const float* r = xp + 16 * c * ST + 16 * q;
        return (f32x4){r[0], r[ST], r[2 * ST], r[3 * ST]};
    };
    f32x4 w = load_w(0), xv[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) xv[q] = load_x(0, q);
#pragma unroll
    for (int c = 0; c < n; ++c) {
        const int cn = min(c + 1, n - 1);
        const f32x4 wn = load_w(cn);
        f32x4 xn[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) xn[q] = load_x(cn, q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < NC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], xv[q][e], acc[q], 0, 0, 0);
        w = wn;
#pragma unroll
        for (int q = 0; q < NC; ++q) xv[q] = xn[q];
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int col = 16 * q + m;
        if (col < HD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + 4 * g + r;
                if (i < T) dst[(int64_t)i * ldd + col] = acc[q][r];
            }
        }
    }
}

// two such products whose row tiles are dealt to the waves as one list (14 items over 8 waves for T = 102)
template <int HD, int NW, bool TRANS0, bool TRANS1>
__device__ __forceinline__ void mma_rows_pair(const float* S, int ss, const float* x0, float* dst0, int64_t ld0, const float* x1,
                                              float* dst1, int64_t ld1, int T, int wave, int lane) {
    const int n = (T + 15) >> 4;
    for (int t = wave; t < 2 * n; t += NW) {
        if (t < n) mma_rows_tile<HD, TRANS0>(S, ss, x0, dst0, ld0, T, t, lane);
        else mma_rows_tile<HD, TRANS1>(S, ss, x1, dst1, ld1, T, t - n, lane);
    }
}

// reductions over the 16 lanes that share g (= over the 16 columns a lane group holds of one row) as DPP lane swaps inside
// the 16-lane row: xor 1, xor 2 (quad permutes), then the half-row and full-row mirrors -- no LDS crossbar round trips
template <int CTRL>
__device__ __forceinline__ float dpp_swap(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp_swap<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_swap<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_swap<0x141>(v));  // row_half_mirror
    return fmaxf(v, dpp_swap<0x140>(v));  // row_mirror
}
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_swap<0xB1>(v);
    v += dpp_swap<0x4E>(v);
    v += dpp_swap<0x141>(v);
    return v + dpp_swap<0x140>(v);
}

// rows of S (T x T, stride sstride) -> softmax in place; 16 lanes per row (4 rows per wave at a time), 8 columns per lane.
// The scores arrive multiplied by log2(e) (folded into the caller's scale): exp(x - max) = 2^(x' - max').
template <int NT>
__device__ __forceinline__ void softmax_rows(float* S, int sstride, int T, int tid) {
    const int l16 = tid & 15, grp = tid >> 4;
    for (int i0 = 0; i0 < T; i0 += NT / 16) {
        const int i = i0 + grp;
        float* row = S + min(i, T - 1) * sstride;  // whole wave stays converged for the DPP reductions
        float x[8];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = l16 + 16 * u;
            x[u] = j < T ? row[j] : -INFINITY;
            mx = fmaxf(mx, x[u]);
        }
        mx = group_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = __builtin_amdgcn_exp2f(x[u] - mx); sum += x[u]; }  // 2^(-inf) = 0 for the padding
        const float inv = 1.0f / group_sum(sum);
        if (i < T) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = l16 + 16 * u;
                if (j < T) row[j] = x[u] * inv;
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ void zero_lds(float* p, int n, int tid) {
    for (int i = tid; i < (n >> 2); i += NT) *(f32x4*)(p + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// 16 score rows of row tile ti against all T16 key columns, entirely in registers: acc[tj][r] = q_{16 ti + 4 g + r} . k_{16 tj + m}
// (NTJ = 8 column tiles cover T <= 128; tiles tj >= n are skipped by a wave-uniform branch).
template <int HD, int NTJ>
__device__ __forceinline__ void score_row_tile(const float* a_rows, const float* b, int n, int lane, f32x4 (&acc)[NTJ]) {
    constexpr int ST = HD + 4, C16 = HD / 16, REM8 = (HD % 16) == 8;
    const int m = lane & 15, g = lane >> 4;
    const float* ap = a_rows + m * ST + 4 * g;
    f32x4 av[C16 + 1];
#pragma unroll
    for (int c = 0; c < C16; ++c) av[c] = *(const f32x4*)(ap + 16 * c);
    f32x2 a2 = {0.f, 0.f};
    if (REM8) a2 = *(const f32x2*)(a_rows + m * ST + 16 * C16 + 2 * g);
#pragma unroll
    for (int tj = 0; tj < NTJ; ++tj) {
        acc[tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tj < n) {
            const float* bp = b + (16 * tj + m) * ST;
            f32x4 bv[C16 + 1];
#pragma unroll
            for (int c = 0; c < C16; ++c) bv[c] = *(const f32x4*)(bp + 4 * g + 16 * c);
            f32x2 b2 = {0.f, 0.f};
            if (REM8) b2 = *(const f32x2*)(bp + 16 * C16 + 2 * g);
#pragma unroll
            for (int c = 0; c < C16; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][e], bv[c][e], acc[tj], 0, 0, 0);
            if (REM8) {
                acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b2.x, acc[tj], 0, 0, 0);
                acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b2.y, acc[tj], 0, 0, 0);
            }
        }
    }
}

// Forward: every wave owns whole ROW tiles of one (sample, head): scores of its 16 rows against all keys stay in the MFMA
// accumulators, the row softmax runs on them in registers (a row's 16 x n columns sit in one 16-lane group), the
// probabilities pass through a wave-private 16 x T16 LDS tile only to change from accumulator to operand layout, and P V
// follows at once -- no workgroup barrier after the q / k / v load, no T x T score matrix in LDS (67 KB per workgroup for
// T = 102, hd = 24: two workgroups per CU, one loading while the other computes).
template <int HD>
__global__ __launch_bounds__(256) void k_attn_mid_fwd(const float* __restrict__ qkv, int64_t ld, float* __restrict__ out, int64_t ldo,
                                                      int H, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 256, NW = NT / 64, ST = HD + 4, NTJ = TMAX / 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = blockIdx.x, h = blockIdx.y, D = H * HD;
    const int T16 = (T + 15) & ~15, n = T16 >> 4, ss = score_stride(T), m = lane & 15, g = lane >> 4;
    float* qs = lds;
    float* ks = qs + T16 * ST;
    float* vs = ks + T16 * ST;
    float* ptile = vs + T16 * ST + wave * 16 * ss;  // this wave's 16 x T16 probabilities
    const float* base = qkv + (int64_t)b * T * ld + h * HD;
    MAE_TS(0)
    {
        const float* const src[3] = {base, base + D, base + 2 * D};
        const int64_t lds_[3] = {ld, ld, ld};
        load_rows<HD, NT, 3>(src, lds_, qs, T, tid);  // q | k | v are consecutive LDS buffers
    }
    __syncthreads();
    MAE_TS(1)
    float* o = out + (int64_t)b * T * ldo + h * HD;
    const float sl2 = scale * 1.44269504088896341f;  // softmax in base 2: exp(s x) = 2^(s x log2 e)
    for (int ti = wave; ti < n; ti += NW) {
        f32x4 acc[NTJ];
        score_row_tile<HD, NTJ>(qs + 16 * ti * ST, ks, n, lane, acc);
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            if (tj < n) {
                const bool live = 16 * tj + m < T;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[tj][r] = live ? acc[tj][r] * sl2 : -INFINITY;
                    mx[r] = fmaxf(mx[r], acc[tj][r]);
                }
            }
        }
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = group_max(mx[r]);
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            if (tj < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[tj][r] = __builtin_amdgcn_exp2f(acc[tj][r] - mx[r]);  // 2^(-inf) = 0 for the padding
                    sum[r] += acc[tj][r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = 1.0f / group_sum(sum[r]);
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            if (tj < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ptile[(4 * g + r) * ss + 16 * tj + m] = acc[tj][r] * sum[r];
            }
        }
        // O rows of this tile = P V; the tile sits at row 0 of ptile: shift the base so that row 16 ti + m lands there
        mma_rows_tile<HD, false>(ptile - 16 * ti * ss, ss, vs, o, ldo, T, ti, lane);
    }
    MAE_TS(4)
}

// Forward, second form (round 4; default for head dims <= 32): the scores are computed TRANSPOSED -- keys as the MFMA's row
// operand, queries as its column operand -- so that the accumulator of key tile tj, acc[tj][r] = S[query 16 ti + m][key 16 tj +
// 4 g + r], is lane for lane the A-operand fragment of the P V product that follows (row = query m, reduction index = key
// 4 g + e): the probabilities never leave the registers.  The first form passed them through a wave-private 16 x T16 LDS tile
// with 28 four-byte stores and as many fragment reads per query tile, and that tile was 30 KB of the workgroup's 67 KB of
// LDS.  V is kept transposed in LDS ([d][key], one 16-byte read per fragment); a query's max and sum are spread over the four
// lane groups g that hold its keys: two cross-row exchanges each.  40 KB of LDS at T = 102, hd = 24.
// Measured at B = 1024, H = 8, hd = 24, T = 102 (8192 (sample, head) units; 73 us of MFMA time at 2.4 GHz): 220 us first form,
// 156 us this one.  What did NOT move it further (profiles/r04_mae_attn.txt): heads per workgroup 1 / 2 / 4 / 8 (+-3 %), six
// instead of four waves per SIMD (160), stores delayed past the next head's rows (0).  Per query tile a wave issues 98 MFMAs
// (3136 clocks of the matrix pipe) and ~260 other instructions; the sum of BOTH over a SIMD's waves, not the larger, is what
// the kernel takes -- the same additive behaviour the GEMM loops show.
// gfx950's lane-swap instructions: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the
// second, v_permlane32_swap the upper half of the first with the lower half of the second; fed the same value twice they
// return (x0 x0 x2 x2 | x1 x1 x3 x3) and (lo lo | hi hi) -- both partners of the xor-16 / xor-32 exchange in two VALU
// registers, without the LDS round trip of a ds_bpermute (four of those sat between the score and the P V products).
// (xrow_max / xrow_sum live in mdt_device.h: the sampler's per-sample attention uses them too)
size_t attn_mid_lds_fwd2(int hd, int T) {
    const int T16 = (T + 15) & ~15, hdp = (hd + 15) & ~15;
    return ((size_t)2 * T16 * (hd + 4) + (size_t)hdp * (T16 + 4)) * sizeof(float);
}
// N = key (and query) tiles of 16, a compile-time count: the N score chains of a query tile are then one straight-line block
// the compiler interleaves (as branches on a run-time count they ran one after the other, each waiting out its MFMA latency).
template <int HD, int N>
__global__ __launch_bounds__(N > 4 ? 512 : 256, 4) void k_attn_mid_fwd2(const float* __restrict__ qkv, int64_t ld, float* __restrict__ out, int64_t ldo,
                                                       int H, int T, float scale, int hpw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = N > 4 ? 512 : 256, NW = NT / 64, ST = HD + 4, NC = (HD + 15) / 16, H4 = HD / 4;
    constexpr int C16 = HD / 16, REM8 = (HD % 16) == 8, T16 = 16 * N, VS = T16 + 4, n16 = T16 * H4,
                  U = (n16 + NT - 1) / NT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = blockIdx.x, h0 = blockIdx.y * hpw, D = H * HD;
    const int m = lane & 15, g = lane >> 4;
    float* qs = lds;                  // [T16][ST], scaled by scale * log2 e
    float* ks = qs + T16 * ST;        // [T16][ST]
    float* vt = ks + T16 * ST;        // [HDP][VS]: V transposed (rows d >= HD are never written: they feed output columns nobody stores)
    const float sl2 = scale * 1.44269504088896341f;  // softmax in base 2
    // One workgroup walks hpw heads of one sample; the rows of head h + 1 are fetched into registers BEFORE the products of
    // head h and written to LDS after them (the two workgroups of a CU start together: without this they wait for memory at
    // the same time).
    f32x4 rq[U], rk[U], rv[U];
    // buffer loads: one 32-bit row offset per slot for all three matrices and all heads (the head is the scalar offset, k and v
    // are immediates) -- the 64-bit pointers of plain loads cost more registers than the rows they fetch
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (int64_t)b * T * ld), 0, 0xffffffffu, 0x00020000);
    unsigned roff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = tid + u * NT, t = i / H4, c = i - t * H4;
        roff[u] = ((unsigned)min(t, T - 1) * (unsigned)ld + 4u * c) << 2;
    }
    auto fetch = [&](int h) {
        const unsigned so = (unsigned)(h * HD) << 2;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rq[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, roff[u], so, 0));
            rk[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, roff[u] + 4u * D, so, 0));
            rv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, roff[u] + 8u * D, so, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * NT, t = i / H4, c = i - t * H4;
            if (i < n16) {   // rows / key columns T .. T16-1: zero (their probabilities are zero; the products must not see NaNs)
                const bool live = t < T;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                *(f32x4*)(qs + t * ST + 4 * c) = live ? rq[u] * sl2 : z;
                *(f32x4*)(ks + t * ST + 4 * c) = live ? rk[u] : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) vt[(4 * c + e) * VS + t] = live ? rv[u][e] : 0.f;
            }
        }
    };
    MAE_TS(0)
    MAE_WG(0)
    fetch(h0);
    // a wave has at most ONE query tile per head (N <= NW); its output waits in registers until the next head's rows are in LDS
    // (the wait for those rows is a wait for every earlier memory operation of the wave, these stores included)
    static_assert(N <= NW, "one query tile per wave and head");
    f32x4 oacc[NC];
    int tq = -1;   // the held tile
    auto store_tile = [&](int h) {
        if (tq < 0) return;
        float* o = out + (int64_t)b * T * ldo + h * HD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 16 * tq + 4 * g + r;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int d = 16 * c + m;
                if (d < HD && q < T) o[(int64_t)q * ldo + d] = oacc[c][r];
            }
        }
    };
    for (int hh = 0; hh < hpw; ++hh) {
        if (hh) __syncthreads();   // every wave is done with the previous head's rows
        commit();
        if (hh) store_tile(h0 + hh - 1);
        tq = -1;
        __syncthreads();
                if (hh + 1 < hpw) fetch(h0 + hh + 1);
        if (hh == 0) { MAE_TS(1) }
        // e.g. 7 query tiles over 8 waves: which wave (= which SIMD) idles rotates with the sample and the head (b >> 8: the two workgroups
        // of a CU are 256 apart)
        const int ti = (wave + b + (b >> 8) + hh) & (NW - 1);
        if (ti < N) {
            tq = ti;
            // the query tile as the COLUMN operand: lane = (query m, d = 4 g + e)
            const float* qp = qs + (16 * ti + m) * ST;
            f32x4 qv[C16 + 1];
#pragma unroll
            for (int c = 0; c < C16; ++c) qv[c] = *(const f32x4*)(qp + 16 * c + 4 * g);
            f32x2 q2 = {0.f, 0.f};
            if (REM8) q2 = *(const f32x2*)(qp + 16 * C16 + 2 * g);
            f32x4 acc[N];
            float mx = -INFINITY;
#pragma unroll
            for (int tj = 0; tj < N; ++tj) {
                acc[tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
                {
                    const float* kp = ks + (16 * tj + m) * ST;   // the key tile as the ROW operand: lane = (key m, d = 4 g + e)
#pragma unroll
                    for (int c = 0; c < C16; ++c) {
                        const f32x4 kv = *(const f32x4*)(kp + 16 * c + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[e], qv[c][e], acc[tj], 0, 0, 0);
                    }
                    if (REM8) {
                        const f32x2 k2 = *(const f32x2*)(kp + 16 * C16 + 2 * g);
                        acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(k2.x, q2.x, acc[tj], 0, 0, 0);
                        acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(k2.y, q2.y, acc[tj], 0, 0, 0);
                    }
                    // acc[tj][r] = log2 e * scale * q_{16 ti + m} . k_{16 tj + 4 g + r}
                    if (tj == N - 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * tj + 4 * g + r >= T) acc[tj][r] = -INFINITY;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[tj][r]);
                }
            }
            mx = xrow_max(mx);   // key 0 is always live: finite
            float sum = 0.f;
#pragma unroll
            for (int tj = 0; tj < N; ++tj) {
                {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[tj][r] = __builtin_amdgcn_exp2f(acc[tj][r] - mx);   // 2^(-inf) = 0 for the padding
                        sum += acc[tj][r];
                    }
                }
            }
            const float inv = 1.0f / xrow_sum(sum);   // of query m, in every lane of column m
            // O[query 4 g + r][d = 16 c + m] = sum_key P[query][key] V[key][d]: the accumulators ARE the A fragments
#pragma unroll
            for (int c = 0; c < NC; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tj = 0; tj < N; ++tj) {
                {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const f32x4 vv = *(const f32x4*)(vt + (16 * c + m) * VS + 16 * tj + 4 * g);   // lane = (d, keys 4 g .. + 3)
#pragma unroll
                        for (int e = 0; e < 4; ++e) oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[tj][e], vv[e], oacc[c], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ir = __shfl(inv, 4 * g + r, 64);   // the normaliser of output row 4 g + r lives in column 4 g + r
#pragma unroll
                for (int c = 0; c < NC; ++c) oacc[c][r] *= ir;
            }
        }
    }
    store_tile(h0 + hpw - 1);
    MAE_TS(4)
    MAE_WG(1)
}

// Backward.  The forward output O arrives from the caller (autograd keeps it: it is the input of the projection that
// follows), so sum_j P_ij dP_ij = dO_i . O_i needs no second P V product here.
template <int HD>
__global__ __launch_bounds__(512) void k_attn_mid_bwd(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ fwd_out,
                                                      int64_t ldf, const float* __restrict__ d_out, int64_t ldd,
                                                      float* __restrict__ d_qkv, int64_t ldg, int H, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512, NW = NT / 64, ST = HD + 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = blockIdx.x, h = blockIdx.y, D = H * HD;
    const int T16 = (T + 15) & ~15, n = T16 >> 4, ss = score_stride(T);
    float* qs = lds;
    float* ks = qs + T16 * ST;
    float* vs = ks + T16 * ST;
    float* dos = vs + T16 * ST;
    float* os = dos + T16 * ST;
    float* S = os + T16 * ST;
    float* rowdot = S + T16 * ss;
    const float* base = qkv + (int64_t)b * T * ld + h * HD;
    MAE_TS(5)
    zero_lds<NT>(S, T16 * ss, tid);
    {
        const float* const src[5] = {base, base + D, base + 2 * D, d_out + (int64_t)b * T * ldd + h * HD,
                                     fwd_out + (int64_t)b * T * ldf + h * HD};
        const int64_t lds_[5] = {ld, ld, ld, ldd, ldf};
        load_rows<HD, NT, 5>(src, lds_, qs, T, tid);  // q | k | v | dO | O are consecutive LDS buffers
    }
    __syncthreads();
    MAE_TS(6)
    for (int i = tid; i < T; i += NT) {  // read again only after three more barriers
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc = fmaf(dos[i * ST + c], os[i * ST + c], acc);
        rowdot[i] = acc;
    }
    mma_scores<HD, NW, 0>(qs, ks, S, ss, T, scale * 1.44269504088896341f, nullptr, wave, lane);  // base-2 softmax
    __syncthreads();
    MAE_TS(7)
    softmax_rows<NT>(S, ss, T, tid);  // S = P
    __syncthreads();
    MAE_TS(8)
    float* g = d_qkv + (int64_t)b * T * ldg + h * HD;
    for (int t = wave; t < n; t += NW) mma_rows_tile<HD, true>(S, ss, dos, g + 2 * D, ldg, T, t, lane);  // dV = P^T dO (before P is overwritten)
    __syncthreads();
    MAE_TS(9)
    MAE_TS(10)
    mma_scores<HD, NW, 1>(dos, vs, S, ss, T, scale, rowdot, wave, lane);  // S = scale * P * (dO v^T - rowdot)
    __syncthreads();
    MAE_TS(11)
    // dQ = dS K and dK = dS^T Q
    mma_rows_pair<HD, NW, false, true>(S, ss, ks, g, ldg, qs, g + D, ldg, T, wave, lane);
    MAE_TS(12)
}

// acc[kt][q] += sum over the 16 rows i of one query tile  W[i][16 kt + .] * x[i][16 q + .]   (kt < n key tiles, q < NC feature
// tiles): the tile's contribution to dV = P^T dO or dK = dS^T Q.  W never touches LDS: a score-row accumulator from
// score_row_tile, w[kt][r] = W[row 4 g + r][key 16 kt + m], is lane for lane the transposed operand (row of the product = key
// m, reduction index = query row 4 g + e).  x: the tile's 16 rows of dO / Q in LDS (stride HD + 4).  (Until round 4 P and dS
// both passed through the wave's LDS tile for this: 28 four-byte stores and 28 four-byte reads per product and query tile.)
template <int HD, int NTJ>
__device__ __forceinline__ void mma_acc_trans_reg(const f32x4 (&w)[NTJ], const float* xrows, int n, int lane,
                                                  f32x4 (&acc)[NTJ][(HD + 15) / 16]) {
    constexpr int ST = HD + 4, NC = (HD + 15) / 16;
    const int m = lane & 15, g = lane >> 4;
    const float* xp = xrows + (4 * g) * ST + m;
    f32x4 xv[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const float* r = xp + 16 * q;
        xv[q] = (f32x4){r[0], r[ST], r[2 * ST], r[3 * ST]};
    }
#pragma unroll
    for (int kt = 0; kt < NTJ; ++kt) {
        if (kt < n) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < NC; ++q) acc[kt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kt][e], xv[q][e], acc[kt][q], 0, 0, 0);
        }
    }
}

// Backward, second form (default): like the forward, every wave owns whole QUERY tiles and keeps their score rows in the MFMA
// accumulators -- P (recomputed, softmax in registers), dP = dO V^T and dS = P (dP - dO.O) never exist as a T x T matrix.  Per
// query tile the wave passes dS through ONE wave-private 16 x T16 LDS tile (accumulator -> row-operand layout) for dQ = dS K,
// whose rows go straight to memory; the tile's contributions to dV = P^T dO and dK = dS^T Q take P and dS from the
// accumulators as they are (mma_acc_trans_reg) and accumulate in registers (T16 x hd each) across the wave's tiles.  One workgroup barrier after the loads, none inside; the four waves' dK / dV partials then meet in the
// (now free) q / k / v / dO buffers in a fixed order.  80 KB of LDS at T = 102, hd = 24 (the first form: 115 KB, five
// barrier-separated phases): two workgroups per CU, one loading or reducing while the other multiplies.
// N = key (and query) tiles of 16, a compile-time count like the forward's (round 5): with a run-time count every key tile's MFMA
// chain sat behind its own branch and waited out its LDS reads and its MFMA latency alone (the ISA: `s_cbranch_vccnz` + `lgkmcnt(0)`
// per key tile in the score rows, the softmax and the two transposed products); as straight-line code the N chains interleave.
template <int HD, int N>
__global__ __launch_bounds__(256, 2) void k_attn_mid_bwd2(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ fwd_out,
                                                       int64_t ldf, const float* __restrict__ d_out, int64_t ldd,
                                                       float* __restrict__ d_qkv, int64_t ldg, int H, int T, float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 256, NW = NT / 64, ST = HD + 4, NTJ = N, NC = (HD + 15) / 16, H4 = HD / 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = blockIdx.x, h = blockIdx.y, D = H * HD;
    constexpr int T16 = 16 * N, n = N;
    const int ss = score_stride(T), m = lane & 15, g = lane >> 4;
    float* qs = lds;
    float* ks = qs + T16 * ST;
    float* vs = ks + T16 * ST;
    float* dos = vs + T16 * ST;
    float* rowdot = dos + T16 * ST;                    // [T16]
    float* ptile = rowdot + T16 + wave * 16 * ss;      // this wave's 16 x T16 tile (P, then dS)
    const float* base = qkv + (int64_t)b * T * ld + h * HD;
    const float* dob = d_out + (int64_t)b * T * ldd + h * HD;
    const float* fob = fwd_out + (int64_t)b * T * ldf + h * HD;
    MAE_TS(5)
    MAE_WG(0)
    {
        const float* const src[4] = {base, base + D, base + 2 * D, dob};
        const int64_t lds_[4] = {ld, ld, ld, ldd};
        load_rows<HD, NT, 4>(src, lds_, qs, T, tid);   // q | k | v | dO are consecutive LDS buffers
    }
    for (int i = tid; i < T16; i += NT) {              // rowdot_i = dO_i . O_i  (= sum_j P_ij dP_ij), straight from memory
        const int64_t ic = min(i, T - 1);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < H4; ++c) {
            const f32x4 a = ldg4(dob + ic * ldd + 4 * c), o = ldg4(fob + ic * ldf + 4 * c);
            acc += (a.x * o.x + a.y * o.y) + (a.z * o.z + a.w * o.w);
        }
        rowdot[i] = i < T ? acc : 0.f;
    }
    __syncthreads();
    MAE_TS(6)
    float* gq = d_qkv + (int64_t)b * T * ldg + h * HD;
    const float sl2 = scale * 1.44269504088896341f;    // softmax in base 2
    f32x4 dK[NTJ][NC], dV[NTJ][NC];
#pragma unroll
    for (int kt = 0; kt < NTJ; ++kt)
#pragma unroll
        for (int q = 0; q < NC; ++q) { dK[kt][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[kt][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // n tiles over 4 waves (7: 2 2 2 1): which wave (= which SIMD) gets the light share rotates with the sample and the head
    for (int ti = (wave + b + (b >> 8) + h) & (NW - 1); ti < n; ti += NW) {
        f32x4 acc[NTJ], dp[NTJ];
        score_row_tile<HD, NTJ>(qs + 16 * ti * ST, ks, n, lane, acc);
        score_row_tile<HD, NTJ>(dos + 16 * ti * ST, vs, n, lane, dp);   // dP rows = dO rows . V rows
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sum[4] = {0.f, 0.f, 0.f, 0.f}, rd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rd[r] = rowdot[16 * ti + 4 * g + r];
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            if (tj < n) {
                const bool live = 16 * tj + m < T;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[tj][r] = live ? acc[tj][r] * sl2 : -INFINITY;
                    mx[r] = fmaxf(mx[r], acc[tj][r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = group_max(mx[r]);
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            if (tj < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[tj][r] = __builtin_amdgcn_exp2f(acc[tj][r] - mx[r]);
                    sum[r] += acc[tj][r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = 1.0f / group_sum(sum[r]);
        // P stays in `acc`, dS = scale * P * (dP - rowdot) in `dp`; only dS passes through the tile (dQ wants it by rows)
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            if (tj < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = acc[tj][r] * sum[r];
                    acc[tj][r] = pv;
                    dp[tj][r] = pv * (dp[tj][r] - rd[r]) * scale;
                    ptile[(4 * g + r) * ss + 16 * tj + m] = dp[tj][r];
                }
            }
        }
        mma_acc_trans_reg<HD, NTJ>(acc, dos + 16 * ti * ST, n, lane, dV);   // dV += P_tile^T dO_tile
        mma_acc_trans_reg<HD, NTJ>(dp, qs + 16 * ti * ST, n, lane, dK);     // dK += dS_tile^T Q_tile
        mma_rows_tile<HD, false, N>(ptile - 16 * ti * ss, ss, ks, gq, ldg, T, ti, lane);   // dQ rows of this tile = dS K
    }
    MAE_TS(7)
    __syncthreads();   // everyone is done with q / k / v / dO: the four buffers now take the partial sums
    // waves 0, 1 store their partials (dK -> buffers 0 / 1, dV -> 2 / 3), waves 2, 3 add theirs to them, then everybody adds
    // the two halves and writes the rows out: ((w0 + w2) + (w1 + w3)), a fixed order.
    // The buffers hold the partials TRANSPOSED, [column][key] with row stride KS: an accumulator register quad is four
    // consecutive KEYS of one column, so a lane moves a quad with ONE 16-byte LDS instruction (round 5; as [key][column] rows it
    // took four 4-byte ones: 112 stores + 112 loads per lane and pass, most of the 15 k clocks this phase took of a workgroup's 70 k).
    // (short sequences: a transposed buffer, HD x KS floats, can be larger than a T16 x ST one; the four of them then reach into the
    //  score-tile area behind, which is free as well by now -- 4 HD KS <= 4 T16 ST + T16 + 64 (T16 + 4) for every HD <= 32)
    const int KS = T16 + 4, SLOT = max(T16 * ST, HD * KS);
    float* bK = lds + (wave & 1) * SLOT;
    float* bV = lds + (2 + (wave & 1)) * SLOT;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if ((wave >> 1) == pass) {
#pragma unroll
            for (int kt = 0; kt < NTJ; ++kt) {
                if (kt < n) {
                    // (the adding pass reads a key tile's quads in one batch, then stores: a read behind a store that may alias
                    //  it as far as the compiler knows would wait for that store)
                    f32x4 tK[NC], tV[NC];
                    if (pass) {
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            const int o = min(16 * q + m, HD - 1) * KS + 16 * kt + 4 * g;
                            tK[q] = *(const f32x4*)(bK + o);
                            tV[q] = *(const f32x4*)(bV + o);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const int col = 16 * q + m;
                        if (col < HD) {
                            const int o = col * KS + 16 * kt + 4 * g;
                            *(f32x4*)(bK + o) = pass ? tK[q] + dK[kt][q] : dK[kt][q];
                            *(f32x4*)(bV + o) = pass ? tV[q] + dV[kt][q] : dV[kt][q];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    MAE_TS(8)
    for (int i = tid; i < T * H4; i += NT) {
        const int t = i / H4, c = i - t * H4;
        f32x4 k4, v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = (4 * c + e) * KS + t;
            k4[e] = lds[o] + lds[SLOT + o];
            v4[e] = lds[2 * SLOT + o] + lds[3 * SLOT + o];
        }
        *(f32x4*)(gq + D + (int64_t)t * ldg + 4 * c) = k4;
        *(f32x4*)(gq + 2 * D + (int64_t)t * ldg + 4 * c) = v4;
    }
    MAE_TS(9)
    MAE_WG(1)
}

size_t attn_mid_lds2(int hd, int T) {
    const int T16 = (T + 15) & ~15, ss = T16 + 4;
    return ((size_t)4 * T16 * (hd + 4) + T16 + (size_t)4 * 16 * ss) * sizeof(float);
}

size_t attn_mid_lds(int hd, int T, bool bwd) {
    const int T16 = (T + 15) & ~15, ss = T16 + 4;
    if (!bwd) return ((size_t)3 * T16 * (hd + 4) + (size_t)4 * 16 * ss + 16) * sizeof(float);  // q k v + 4 waves' P tiles
    return ((size_t)5 * T16 * (hd + 4) + (size_t)T16 * ss + TMAX + 16) * sizeof(float);
}

template <int HD>
hipError_t launch_fwd(const float* qkv, int64_t ld, float* out, int64_t ldo, int64_t B, int H, int T, float scale, hipStream_t s) {
    static int form = -1;  // MDT_HIP_ATTN_FWD=1: the first form (probabilities through a wave-private LDS tile), for A/B runs
    if (form < 0) { const char* e = getenv("MDT_HIP_ATTN_FWD"); form = e ? atoi(e) : 2; }
    // (the second form addresses a sample's rows with 32-bit byte offsets: a sample must span less than 2 GiB)
    if (form == 2 && HD <= 32 && (int64_t)T * ld + 3ll * H * HD < (1ll << 29)) {
        const size_t lds2 = attn_mid_lds_fwd2(HD, T);
        // heads per workgroup: all of them once that still leaves four waves per SIMD on every CU, else the largest divisor of H
        // that does
        static int hpw_env = -1;
        if (hpw_env < 0) { const char* e = getenv("MDT_HIP_ATTN_HPW"); hpw_env = e ? atoi(e) : 0; }
        const int n = (T + 15) >> 4, per_cu = n > 4 ? 2 : 4;
        int hpw = H;
        while (hpw > 1 && (B * (H / hpw) < 256 * per_cu || H % hpw)) --hpw;
        if (hpw_env > 0 && H % hpw_env == 0) hpw = hpw_env;
        const dim3 grid((unsigned)B, H / hpw);
#define MDT_FWD2(N_)                                                                                                              \
    case N_: {                                                                                                                    \
        hipError_t e2 = hipFuncSetAttribute((const void*)k_attn_mid_fwd2<HD, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        if (e2 != hipSuccess) return e2;                                                                                          \
        hipLaunchKernelGGL((k_attn_mid_fwd2<HD, N_>), grid, dim3(N_ > 4 ? 512 : 256), lds2, s, qkv, ld, out, ldo, H, T, scale, hpw); \
        return hipGetLastError();                                                                                                 \
    }
        switch (n) {
            MDT_FWD2(1) MDT_FWD2(2) MDT_FWD2(3) MDT_FWD2(4) MDT_FWD2(5) MDT_FWD2(6) MDT_FWD2(7) MDT_FWD2(8)
        }
#undef MDT_FWD2
    }
    const size_t lds = attn_mid_lds(HD, T, false);
    hipError_t e = hipFuncSetAttribute((const void*)k_attn_mid_fwd<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_attn_mid_fwd<HD>), dim3((unsigned)B, H), dim3(256), lds, s, qkv, ld, out, ldo, H, T, scale);
    return hipGetLastError();
}
template <int HD>
hipError_t launch_bwd(const float* qkv, int64_t ld, const float* fo, int64_t ldf, const float* d_out, int64_t ldd, float* d_qkv,
                      int64_t ldg, int64_t B, int H, int T, float scale, hipStream_t s) {
    static int form = -1;  // MDT_HIP_ATTN_BWD=1: the first form (T x T matrices in LDS, five phases), for A/B runs
    if (form < 0) { const char* e = getenv("MDT_HIP_ATTN_BWD"); form = e ? atoi(e) : 2; }
    if constexpr (HD <= 32) {  // (hd 48 / 64: the T16 x hd accumulators of dK and dV no longer fit the registers)
        if (form == 2) {
            const size_t lds2 = attn_mid_lds2(HD, T);
#define MDT_BWD2(N_)                                                                                                              \
    case N_: {                                                                                                                    \
        hipError_t e2 = hipFuncSetAttribute((const void*)k_attn_mid_bwd2<HD, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        if (e2 != hipSuccess) return e2;                                                                                          \
        hipLaunchKernelGGL((k_attn_mid_bwd2<HD, N_>), dim3((unsigned)B, H), dim3(256), lds2, s, qkv, ld, fo, ldf, d_out, ldd, d_qkv, ldg, \
                           H, T, scale);                                                                                          \
        return hipGetLastError();                                                                                                 \
    }
            switch ((T + 15) >> 4) {
                MDT_BWD2(1) MDT_BWD2(2) MDT_BWD2(3) MDT_BWD2(4) MDT_BWD2(5) MDT_BWD2(6) MDT_BWD2(7) MDT_BWD2(8)
            }
#undef MDT_BWD2
        }
    }
    const size_t lds = attn_mid_lds(HD, T, true);
    hipError_t e = hipFuncSetAttribute((const void*)k_attn_mid_bwd<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_attn_mid_bwd<HD>), dim3((unsigned)B, H), dim3(512), lds, s, qkv, ld, fo, ldf, d_out, ldd, d_qkv, ldg, H, T,
                       scale);
    return hipGetLastError();
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// LayerScale + residual: out = x + gamma * z over rows of D (voltron Block: x + layer_scale(branch(x))); one 16-byte column
// group per thread, the row loop strided by the rows a workgroup holds at once.
__global__ __launch_bounds__(256) void k_scale_residual_fwd(const float* __restrict__ x, const float* __restrict__ z,
                                                            const float* __restrict__ gamma, float* __restrict__ out, int64_t n4,
                                                            int D4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 g = ((const f32x4*)gamma)[i % D4];
    ((f32x4*)out)[i] = ((const f32x4*)x)[i] + g * ((const f32x4*)z)[i];
}

// backward of the branch: dz = gamma * g, and the partial sums of dgamma = sum_rows g * z over this workgroup's row slice
// (part[slice][D], added up by k_colsum).  256 threads = RL row lanes x D4 column groups.
constexpr int SR_ROWS = 64;  // rows per workgroup
__global__ __launch_bounds__(256) void k_scale_residual_bwd(const float* __restrict__ g, const float* __restrict__ z,
                                                            const float* __restrict__ gamma, float* __restrict__ dz,
                                                            float* __restrict__ part, int64_t M, int D4) {
    __shared__ f32x4 red[256];
    const int RL = 256 / D4, c = threadIdx.x % D4, rl = threadIdx.x / D4;
    const int64_t r0 = (int64_t)blockIdx.x * SR_ROWS, r1 = min(r0 + SR_ROWS, M);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rl < RL) {
        const f32x4 gm = ((const f32x4*)gamma)[c];
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            const f32x4 gv = ((const f32x4*)g)[r * D4 + c], zv = ((const f32x4*)z)[r * D4 + c];
            ((f32x4*)dz)[r * D4 + c] = gm * gv;
            acc += gv * zv;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int q = 1; q < RL; ++q) acc += red[q * D4 + c];
        ((f32x4*)part)[(int64_t)blockIdx.x * D4 + c] = acc;
    }
}

// LayerScale + residual AND the RMSNorm at the head of the NEXT branch in one pass each way (round 4): the block's
//   x' = x + gamma z ;  h = x' / max(||x'|| D^-1/2, eps) * gn
// read x and z once and write x' and h (the two stand-alone kernels read x' again); backward, with G the whole gradient at x'
// (d_res, what arrives on the residual path, + the norm's backward of d_h):
//   G = d_res + gn d_h / n - x' (sum_j gn_j d_h_j x'_j) / (D n^3) ;  dz = gamma G ;  dgamma = sum_rows G z ;  dgn = sum_rows d_h x' / n
// one wave per row, 64 rows per workgroup (16 per wave), the two parameter gradients as per-workgroup partials (k_colsum).
constexpr int SRN_C4 = 2, SRN_ROWS = 64;   // 16-byte column groups per lane (D <= 512, D % 4 == 0); rows per workgroup (backward)
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__global__ __launch_bounds__(256) void k_sr_rms_fwd(const float* __restrict__ x, const float* __restrict__ z,
                                                    const float* __restrict__ gamma, const float* __restrict__ gn,
                                                    float* __restrict__ xn, float* __restrict__ h, int M, int D4, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[SRN_C4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < SRN_C4; ++i) {
        const int c = lane + 64 * i;
        const int64_t o = (int64_t)row * D4 + c;
        v[i] = c < D4 ? ((const f32x4*)x)[o] + ((const f32x4*)gamma)[c] * ((const f32x4*)z)[o] : zero4;
        if (c < D4) ((f32x4*)xn)[o] = v[i];
        ss += dot4(v[i], v[i]);
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)) * rsqrtf((float)(4 * D4)), eps);
#pragma unroll
    for (int i = 0; i < SRN_C4; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) ((f32x4*)h)[(int64_t)row * D4 + c] = v[i] * inv * ((const f32x4*)gn)[c];
    }
}
__global__ __launch_bounds__(256) void k_sr_rms_bwd(const float* __restrict__ xn, const float* __restrict__ gn,
                                                    const float* __restrict__ dh, const float* __restrict__ res,
                                                    const float* __restrict__ z, const float* __restrict__ gamma,
                                                    float* __restrict__ G, float* __restrict__ dz, float* __restrict__ pgn,
                                                    float* __restrict__ pgam, int M, int D4, float eps) {
    __shared__ f32x4 red[2][4][64 * SRN_C4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * SRN_ROWS, r1 = min(r0 + SRN_ROWS, M);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float Df = (float)(4 * D4);
    f32x4 gnv[SRN_C4], gmv[SRN_C4], dgn[SRN_C4], dgm[SRN_C4];
#pragma unroll
    for (int i = 0; i < SRN_C4; ++i) {
        const int c = lane + 64 * i;
        gnv[i] = c < D4 ? ((const f32x4*)gn)[c] : zero4;
        gmv[i] = c < D4 ? ((const f32x4*)gamma)[c] : zero4;
        dgn[i] = zero4;
        dgm[i] = zero4;
    }
    for (int row = r0 + wv; row < r1; row += 4) {
        f32x4 xv[SRN_C4], d[SRN_C4], rv[SRN_C4], zv[SRN_C4];
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < SRN_C4; ++i) {
            const int c = lane + 64 * i;
            const int64_t o = (int64_t)row * D4 + c;
            xv[i] = c < D4 ? ((const f32x4*)xn)[o] : zero4;
            d[i] = c < D4 ? ((const f32x4*)dh)[o] : zero4;
            rv[i] = (res != nullptr && c < D4) ? ((const f32x4*)res)[o] : zero4;
            zv[i] = c < D4 ? ((const f32x4*)z)[o] : zero4;
            ss += dot4(xv[i], xv[i]);
            dot += dot4(gnv[i] * d[i], xv[i]);
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float raw = sqrtf(ss) * rsqrtf(Df);
        const bool clamped = raw < eps;
        const float n = clamped ? eps : raw, inv = 1.0f / n;
        const float k = clamped ? 0.f : dot / (Df * n * n * n);
#pragma unroll
        for (int i = 0; i < SRN_C4; ++i) {
            const int c = lane + 64 * i;
            const f32x4 v = gnv[i] * d[i] * inv - xv[i] * k + rv[i];
            if (c < D4) {
                const int64_t o = (int64_t)row * D4 + c;
                ((f32x4*)G)[o] = v;
                ((f32x4*)dz)[o] = gmv[i] * v;
            }
            dgn[i] += d[i] * xv[i] * inv;
            dgm[i] += v * zv[i];
        }
    }
#pragma unroll
    for (int i = 0; i < SRN_C4; ++i) {
        red[0][wv][lane + 64 * i] = dgn[i];
        red[1][wv][lane + 64 * i] = dgm[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D4; c += 256) {
        ((f32x4*)pgn)[(int64_t)blockIdx.x * D4 + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        ((f32x4*)pgam)[(int64_t)blockIdx.x * D4 + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

}  // namespace

#ifdef MDT_DEBUG_TIMING
extern "C" int mdt_mae_debug_ts(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mae_ts), 16 * sizeof(unsigned long long));
}
extern "C" int mdt_mae_debug_wg(unsigned long long* out, int n_wg) {   // (start, end, hw id) per workgroup, 100 MHz ticks
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mae_wg), (size_t)3 * (n_wg < 8192 ? n_wg : 8192) * sizeof(unsigned long long));
}
#endif

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

extern "C" mdt_status mdt_op_attn_mid_bwd(const float* qkv, int64_t ld_qkv, const float* out, int64_t ld_out, const float* d_out,
                                          int64_t ld_do, float* d_qkv, int64_t ld_dqkv, int64_t B, int32_t H, int32_t hd, int32_t T,
                                          float scale, void* stream) {
    if (!qkv || !out || !d_out || !d_qkv || B < 1 || H < 1 || T < 1)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_mid_bwd: bad argument");
    if (T > TMAX) return mdt_fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_mid_bwd: T must be <= %d", TMAX);
    if (!aligned16(qkv) || !aligned16(out) || !aligned16(d_out) || !aligned16(d_qkv) || ld_qkv % 4 || ld_out % 4 || ld_do % 4 ||
        ld_dqkv % 4)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_mid_bwd: pointers 16-byte aligned, strides multiples of 4");
    if (attn_mid_lds(hd, T, true) > 160 * 1024) return mdt_fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_mid_bwd: (T, hd) does not fit LDS");
    hipStream_t s = (hipStream_t)stream;
    switch (hd) {
        case 16: LAUNCH(launch_bwd<16>(qkv, ld_qkv, out, ld_out, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        case 24: LAUNCH(launch_bwd<24>(qkv, ld_qkv, out, ld_out, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        case 32: LAUNCH(launch_bwd<32>(qkv, ld_qkv, out, ld_out, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        case 48: LAUNCH(launch_bwd<48>(qkv, ld_qkv, out, ld_out, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
        case 64: LAUNCH(launch_bwd<64>(qkv, ld_qkv, out, ld_out, d_out, ld_do, d_qkv, ld_dqkv, B, H, T, scale, s)); break;
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

extern "C" mdt_status mdt_op_rms_bwd_res(const float* x, const float* g, const float* dy, const float* d_res, float* dx, float* dg,
                                         int32_t accumulate_dg, int64_t M, int32_t D, float eps, float* scratch, void* stream) {
    if (!x || !g || !dy || !d_res || !dx || !scratch || M < 1 || D < 1 || D > 512)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_rms_bwd_res: bad argument (D <= 512)");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(mdt_launch_rms_bwd(x, g, dy, dx, 0, scratch, M, D, eps, s, d_res));
    if (dg) LAUNCH(mdt_launch_colsum(scratch, D, (int)((M + 3) / 4), D, dg, accumulate_dg, s));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_scale_residual_fwd(const float* x, const float* z, const float* gamma, float* out, int64_t M, int32_t D,
                                                void* stream) {
    if (!x || !z || !gamma || !out || M < 1 || D < 4 || D % 4 || D > 1024)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_scale_residual_fwd: bad argument (D a multiple of 4, <= 1024)");
    if (!aligned16(x) || !aligned16(z) || !aligned16(gamma) || !aligned16(out))
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_scale_residual_fwd: pointers must be 16-byte aligned");
    const int64_t n4 = M * (D / 4);
    hipLaunchKernelGGL(k_scale_residual_fwd, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, z, gamma, out,
                       n4, D / 4);
    LAUNCH(hipGetLastError());
    return MDT_OK;
}

extern "C" mdt_status mdt_op_scale_residual_rms_fwd(const float* x, const float* z, const float* gamma, const float* g_norm, float* x_new,
                                                    float* h, int64_t M, int32_t D, float eps, void* stream) {
    if (!x || !z || !gamma || !g_norm || !x_new || !h || M < 1 || D < 4 || D % 4 || D > 256 * SRN_C4)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_scale_residual_rms_fwd: bad argument (D a multiple of 4, <= 512)");
    if (!aligned16(x) || !aligned16(z) || !aligned16(gamma) || !aligned16(g_norm) || !aligned16(x_new) || !aligned16(h))
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_scale_residual_rms_fwd: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(k_sr_rms_fwd, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, z, gamma, g_norm, x_new, h, (int)M,
                       D / 4, eps);
    LAUNCH(hipGetLastError());
    return MDT_OK;
}

extern "C" int64_t mdt_op_scale_residual_rms_bwd_scratch(int64_t M, int32_t D) { return 2 * ((M + SRN_ROWS - 1) / SRN_ROWS) * (int64_t)D; }

extern "C" mdt_status mdt_op_scale_residual_rms_bwd(const float* x_new, const float* g_norm, const float* d_h, const float* d_res,
                                                    const float* z, const float* gamma, float* d_x, float* d_z, float* d_gamma,
                                                    float* d_gnorm, int64_t M, int32_t D, float eps, float* scratch, void* stream) {
    if (!x_new || !g_norm || !d_h || !z || !gamma || !d_x || !d_z || !d_gamma || !d_gnorm || !scratch || M < 1 || D < 4 || D % 4 ||
        D > 256 * SRN_C4)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_scale_residual_rms_bwd: bad argument (D a multiple of 4, <= 512; d_res alone may be NULL)");
    if (!aligned16(x_new) || !aligned16(g_norm) || !aligned16(d_h) || !aligned16(d_res) || !aligned16(z) || !aligned16(gamma) ||
        !aligned16(d_x) || !aligned16(d_z) || !aligned16(scratch))
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_scale_residual_rms_bwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t slices = (M + SRN_ROWS - 1) / SRN_ROWS;
    float* pgn = scratch;
    float* pgam = scratch + slices * D;
    hipLaunchKernelGGL(k_sr_rms_bwd, dim3((unsigned)slices), dim3(256), 0, s, x_new, g_norm, d_h, d_res, z, gamma, d_x, d_z, pgn, pgam,
                       (int)M, D / 4, eps);
    LAUNCH(hipGetLastError());
    LAUNCH(mdt_launch_colsum(pgn, D, (int)slices, D, d_gnorm, 0, s));
    LAUNCH(mdt_launch_colsum(pgam, D, (int)slices, D, d_gamma, 0, s));
    return MDT_OK;
}

extern "C" int64_t mdt_op_scale_residual_bwd_scratch(int64_t M, int32_t D) { return ((M + SR_ROWS - 1) / SR_ROWS) * (int64_t)D; }

extern "C" mdt_status mdt_op_scale_residual_bwd(const float* g, const float* z, const float* gamma, float* dz, float* dgamma,
                                                int64_t M, int32_t D, float* scratch, void* stream) {
    if (!g || !z || !gamma || !dz || !dgamma || !scratch || M < 1 || D < 4 || D % 4 || D > 1024)
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_scale_residual_bwd: bad argument (D a multiple of 4, <= 1024)");
    if (!aligned16(g) || !aligned16(z) || !aligned16(gamma) || !aligned16(dz) || !aligned16(scratch))
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_scale_residual_bwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t slices = (M + SR_ROWS - 1) / SR_ROWS;
    hipLaunchKernelGGL(k_scale_residual_bwd, dim3((unsigned)slices), dim3(256), 0, s, g, z, gamma, dz, scratch, M, D / 4);
    LAUNCH(hipGetLastError());
    LAUNCH(mdt_launch_colsum(scratch, D, (int)slices, D, dgamma, 0, s));
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


// ------------------------------------------------------------------------------------------------
// masked per-patch MSE (compute_loss, masked_transformer_decoder.py:228-262) without the patchified copy of the images and
// the five elementwise passes over (B, 2, n, 768) tensors: one workgroup per (b, x, patch) reads its 768 reconstruction
// values (coalesced) and the patch's pixels out of the image planes, sums the squares; one more workgroup adds the per-patch
// sums in a fixed order.  Visible patches (mask 0) are skipped.
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ int64_t img_index(int64_t bx, int patch, int e, int C_, int R, int P) {
    const int g = R / P, gh = patch / g, gw = patch - gh * g;
    const int c = e % C_, pp = e / C_, ph = pp / P, pw = pp - ph * P;
    return ((bx * C_ + c) * R + gh * P + ph) * (int64_t)R + gw * P + pw;
}
__global__ __launch_bounds__(256) void k_patch_mse_fwd(const float* __restrict__ rec, const float* __restrict__ imgs,
                                                       const float* __restrict__ mask, float* __restrict__ partial, int X, int n,
                                                       int C_, int R, int P) {
    const int64_t blk = blockIdx.x;  // (b * X + x) * n + patch
    const int patch = (int)(blk % n);
    const int64_t bx = blk / n, b = bx / X;
    __shared__ float part[4];
    if (mask[b * n + patch] == 0.f) {
        if (threadIdx.x == 0) partial[blk] = 0.f;
        return;
    }
    const int E = P * P * C_;
    float s = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) {
        const float d = rec[blk * E + e] - imgs[img_index(bx, patch, e, C_, R, P)];
        s = fmaf(d, d, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blk] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)E;
}
// one workgroup: loss and sum(mask) (fixed order: deterministic)
__global__ __launch_bounds__(1024) void k_patch_mse_reduce(const float* __restrict__ partial, const float* __restrict__ mask,
                                                           float* __restrict__ loss, float* __restrict__ mask_sum, int64_t B, int X,
                                                           int n) {
    __shared__ float ps[16], ms[16];
    float s = 0.f, m = 0.f;
    for (int64_t i = threadIdx.x; i < B * X * n; i += 1024) s += partial[i];  // visible patches hold 0
    for (int64_t i = threadIdx.x; i < B * n; i += 1024) m += mask[i];
    s = wave_sum(s); m = wave_sum(m);
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; ms[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tm = 0.f;
        for (int w = 0; w < 16; ++w) { ts += ps[w]; tm += ms[w]; }
        *mask_sum = tm;
        *loss = ts / tm / (float)X;
    }
}
__global__ __launch_bounds__(256) void k_patch_mse_bwd(const float* __restrict__ rec, const float* __restrict__ imgs,
                                                       const float* __restrict__ mask, const float* __restrict__ mask_sum,
                                                       const float* __restrict__ g, float* __restrict__ d_rec, int X, int n, int C_,
                                                       int R, int P) {
    const int64_t blk = blockIdx.x;
    const int patch = (int)(blk % n);
    const int64_t bx = blk / n, b = bx / X;
    const int E = P * P * C_;
    const bool on = mask[b * n + patch] != 0.f;
    const float k = on ? 2.0f * g[0] / ((float)E * mask_sum[0] * (float)X) : 0.f;
    for (int e = threadIdx.x; e < E; e += 256)
        d_rec[blk * E + e] = on ? k * (rec[blk * E + e] - imgs[img_index(bx, patch, e, C_, R, P)]) : 0.f;
}
}  // namespace

extern "C" mdt_status mdt_op_patch_mse_fwd(const float* rec, const float* imgs, const float* mask, float* partial, float* loss,
                                           float* mask_sum, int64_t B, int32_t X, int32_t C_, int32_t R, int32_t P, void* stream) {
    if (!rec || !imgs || !mask || !partial || !loss || !mask_sum || B < 1 || X < 1 || C_ < 1 || P < 1 || R < P || (R % P))
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_patch_mse_fwd: bad argument");
    const int n = (R / P) * (R / P);
    if (B * X * n > ((int64_t)1 << 31) - 1) return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_patch_mse_fwd: batch too large");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_patch_mse_fwd, dim3((unsigned)(B * X * n)), dim3(256), 0, s, rec, imgs, mask, partial, X, n, C_, R, P);
    LAUNCH(hipGetLastError());
    hipLaunchKernelGGL(k_patch_mse_reduce, dim3(1), dim3(1024), 0, s, partial, mask, loss, mask_sum, B, X, n);
    LAUNCH(hipGetLastError());
    return MDT_OK;
}

extern "C" mdt_status mdt_op_patch_mse_bwd(const float* rec, const float* imgs, const float* mask, const float* mask_sum,
                                           const float* g, float* d_rec, int64_t B, int32_t X, int32_t C_, int32_t R, int32_t P,
                                           void* stream) {
    if (!rec || !imgs || !mask || !mask_sum || !g || !d_rec || B < 1 || X < 1 || C_ < 1 || P < 1 || R < P || (R % P))
        return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_patch_mse_bwd: bad argument");
    const int n = (R / P) * (R / P);
    if (B * X * n > ((int64_t)1 << 31) - 1) return mdt_fail(MDT_ERR_INVALID_ARG, "mdt_op_patch_mse_bwd: batch too large");
    hipLaunchKernelGGL(k_patch_mse_bwd, dim3((unsigned)(B * X * n)), dim3(256), 0, (hipStream_t)stream, rec, imgs, mask, mask_sum, g,
                       d_rec, X, n, C_, R, P);
    LAUNCH(hipGetLastError());
    return MDT_OK;
}
