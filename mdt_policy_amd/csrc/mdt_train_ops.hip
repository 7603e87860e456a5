// mdt_train_ops.hip -- the Linear backward built from the forward fp32-MFMA GEMM, and the kernel-level C ABI of
// the training path (include/mdt_hip_train.h, "kernel level").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <mutex>
#include <vector>

#include "mdt_internal.h"

#define fail mdt_fail

// out = X W^T (+ b):
//   dW[n][k] = sum_m dY[m][n] X[m][k]  -> GEMM with row operand dY^T (N x Mp, materialised by a transpose) and the
//              "weight" X^T packed as an (N' = K, K' = Mp) image: out'(N x K) = dY^T . (X^T)^T
//   dX[m][k] = sum_n dY[m][n] W[n][k]  -> GEMM with row operand dY and the packed image of W^T (N' = K, K' = N)
// Mp = M rounded up to 16; the pad columns / k-slices are zero, so they add nothing.
// Split of the row reduction of dW = dY^T X into S parallel slices of L rows (multiples of 32), so that the product
// fills the chip with WIDE tiles however small N x K is: slice s multiplies its own (N, L) piece of dY^T with its own
// packed (K, L) piece of X^T in ONE batched launch (grid.z = S), and the S partial (N, K) products are summed in a
// fixed order by a column sum -- deterministic, unlike atomics.  Only worth it for deep reductions.
static bool no_splitk() {
    static int v = -1;
    if (v < 0) v = getenv("MDT_HIP_NO_SPLITK") ? 1 : 0;
    return v == 1;
}
static void split_rows(int64_t M, int64_t N, int64_t K, int* S, int* L) {
    const int64_t tiles = ((N + 31) / 32) * ((K + 127) / 128);   // 32 x 128 tiles of one product
    static int64_t target = -1, min_depth = -1;  // tuning knobs (A/B runs): workgroups aimed for, least slice depth
    if (target < 0) { const char* e = getenv("MDT_HIP_SPLIT_TARGET"); target = e ? atoll(e) : 1000; }
    if (min_depth < 0) { const char* e = getenv("MDT_HIP_SPLIT_DEPTH"); min_depth = e ? atoll(e) : 512; }
    // measured (tools/gpu_splitk_ab.sh, B = 1024 step): aiming for ~1000 workgroups of >= 512-deep slices instead of 400
    // of >= 1024 is 11 % faster per step (several 4-wave workgroups share a CU); reductions under 4096 rows (B = 128:
    // 1280 rows) were 2.6 % slower when cut in two, so they keep the 1024-row floor
    const int64_t depth = M >= 4096 ? min_depth : std::max<int64_t>(min_depth, 1024);
    int64_t s = std::max<int64_t>(1, std::min<int64_t>((target + tiles - 1) / tiles, M / depth));
    s = std::min<int64_t>(s, 32);
    if (no_splitk()) s = 1;
    int64_t l = ((M + s - 1) / s + 31) / 32 * 32;                // equal slices: at most 31 pad rows each
    while (l > 16384) { ++s; l = ((M + s - 1) / s + 31) / 32 * 32; }  // keeps the GEMM's K' well inside its limit
    *S = (int)((M + l - 1) / l);
    *L = (int)l;
}

// The same for k_gemm_tn (64 x 128 tiles, no transposed / packed copies to amortise): ~6 workgroups per CU queued (3 resident), slices of at
// least 128 rows -- small batches (B = 128: 1280 rows) need the split even more than large ones.
static void split_rows_tn(int64_t M, int64_t N, int64_t K, int* S, int* L) {
    int tn, tk;
    mdt_gemm_tn_tile(M, (int)N, (int)K, &tn, &tk);
    // in units of 64-column tiles whatever the n-tile: a 128- / 192-wide workgroup counts two / three times (it has as many waves)
    const int64_t tiles = ((N + 63) / 64) * ((K + tk - 1) / tk);
    static int64_t target = -1;
    if (target < 0) { const char* e = getenv("MDT_HIP_TN_TARGET"); target = e ? atoll(e) : 512; }  // measured at B = 1024: round 2, everything in one stream: 768 -> 11.27 ms step / 42.1 ms head, 1536 -> 11.18 / 41.4, 2304 -> 11.22 / 41.4; round 6, the weight gradients beside the chain on a side stream: 1536 -> 9.34, 768 -> 9.25, 512 -> 9.23 (fewer, deeper slices: less partial-sum traffic beside the chain)
    int64_t s = std::max<int64_t>(1, std::min<int64_t>((target + tiles - 1) / tiles, M / 128));
    if (tn == 192) {
        // twelve-wave workgroups, one per CU: ONE round of them (tools/dw_bench.py at M = 104448, us at 32 / 64 / 128 slices:
        // 1536 x 192 (8 tiles) 597 / 624 / 616, 192 x 768 (4) 495 / 280 / 302, 576 x 192 (3) 483 / 263 / 281)
        const int64_t tiles192 = (N / 192) * (K / 192);
        static int64_t round = -1;  // MDT_HIP_TN_ROUND: workgroups aimed for (A/B runs; beside the backward chain fewer may pay)
        if (round < 0) { const char* e = getenv("MDT_HIP_TN_ROUND"); round = e ? atoll(e) : 256; if (round < 1 || round > 256) round = 256; }
        s = std::max<int64_t>(1, std::min<int64_t>((round + tiles192 / 2) / tiles192, M / 128));
    }
    static int64_t cap = -1;
    if (cap < 0) { const char* e = getenv("MDT_HIP_TN_SLICES"); cap = e ? atoll(e) : 128; }  // measured: masked-image head 37.7 ms (64) -> 37.2 (128) = (192, 256); denoiser step unchanged
    s = std::min<int64_t>(s, cap);
    if (tn != 192) {
        // whole rounds of workgroups here too where that is within a factor 1.5 of the above (192 x 192 at M = 104448, three
        // 64 x 192 tiles: 128 slices = 384 workgroups 118 us, 170 = 510 of the 512 that are resident at once 99 us, 256 108 us)
        const int64_t tiles_wg = ((N + tn - 1) / tn) * ((K + tk - 1) / tk);
        const int64_t slots = 256 * (tn == 64 ? (tk == 192 ? 2 : 3) : (tk == 192 ? 1 : 2));   // resident workgroups: LDS 51 / 68 / 68 / 84 KB
        const int64_t rounds = std::max<int64_t>(1, (tiles_wg * s + slots / 2) / slots);
        const int64_t s2 = std::min<int64_t>((rounds * slots) / tiles_wg, M / 128);
        if (s2 >= 1 && 2 * s2 <= 3 * s && 3 * s2 >= 2 * s) s = s2;
    }
    if (no_splitk()) s = 1;
    const int64_t l = ((M + s - 1) / s + 31) / 32 * 32;
    *S = (int)((M + l - 1) / l);
    *L = (int)l;
}

// Floats of scratch ONE product of exactly M rows needs (what mdt_linear_bwd carves out of `scratch` for it).
static int64_t linear_bwd_scratch_exact(int64_t M, int64_t N, int64_t K) {
    int S, L, St, Lt;
    split_rows(M, N, K, &S, &L);
    split_rows_tn(M, N, K, &St, &Lt);
    const int64_t Mp = (int64_t)S * L;
    return std::max<int64_t>((N + K) * Mp + (int64_t)S * N * K + N * (Mp / 32 + 2), (int64_t)St * N * K + (int64_t)St * N + 64);
}

// Floats of scratch that serve EVERY row count up to M.  The callers size their scratch once, for the largest batch seen, and
// reuse it for every smaller one -- and the slice count is not monotone in M (the n-tile changes at 8192 rows, the 192-wide tile
// wants one round of workgroups, the narrower ones whole rounds of the resident workgroups: a smaller batch can be cut into MORE
// slices than the capacity batch), so the exact figure at M is not enough.  Closed-form bounds of both paths:
//   k_gemm_tn: S <= M / 128 always; S <= 3/2 x min(ceil(target / tiles), cap) for the 64- / 128-wide tiles (`tiles` counted with
//              the k-tile that gives fewer of them), S <= (256 + t / 2) / t one-round slices for the 192-wide one;
//   transposed-copy path (MDT_HIP_DW_TN=0 or odd shapes): its slice count s only grows with M and Mp = S L <= M + 32 s.
// tests/test_cpu_abi.py sweeps the exact need of every row count below a capacity against this bound.
int64_t mdt_linear_bwd_scratch(int64_t M, int64_t N, int64_t K) {
    if (M < 1) M = 1;
    // k_gemm_tn
    static int64_t target = -1, cap = -1;
    if (target < 0) { const char* e = getenv("MDT_HIP_TN_TARGET"); target = e ? atoll(e) : 1536; }
    if (cap < 0) { const char* e = getenv("MDT_HIP_TN_SLICES"); cap = e ? atoll(e) : 128; }
    const int64_t tiles_min = ((N + 63) / 64) * ((K + 191) / 192);
    int64_t s_tn = (3 * std::min<int64_t>((target + tiles_min - 1) / tiles_min, cap) + 1) / 2;
    if (N % 192 == 0 && K % 192 == 0) {
        const int64_t t192 = (N / 192) * (K / 192);
        s_tn = std::max<int64_t>(s_tn, (256 + t192 / 2) / t192);
    }
    s_tn = std::max<int64_t>(1, std::min<int64_t>(s_tn, M / 128));
    const int64_t need_tn = s_tn * N * K + s_tn * N + 64;
    // transposed-copy path: split_rows' slice count before it is rounded to whole slices -- non-decreasing in M
    int S, L;
    split_rows(M, N, K, &S, &L);
    int64_t s_old = std::max<int64_t>(S, std::min<int64_t>(32, std::max<int64_t>(1, M / 512)));
    while ((M + s_old - 1) / s_old > 16384) ++s_old;
    const int64_t Mp = M + 32 * s_old;
    const int64_t need_old = (N + K) * Mp + s_old * N * K + N * (Mp / 32 + 2);
    return std::max(std::max(need_tn, need_old), linear_bwd_scratch_exact(M, N, K));
}

mdt_status mdt_linear_bwd(const mdt_linear_bwd_args& a, hipStream_t s, mdt_colsum_entry* defer_bias, float* bias_space) {
    if (a.M < 1 || a.N < 1 || a.K < 1 || (a.K % 16)) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: bad shape");
    if (defer_bias) defer_bias->src = nullptr;
    // the bias gradient rides on the transpose the dW path needs anyway (per-32-row column partials)
    const bool bias_from_partials = a.dbias && a.dW && !(a.N % 16);
    if (a.dbias && !bias_from_partials) LAUNCH(mdt_launch_colsum(a.dY, a.ldy, a.M, a.N, a.dbias, a.accumulate_dw, s));
    // (bias_from_partials: the dW path below also leaves the bias gradient -- per-slice column sums of dY from k_gemm_tn,
    //  or per-32-row partials from the transpose of the older path)
    static int use_tn = -1;  // MDT_HIP_DW_TN=0: the transposed-copy + packed-copy + forward-GEMM path (A/B runs)
    if (use_tn < 0) { const char* e = getenv("MDT_HIP_DW_TN"); use_tn = e ? atoi(e) : 1; }
    if (a.dW && use_tn && !(a.N % 16) && !(a.ldy % 4) && !(a.ldx % 4)) {
        // dW straight from dY and X (k_gemm_tn): S slices of the row reduction as one batched launch, partial products and
        // the bias gradient's per-slice column sums added up in a fixed order afterwards
        int S, L;
        split_rows_tn(a.M, a.N, a.K, &S, &L);
        // from 8192 rows on: the bf16 split kernel, ONE round of its one-per-CU workgroups (never more slices than the fp32 tiling
        // takes: the scratch bound is that tiling's)
        // ... where its 128-wide n-tiles pad N by at most an eighth (tools/dw_bench.py, us fp32 / split: 12288 x 1536 x 384 127 / 101,
        // x 384 x 1536 128 / 100, x 1152 x 384 105 / 80, x 384 x 384 47 / 36; 104448 x 1536 x 192 598 / 510, x 576 x 192 217 / 185;
        // N = 192 -- a quarter of its second tile empty -- loses: x 192 x 768 286 / 311, x 192 x 1536 507 / 590)
        int stn = 128, stk = 128;
        mdt_gemm_tn_split_tile(a.N, a.K, &stn, &stk);
        const bool tn_split = a.M >= 8192 && mdt_gemm_tn_split_on() && 8 * ((a.N + stn - 1) / stn * stn) <= 9 * a.N;
        if (tn_split) {
            const int64_t tiles = ((a.N + stn - 1) / stn) * ((a.K + stk - 1) / stk);
            int64_t s2 = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(256 / tiles, S), a.M / 256));   // never a second round
            if (no_splitk()) s2 = 1;
            const int64_t l2 = ((a.M + s2 - 1) / s2 + 31) / 32 * 32;
            S = (int)((a.M + l2 - 1) / l2);
            L = (int)l2;
        }
        float* parts = a.scratch;                              // [S][N][K] (S > 1)
        const bool defer = a.dbias && defer_bias && bias_space && S <= 256;
        float* bpart = defer ? bias_space : parts + (int64_t)S * a.N * a.K;  // [S][N]
        if (tn_split)
            LAUNCH(mdt_launch_gemm_tn_split(a.dY, a.ldy, a.X, a.ldx, S > 1 ? parts : a.dW, (int64_t)a.N * a.K, a.M, a.N, a.K, S, L,
                                            S > 1 ? 0 : a.accumulate_dw, a.dbias ? bpart : nullptr, s));
        else
        LAUNCH(mdt_launch_gemm_tn(a.dY, a.ldy, a.X, a.ldx, S > 1 ? parts : a.dW, (int64_t)a.N * a.K, a.M, a.N, a.K, S, L,
                                  S > 1 ? 0 : a.accumulate_dw, a.dbias ? bpart : nullptr, s));
        if (S > 1) LAUNCH(mdt_launch_colsum(parts, (int64_t)a.N * a.K, S, a.N * a.K, a.dW, a.accumulate_dw, s));
        if (defer) *defer_bias = mdt_colsum_entry{bpart, a.dbias, (int64_t)a.N, S, a.N, a.accumulate_dw};
        else if (a.dbias) LAUNCH(mdt_launch_colsum(bpart, a.N, S, a.N, a.dbias, a.accumulate_dw, s));
    } else if (a.dW) {
        if (a.N % 16) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: N must be a multiple of 16 for dW");
        int S, L;
        split_rows(a.M, a.N, a.K, &S, &L);
        const int64_t Mp = (int64_t)S * L;
        float* dYt = a.scratch;                                  // [S][N][L]
        float* Xt = dYt + (int64_t)a.N * Mp;                     // [S] packed (N' = K, K' = L)
        float* parts = Xt + (int64_t)a.K * Mp;                   // [S][N][K] partial products (S > 1)
        float* bpart = parts + (int64_t)S * a.N * a.K;           // (ceil(M/32), N) column partials of dY
        if (Mp != a.M) {  // zero pad rows of the last slice
            HIP_TRY(hipMemsetAsync(dYt + (int64_t)(S - 1) * a.N * L, 0, (size_t)a.N * L * sizeof(float), s));
            HIP_TRY(hipMemsetAsync(Xt + (int64_t)(S - 1) * a.K * L, 0, (size_t)a.K * L * sizeof(float), s));
        }
        LAUNCH(mdt_launch_transpose_ld(a.dY, a.ldy, dYt, L, a.M, a.N, bias_from_partials ? bpart : nullptr, s, L));
        if (bias_from_partials) LAUNCH(mdt_launch_colsum(bpart, a.N, (a.M + 31) / 32, a.N, a.dbias, a.accumulate_dw, s));
        LAUNCH(mdt_launch_pack_weight_t(a.X, a.M, a.K, a.ldx, Xt, 0, L / 16, s, L));
        Lin w;
        w.wp = Xt; w.bias = nullptr; w.N = a.K; w.K = L;
        mdt_gemm_args g = gemm_args(dYt, L, w, S > 1 ? parts : a.dW, a.K, a.N);
        if (S > 1) {
            g.batch = S; g.bs_a = (int64_t)a.N * L; g.bs_w = (int64_t)a.K * L; g.bs_out = (int64_t)a.N * a.K;
        } else {
            g.residual = a.accumulate_dw;
        }
        LAUNCH(mdt_launch_gemm(g, s));
        if (S > 1) LAUNCH(mdt_launch_colsum(parts, (int64_t)a.N * a.K, S, a.N * a.K, a.dW, a.accumulate_dw, s));
    }
    if (a.dX) {
        if (!a.Wt || (a.N % 16)) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: dX needs the packed W^T and N % 16 == 0");
        Lin w;
        w.wp = const_cast<float*>(a.Wt); w.bias = nullptr; w.N = a.K; w.K = a.N;
        mdt_gemm_args g = gemm_args(a.dY, a.ldy, w, a.dX, a.ldxo, a.M);
        g.residual = a.accumulate_dx;
        if (a.dx_act_u) {  // dX = (dY W) * act'(u): the activation's backward rides on this product's epilogue
            if (a.accumulate_dx || a.N > 512) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: dx_act_u needs accumulate_dx = 0 and N <= 512");
            g.aux = a.dx_act_u; g.aux_mode = 2; g.act = a.dx_act;
            if (a.dx_act == MDT_ACT_SWIGLU) {  // dX (M, 2K; ldxo) from d = dY W (M, K) and u (M, 2K; ldxo)
                if (a.ldxo < 2 * (int64_t)a.K) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: SwishGLU backward writes 2 K columns (ldxo >= 2 K)");
                g.aux_mode = 4; g.act = MDT_ACT_NONE;
            }
        }
        LAUNCH(mdt_launch_gemm(g, s));
    }
    return MDT_OK;
}

extern "C" int64_t mdt_op_linear_bwd_scratch(int64_t M, int64_t N, int64_t K) { return mdt_linear_bwd_scratch(M, N, K); }
extern "C" int64_t mdt_op_linear_bwd_scratch_exact(int64_t M, int64_t N, int64_t K) { return linear_bwd_scratch_exact(M, N, K); }

extern "C" mdt_status mdt_op_linear_bwd(const mdt_linear_bwd_args* a, void* stream) {
    if (!a || !a->X || !a->dY || !a->scratch) return fail(MDT_ERR_INVALID_ARG, "mdt_op_linear_bwd: null argument");
    return mdt_linear_bwd(*a, (hipStream_t)stream);
}

extern "C" mdt_status mdt_op_pack_weight_t(const float* src, int64_t rows, int64_t cols, int64_t ld, float* packed,
                                           int64_t k_off, int64_t k_total, void* stream) {
    if (!src || !packed || rows < 0 || cols < 1 || (cols % 16) || (k_total % 16) || k_off < 0 || k_off + rows > k_total)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight_t: bad argument");
    LAUNCH(mdt_launch_pack_weight_t(src, (int)rows, (int)cols, ld, packed, (int)k_off, (int)(k_total / 16),
                                    (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_ln_fwd_train(const mdt_ln_train_args* a, void* stream) {
    if (!a || !a->x || !a->w || !a->out || a->M < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_op_ln_fwd_train: bad argument");
    LAUNCH(mdt_launch_ln_fwd_train(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_ln_bwd(const mdt_ln_bwd_args* a, void* stream) {
    if (!a || !a->x || !a->stats || !a->w || !a->dh || !a->dx || !a->pw || a->B < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_ln_bwd: bad argument");
    LAUNCH(mdt_launch_ln_bwd(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_attn_bwd(const mdt_attn_bwd_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->v || !a->d_out || !a->dq || !a->dk || !a->dv)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_bwd: null argument");
    LAUNCH(mdt_launch_attn_bwd(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_act_fwd(const float* u, float* out, int64_t n, int32_t act, void* stream) {
    if (!u || !out || n < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_op_act_fwd: bad argument");
    LAUNCH(mdt_launch_act_fwd(u, out, n, act, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_act_bwd(const float* u, const float* dy, float* du, int64_t n, int32_t act, void* stream) {
    if (!u || !dy || !du || n < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_op_act_bwd: bad argument");
    LAUNCH(mdt_launch_act_bwd(u, dy, du, n, act, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_merge_fwd(const mdt_merge_args* a, void* stream) {
    if (!a || !a->x || !a->a || !a->out || a->B < 1 || a->rows_per_sample < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_merge_fwd: bad argument");
    LAUNCH(mdt_launch_merge_fwd(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_merge_bwd(const mdt_merge_args* a, void* stream) {
    if (!a || !a->x || !a->a || !a->out || a->B < 1 || a->rows_per_sample < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_merge_bwd: bad argument");
    LAUNCH(mdt_launch_merge_bwd(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_merge_ln_fwd(const mdt_merge_args* g, const mdt_ln_train_args* l, void* stream) {
    if (!g || !l || !g->x || !g->a || !g->out || !l->w || !l->out || g->B < 1 || g->rows_per_sample < 1 ||
        (int64_t)g->B * g->rows_per_sample != l->M || g->D != l->D)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_merge_ln_fwd: bad argument");
    LAUNCH(mdt_launch_merge_ln_fwd(*g, *l, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_ln_bwd_merge(const mdt_ln_bwd_args* a, const mdt_merge_args* g, void* stream) {
    if (!a || !g || !a->x || !a->stats || !a->w || !a->dh || !a->dx || !a->pw || a->B < 1 || !g->a || !g->out ||
        g->B != a->B || g->rows_per_sample != a->rows_per_sample || g->D != a->D)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_ln_bwd_merge: bad argument");
    LAUNCH(mdt_launch_ln_bwd_merge(*a, *g, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_attn_fwd_train(const mdt_attn_train_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->v || !a->out) return fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_fwd_train: null argument");
    LAUNCH(mdt_launch_attn_fwd_train(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, int32_t accumulate,
                                    void* stream) {
    if (!X || !out || M < 1 || N < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_op_colsum: bad argument");
    LAUNCH(mdt_launch_colsum(X, ldx, (int)M, (int)N, out, accumulate, (hipStream_t)stream));
    return MDT_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-tensor optimizer updates: the (tensor, chunk) table of a call lives in device memory.  A training loop passes the
// same pointers step after step, so the tables are cached by content (16 slots per device, least recently used replaced -- the reference's
// configure_optimizers builds 6-7 parameter groups plus the EMA list --, pinned staging): a repeated
// table costs nothing, a new one is one asynchronous copy -- the call never waits for the stream (a stream
// synchronisation here cost 1.5 ms per B = 1024 step: the host lost its run-ahead over the backward's launches).
// ------------------------------------------------------------------------------------------------
namespace {
struct OptSlot {
    void* dev = nullptr;
    void* host = nullptr;  // pinned
    size_t cap = 0;
    hipEvent_t ev = nullptr;  // recorded behind the kernel that last read `dev`
    hipStream_t stream = nullptr;  // ... and the stream it was recorded on (the upload was enqueued there as well)
    bool busy = false;
    std::vector<char> key;    // the caller's table bytes this slot holds
    size_t tab_bytes = 0;
    int n_blocks = 0;
    uint64_t last_use = 0;
};
struct OptTable {  // one per device: the tables live in that device's memory
    std::mutex mu;
    OptSlot slots[16];
    uint64_t clock = 0;
};
OptTable g_opt_dev[32];
}  // namespace

static mdt_status upload_opt_table(OptTable& g_opt, const mdt_opt_tensor* tensors, int n, const mdt_opt_tensor** d_tab,
                                   const int2** d_blocks, int* n_blocks, OptSlot** used, hipStream_t s) {
    const int CH = 4096;  // OPT_CHUNK of the kernels
    const size_t key_bytes = (size_t)n * sizeof(mdt_opt_tensor);
    for (OptSlot& sl : g_opt.slots)
        if (sl.dev && sl.key.size() == key_bytes && memcmp(sl.key.data(), tensors, key_bytes) == 0) {
            *d_tab = (const mdt_opt_tensor*)sl.dev;
            *d_blocks = (const int2*)((const char*)sl.dev + sl.tab_bytes);
            *n_blocks = sl.n_blocks;
            *used = &sl;
            sl.last_use = ++g_opt.clock;
            // a hit from ANOTHER stream: the table's copy (and its last reader) were enqueued elsewhere -- order behind them
            if (sl.busy && sl.stream != s) HIP_TRY(hipStreamWaitEvent(s, sl.ev, 0));
            return MDT_OK;
        }
    std::vector<int2> blocks;
    for (int i = 0; i < n; ++i) {
        if (tensors[i].numel < 0 || tensors[i].numel > ((int64_t)1 << 31) - CH)
            return fail(MDT_ERR_INVALID_ARG, "multi-tensor update: tensor %d has an unsupported size", i);
        for (int64_t off = 0; off < tensors[i].numel; off += CH) blocks.push_back(make_int2(i, (int)off));
    }
    const size_t tab_bytes = (key_bytes + 255) & ~(size_t)255;
    const size_t total = tab_bytes + blocks.size() * sizeof(int2);
    OptSlot* lru = &g_opt.slots[0];
    for (OptSlot& c : g_opt.slots)
        if (c.last_use < lru->last_use) lru = &c;
    OptSlot& sl = *lru;
    sl.last_use = ++g_opt.clock;
    if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.busy) HIP_TRY(hipEventSynchronize(sl.ev));  // the last kernel that read this slot (long done in practice)
    if (total > sl.cap) {
        if (sl.dev) HIP_TRY(hipFree(sl.dev));
        if (sl.host) HIP_TRY(hipHostFree(sl.host));
        sl.dev = sl.host = nullptr; sl.cap = 0;
        HIP_TRY(hipMalloc(&sl.dev, total * 2));
        HIP_TRY(hipHostMalloc(&sl.host, total * 2, hipHostMallocDefault));
        sl.cap = total * 2;
    }
    memcpy(sl.host, tensors, key_bytes);
    memcpy((char*)sl.host + tab_bytes, blocks.data(), blocks.size() * sizeof(int2));
    HIP_TRY(hipMemcpyAsync(sl.dev, sl.host, total, hipMemcpyHostToDevice, s));
    sl.key.assign((const char*)tensors, (const char*)tensors + key_bytes);
    sl.tab_bytes = tab_bytes;
    sl.n_blocks = (int)blocks.size();
    *d_tab = (const mdt_opt_tensor*)sl.dev;
    *d_blocks = (const int2*)((const char*)sl.dev + tab_bytes);
    *n_blocks = sl.n_blocks;
    *used = &sl;
    return MDT_OK;
}

// a byte blob (move table + block list of k_multi_load) through the same content-keyed slots
static mdt_status upload_blob(OptTable& g_opt, const std::vector<char>& bytes, const void** dev, OptSlot** used, hipStream_t s) {
    for (OptSlot& sl : g_opt.slots)
        if (sl.dev && sl.key.size() == bytes.size() && memcmp(sl.key.data(), bytes.data(), bytes.size()) == 0) {
            *dev = sl.dev; *used = &sl;
            sl.last_use = ++g_opt.clock;
            if (sl.busy && sl.stream != s) HIP_TRY(hipStreamWaitEvent(s, sl.ev, 0));  // see upload_opt_table
            return MDT_OK;
        }
    OptSlot* lru = &g_opt.slots[0];
    for (OptSlot& c : g_opt.slots)
        if (c.last_use < lru->last_use) lru = &c;
    OptSlot& sl = *lru;
    sl.last_use = ++g_opt.clock;
    if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.busy) HIP_TRY(hipEventSynchronize(sl.ev));
    if (bytes.size() > sl.cap) {
        if (sl.dev) HIP_TRY(hipFree(sl.dev));
        if (sl.host) HIP_TRY(hipHostFree(sl.host));
        sl.dev = sl.host = nullptr; sl.cap = 0;
        HIP_TRY(hipMalloc(&sl.dev, bytes.size() * 2));
        HIP_TRY(hipHostMalloc(&sl.host, bytes.size() * 2, hipHostMallocDefault));
        sl.cap = bytes.size() * 2;
    }
    memcpy(sl.host, bytes.data(), bytes.size());
    HIP_TRY(hipMemcpyAsync(sl.dev, sl.host, bytes.size(), hipMemcpyHostToDevice, s));
    sl.key = bytes;
    sl.tab_bytes = 0; sl.n_blocks = 0;
    *dev = sl.dev; *used = &sl;
    return MDT_OK;
}

// Fragment-packed images of n Linear weights (N_i, K_i) -- the forward operand into wp[i] and, where wt[i] is given, the image
// of W^T for dX = dY W -- in ONE launch (a training step re-packs every weight of a module after its optimizer step).
extern "C" mdt_status mdt_op_pack_many(int32_t n, const float* const* srcs, const int32_t* N, const int32_t* K, float* const* wp,
                                       float* const* wt, void* stream) {
    if (n < 0 || (n > 0 && (!srcs || !N || !K || !wp))) return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_many: bad argument");
    if (n == 0) return MDT_OK;
    std::vector<mdt_load_entry> tab;
    std::vector<int2> blocks;
    auto add = [&](const float* src, float* dst, int kind, int rows, int Kc, int p1) {
        mdt_load_entry e;
        memset(&e, 0, sizeof e);
        e.src = src; e.dst = dst; e.kind = kind; e.rows = rows; e.K = Kc; e.p0 = 0; e.p1 = p1;
        const int64_t work = kind == MDT_LOAD_PACK_T ? (int64_t)((rows + 3) / 4) * Kc : (int64_t)rows * (Kc / 4);
        for (int64_t c = 0; c * 1024 < work; ++c) blocks.push_back(make_int2((int)tab.size(), (int)c));
        tab.push_back(e);
    };
    for (int i = 0; i < n; ++i) {
        if (!srcs[i] || !wp[i] || N[i] < 16 || K[i] < 16 || (N[i] % 16) || (K[i] % 16) || ((uintptr_t)srcs[i] & 15) || ((uintptr_t)wp[i] & 15))
            return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_many: entry %d (N, K multiples of 16; 16-byte aligned pointers)", i);
        add(srcs[i], wp[i], MDT_LOAD_PACK, N[i], K[i], 0);
        if (wt && wt[i]) add(srcs[i], wt[i], MDT_LOAD_PACK_T, N[i], K[i], N[i] / 16);
    }
    const size_t tab_bytes = (tab.size() * sizeof(mdt_load_entry) + 255) & ~(size_t)255;
    std::vector<char> bytes(tab_bytes + blocks.size() * sizeof(int2), 0);
    memcpy(bytes.data(), tab.data(), tab.size() * sizeof(mdt_load_entry));
    memcpy(bytes.data() + tab_bytes, blocks.data(), blocks.size() * sizeof(int2));
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 32) return fail(MDT_ERR_INVALID_ARG, "device index %d out of range", dev);
    OptTable& g_opt = g_opt_dev[dev];
    std::lock_guard<std::mutex> lock(g_opt.mu);
    hipStream_t s = (hipStream_t)stream;
    const void* d = nullptr;
    OptSlot* slot = nullptr;
    MDT_TRY(upload_blob(g_opt, bytes, &d, &slot, s));
    LAUNCH(mdt_launch_multi_load((const mdt_load_entry*)d, (const int2*)((const char*)d + tab_bytes), (int)blocks.size(), s));
    HIP_TRY(hipEventRecord(slot->ev, s));
    slot->busy = true;
    slot->stream = s;
    return MDT_OK;
}

extern "C" mdt_status mdt_op_multi_adamw(const mdt_opt_tensor* tensors, int32_t n, float lr, float beta1, float beta2,
                                         float eps, float weight_decay, int64_t step, void* stream) {
    if (!tensors || n < 0 || step < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_op_multi_adamw: bad argument");
    for (int i = 0; i < n; ++i)
        if (!tensors[i].p || !tensors[i].g || !tensors[i].m || !tensors[i].v)
            return fail(MDT_ERR_INVALID_ARG, "mdt_op_multi_adamw: tensor %d lacks p / g / m / v", i);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 32) return fail(MDT_ERR_INVALID_ARG, "device index %d out of range", dev);
    OptTable& g_opt = g_opt_dev[dev];
    std::lock_guard<std::mutex> lock(g_opt.mu);
    hipStream_t s = (hipStream_t)stream;
    const mdt_opt_tensor* tab; const int2* blocks; int nb;
    OptSlot* slot = nullptr;
    MDT_TRY(upload_opt_table(g_opt, tensors, n, &tab, &blocks, &nb, &slot, s));
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    LAUNCH(mdt_launch_multi_adamw(tab, blocks, nb, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)std::sqrt(bc2), s));
    HIP_TRY(hipEventRecord(slot->ev, s));
    slot->busy = true;
    slot->stream = s;
    return MDT_OK;
}

extern "C" mdt_status mdt_op_multi_ema(const mdt_opt_tensor* tensors, int32_t n, float decay, void* stream) {
    if (!tensors || n < 0 || !(decay >= 0.f && decay <= 1.f)) return fail(MDT_ERR_INVALID_ARG, "mdt_op_multi_ema: bad argument");
    for (int i = 0; i < n; ++i)
        if (!tensors[i].p || !tensors[i].ema) return fail(MDT_ERR_INVALID_ARG, "mdt_op_multi_ema: tensor %d lacks p / ema", i);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 32) return fail(MDT_ERR_INVALID_ARG, "device index %d out of range", dev);
    OptTable& g_opt = g_opt_dev[dev];
    std::lock_guard<std::mutex> lock(g_opt.mu);
    hipStream_t s = (hipStream_t)stream;
    const mdt_opt_tensor* tab; const int2* blocks; int nb;
    OptSlot* slot = nullptr;
    MDT_TRY(upload_opt_table(g_opt, tensors, n, &tab, &blocks, &nb, &slot, s));
    LAUNCH(mdt_launch_multi_axpby(tab, blocks, nb, decay, 1.0f - decay, s));
    HIP_TRY(hipEventRecord(slot->ev, s));
    slot->busy = true;
    slot->stream = s;
    return MDT_OK;
}
