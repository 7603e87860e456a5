// mdt_train_ops.hip -- the Linear backward built from the forward fp32-MFMA GEMM, and the kernel-level C ABI of
// the training path (include/mdt_hip_train.h, "kernel level").
#include <hip/hip_runtime.h>

#include <algorithm>
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
mdt_status mdt_linear_bwd(const mdt_linear_bwd_args& a, hipStream_t s) {
    if (a.M < 1 || a.N < 1 || a.K < 1 || (a.K % 16)) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: bad shape");
    // the bias gradient rides on the transpose the dW path needs anyway (per-32-row column partials)
    const bool bias_from_partials = a.dbias && a.dW && !(a.N % 16);
    if (a.dbias && !bias_from_partials) LAUNCH(mdt_launch_colsum(a.dY, a.ldy, a.M, a.N, a.dbias, a.accumulate_dw, s));
    if (a.dW) {
        if (a.N % 16) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: N must be a multiple of 16 for dW");
        // the reduction runs over the rows; very tall inputs (Perceiver media tokens) go through in slices that
        // keep the GEMM's K' within its limit, each accumulating into dW
        const int SLICE = 32768;
        for (int m0 = 0; m0 < a.M; m0 += SLICE) {
            const int Ms = std::min(SLICE, a.M - m0);
            const int Mp = (Ms + 15) & ~15;
            float* dYt = a.scratch;                          // (N, Mp)
            float* Xt = a.scratch + (size_t)a.N * Mp;        // packed (N' = K, K' = Mp)
            // column partials live in the tail of the packed-X region's slack: (Mp/32 + 1) x N floats behind it
            float* part = bias_from_partials ? Xt + (size_t)a.K * Mp : nullptr;
            if (Mp != Ms) {
                HIP_TRY(hipMemsetAsync(dYt, 0, (size_t)a.N * Mp * sizeof(float), s));
                HIP_TRY(hipMemsetAsync(Xt, 0, (size_t)a.K * Mp * sizeof(float), s));
            }
            LAUNCH(mdt_launch_transpose_ld(a.dY + (int64_t)m0 * a.ldy, a.ldy, dYt, Mp, Ms, a.N, part, s));
            if (part)
                LAUNCH(mdt_launch_colsum(part, a.N, (Ms + 31) / 32, a.N, a.dbias, (a.accumulate_dw || m0 > 0) ? 1 : 0, s));
            LAUNCH(mdt_launch_pack_weight_t(a.X + (int64_t)m0 * a.ldx, Ms, a.K, a.ldx, Xt, 0, Mp / 16, s));
            Lin w;
            w.wp = Xt; w.bias = nullptr; w.N = a.K; w.K = Mp;
            mdt_gemm_args g = gemm_args(dYt, Mp, w, a.dW, a.K, a.N);
            g.residual = (a.accumulate_dw || m0 > 0) ? 1 : 0;
            LAUNCH(mdt_launch_gemm(g, s));
        }
    }
    if (a.dX) {
        if (!a.Wt || (a.N % 16)) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: dX needs the packed W^T and N % 16 == 0");
        Lin w;
        w.wp = const_cast<float*>(a.Wt); w.bias = nullptr; w.N = a.K; w.K = a.N;
        mdt_gemm_args g = gemm_args(a.dY, a.ldy, w, a.dX, a.ldxo, a.M);
        g.residual = a.accumulate_dx;
        LAUNCH(mdt_launch_gemm(g, s));
    }
    return MDT_OK;
}

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
// multi-tensor optimizer updates: the (tensor, chunk) table is rebuilt on the host per call (a few KB) and
// uploaded stream-ordered into a per-process device buffer
// ------------------------------------------------------------------------------------------------
namespace {
struct OptTable {
    std::mutex mu;
    void* dev = nullptr;
    size_t cap = 0;
    std::vector<char> host;
};
OptTable g_opt;
}  // namespace

static mdt_status upload_opt_table(const mdt_opt_tensor* tensors, int n, const mdt_opt_tensor** d_tab, const int2** d_blocks,
                                   int* n_blocks, hipStream_t s) {
    const int CH = 4096;  // OPT_CHUNK of the kernels
    std::vector<int2> blocks;
    for (int i = 0; i < n; ++i) {
        if (tensors[i].numel < 0 || tensors[i].numel > ((int64_t)1 << 31) - CH)
            return fail(MDT_ERR_INVALID_ARG, "multi-tensor update: tensor %d has an unsupported size", i);
        for (int64_t off = 0; off < tensors[i].numel; off += CH) blocks.push_back(make_int2(i, (int)off));
    }
    const size_t tab_bytes = ((size_t)n * sizeof(mdt_opt_tensor) + 255) & ~(size_t)255;
    const size_t total = tab_bytes + blocks.size() * sizeof(int2);
    if (total > g_opt.cap) {
        if (g_opt.dev) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g_opt.dev)); g_opt.dev = nullptr; g_opt.cap = 0; }
        HIP_TRY(hipMalloc(&g_opt.dev, total * 2));
        g_opt.cap = total * 2;
    }
    // the staging copy must stay valid until the async copy ran: synchronise before reusing it
    HIP_TRY(hipStreamSynchronize(s));
    g_opt.host.resize(total);
    memcpy(g_opt.host.data(), tensors, (size_t)n * sizeof(mdt_opt_tensor));
    memcpy(g_opt.host.data() + tab_bytes, blocks.data(), blocks.size() * sizeof(int2));
    HIP_TRY(hipMemcpyAsync(g_opt.dev, g_opt.host.data(), total, hipMemcpyHostToDevice, s));
    *d_tab = (const mdt_opt_tensor*)g_opt.dev;
    *d_blocks = (const int2*)((const char*)g_opt.dev + tab_bytes);
    *n_blocks = (int)blocks.size();
    return MDT_OK;
}

extern "C" mdt_status mdt_op_multi_adamw(const mdt_opt_tensor* tensors, int32_t n, float lr, float beta1, float beta2,
                                         float eps, float weight_decay, int64_t step, void* stream) {
    if (!tensors || n < 0 || step < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_op_multi_adamw: bad argument");
    for (int i = 0; i < n; ++i)
        if (!tensors[i].p || !tensors[i].g || !tensors[i].m || !tensors[i].v)
            return fail(MDT_ERR_INVALID_ARG, "mdt_op_multi_adamw: tensor %d lacks p / g / m / v", i);
    std::lock_guard<std::mutex> lock(g_opt.mu);
    hipStream_t s = (hipStream_t)stream;
    const mdt_opt_tensor* tab; const int2* blocks; int nb;
    MDT_TRY(upload_opt_table(tensors, n, &tab, &blocks, &nb, s));
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    LAUNCH(mdt_launch_multi_adamw(tab, blocks, nb, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)std::sqrt(bc2), s));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_multi_ema(const mdt_opt_tensor* tensors, int32_t n, float decay, void* stream) {
    if (!tensors || n < 0 || !(decay >= 0.f && decay <= 1.f)) return fail(MDT_ERR_INVALID_ARG, "mdt_op_multi_ema: bad argument");
    for (int i = 0; i < n; ++i)
        if (!tensors[i].p || !tensors[i].ema) return fail(MDT_ERR_INVALID_ARG, "mdt_op_multi_ema: tensor %d lacks p / ema", i);
    std::lock_guard<std::mutex> lock(g_opt.mu);
    hipStream_t s = (hipStream_t)stream;
    const mdt_opt_tensor* tab; const int2* blocks; int nb;
    MDT_TRY(upload_opt_table(tensors, n, &tab, &blocks, &nb, s));
    LAUNCH(mdt_launch_multi_axpby(tab, blocks, nb, decay, 1.0f - decay, s));
    return MDT_OK;
}
