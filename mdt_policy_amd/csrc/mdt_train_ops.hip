// mdt_train_ops.hip -- the Linear backward built from the forward fp32-MFMA GEMM, and the kernel-level C ABI of
// the training path (include/mdt_hip_train.h, "kernel level").
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mdt_internal.h"

#define fail mdt_fail

// out = X W^T (+ b):
//   dW[n][k] = sum_m dY[m][n] X[m][k]  -> GEMM with row operand dY^T (N x Mp, materialised by a transpose) and the
//              "weight" X^T packed as an (N' = K, K' = Mp) image: out'(N x K) = dY^T . (X^T)^T
//   dX[m][k] = sum_n dY[m][n] W[n][k]  -> GEMM with row operand dY and the packed image of W^T (N' = K, K' = N)
// Mp = M rounded up to 16; the pad columns / k-slices are zero, so they add nothing.
mdt_status mdt_linear_bwd(const mdt_linear_bwd_args& a, hipStream_t s) {
    if (a.M < 1 || a.N < 1 || a.K < 1 || (a.K % 16)) return fail(MDT_ERR_INVALID_ARG, "linear_bwd: bad shape");
    if (a.dbias) LAUNCH(mdt_launch_colsum(a.dY, a.ldy, a.M, a.N, a.dbias, a.accumulate_dw, s));
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
            if (Mp != Ms) {
                HIP_TRY(hipMemsetAsync(dYt, 0, (size_t)a.N * Mp * sizeof(float), s));
                HIP_TRY(hipMemsetAsync(Xt, 0, (size_t)a.K * Mp * sizeof(float), s));
            }
            LAUNCH(mdt_launch_transpose_ld(a.dY + (int64_t)m0 * a.ldy, a.ldy, dYt, Mp, Ms, a.N, s));
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
