// mdt_map_pool.hip -- the attention-pooling head of the contrastive (CLA) auxiliary loss (include/mdt_map_pool.h):
// MAPBlock = ClipStyleProjection('map').latent_proj, forward and backward.
//
// Reference replaced: mdt/models/networks/transformers/transformer_blocks.py:746-791 (MAPBlock), :716-743
// (MAPAttention), :42-62 (RMSNorm, SwishGLU); called on latent_encoder_emb of both goal modalities by
// MDTVAgent.compute_contrastive_loss (mdtv_agent.py:440-484).  The reference's forward
//   latents = rms(latents + proj(attn(q(latents), kv(projection(x)))))        seed vectors attend over the tokens
//   latents = rms(latents + mlp.1(swish_glu(mlp.0.project(latents))))
// runs here as the fp32-MFMA GEMMs of the denoiser (projection, kv, q, proj, mlp.0, mlp.1: bias / residual fused in
// their epilogues) plus three small VALU kernels: seed-vector attention over <= 16 tokens (one workgroup per sample,
// q / k / v of the sample in LDS), RMSNorm, SwishGLU.  The backward mirrors resampler / denoiser training: every dense
// contraction through mdt_linear_bwd, the row-local pieces as VALU kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "mdt_device.h"
#include "mdt_internal.h"
#include "mdt_map_pool.h"

#define fail mdt_fail

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// MAPAttention core (transformer_blocks.py:733-740): per sample, Q seed queries (shared by all samples) over N tokens,
// H heads of hd channels.  LDS: q (Q, D) | k|v (N, 2D) | scores (Q, H, N).  probs (B, Q, H, N) is kept for training.
__global__ __launch_bounds__(256) void k_map_attn_fwd(const float* __restrict__ q, const float* __restrict__ kv,
                                                      float* __restrict__ out, float* __restrict__ probs, int Q, int N, int D,
                                                      int H, float scale) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, hd = D / H;
    float* qs = sm;
    float* kvs = qs + Q * D;
    float* sc = kvs + N * 2 * D;
    for (int i = tid; i < Q * D; i += 256) qs[i] = q[i];
    for (int i = tid; i < N * 2 * D; i += 256) kvs[i] = kv[(int64_t)b * N * 2 * D + i];
    __syncthreads();
    for (int i = tid; i < Q * H * N; i += 256) {
        const int n = i % N, h = (i / N) % H, qi = i / (N * H);
        const float* qp = qs + qi * D + h * hd;
        const float* kp = kvs + n * 2 * D + h * hd;
        float s = 0.f;
        for (int c = 0; c < hd; ++c) s = fmaf(qp[c], kp[c] * scale, s);  // the reference scales k, then q k^T
        sc[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < Q * H; i += 256) {
        float* row = sc + i * N;
        float mx = row[0];
        for (int n = 1; n < N; ++n) mx = fmaxf(mx, row[n]);
        float sum = 0.f;
        for (int n = 0; n < N; ++n) { row[n] = expf(row[n] - mx); sum += row[n]; }
        const float inv = 1.0f / sum;
        for (int n = 0; n < N; ++n) {
            row[n] *= inv;
            if (probs) probs[((int64_t)b * Q * H + i) * N + n] = row[n];
        }
    }
    __syncthreads();
    for (int i = tid; i < Q * D; i += 256) {
        const int c = i % D, qi = i / D, h = c / hd;
        const float* row = sc + (qi * H + h) * N;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc = fmaf(row[n], kvs[n * 2 * D + D + c], acc);
        out[((int64_t)b * Q + qi) * D + c] = acc;
    }
}

// backward of k_map_attn_fwd.  dq_part (B, Q*D): per-sample partial of the shared queries' gradient.
__global__ __launch_bounds__(256) void k_map_attn_bwd(const float* __restrict__ q, const float* __restrict__ kv,
                                                      const float* __restrict__ probs, const float* __restrict__ d_out,
                                                      float* __restrict__ dq_part, float* __restrict__ d_kv, int Q, int N, int D,
                                                      int H, float scale) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, hd = D / H;
    float* qs = sm;
    float* kvs = qs + Q * D;
    float* dos = kvs + N * 2 * D;
    float* P = dos + Q * D;
    float* dS = P + Q * H * N;
    for (int i = tid; i < Q * D; i += 256) { qs[i] = q[i]; dos[i] = d_out[(int64_t)b * Q * D + i]; }
    for (int i = tid; i < N * 2 * D; i += 256) kvs[i] = kv[(int64_t)b * N * 2 * D + i];
    for (int i = tid; i < Q * H * N; i += 256) P[i] = probs[(int64_t)b * Q * H * N + i];
    __syncthreads();
    for (int i = tid; i < Q * H * N; i += 256) {  // dP = dO . v
        const int n = i % N, h = (i / N) % H, qi = i / (N * H);
        const float* dp = dos + qi * D + h * hd;
        const float* vp = kvs + n * 2 * D + D + h * hd;
        float s = 0.f;
        for (int c = 0; c < hd; ++c) s = fmaf(dp[c], vp[c], s);
        dS[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < Q * H; i += 256) {  // dS = P * (dP - sum_n P dP)
        float dot = 0.f;
        for (int n = 0; n < N; ++n) dot = fmaf(P[i * N + n], dS[i * N + n], dot);
        for (int n = 0; n < N; ++n) dS[i * N + n] = P[i * N + n] * (dS[i * N + n] - dot);
    }
    __syncthreads();
    for (int i = tid; i < Q * D; i += 256) {  // dq = scale * sum_n dS k
        const int c = i % D, qi = i / D, h = c / hd;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc = fmaf(dS[(qi * H + h) * N + n], kvs[n * 2 * D + c], acc);
        dq_part[(int64_t)b * Q * D + i] = acc * scale;
    }
    for (int i = tid; i < N * D; i += 256) {  // dk = scale * sum_q dS q ; dv = sum_q P dO
        const int c = i % D, n = i / D, h = c / hd;
        float ak = 0.f, av = 0.f;
        for (int qi = 0; qi < Q; ++qi) {
            ak = fmaf(dS[(qi * H + h) * N + n], qs[qi * D + c], ak);
            av = fmaf(P[(qi * H + h) * N + n], dos[qi * D + c], av);
        }
        d_kv[((int64_t)b * N + n) * 2 * D + c] = ak * scale;
        d_kv[((int64_t)b * N + n) * 2 * D + D + c] = av;
    }
}

// RMSNorm (transformer_blocks.py:43-51): y = x / max(||x|| * D^-1/2, eps) * g ; one wave per row, D <= 512
#define RMS_MAXC 8
__global__ __launch_bounds__(256) void k_rms_fwd(const float* __restrict__ x, const float* __restrict__ g,
                                                 float* __restrict__ out, int M, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float v[RMS_MAXC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? x[(int64_t)row * D + c] : 0.f;
        ss = fmaf(v[i], v[i], ss);
    }
    const float nrm = fmaxf(sqrtf(wave_sum(ss)) * rsqrtf((float)D), eps);
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < D) out[(int64_t)row * D + c] = v[i] / nrm * g[c];
    }
}

// dx (+)= g dy / n - x * (sum_j g_j dy_j x_j) / (D n^3)   (clamped rows: dx = g dy / eps);  pg: per-workgroup partial
// (4 rows) of dg = sum_rows dy x / n, reduced by k_colsum afterwards.  res != nullptr: dx = res + that (the gradient that
// reaches x past the norm, on the residual path: autograd's separate add of the two becomes one more read here)
__global__ __launch_bounds__(256) void k_rms_bwd(const float* __restrict__ x, const float* __restrict__ g,
                                                 const float* __restrict__ dy, const float* __restrict__ res, float* __restrict__ dx,
                                                 int accumulate, float* __restrict__ pg, int M, int D, float eps) {
    __shared__ float red[4][64 * RMS_MAXC];
    const int wv = threadIdx.x >> 6, row = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    float dgv[RMS_MAXC];
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) dgv[i] = 0.f;
    if (row < M) {
        float xv[RMS_MAXC], gd[RMS_MAXC], rv[RMS_MAXC];
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < RMS_MAXC; ++i) {
            const int c = lane + 64 * i;
            xv[i] = c < D ? x[(int64_t)row * D + c] : 0.f;
            rv[i] = (res != nullptr && c < D) ? res[(int64_t)row * D + c] : 0.f;
            const float d = c < D ? dy[(int64_t)row * D + c] : 0.f;
            gd[i] = c < D ? g[c] * d : 0.f;
            ss = fmaf(xv[i], xv[i], ss);
            dot = fmaf(gd[i], xv[i], dot);
            dgv[i] = d * xv[i];
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float raw = sqrtf(ss) * rsqrtf((float)D);
        const bool clamped = raw < eps;
        const float n = clamped ? eps : raw;
        const float k = clamped ? 0.f : dot / ((float)D * n * n * n);
#pragma unroll
        for (int i = 0; i < RMS_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float v = gd[i] / n - xv[i] * k + rv[i];
                float* p = dx + (int64_t)row * D + c;
                *p = accumulate ? *p + v : v;
            }
            dgv[i] /= n;
        }
    }
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) red[wv][lane + 64 * i] = dgv[i];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) pg[(int64_t)blockIdx.x * D + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// SwishGLU (transformer_blocks.py:55-62): u = [projected | gate] (M, 2H) -> projected * silu(gate)
__global__ void k_swiglu_fwd(const float* __restrict__ u, float* __restrict__ out, int64_t n, int Hm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t m = i / Hm;
    const int c = (int)(i - m * Hm);
    out[i] = u[m * 2 * Hm + c] * act_silu(u[m * 2 * Hm + Hm + c]);
}
__global__ void k_swiglu_bwd(const float* __restrict__ u, const float* __restrict__ d_out, float* __restrict__ du, int64_t n,
                             int Hm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t m = i / Hm;
    const int c = (int)(i - m * Hm);
    const float p = u[m * 2 * Hm + c], gt = u[m * 2 * Hm + Hm + c], d = d_out[i];
    du[m * 2 * Hm + c] = d * act_silu(gt);
    du[m * 2 * Hm + Hm + c] = d * p * act_silu_grad(gt);
}

// launchers shared with the masked-image decoder's op-level ABI (mdt_mae.hip)
hipError_t mdt_launch_rms_fwd(const float* x, const float* g, float* out, int64_t M, int D, float eps, hipStream_t s) {
    if (D > 64 * RMS_MAXC) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_rms_fwd, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, g, out, (int)M, D, eps);
    return hipGetLastError();
}
// pg: ceil(M / 4) x D floats of scratch (per-workgroup partials of dg; reduce with mdt_launch_colsum)
hipError_t mdt_launch_rms_bwd(const float* x, const float* g, const float* dy, float* dx, int accumulate, float* pg, int64_t M,
                              int D, float eps, hipStream_t s, const float* res) {
    if (D > 64 * RMS_MAXC) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_rms_bwd, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, g, dy, res, dx, accumulate, pg, (int)M, D, eps);
    return hipGetLastError();
}
hipError_t mdt_launch_swiglu_fwd(const float* u, float* out, int64_t M, int Hm, hipStream_t s) {
    const int64_t n = M * Hm;
    hipLaunchKernelGGL(k_swiglu_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, out, n, Hm);
    return hipGetLastError();
}
hipError_t mdt_launch_swiglu_bwd(const float* u, const float* d_out, float* du, int64_t M, int Hm, hipStream_t s) {
    const int64_t n = M * Hm;
    hipLaunchKernelGGL(k_swiglu_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, d_out, du, n, Hm);
    return hipGetLastError();
}

namespace {

const float RMS_EPS = 1e-8f;

struct MSlot {
    std::string name;
    int64_t numel = 0;
    bool pack = false;
    float* dst = nullptr;
    int rows = 0, K = 0;
    bool loaded = false;
    Lin* lin = nullptr;
};

struct MTape {
    bool in_use = false;
    int64_t B = 0, cap = 0;
    int N = 0, cap_n = 0;
    float* buf = nullptr;
    float *x, *xp, *kv, *probs, *att, *x1, *lat1, *u, *hid, *x2;
};

}  // namespace

struct mdt_map_pool {
    mdt_map_pool_config cfg;
    int Din, D, H, hd, Hm, Q;
    float* arena = nullptr;
    std::vector<MSlot> slots;
    float *latents, *an_g, *mn_g;
    Lin projection, q, kv, aproj, m0, m1;
    float* staging = nullptr;
    float* qv = nullptr;  // q(latents): (Q, D), recomputed per forward (the weights may have changed)
    // inference workspace
    float* ws = nullptr;
    int64_t cap_b = 0;
    int cap_n = 0;
    float *xp, *kvb, *att, *x1, *lat1, *u, *hid, *x2;
    // training
    float* wt_arena = nullptr;
    std::vector<int64_t> grad_off;
    int64_t grad_numel = 0;
    std::vector<MTape> tapes;
    float* tscratch = nullptr;
    int64_t ts_b = 0;
    int ts_n = 0;
    float *g_a, *g_b, *g_u, *g_hid, *g_dq, *g_dqs, *g_dkv, *g_dxp, *g_pg, *g_lin;
};

static void build(mdt_map_pool* p, Bump& b, bool fill) {
    auto raw = [&](float*& dst, const std::string& name, int64_t n) {
        dst = b.take(n);
        if (!fill) return;
        MSlot s;
        s.name = name; s.numel = n; s.dst = dst;
        p->slots.push_back(s);
    };
    auto lin = [&](Lin& l, const std::string& name, int N, int K, bool bias) {
        l.N = N; l.K = K;
        l.wp = b.take((size_t)N * K);
        if (fill) {
            MSlot s;
            s.name = name + ".weight"; s.numel = (int64_t)N * K; s.pack = true; s.dst = l.wp; s.rows = N; s.K = K; s.lin = &l;
            p->slots.push_back(s);
        }
        if (bias) raw(l.bias, name + ".bias", N);
        else l.bias = nullptr;
    };
    // registration order of the reference module (its own Parameter first, then the sub-modules)
    raw(p->latents, "latents", (int64_t)p->Q * p->D);
    lin(p->projection, "projection", p->D, p->Din, true);
    raw(p->an_g, "attn_norm.g", p->D);
    lin(p->q, "attn.q", p->D, p->D, false);
    lin(p->kv, "attn.kv", 2 * p->D, p->D, false);
    lin(p->aproj, "attn.proj", p->D, p->D, true);
    raw(p->mn_g, "mlp_norm.g", p->D);
    lin(p->m0, "mlp.0.project", 2 * p->Hm, p->D, true);
    lin(p->m1, "mlp.1", p->D, p->Hm, true);
    p->qv = b.take((size_t)p->Q * p->D);
}

static size_t attn_lds_floats(const mdt_map_pool* p, int N, bool bwd) {
    const size_t base = (size_t)p->Q * p->D + (size_t)N * 2 * p->D + (size_t)p->Q * p->H * N;
    return bwd ? base + (size_t)p->Q * p->D + (size_t)p->Q * p->H * N : base;
}

extern "C" mdt_status mdt_map_pool_create(const mdt_map_pool_config* cfg, mdt_map_pool** out) {
    if (!cfg || !out) return fail(MDT_ERR_INVALID_ARG, "mdt_map_pool_create: null argument");
    const mdt_map_pool_config& c = *cfg;
    if (c.n_latents < 1 || c.n_latents > 16) return fail(MDT_ERR_UNSUPPORTED, "map pool n_latents %d: supported 1..16", c.n_latents);
    if (c.embed_dim <= 0 || c.embed_dim % 16 || c.output_dim <= 0 || c.output_dim % 16 || c.output_dim > 512)
        return fail(MDT_ERR_UNSUPPORTED, "map pool: embed_dim / output_dim must be multiples of 16, output_dim <= 512");
    if (c.n_heads < 1 || c.output_dim % (2 * c.n_heads))
        return fail(MDT_ERR_INVALID_ARG, "map pool: output_dim must be divisible by 2 * n_heads (transformer_blocks.py:721,759)");
    if (c.mlp_hidden <= 0 || c.mlp_hidden % 16) return fail(MDT_ERR_UNSUPPORTED, "map pool: mlp_hidden must be a multiple of 16");
    mdt_map_pool* p = new mdt_map_pool();
    p->cfg = c;
    p->Din = c.embed_dim; p->D = c.output_dim; p->H = 2 * c.n_heads; p->hd = p->D / p->H; p->Hm = c.mlp_hidden; p->Q = c.n_latents;
    Bump count;
    build(p, count, false);
    hipError_t e = hipMalloc((void**)&p->arena, count.off * sizeof(float));
    if (e != hipSuccess) { delete p; return fail(MDT_ERR_HIP, "hipMalloc(map pool arena) failed: %s", hipGetErrorString(e)); }
    Bump real;
    real.base = p->arena;
    build(p, real, true);
    size_t mx = 0;
    for (const MSlot& s : p->slots) mx = std::max(mx, (size_t)s.numel);
    e = hipMalloc((void**)&p->staging, mx * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(p->arena); delete p; return fail(MDT_ERR_HIP, "hipMalloc(staging) failed: %s", hipGetErrorString(e)); }
    *out = p;
    return MDT_OK;
}

extern "C" mdt_status mdt_map_pool_destroy(mdt_map_pool* p) {
    if (!p) return MDT_OK;
    (void)hipDeviceSynchronize();
    for (MTape& t : p->tapes) (void)mdt_dev_free(t.buf);
    (void)mdt_dev_free(p->tscratch);
    (void)hipFree(p->wt_arena);
    (void)hipFree(p->arena);
    (void)hipFree(p->staging);
    (void)mdt_dev_free(p->ws);
    delete p;
    return MDT_OK;
}

extern "C" int64_t mdt_map_pool_param_count(const mdt_map_pool* p) { return p ? (int64_t)p->slots.size() : 0; }
extern "C" const char* mdt_map_pool_param_name(const mdt_map_pool* p, int64_t i) {
    return (p && i >= 0 && i < (int64_t)p->slots.size()) ? p->slots[i].name.c_str() : nullptr;
}
extern "C" int64_t mdt_map_pool_param_numel(const mdt_map_pool* p, int64_t i) {
    return (p && i >= 0 && i < (int64_t)p->slots.size()) ? p->slots[i].numel : -1;
}

extern "C" mdt_status mdt_map_pool_load_param(mdt_map_pool* p, const char* name, const float* src, int64_t numel,
                                              void* stream) {
    if (!p || !name || !src) return fail(MDT_ERR_INVALID_ARG, "mdt_map_pool_load_param: null argument");
    hipStream_t s = (hipStream_t)stream;
    MSlot* slot = nullptr;
    for (MSlot& c : p->slots)
        if (c.name == name) { slot = &c; break; }
    if (!slot) return fail(MDT_ERR_INVALID_ARG, "mdt_map_pool_load_param: unknown parameter '%s'", name);
    if (numel != slot->numel)
        return fail(MDT_ERR_INVALID_ARG, "mdt_map_pool_load_param: '%s' has %lld elements, expected %lld", name,
                    (long long)numel, (long long)slot->numel);
    if (!slot->pack) {
        HIP_TRY(hipMemcpyAsync(slot->dst, src, numel * sizeof(float), hipMemcpyDefault, s));
    } else {
        const float* dev = src;
        hipPointerAttribute_t attr;
        hipError_t pe = hipPointerGetAttributes(&attr, src);
        if (!(pe == hipSuccess && attr.type == hipMemoryTypeDevice)) {
            (void)hipGetLastError();  // unregistered host memory reports an error: clear it
            HIP_TRY(hipMemcpyAsync(p->staging, src, numel * sizeof(float), hipMemcpyHostToDevice, s));
            dev = p->staging;
        }
        LAUNCH(mdt_launch_pack_weight(dev, slot->rows, slot->K, slot->dst, 0, s));
        if (slot->lin->wt)  // training: image of W^T for dX = dY W
            LAUNCH(mdt_launch_pack_weight_t(dev, slot->rows, slot->K, slot->K, slot->lin->wt, 0, slot->lin->N / 16, s));
        if (dev == p->staging) HIP_TRY(hipStreamSynchronize(s));  // the staging buffer is reused by the next upload
    }
    slot->loaded = true;
    return MDT_OK;
}

static mdt_status check_call(const mdt_map_pool* p, const float* x, const float* out, int64_t batch, int n_tokens, bool bwd) {
    if (!p || !x || !out || batch < 1) return fail(MDT_ERR_INVALID_ARG, "map pool: bad argument");
    if (n_tokens < 1 || n_tokens > 16) return fail(MDT_ERR_UNSUPPORTED, "map pool: 1..16 tokens per sample, got %d", n_tokens);
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return fail(MDT_ERR_INVALID_ARG, "map pool: pointers must be 16-byte aligned");
    if (batch * n_tokens > ((int64_t)1 << 24)) return fail(MDT_ERR_INVALID_ARG, "map pool: batch too large");
    if (attn_lds_floats(p, n_tokens, bwd) * sizeof(float) > 64 * 1024)
        return fail(MDT_ERR_UNSUPPORTED, "map pool: %d latents x %d tokens x %d channels exceed the attention kernel's LDS budget",
                    p->Q, n_tokens, p->D);
    for (const MSlot& sl : p->slots)
        if (!sl.loaded) return fail(MDT_ERR_NOT_LOADED, "map pool parameter '%s' was never loaded", sl.name.c_str());
    return MDT_OK;
}

struct MBuf { float *xp, *kv, *probs, *att, *x1, *lat1, *u, *hid, *x2; };

// the launch sequence shared by inference and the taped forward
static mdt_status run_forward(mdt_map_pool* p, const float* x, int64_t B, int N, const MBuf& w, float* out, hipStream_t s) {
    const int D = p->D, Q = p->Q, Hm = p->Hm;
    const int64_t rows = B * N, lr = B * Q;
    LAUNCH(mdt_launch_gemm(gemm_args(x, p->Din, p->projection, w.xp, D, (int)rows), s));
    LAUNCH(mdt_launch_gemm(gemm_args(w.xp, D, p->kv, w.kv, 2 * D, (int)rows), s));
    LAUNCH(mdt_launch_gemm(gemm_args(p->latents, D, p->q, p->qv, D, Q), s));
    const float scale = 1.0f / sqrtf((float)p->hd);
    hipLaunchKernelGGL(k_map_attn_fwd, dim3((unsigned)B), dim3(256), attn_lds_floats(p, N, false) * sizeof(float), s, p->qv, w.kv,
                       w.att, w.probs, Q, N, D, p->H, scale);
    LAUNCH(hipGetLastError());
    // x1 = latents + proj(att)
    LAUNCH(mdt_launch_bcast_rows(p->latents, w.x1, B, Q, D, s));
    mdt_gemm_args o = gemm_args(w.att, D, p->aproj, w.x1, D, (int)lr);
    o.residual = 1;
    LAUNCH(mdt_launch_gemm(o, s));
    hipLaunchKernelGGL(k_rms_fwd, dim3((unsigned)((lr + 3) / 4)), dim3(256), 0, s, w.x1, p->an_g, w.lat1, (int)lr, D, RMS_EPS);
    LAUNCH(hipGetLastError());
    // x2 = lat1 + mlp.1(swish_glu(mlp.0.project(lat1)))
    LAUNCH(mdt_launch_gemm(gemm_args(w.lat1, D, p->m0, w.u, 2 * Hm, (int)lr), s));
    const int64_t nh = lr * Hm;
    hipLaunchKernelGGL(k_swiglu_fwd, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, w.u, w.hid, nh, Hm);
    LAUNCH(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(w.x2, w.lat1, (size_t)lr * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    mdt_gemm_args f = gemm_args(w.hid, Hm, p->m1, w.x2, D, (int)lr);
    f.residual = 1;
    LAUNCH(mdt_launch_gemm(f, s));
    hipLaunchKernelGGL(k_rms_fwd, dim3((unsigned)((lr + 3) / 4)), dim3(256), 0, s, w.x2, p->mn_g, out, (int)lr, D, RMS_EPS);
    LAUNCH(hipGetLastError());
    return MDT_OK;
}

static void carve_ws(mdt_map_pool* p, Bump& b, int64_t B, int N) {
    const int64_t rows = B * N, lr = B * p->Q;
    p->xp = b.take(rows * p->D); p->kvb = b.take(rows * 2 * p->D); p->att = b.take(lr * p->D); p->x1 = b.take(lr * p->D);
    p->lat1 = b.take(lr * p->D); p->u = b.take(lr * 2 * p->Hm); p->hid = b.take(lr * p->Hm); p->x2 = b.take(lr * p->D);
}

extern "C" mdt_status mdt_map_pool_forward(mdt_map_pool* p, const float* x, int64_t batch, int32_t n_tokens, float* out,
                                           void* stream) {
    MDT_TRY(check_call(p, x, out, batch, n_tokens, false));
    if (batch > p->cap_b || n_tokens > p->cap_n) {
        const int64_t B = std::max(batch, p->cap_b);
        const int N = std::max((int)n_tokens, p->cap_n);
        if (p->ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(mdt_dev_free(p->ws)); p->ws = nullptr; p->cap_b = 0; p->cap_n = 0; }
        Bump count;
        carve_ws(p, count, B, N);
        HIP_TRY(mdt_dev_malloc((void**)&p->ws, count.off * sizeof(float)));
        p->cap_b = B; p->cap_n = N;
    }
    Bump real;
    real.base = p->ws;
    carve_ws(p, real, p->cap_b, p->cap_n);
    MBuf w = {p->xp, p->kvb, nullptr, p->att, p->x1, p->lat1, p->u, p->hid, p->x2};
    return run_forward(p, x, batch, n_tokens, w, out, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// training
// ------------------------------------------------------------------------------------------------
extern "C" mdt_status mdt_map_pool_train_prepare(mdt_map_pool* p) {
    if (!p) return fail(MDT_ERR_INVALID_ARG, "mdt_map_pool_train_prepare: null handle");
    if (p->wt_arena) return MDT_OK;
    Lin* lins[] = {&p->projection, &p->q, &p->kv, &p->aproj, &p->m0, &p->m1};
    Bump count;
    for (Lin* l : lins) count.take((size_t)l->N * l->K);
    HIP_TRY(hipMalloc((void**)&p->wt_arena, count.off * sizeof(float)));
    Bump real;
    real.base = p->wt_arena;
    for (Lin* l : lins) l->wt = real.take((size_t)l->N * l->K);
    int64_t off = 0;
    p->grad_off.clear();
    for (const MSlot& sl : p->slots) { p->grad_off.push_back(off); off += (sl.numel + 3) & ~(int64_t)3; }
    p->grad_numel = off;
    for (MSlot& sl : p->slots) sl.loaded = false;
    return MDT_OK;
}

extern "C" int64_t mdt_map_pool_grad_numel(const mdt_map_pool* p) { return (p && p->wt_arena) ? p->grad_numel : -1; }
extern "C" int64_t mdt_map_pool_grad_offset(const mdt_map_pool* p, int64_t i) {
    return (p && p->wt_arena && i >= 0 && i < (int64_t)p->grad_off.size()) ? p->grad_off[i] : -1;
}

static void carve_tape(mdt_map_pool* p, Bump& b, MTape& t, int64_t B, int N) {
    const int64_t rows = B * N, lr = B * p->Q;
    const int D = p->D;
    t.x = b.take(rows * p->Din); t.xp = b.take(rows * D); t.kv = b.take(rows * 2 * D); t.probs = b.take(lr * p->H * N);
    t.att = b.take(lr * D); t.x1 = b.take(lr * D); t.lat1 = b.take(lr * D); t.u = b.take(lr * 2 * p->Hm);
    t.hid = b.take(lr * p->Hm); t.x2 = b.take(lr * D);
}

static void carve_scratch(mdt_map_pool* p, Bump& b, int64_t B, int N) {
    const int64_t rows = B * N, lr = B * p->Q;
    const int D = p->D, Hm = p->Hm;
    p->g_a = b.take(lr * D); p->g_b = b.take(lr * D); p->g_u = b.take(lr * 2 * Hm); p->g_hid = b.take(lr * Hm);
    p->g_dq = b.take(lr * D); p->g_dqs = b.take((size_t)p->Q * D); p->g_dkv = b.take(rows * 2 * D); p->g_dxp = b.take(rows * D);
    p->g_pg = b.take(((lr + 3) / 4) * D);
    int64_t need = 0;
    for (auto nk : {std::pair<int, int>(D, p->Din), std::pair<int, int>(2 * D, D)})
        need = std::max(need, mdt_linear_bwd_scratch(rows, nk.first, nk.second));
    for (auto nk : {std::pair<int, int>(D, D), std::pair<int, int>(2 * Hm, D), std::pair<int, int>(D, Hm)})
        need = std::max(need, mdt_linear_bwd_scratch(lr, nk.first, nk.second));
    need = std::max(need, mdt_linear_bwd_scratch(p->Q, D, D));
    p->g_lin = b.take(need);
}

static mdt_status reserve_scratch(mdt_map_pool* p, int64_t B, int N) {
    if (B > p->ts_b || N > p->ts_n) {
        B = std::max(B, p->ts_b); N = std::max(N, p->ts_n);
        if (p->tscratch) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(mdt_dev_free(p->tscratch)); p->tscratch = nullptr; }
        Bump count;
        carve_scratch(p, count, B, N);
        HIP_TRY(mdt_dev_malloc((void**)&p->tscratch, count.off * sizeof(float)));
        p->ts_b = B; p->ts_n = N;
    }
    Bump real;
    real.base = p->tscratch;
    carve_scratch(p, real, p->ts_b, p->ts_n);
    return MDT_OK;
}

extern "C" mdt_status mdt_map_pool_forward_train(mdt_map_pool* p, const float* x, int64_t batch, int32_t n_tokens, float* out,
                                                 int32_t* tape, void* stream) {
    if (!tape) return fail(MDT_ERR_INVALID_ARG, "mdt_map_pool_forward_train: null tape pointer");
    if (p && !p->wt_arena) return fail(MDT_ERR_STATE, "map pool training was not prepared (mdt_map_pool_train_prepare)");
    MDT_TRY(check_call(p, x, out, batch, n_tokens, true));
    hipStream_t s = (hipStream_t)stream;
    int pick = -1;
    for (size_t i = 0; i < p->tapes.size(); ++i)
        if (!p->tapes[i].in_use && p->tapes[i].cap >= batch && p->tapes[i].cap_n >= n_tokens) { pick = (int)i; break; }
    if (pick < 0)
        for (size_t i = 0; i < p->tapes.size(); ++i)
            if (!p->tapes[i].in_use) { pick = (int)i; break; }
    if (pick < 0) {
        if (p->tapes.size() >= 8) return fail(MDT_ERR_STATE, "more than 8 map pool tapes alive: release tapes after their backward");
        p->tapes.emplace_back();
        pick = (int)p->tapes.size() - 1;
    }
    MTape& t = p->tapes[pick];
    if (t.cap < batch || t.cap_n < n_tokens) {
        const int64_t B = std::max(batch, t.cap);
        const int N = std::max((int)n_tokens, t.cap_n);
        if (t.buf) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(mdt_dev_free(t.buf)); t.buf = nullptr; t.cap = 0; t.cap_n = 0; }
        Bump count;
        carve_tape(p, count, t, B, N);
        HIP_TRY(mdt_dev_malloc((void**)&t.buf, count.off * sizeof(float)));
        t.cap = B; t.cap_n = N;
    }
    Bump real;
    real.base = t.buf;
    carve_tape(p, real, t, t.cap, t.cap_n);
    t.B = batch; t.N = n_tokens;
    HIP_TRY(hipMemcpyAsync(t.x, x, (size_t)batch * n_tokens * p->Din * sizeof(float), hipMemcpyDeviceToDevice, s));
    MBuf w = {t.xp, t.kv, t.probs, t.att, t.x1, t.lat1, t.u, t.hid, t.x2};
    MDT_TRY(run_forward(p, t.x, batch, n_tokens, w, out, s));
    t.in_use = true;
    *tape = pick;
    return MDT_OK;
}

extern "C" mdt_status mdt_map_pool_tape_release(mdt_map_pool* p, int32_t tape) {
    if (!p || tape < 0 || tape >= (int)p->tapes.size() || !p->tapes[tape].in_use)
        return fail(MDT_ERR_INVALID_ARG, "invalid or released map pool tape %d", tape);
    p->tapes[tape].in_use = false;
    return MDT_OK;
}

static float* grad_of(mdt_map_pool* p, float* grads, const float* dst) {
    for (size_t i = 0; i < p->slots.size(); ++i)
        if (p->slots[i].dst == dst) return grads + p->grad_off[i];
    return nullptr;
}

static mdt_status m_lin_bwd(mdt_map_pool* p, float* grads, const Lin& l, const float* X, int64_t ldx, const float* dY, int64_t ldy,
                            int64_t M, float* dX, int64_t ldxo, int acc_dx, hipStream_t s) {
    mdt_linear_bwd_args a;
    memset(&a, 0, sizeof a);
    a.X = X; a.ldx = ldx; a.dY = dY; a.ldy = ldy;
    a.dW = grad_of(p, grads, l.wp);
    a.dbias = l.bias ? grad_of(p, grads, l.bias) : nullptr;
    a.accumulate_dw = 1; a.Wt = l.wt; a.dX = dX; a.ldxo = ldxo; a.accumulate_dx = acc_dx;
    a.M = (int)M; a.N = l.N; a.K = l.K; a.scratch = p->g_lin;
    return mdt_linear_bwd(a, s);
}

static mdt_status m_rms_bwd(mdt_map_pool* p, float* grads, const float* x, const float* g, const float* dy, float* dx, int acc,
                            int64_t M, hipStream_t s) {
    const int blocks = (int)((M + 3) / 4);
    hipLaunchKernelGGL(k_rms_bwd, dim3(blocks), dim3(256), 0, s, x, g, dy, (const float*)nullptr, dx, acc, p->g_pg, (int)M, p->D, RMS_EPS);
    LAUNCH(hipGetLastError());
    LAUNCH(mdt_launch_colsum(p->g_pg, p->D, blocks, p->D, grad_of(p, grads, g), 1, s));
    return MDT_OK;
}

extern "C" mdt_status mdt_map_pool_backward(mdt_map_pool* p, int32_t tape, const float* g_out, float* grads, float* d_x,
                                            void* stream) {
    if (!p || !g_out || !grads) return fail(MDT_ERR_INVALID_ARG, "mdt_map_pool_backward: null argument");
    if (tape < 0 || tape >= (int)p->tapes.size() || !p->tapes[tape].in_use)
        return fail(MDT_ERR_INVALID_ARG, "invalid or released map pool tape %d", tape);
    MTape& t = p->tapes[tape];
    hipStream_t s = (hipStream_t)stream;
    const int D = p->D, Q = p->Q, Hm = p->Hm, N = t.N;
    const int64_t B = t.B, rows = B * N, lr = B * Q;
    MDT_TRY(reserve_scratch(p, B, N));
    // out = rms(x2) ; x2 = lat1 + mlp.1(hid) ; hid = swish_glu(u) ; u = mlp.0.project(lat1)
    MDT_TRY(m_rms_bwd(p, grads, t.x2, p->mn_g, g_out, p->g_a, 0, lr, s));                         // g_a = d(x2) = d(lat1) so far
    MDT_TRY(m_lin_bwd(p, grads, p->m1, t.hid, Hm, p->g_a, D, lr, p->g_hid, Hm, 0, s));
    const int64_t nh = lr * Hm;
    hipLaunchKernelGGL(k_swiglu_bwd, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, t.u, p->g_hid, p->g_u, nh, Hm);
    LAUNCH(hipGetLastError());
    MDT_TRY(m_lin_bwd(p, grads, p->m0, t.lat1, D, p->g_u, 2 * Hm, lr, p->g_a, D, 1, s));          // g_a += dX
    // lat1 = rms(x1) ; x1 = latents + attn.proj(att)
    MDT_TRY(m_rms_bwd(p, grads, t.x1, p->an_g, p->g_a, p->g_b, 0, lr, s));                        // g_b = d(x1)
    float* g_lat = grad_of(p, grads, p->latents);
    LAUNCH(mdt_launch_colsum(p->g_b, (int64_t)Q * D, (int)B, Q * D, g_lat, 1, s));                 // latents were repeated
    MDT_TRY(m_lin_bwd(p, grads, p->aproj, t.att, D, p->g_b, D, lr, p->g_a, D, 0, s));             // g_a = d(att)
    const float scale = 1.0f / sqrtf((float)p->hd);
    hipLaunchKernelGGL(k_map_attn_bwd, dim3((unsigned)B), dim3(256), attn_lds_floats(p, N, true) * sizeof(float), s, p->qv, t.kv,
                       t.probs, p->g_a, p->g_dq, p->g_dkv, Q, N, D, p->H, scale);
    LAUNCH(hipGetLastError());
    // q = attn.q(latents): the same Q rows for every sample
    LAUNCH(mdt_launch_colsum(p->g_dq, (int64_t)Q * D, (int)B, Q * D, p->g_dqs, 0, s));
    MDT_TRY(m_lin_bwd(p, grads, p->q, p->latents, D, p->g_dqs, D, Q, g_lat, D, 1, s));
    // k|v = attn.kv(xp) ; xp = projection(x)
    MDT_TRY(m_lin_bwd(p, grads, p->kv, t.xp, D, p->g_dkv, 2 * D, rows, p->g_dxp, D, 0, s));
    MDT_TRY(m_lin_bwd(p, grads, p->projection, t.x, p->Din, p->g_dxp, D, rows, d_x, p->Din, 0, s));
    return MDT_OK;
}
