// mdt_resampler.hip -- host side of the Perceiver resampler handle (include/mdt_resampler.h): packed parameter
// arena, workspace, and the launch sequence of PerceiverResampler.forward.
//
// Reference replaced: mdt/models/networks/transformers/perceiver_resampler.py:85-162 (+ :11-82 attention layer,
// utils.py:16-29 feed-forward).  Per layer the reference runs  LN(media) , LN(latents) , to_q , cat , to_k , to_v ,
// softmax(q k^T) v , to_out , + , LN , Linear , GELU , Linear , + ; here:
//   k|v of the media tokens : ONE GEMM, norm_media fused into its prologue, rows scattered to (b, f) of the K|V
//                             buffer (99 % of the module's FLOPs: B*T*n rows x dim x 2*inner on fp32 MFMA)
//   k|v and q of the latents: two small GEMMs with norm_latents fused, K|V rows land behind the media rows, so
//                             the reference's torch.cat never materialises
//   attention               : k_attn_long (few queries over ~400 keys)
//   to_out / feed-forward   : GEMMs with fused residual, LayerNorm prologue and GELU epilogue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "mdt_internal.h"

#define fail mdt_fail

namespace {

struct RSlot {
    std::string name;
    int64_t numel = 0;
    bool pack = false;
    float* dst = nullptr;
    int rows = 0, K = 0, n_off = 0;
    bool loaded = false;
};

struct RLayer {
    float *nm_w, *nm_b, *nl_w, *nl_b, *ff_w, *ff_b;
    Lin kv, q, o, f1, f2;
};

}  // namespace

struct mdt_resampler {
    mdt_resampler_config cfg;
    int D, inner, ff, Q, H, hd;
    float* arena = nullptr;
    size_t arena_floats = 0;
    std::vector<RSlot> slots;
    std::vector<RLayer> layers;
    float *latents = nullptr, *tpe = nullptr, *norm_w = nullptr, *norm_b = nullptr;
    float* staging = nullptr;
    size_t staging_floats = 0;
    // workspace for up to cap_rows media rows (B*T*n) and cap_b samples
    float* ws = nullptr;
    int64_t cap_rows = 0, cap_b = 0;
    float *xf, *kv, *x, *qb, *att, *hid;
};

static void build(mdt_resampler* r, Bump& b, bool fill) {
    const int D = r->D, inner = r->inner, ff = r->ff;
    auto raw = [&](float*& p, const std::string& name, int64_t n) {
        p = b.take(n);
        if (!fill) return;
        RSlot s;
        s.name = name; s.numel = n; s.pack = false; s.dst = p;
        r->slots.push_back(s);
    };
    auto lin = [&](Lin& l, int N, int K) {
        l.N = N; l.K = K; l.bias = nullptr;
        l.wp = b.take((size_t)N * K);
    };
    auto part = [&](Lin& l, const std::string& name, int rows, int n_off) {
        if (!fill) return;
        RSlot s;
        s.name = name; s.numel = (int64_t)rows * l.K; s.pack = true; s.dst = l.wp; s.rows = rows; s.K = l.K; s.n_off = n_off;
        r->slots.push_back(s);
    };
    raw(r->latents, "latents", (int64_t)r->Q * D);
    raw(r->tpe, "time_pos_emb", (int64_t)r->cfg.num_time_embeds * D);
    if (!fill) r->layers.assign(r->cfg.depth, RLayer());
    for (int i = 0; i < r->cfg.depth; ++i) {
        RLayer& L = r->layers[i];
        const std::string a = "layers." + std::to_string(i) + ".0.", f = "layers." + std::to_string(i) + ".1.";
        raw(L.nm_w, a + "norm_media.weight", D);
        raw(L.nm_b, a + "norm_media.bias", D);
        raw(L.nl_w, a + "norm_latents.weight", D);
        raw(L.nl_b, a + "norm_latents.bias", D);
        lin(L.q, inner, D);
        part(L.q, a + "to_q.weight", inner, 0);
        lin(L.kv, 2 * inner, D);  // rows: to_k | to_v
        part(L.kv, a + "to_k.weight", inner, 0);
        part(L.kv, a + "to_v.weight", inner, inner);
        lin(L.o, D, inner);
        part(L.o, a + "to_out.weight", D, 0);
        raw(L.ff_w, f + "0.weight", D);
        raw(L.ff_b, f + "0.bias", D);
        lin(L.f1, ff, D);
        part(L.f1, f + "1.weight", ff, 0);
        lin(L.f2, D, ff);
        part(L.f2, f + "3.weight", D, 0);
    }
    raw(r->norm_w, "norm.weight", D);
    raw(r->norm_b, "norm.bias", D);
}

extern "C" mdt_status mdt_resampler_create(const mdt_resampler_config* cfg, mdt_resampler** out) {
    if (!cfg || !out) return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_create: null argument");
    const mdt_resampler_config& c = *cfg;
    if (c.dim <= 0 || c.dim % 16 || c.dim > 512)
        return fail(MDT_ERR_UNSUPPORTED, "resampler dim %d: need a multiple of 16, <= 512", c.dim);
    if (c.depth < 1 || c.heads < 1) return fail(MDT_ERR_INVALID_ARG, "resampler: depth and heads must be >= 1");
    if (c.dim_head != 16 && c.dim_head != 32 && c.dim_head != 64)
        return fail(MDT_ERR_UNSUPPORTED, "resampler dim_head %d: supported 16/32/64", c.dim_head);
    if (c.num_latents < 1 || c.num_latents > 16)
        return fail(MDT_ERR_UNSUPPORTED, "resampler num_latents %d: supported 1..16", c.num_latents);
    if (c.num_time_embeds < 1) return fail(MDT_ERR_INVALID_ARG, "resampler: num_time_embeds must be >= 1");
    if (c.ff_mult < 1 || (c.ff_mult * c.dim) % 16) return fail(MDT_ERR_INVALID_ARG, "resampler: bad ff_mult");
    if (c.activation != 0) return fail(MDT_ERR_UNSUPPORTED, "resampler activation: only 'gelu' is implemented");
    mdt_resampler* r = new mdt_resampler();
    r->cfg = c;
    r->D = c.dim; r->H = c.heads; r->hd = c.dim_head; r->inner = c.heads * c.dim_head; r->ff = c.ff_mult * c.dim;
    r->Q = c.num_latents;
    Bump count;
    build(r, count, false);
    r->arena_floats = count.off;
    hipError_t e = hipMalloc((void**)&r->arena, r->arena_floats * sizeof(float));
    if (e != hipSuccess) { delete r; return fail(MDT_ERR_HIP, "hipMalloc(resampler arena) failed: %s", hipGetErrorString(e)); }
    Bump real;
    real.base = r->arena;
    build(r, real, true);
    size_t mx = 0;
    for (const RSlot& s : r->slots) mx = std::max(mx, (size_t)s.numel);
    r->staging_floats = mx;
    e = hipMalloc((void**)&r->staging, mx * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(r->arena); delete r; return fail(MDT_ERR_HIP, "hipMalloc(staging) failed: %s", hipGetErrorString(e)); }
    *out = r;
    return MDT_OK;
}

extern "C" mdt_status mdt_resampler_destroy(mdt_resampler* r) {
    if (!r) return MDT_OK;
    (void)hipDeviceSynchronize();
    (void)hipFree(r->arena);
    (void)hipFree(r->staging);
    (void)hipFree(r->ws);
    delete r;
    return MDT_OK;
}

extern "C" int64_t mdt_resampler_param_count(const mdt_resampler* r) { return r ? (int64_t)r->slots.size() : 0; }
extern "C" const char* mdt_resampler_param_name(const mdt_resampler* r, int64_t i) {
    return (r && i >= 0 && i < (int64_t)r->slots.size()) ? r->slots[i].name.c_str() : nullptr;
}
extern "C" int64_t mdt_resampler_param_numel(const mdt_resampler* r, int64_t i) {
    return (r && i >= 0 && i < (int64_t)r->slots.size()) ? r->slots[i].numel : -1;
}

extern "C" mdt_status mdt_resampler_load_param(mdt_resampler* r, const char* name, const float* src, int64_t numel,
                                               void* stream) {
    if (!r || !name || !src) return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_load_param: null argument");
    hipStream_t s = (hipStream_t)stream;
    RSlot* slot = nullptr;
    for (RSlot& c : r->slots)
        if (c.name == name) { slot = &c; break; }
    if (!slot) return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_load_param: unknown parameter '%s'", name);
    if (numel != slot->numel)
        return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_load_param: '%s' has %lld elements, expected %lld", name,
                    (long long)numel, (long long)slot->numel);
    if (!slot->pack) {
        HIP_TRY(hipMemcpyAsync(slot->dst, src, numel * sizeof(float), hipMemcpyDefault, s));
    } else {
        const float* dev = src;
        hipPointerAttribute_t attr;
        hipError_t pe = hipPointerGetAttributes(&attr, src);
        if (!(pe == hipSuccess && attr.type == hipMemoryTypeDevice)) {
            (void)hipGetLastError();  // unregistered host memory reports an error: clear it
            HIP_TRY(hipMemcpyAsync(r->staging, src, numel * sizeof(float), hipMemcpyHostToDevice, s));
            dev = r->staging;
        }
        LAUNCH(mdt_launch_pack_weight(dev, slot->rows, slot->K, slot->dst, slot->n_off, s));
        if (dev == r->staging) HIP_TRY(hipStreamSynchronize(s));  // the staging buffer is reused by the next upload
    }
    slot->loaded = true;
    return MDT_OK;
}

static void carve(mdt_resampler* r, Bump& b, int64_t rows, int64_t B) {
    const int64_t kvrows = rows + B * r->Q, lr = B * r->Q;
    r->xf = b.take(rows * r->D);
    r->kv = b.take(kvrows * 2 * r->inner);
    r->x = b.take(lr * r->D);
    r->qb = b.take(lr * r->inner);
    r->att = b.take(lr * r->inner);
    r->hid = b.take(lr * r->ff);
}

static mdt_status reserve(mdt_resampler* r, int64_t rows, int64_t B) {
    if (rows <= r->cap_rows && B <= r->cap_b) return MDT_OK;
    rows = std::max(rows, r->cap_rows);
    B = std::max(B, r->cap_b);
    if (r->ws) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(r->ws));
        r->ws = nullptr;
        r->cap_rows = r->cap_b = 0;
    }
    Bump count;
    carve(r, count, rows, B);
    HIP_TRY(hipMalloc((void**)&r->ws, count.off * sizeof(float)));
    Bump real;
    real.base = r->ws;
    carve(r, real, rows, B);
    r->cap_rows = rows;
    r->cap_b = B;
    return MDT_OK;
}

extern "C" mdt_status mdt_resampler_forward(mdt_resampler* r, const float* x_f, const uint8_t* mask, int64_t batch,
                                            int32_t n_frames, int32_t n_tokens, float* out, void* stream) {
    if (!r || !x_f || !out || batch < 1 || n_frames < 1 || n_tokens < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_forward: bad argument");
    if (n_frames > r->cfg.num_time_embeds)
        return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_forward: %d frames but only %d time embeddings", n_frames,
                    r->cfg.num_time_embeds);
    if (((uintptr_t)x_f & 15) || ((uintptr_t)out & 15))
        return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_forward: pointers must be 16-byte aligned");
    for (const RSlot& sl : r->slots)
        if (!sl.loaded) return fail(MDT_ERR_NOT_LOADED, "resampler parameter '%s' was never loaded", sl.name.c_str());
    const int D = r->D, inner = r->inner, Q = r->Q;
    const int F = n_frames * n_tokens, Tk = F + Q;
    if (!mdt_attention_long_supported(r->hd, Q, Tk))
        return fail(MDT_ERR_UNSUPPORTED, "resampler: %d keys x %d latents exceeds the attention kernel's LDS budget", Tk, Q);
    const int64_t rows = batch * F, lr = batch * Q;
    if (rows > ((int64_t)1 << 30) / D) return fail(MDT_ERR_INVALID_ARG, "resampler: batch too large");
    hipStream_t s = (hipStream_t)stream;
    MDT_TRY(reserve(r, rows, batch));
    // x_f + time_pos_emb (masked per frame), frames flattened; latents repeated over the batch      (:141-154)
    LAUNCH(mdt_launch_add_time_emb(x_f, r->tpe, mask, r->xf, batch, n_frames, n_tokens, D, s));
    LAUNCH(mdt_launch_bcast_rows(r->latents, r->x, batch, Q, D, s));
    const float scale = 1.0f / sqrtf((float)r->hd);
    for (const RLayer& L : r->layers) {
        // k|v = to_k|to_v(cat(norm_media(x_f), norm_latents(x))): media rows, then the latent rows behind them
        mdt_gemm_args a = gemm_args(r->xf, D, L.kv, r->kv, 2 * inner, (int)rows);
        a.ln = 1; a.ln_w = L.nm_w; a.ln_b = L.nm_b;
        a.gin = F; a.gout = Tk; a.goff = 0;
        LAUNCH(mdt_launch_gemm(a, s));
        mdt_gemm_args b = gemm_args(r->x, D, L.kv, r->kv, 2 * inner, (int)lr);
        b.ln = 1; b.ln_w = L.nl_w; b.ln_b = L.nl_b;
        b.gin = Q; b.gout = Tk; b.goff = F;
        LAUNCH(mdt_launch_gemm(b, s));
        mdt_gemm_args q = gemm_args(r->x, D, L.q, r->qb, inner, (int)lr);
        q.ln = 1; q.ln_w = L.nl_w; q.ln_b = L.nl_b;
        LAUNCH(mdt_launch_gemm(q, s));
        LAUNCH(mdt_launch_attention_long(r->qb, inner, r->kv, r->kv + inner, 2 * inner, r->att, inner, (int)batch, r->H,
                                         r->hd, Q, Tk, scale, s));
        mdt_gemm_args o = gemm_args(r->att, inner, L.o, r->x, D, (int)lr);
        o.residual = 1;
        LAUNCH(mdt_launch_gemm(o, s));
        mdt_gemm_args f1 = gemm_args(r->x, D, L.f1, r->hid, r->ff, (int)lr);
        f1.ln = 1; f1.ln_w = L.ff_w; f1.ln_b = L.ff_b; f1.act = MDT_ACT_GELU;
        LAUNCH(mdt_launch_gemm(f1, s));
        mdt_gemm_args f2 = gemm_args(r->hid, r->ff, L.f2, r->x, D, (int)lr);
        f2.residual = 1;
        LAUNCH(mdt_launch_gemm(f2, s));
    }
    LAUNCH(mdt_launch_layernorm(r->x, r->norm_w, r->norm_b, out, (int)lr, D, s));
    return MDT_OK;
}

extern "C" double mdt_resampler_flops(const mdt_resampler* r, int32_t n_frames, int32_t n_tokens) {
    if (!r) return 0.0;
    const double D = r->D, I = r->inner, Q = r->Q, F = (double)n_frames * n_tokens, FF = r->ff;
    const double layer = 2.0 * (F + Q) * D * 2 * I + 2.0 * Q * D * I + 2.0 * 2.0 * Q * (F + Q) * I + 2.0 * Q * I * D +
                         2.0 * 2.0 * Q * D * FF;
    return r->cfg.depth * layer;
}
