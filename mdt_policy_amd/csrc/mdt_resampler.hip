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
#include <utility>
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
    Lin* lin = nullptr;
};

struct RLayer {
    float *nm_w, *nm_b, *nl_w, *nl_b, *ff_w, *ff_b;
    Lin kv, q, o, f1, f2;
};

struct RLayerTape {  // rows: media B*F, latents B*Q
    float *st_m, *kv, *x_in, *st_l, *lat_n, *q, *att, *x_mid, *st_f, *h, *u, *hid, *x_out;
};

struct RTape {
    bool in_use = false;
    int64_t B = 0, cap_rows = 0, cap_b = 0;
    int T = 0, n = 0, has_mask = 0;
    float* buf = nullptr;
    float *xf, *x0, *st_out;
    uint8_t* mask;
    std::vector<RLayerTape> layers;
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
    // training
    float* wt_arena = nullptr;
    std::vector<int64_t> grad_off;
    int64_t grad_numel = 0;
    std::vector<RTape> tapes;
    float* tscratch = nullptr;
    int64_t ts_rows = 0, ts_b = 0;
    float *g_dx, *g_dxf, *g_nm, *g_dnm, *g_dkv, *g_dkvm, *g_dkvl, *g_dq, *g_datt, *g_dlat, *g_ff, *g_td, *g_pw, *g_pb, *g_lin;
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
        s.lin = &l;
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
    for (RTape& t : r->tapes) (void)mdt_dev_free(t.buf);
    (void)mdt_dev_free(r->tscratch);
    (void)hipFree(r->wt_arena);
    (void)hipFree(r->arena);
    (void)hipFree(r->staging);
    (void)mdt_dev_free(r->ws);
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
        if (slot->lin && slot->lin->wt)  // training: image of W^T for dX = dY W
            LAUNCH(mdt_launch_pack_weight_t(dev, slot->rows, slot->K, slot->K, slot->lin->wt, slot->n_off, slot->lin->N / 16, s));
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
        HIP_TRY(mdt_dev_free(r->ws));
        r->ws = nullptr;
        r->cap_rows = r->cap_b = 0;
    }
    Bump count;
    carve(r, count, rows, B);
    HIP_TRY(mdt_dev_malloc((void**)&r->ws, count.off * sizeof(float)));
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

// ------------------------------------------------------------------------------------------------
// training: forward with a tape, backward (autograd through perceiver_resampler.py:124-162)
// ------------------------------------------------------------------------------------------------
extern "C" mdt_status mdt_resampler_train_prepare(mdt_resampler* r) {
    if (!r) return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_train_prepare: null handle");
    if (r->wt_arena) return MDT_OK;
    std::vector<Lin*> lins;
    for (RSlot& sl : r->slots)
        if (sl.lin && std::find(lins.begin(), lins.end(), sl.lin) == lins.end()) lins.push_back(sl.lin);
    Bump count;
    for (Lin* l : lins) count.take((size_t)l->N * l->K);
    HIP_TRY(hipMalloc((void**)&r->wt_arena, count.off * sizeof(float)));
    Bump real;
    real.base = r->wt_arena;
    for (Lin* l : lins) l->wt = real.take((size_t)l->N * l->K);
    // gradient layout: slot order (to_k | to_v are adjacent slots, so the stacked K|V weight is contiguous)
    int64_t off = 0;
    r->grad_off.clear();
    for (const RSlot& sl : r->slots) { r->grad_off.push_back(off); off += sl.numel; }
    r->grad_numel = off;
    for (RSlot& sl : r->slots) sl.loaded = false;
    return MDT_OK;
}

extern "C" int64_t mdt_resampler_grad_numel(const mdt_resampler* r) { return (r && r->wt_arena) ? r->grad_numel : -1; }
extern "C" int64_t mdt_resampler_grad_offset(const mdt_resampler* r, int64_t i) {
    return (r && r->wt_arena && i >= 0 && i < (int64_t)r->grad_off.size()) ? r->grad_off[i] : -1;
}

static void carve_rtape(mdt_resampler* r, Bump& b, RTape& t, int64_t rows, int64_t B) {
    const int D = r->D, I = r->inner, Q = r->Q;
    const int64_t lr = B * Q, kvrows = rows + lr;
    t.xf = b.take(rows * D); t.x0 = b.take(lr * D); t.st_out = b.take(lr * 2);
    t.mask = reinterpret_cast<uint8_t*>(b.take((B * r->cfg.num_time_embeds + 3) / 4 + 1));
    t.layers.resize(r->cfg.depth);
    for (RLayerTape& L : t.layers) {
        L.st_m = b.take(rows * 2); L.kv = b.take(kvrows * 2 * I); L.x_in = nullptr; L.st_l = b.take(lr * 2);
        L.lat_n = b.take(lr * D); L.q = b.take(lr * I); L.att = b.take(lr * I); L.x_mid = b.take(lr * D);
        L.st_f = b.take(lr * 2); L.h = b.take(lr * D); L.u = b.take(lr * r->ff); L.hid = b.take(lr * r->ff);
        L.x_out = b.take(lr * D);
    }
}

static mdt_ln_train_args rln(const float* x, const float* w, const float* b, float* out, float* stats, int64_t M, int D) {
    mdt_ln_train_args a;
    memset(&a, 0, sizeof a);
    a.x = x; a.w = w; a.b = b; a.out = out; a.stats = stats; a.M = (int)M; a.D = D; a.rows_per_sample = 1;
    a.shift_off = a.scale_off = -1;
    return a;
}

static const int LN_CHUNKS = 8;  // row chunks per sample of the LayerNorm backward over the media tokens

static void carve_rscratch(mdt_resampler* r, Bump& b, int64_t rows, int64_t B) {
    const int D = r->D, I = r->inner, Q = r->Q;
    const int64_t lr = B * Q;
    r->g_dx = b.take(lr * D); r->g_dxf = b.take(rows * D); r->g_nm = b.take(rows * D); r->g_dnm = b.take(rows * D);
    r->g_dkv = nullptr; r->g_dkvm = b.take(rows * 2 * I); r->g_dkvl = b.take(lr * 2 * I);
    r->g_dq = b.take(lr * I); r->g_datt = b.take(lr * I); r->g_dlat = b.take(lr * D); r->g_ff = b.take(lr * r->ff);
    r->g_td = b.take(lr * D); r->g_pw = b.take(B * LN_CHUNKS * D); r->g_pb = b.take(B * LN_CHUNKS * D);
    int64_t need = mdt_linear_bwd_scratch(rows, 2 * I, D);
    for (auto nk : {std::pair<int, int>(2 * I, D), std::pair<int, int>(I, D), std::pair<int, int>(D, I), std::pair<int, int>(r->ff, D),
                    std::pair<int, int>(D, r->ff)})
        need = std::max(need, mdt_linear_bwd_scratch(lr, nk.first, nk.second));
    r->g_lin = b.take(need);
}

static mdt_status rscratch(mdt_resampler* r, int64_t rows, int64_t B) {
    if (rows > r->ts_rows || B > r->ts_b) {
        rows = std::max(rows, r->ts_rows); B = std::max(B, r->ts_b);
        if (r->tscratch) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(mdt_dev_free(r->tscratch)); r->tscratch = nullptr; }
        Bump count;
        carve_rscratch(r, count, rows, B);
        HIP_TRY(mdt_dev_malloc((void**)&r->tscratch, count.off * sizeof(float)));
        r->ts_rows = rows; r->ts_b = B;
    }
    Bump real;
    real.base = r->tscratch;
    carve_rscratch(r, real, r->ts_rows, r->ts_b);
    return MDT_OK;
}

extern "C" mdt_status mdt_resampler_forward_train(mdt_resampler* r, const float* x_f, const uint8_t* mask, int64_t batch,
                                                  int32_t n_frames, int32_t n_tokens, float* out, int32_t* tape,
                                                  void* stream) {
    if (!r || !x_f || !out || !tape || batch < 1 || n_frames < 1 || n_tokens < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_forward_train: bad argument");
    if (!r->wt_arena) return fail(MDT_ERR_STATE, "resampler training was not prepared (mdt_resampler_train_prepare)");
    if (n_frames > r->cfg.num_time_embeds) return fail(MDT_ERR_INVALID_ARG, "more frames than time embeddings");
    for (const RSlot& sl : r->slots)
        if (!sl.loaded) return fail(MDT_ERR_NOT_LOADED, "resampler parameter '%s' was not loaded after train_prepare", sl.name.c_str());
    const int D = r->D, I = r->inner, Q = r->Q, F = n_frames * n_tokens, Tk = F + Q;
    if (!mdt_attention_long_bwd_supported(r->hd, Q, Tk))
        return fail(MDT_ERR_UNSUPPORTED, "resampler: %d keys x %d latents exceeds the attention kernels' LDS budget", Tk, Q);
    const int64_t rows = batch * F, lr = batch * Q;
    hipStream_t s = (hipStream_t)stream;
    int pick = -1;
    for (size_t i = 0; i < r->tapes.size(); ++i)
        if (!r->tapes[i].in_use && r->tapes[i].cap_rows >= rows && r->tapes[i].cap_b >= batch) { pick = (int)i; break; }
    if (pick < 0)
        for (size_t i = 0; i < r->tapes.size(); ++i)
            if (!r->tapes[i].in_use) { pick = (int)i; break; }
    if (pick < 0) {
        if (r->tapes.size() >= 8) return fail(MDT_ERR_STATE, "more than 8 resampler tapes alive");
        r->tapes.emplace_back();
        pick = (int)r->tapes.size() - 1;
    }
    RTape& t = r->tapes[pick];
    if (t.cap_rows < rows || t.cap_b < batch) {
        if (t.buf) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(mdt_dev_free(t.buf)); t.buf = nullptr; }
        Bump count;
        carve_rtape(r, count, t, rows, batch);
        HIP_TRY(mdt_dev_malloc((void**)&t.buf, count.off * sizeof(float)));
        t.cap_rows = rows; t.cap_b = batch;
    }
    Bump real;
    real.base = t.buf;
    carve_rtape(r, real, t, t.cap_rows, t.cap_b);
    t.B = batch; t.T = n_frames; t.n = n_tokens; t.has_mask = mask != nullptr;
    MDT_TRY(rscratch(r, rows, batch));
    if (mask) HIP_TRY(hipMemcpyAsync(t.mask, mask, (size_t)batch * n_frames, hipMemcpyDeviceToDevice, s));
    LAUNCH(mdt_launch_add_time_emb(x_f, r->tpe, mask, t.xf, batch, n_frames, n_tokens, D, s));
    LAUNCH(mdt_launch_bcast_rows(r->latents, t.x0, batch, Q, D, s));
    const float scale = 1.0f / sqrtf((float)r->hd);
    const float* x = t.x0;
    for (int li = 0; li < r->cfg.depth; ++li) {
        const RLayer& L = r->layers[li];
        RLayerTape& P = t.layers[li];
        P.x_in = const_cast<float*>(x);
        LAUNCH(mdt_launch_ln_fwd_train(rln(t.xf, L.nm_w, L.nm_b, r->g_nm, P.st_m, rows, D), s));
        mdt_gemm_args a = gemm_args(r->g_nm, D, L.kv, P.kv, 2 * I, (int)rows);
        a.gin = F; a.gout = Tk; a.goff = 0;
        LAUNCH(mdt_launch_gemm(a, s));
        LAUNCH(mdt_launch_ln_fwd_train(rln(x, L.nl_w, L.nl_b, P.lat_n, P.st_l, lr, D), s));
        mdt_gemm_args b = gemm_args(P.lat_n, D, L.kv, P.kv, 2 * I, (int)lr);
        b.gin = Q; b.gout = Tk; b.goff = F;
        LAUNCH(mdt_launch_gemm(b, s));
        LAUNCH(mdt_launch_gemm(gemm_args(P.lat_n, D, L.q, P.q, I, (int)lr), s));
        LAUNCH(mdt_launch_attention_long(P.q, I, P.kv, P.kv + I, 2 * I, P.att, I, (int)batch, r->H, r->hd, Q, Tk, scale, s));
        HIP_TRY(hipMemcpyAsync(P.x_mid, x, (size_t)lr * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        mdt_gemm_args o = gemm_args(P.att, I, L.o, P.x_mid, D, (int)lr);
        o.residual = 1;
        LAUNCH(mdt_launch_gemm(o, s));
        LAUNCH(mdt_launch_ln_fwd_train(rln(P.x_mid, L.ff_w, L.ff_b, P.h, P.st_f, lr, D), s));
        LAUNCH(mdt_launch_gemm(gemm_args(P.h, D, L.f1, P.u, r->ff, (int)lr), s));
        LAUNCH(mdt_launch_act_fwd(P.u, P.hid, lr * r->ff, MDT_ACT_GELU, s));
        HIP_TRY(hipMemcpyAsync(P.x_out, P.x_mid, (size_t)lr * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        mdt_gemm_args f2 = gemm_args(P.hid, r->ff, L.f2, P.x_out, D, (int)lr);
        f2.residual = 1;
        LAUNCH(mdt_launch_gemm(f2, s));
        x = P.x_out;
    }
    LAUNCH(mdt_launch_ln_fwd_train(rln(x, r->norm_w, r->norm_b, out, t.st_out, lr, D), s));
    t.in_use = true;
    *tape = pick;
    return MDT_OK;
}

extern "C" mdt_status mdt_resampler_tape_release(mdt_resampler* r, int32_t tape) {
    if (!r || tape < 0 || tape >= (int)r->tapes.size() || !r->tapes[tape].in_use)
        return fail(MDT_ERR_INVALID_ARG, "invalid or released resampler tape %d", tape);
    r->tapes[tape].in_use = false;
    return MDT_OK;
}

static int rslot_of(const mdt_resampler* r, const float* dst) {
    for (size_t i = 0; i < r->slots.size(); ++i)
        if (r->slots[i].dst == dst) return (int)i;
    return -1;
}

static mdt_status r_lin_bwd(mdt_resampler* r, float* grads, const Lin& l, const float* X, int64_t ldx, const float* dY,
                            int64_t ldy, int64_t M, float* dX, int64_t ldxo, int acc_dx, hipStream_t s) {
    mdt_linear_bwd_args a;
    memset(&a, 0, sizeof a);
    a.X = X; a.ldx = ldx; a.dY = dY; a.ldy = ldy;
    a.dW = grads + r->grad_off[rslot_of(r, l.wp)];  // first part of the stack (rows from 0)
    a.accumulate_dw = 1; a.Wt = l.wt; a.dX = dX; a.ldxo = ldxo; a.accumulate_dx = acc_dx;
    a.M = (int)M; a.N = l.N; a.K = l.K; a.scratch = r->g_lin;
    return mdt_linear_bwd(a, s);
}

static mdt_status r_ln_bwd(mdt_resampler* r, float* grads, const float* x, const float* stats, const float* w, const float* b,
                           const float* dh, float* dx, int acc, int64_t B, int rps, hipStream_t s) {
    mdt_ln_bwd_args a;
    memset(&a, 0, sizeof a);
    a.x = x; a.stats = stats; a.w = w; a.b = b; a.shift_off = a.scale_off = -1; a.dh = dh; a.ld_dh = r->D; a.dx = dx;
    a.accumulate = acc; a.pw = r->g_pw; a.pb = r->g_pb; a.B = (int)B; a.rows_per_sample = rps; a.D = r->D;
    a.row_chunks = std::max(1, std::min(LN_CHUNKS, rps / 32));  // the ~400 media rows of a sample: 8 workgroups
    LAUNCH(mdt_launch_ln_bwd(a, s));
    LAUNCH(mdt_launch_colsum2(r->g_pw, r->g_pb, r->D, (int)B * a.row_chunks, r->D, grads + r->grad_off[rslot_of(r, w)],
                              grads + r->grad_off[rslot_of(r, b)], 1, s));
    return MDT_OK;
}

extern "C" mdt_status mdt_resampler_backward(mdt_resampler* r, int32_t tape, const float* g_out, float* grads, float* d_x_f,
                                             void* stream) {
    if (!r || !g_out || !grads) return fail(MDT_ERR_INVALID_ARG, "mdt_resampler_backward: null argument");
    if (tape < 0 || tape >= (int)r->tapes.size() || !r->tapes[tape].in_use)
        return fail(MDT_ERR_INVALID_ARG, "invalid or released resampler tape %d", tape);
    RTape& t = r->tapes[tape];
    hipStream_t s = (hipStream_t)stream;
    const int D = r->D, I = r->inner, Q = r->Q, F = t.T * t.n, Tk = F + Q;
    const int64_t B = t.B, rows = B * F, lr = B * Q;
    MDT_TRY(rscratch(r, rows, B));
    const float scale = 1.0f / sqrtf((float)r->hd);
    HIP_TRY(hipMemsetAsync(r->g_dxf, 0, (size_t)rows * D * sizeof(float), s));
    // final LayerNorm
    const float* x_last = t.layers[r->cfg.depth - 1].x_out;
    MDT_TRY(r_ln_bwd(r, grads, x_last, t.st_out, r->norm_w, r->norm_b, g_out, r->g_dx, 0, B, Q, s));
    for (int li = r->cfg.depth - 1; li >= 0; --li) {
        const RLayer& L = r->layers[li];
        RLayerTape& P = t.layers[li];
        // feed-forward: x_out = x_mid + f2(gelu(f1(ln(x_mid))))
        MDT_TRY(r_lin_bwd(r, grads, L.f2, P.hid, r->ff, r->g_dx, D, lr, r->g_ff, r->ff, 0, s));
        LAUNCH(mdt_launch_act_bwd(P.u, r->g_ff, r->g_ff, lr * r->ff, MDT_ACT_GELU, s));
        MDT_TRY(r_lin_bwd(r, grads, L.f1, P.h, D, r->g_ff, r->ff, lr, r->g_td, D, 0, s));
        MDT_TRY(r_ln_bwd(r, grads, P.x_mid, P.st_f, L.ff_w, L.ff_b, r->g_td, r->g_dx, 1, B, Q, s));
        // attention: x_mid = x_in + to_out(attn(to_q(ln_l(x_in)), to_k|to_v(cat(ln_m(x_f), ln_l(x_in)))))
        MDT_TRY(r_lin_bwd(r, grads, L.o, P.att, I, r->g_dx, D, lr, r->g_datt, I, 0, s));
        // dK|dV of the media rows and of the latent rows leave the kernel in their own dense buffers
        LAUNCH(mdt_launch_attention_long_bwd(P.q, I, P.kv, P.kv + I, 2 * I, r->g_datt, I, r->g_dq, I, r->g_dkvm, r->g_dkvm + I,
                                             2 * I, (int)B, r->H, r->hd, Q, Tk, scale, s, r->g_dkvl, F));
        MDT_TRY(r_lin_bwd(r, grads, L.q, P.lat_n, D, r->g_dq, I, lr, r->g_dlat, D, 0, s));
        MDT_TRY(r_lin_bwd(r, grads, L.kv, P.lat_n, D, r->g_dkvl, 2 * I, lr, r->g_dlat, D, 1, s));
        LAUNCH(mdt_launch_ln_fwd_train(rln(t.xf, L.nm_w, L.nm_b, r->g_nm, nullptr, rows, D), s));  // recomputed, not kept
        MDT_TRY(r_lin_bwd(r, grads, L.kv, r->g_nm, D, r->g_dkvm, 2 * I, rows, r->g_dnm, D, 0, s));
        MDT_TRY(r_ln_bwd(r, grads, t.xf, P.st_m, L.nm_w, L.nm_b, r->g_dnm, r->g_dxf, 1, B, F, s));
        MDT_TRY(r_ln_bwd(r, grads, P.x_in, P.st_l, L.nl_w, L.nl_b, r->g_dlat, r->g_dx, 1, B, Q, s));
    }
    // latents were repeated over the batch; the time embedding was added (masked) to every token of its frame
    LAUNCH(mdt_launch_colsum(r->g_dx, (int64_t)Q * D, (int)B, Q * D, grads + r->grad_off[rslot_of(r, r->latents)], 1, s));
    LAUNCH(mdt_launch_time_emb_grad(r->g_dxf, t.has_mask ? t.mask : nullptr, grads + r->grad_off[rslot_of(r, r->tpe)],
                                    r->g_dnm /* free by now: B*T*D <= rows*D floats */, B, t.T, t.n, D, 1, s));
    if (d_x_f) HIP_TRY(hipMemcpyAsync(d_x_f, r->g_dxf, (size_t)rows * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    return MDT_OK;
}
