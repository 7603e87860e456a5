// mdt_model.hip -- host side of libmdt_hip.so: model handle, fragment-packed weight arena, workspace, and the
// launch sequences behind the C ABI of include/mdt_hip.h (encoder, adaLN decoder step, fused DDIM loop, loss).
//
// Reference call stack replaced:
//   sample_ddim (gc_sampling.py:922) -> GCDenoiser.forward (score_wrappers.py:65) ->
//   MDTVTransformer.forward (mdtv_transformer.py:208) = forward_enc_only (:213) + forward_dec_only (:224)
// Structural facts exploited: with adaLN conditioning the encoder output and the
// cross-attention K/V do not depend on sigma or the noisy actions -> computed once per sample call; in a
// sampler sigma is one scalar per step -> sigma-embedding + all adaLN modulation vectors for all steps are
// one batched M = n_steps GEMM chain before the loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "mdt_internal.h"

// ------------------------------------------------------------------------------------------------
// error plumbing (shared with mdt_resampler.hip through mdt_internal.h)
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

mdt_status mdt_fail(mdt_status st, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}
#define fail mdt_fail

extern "C" const char* mdt_last_error(void) { return g_err.c_str(); }
extern "C" uint32_t mdt_fnv1_32(const void* buf, uint64_t len, uint32_t seed) {
    const unsigned char* p = static_cast<const unsigned char*>(buf);
    uint32_t h = seed;
    for (uint64_t i = 0; i < len; ++i) {
        h *= 0x01000193u;
        h ^= p[i];
    }
    return h;
}

// ---- pluggable allocator of the batch-sized buffers (include/mdt_hip.h: mdt_set_allocator) ----
#include <mutex>
namespace {
std::mutex g_alloc_mu;
mdt_alloc_fn g_alloc = nullptr;
mdt_free_fn g_free = nullptr;
void* g_alloc_user = nullptr;
std::unordered_map<void*, std::pair<mdt_free_fn, void*>>* g_custom_free = nullptr;
}  // namespace

extern "C" mdt_status mdt_set_allocator(mdt_alloc_fn alloc, mdt_free_fn free_, void* user) {
    if ((alloc == nullptr) != (free_ == nullptr)) return fail(MDT_ERR_INVALID_ARG, "mdt_set_allocator: give both functions or neither");
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    g_alloc = alloc; g_free = free_; g_alloc_user = user;
    return MDT_OK;
}

extern "C" mdt_status mdt_allocator_detach(void) {
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    g_alloc = nullptr; g_free = nullptr; g_alloc_user = nullptr;
    // the entries stay, with a null callback: mdt_dev_free must still tell these pointers from hipMalloc'ed ones
    if (g_custom_free) for (auto& kv : *g_custom_free) kv.second = std::make_pair((mdt_free_fn)nullptr, (void*)nullptr);
    return MDT_OK;
}

hipError_t mdt_dev_malloc(void** p, size_t bytes) {
    mdt_alloc_fn fn; mdt_free_fn ff; void* user;
    { std::lock_guard<std::mutex> lock(g_alloc_mu); fn = g_alloc; ff = g_free; user = g_alloc_user; }
    if (fn == nullptr) return hipMalloc(p, bytes);
    void* q = fn(bytes, user);
    if (q == nullptr) { *p = nullptr; return hipErrorOutOfMemory; }
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    if (!g_custom_free) g_custom_free = new std::unordered_map<void*, std::pair<mdt_free_fn, void*>>();
    (*g_custom_free)[q] = std::make_pair(ff, user);
    *p = q;
    return hipSuccess;
}

hipError_t mdt_dev_free(void* p) {
    if (p == nullptr) return hipSuccess;
    std::pair<mdt_free_fn, void*> how(nullptr, nullptr);
    bool custom = false;
    {
        std::lock_guard<std::mutex> lock(g_alloc_mu);
        if (g_custom_free) {
            auto it = g_custom_free->find(p);
            if (it != g_custom_free->end()) { how = it->second; custom = true; g_custom_free->erase(it); }
        }
    }
    if (custom) {  // from an installed allocator: its callback, or nothing at all once the host detached (mdt_allocator_detach)
        if (how.first) how.first(p, how.second);
        return hipSuccess;
    }
    return hipFree(p);
}

extern "C" const char* mdt_version(void) { return "mdt_hip 0.1 (gfx950, v_mfma_f32_16x16x4_f32)"; }

#include "mdt_model_types.h"

static const int MAX_WAYS = 4;

// workspace view of a contiguous slice of samples [b0, b0 + nb) for the decoder (Ta rows per sample)
struct View {
    float *y, *qkv, *att, *hid, *qx, *kvx;
    int64_t b0;  // first sample of the slice
};

static View decoder_view(const mdt_model* m, int64_t b0) {
    View v;
    const int64_t r = b0 * m->Ta;
    v.y = m->y + r * m->D;
    v.qkv = m->qkv + r * 3 * m->D;
    v.att = m->att + r * m->D;
    v.hid = m->hid + r * 4 * m->D;
    v.qx = m->qx + r * m->D;
    v.kvx = m->kvx + b0 * m->Te * (int64_t)m->Ld * 2 * m->D;
    v.b0 = b0;
    return v;
}

static View encoder_view(const mdt_model* m) {
    View v;
    v.y = m->h_enc; v.qkv = m->qkv; v.att = m->att; v.hid = m->hid; v.qx = nullptr; v.kvx = nullptr;
    v.b0 = 0;
    return v;
}

static const int MAX_STEPS = MDT_SCHED_MAX;  // (a host schedule travels in k_sample_prep's kernel arguments)

// Parameter map: called twice (count pass with base == nullptr, then with the allocated arena).
static void build_params(mdt_model* m, Bump& b, bool fill_slots) {
    const mdt_config& c = m->cfg;
    const int D = m->D, G = m->G, O = m->O, A = m->A;
    auto add_slot = [&](const std::string& name, int64_t numel, int kind, float* dst, int rows, int K, int n_off) {
        if (!fill_slots) return;
        Slot s;
        s.name = name; s.numel = numel; s.kind = kind; s.dst = dst; s.rows = rows; s.K = K; s.n_off = n_off;
        m->slots.push_back(s);
    };
    // a Linear made of `parts` reference Linears stacked along N (e.g. query|key|value)
    auto lin_begin = [&](Lin& l, int N, int K, bool bias) {
        l.N = N; l.K = K;
        l.wp = b.take((size_t)N * K);
        l.bias = bias ? b.take(N) : nullptr;
    };
    auto lin_part = [&](Lin& l, const std::string& prefix, int rows, int n_off, bool bias) {
        add_slot(prefix + ".weight", (int64_t)rows * l.K, SLOT_PACK, l.wp, rows, l.K, n_off);
        if (fill_slots) {
            m->slots.back().lin = &l;
            LinPart p;
            p.lin = &l; p.w_slot = (int)m->slots.size() - 1; p.b_slot = bias ? p.w_slot + 1 : -1; p.rows = rows; p.n_off = n_off;
            m->parts.push_back(p);
        }
        if (bias) add_slot(prefix + ".bias", rows, SLOT_RAW, l.bias ? l.bias + n_off : nullptr, 0, 0, 0);
    };
    auto raw = [&](float*& p, const std::string& name, int64_t n) {
        p = b.take(n);
        add_slot(name, n, SLOT_RAW, p, 0, 0, 0);
    };
    auto add_extra = [&](const std::string& name, int64_t numel, int kind, float* dst, int rows, int K, int n_off = 0) {
        if (!fill_slots) return;
        Slot s;
        s.name = name; s.numel = numel; s.kind = kind; s.dst = dst; s.rows = rows; s.K = K; s.n_off = n_off;
        m->extra.push_back(s);
    };
    const std::string P = "inner_model.";
    const bool xb = c.bias != 0;  // reference `bias` flag: c_proj / MLP / custom LayerNorm biases

    if (c.arch == MDT_ARCH_MDT && c.use_abs_pos_emb)
        raw(m->pos_emb, P + "pos_emb", (int64_t)(c.goal_seq_len + c.action_seq_len) * D);
    lin_begin(m->tok, D, O, true);
    lin_part(m->tok, P + "tok_emb", D, 0, true);
    if (c.arch == MDT_ARCH_MDT) {
        lin_begin(m->incam, D, O, true);
        lin_part(m->incam, P + "incam_embed", D, 0, true);
    }
    auto goal_mlp = [&](Lin& l0, Lin& l2, const std::string& nm) {
        if (c.use_mlp_goal) {
            lin_begin(l0, 2 * D, G, true);
            lin_part(l0, P + nm + ".0", 2 * D, 0, true);
            lin_begin(l2, D, 2 * D, true);
            lin_part(l2, P + nm + ".2", D, 0, true);
        } else {
            lin_begin(l2, D, G, true);
            lin_part(l2, P + nm, D, 0, true);
        }
    };
    goal_mlp(m->goal0, m->goal2, "goal_emb");
    if (c.use_modality_encoder) goal_mlp(m->lang0, m->lang2, "lang_emb");

    auto attn_qkv = [&](Lin& qkv, const std::string& pre) {
        // packed row order is q | k | v; the reference registers key, query, value (transformer_blocks.py:85-87)
        lin_begin(qkv, 3 * D, D, true);
        lin_part(qkv, pre + ".key", D, D, true);
        lin_part(qkv, pre + ".query", D, 0, true);
        lin_part(qkv, pre + ".value", D, 2 * D, true);
        if (D % 128 == 0 && D <= 512) {   // the split form of the LayerNorm-prologue product (mdt_mlp_split.h): 6 bytes per weight
            qkv.ws = b.take((size_t)3 * D * D * 6 / 4);
            add_extra(pre + ".key.weight", (int64_t)D * D, SLOT_PACK_SPLIT, (float*)qkv.ws, D, D, D);
            add_extra(pre + ".query.weight", (int64_t)D * D, SLOT_PACK_SPLIT, (float*)qkv.ws, D, D, 0);
            add_extra(pre + ".value.weight", (int64_t)D * D, SLOT_PACK_SPLIT, (float*)qkv.ws, D, D, 2 * D);
        }
    };
    auto block_common_a = [&](EncBlock& e, const std::string& pre) {
        raw(e.ln1_w, pre + ".ln_1.weight", D);
        if (xb) raw(e.ln1_b, pre + ".ln_1.bias", D);
        attn_qkv(e.qkv, pre + ".attn");
        lin_begin(e.proj, D, D, xb);
        lin_part(e.proj, pre + ".attn.c_proj", D, 0, xb);
    };
    auto block_common_b = [&](EncBlock& e, const std::string& pre) {
        raw(e.ln2_w, pre + ".ln_2.weight", D);
        if (xb) raw(e.ln2_b, pre + ".ln_2.bias", D);
        lin_begin(e.fc, 4 * D, D, xb);
        lin_part(e.fc, pre + ".mlp.c_fc", 4 * D, 0, xb);
        lin_begin(e.proj2, D, 4 * D, xb);
        lin_part(e.proj2, pre + ".mlp.c_proj", D, 0, xb);
        if (D % 128 == 0 && D <= 512) {   // the fused MLP launch's split form (mdt_mlp_split.h): 6 bytes per weight
            e.fc.ws = b.take((size_t)4 * D * D * 6 / 4);
            e.proj2.ws = b.take((size_t)4 * D * D * 6 / 4);
            add_extra(pre + ".mlp.c_fc.weight", (int64_t)4 * D * D, SLOT_PACK_SPLIT, (float*)e.fc.ws, 4 * D, D);
            add_extra(pre + ".mlp.c_proj.weight", (int64_t)4 * D * D, SLOT_PACK_SPLIT, (float*)e.proj2.ws, D, 4 * D);
        }
    };
    if (!fill_slots) { m->enc.assign(m->Le, EncBlock()); m->dec.assign(m->Ld, DecBlock()); }
    for (int l = 0; l < m->Le; ++l) {
        const std::string pre = P + "encoder.blocks." + std::to_string(l);
        block_common_a(m->enc[l], pre);
        block_common_b(m->enc[l], pre);
    }
    raw(m->enc_ln_w, P + "encoder.ln.weight", D);
    if (xb) raw(m->enc_ln_b, P + "encoder.ln.bias", D);

    // stacked across decoder blocks: cross-attention K|V projections of the context, adaLN modulation
    lin_begin(m->kv_all, m->Ld * 2 * D, D, true);
    if (m->cond == COND_ADALN) lin_begin(m->mod_all, m->Ld * 6 * D, D, true);
    for (int l = 0; l < m->Ld; ++l) {
        const std::string pre = P + "decoder.blocks." + std::to_string(l);
        DecBlock& d = m->dec[l];
        block_common_a(d, pre);
        lin_part(m->kv_all, pre + ".cross_att.key", D, l * 2 * D, true);
        lin_begin(d.xq, D, D, true);
        lin_part(d.xq, pre + ".cross_att.query", D, 0, true);
        d.xq_pT = b.take((size_t)D * D);
        add_extra(pre + ".cross_att.query.weight", (int64_t)D * D, SLOT_PACK_T, d.xq_pT, D, D);
        lin_part(m->kv_all, pre + ".cross_att.value", D, l * 2 * D + D, true);
        lin_begin(d.xproj, D, D, xb);
        lin_part(d.xproj, pre + ".cross_att.c_proj", D, 0, xb);
        raw(d.ln3_w, pre + ".ln3.weight", D);
        raw(d.ln3_b, pre + ".ln3.bias", D);
        block_common_b(d, pre);
        if (m->cond == COND_ADALN) lin_part(m->mod_all, pre + ".adaLN_zero.modulation.1", 6 * D, l * 6 * D, true);
    }
    raw(m->dec_ln_w, P + "decoder.ln.weight", D);
    if (xb) raw(m->dec_ln_b, P + "decoder.ln.bias", D);
    if (c.use_proprio) {  // registered between decoder and sigma_emb (mdtv_transformer.py:160-164)
        m->prop0_T = b.take((size_t)2 * D * c.proprio_dim);  // (Pd, 2D): k_narrow_linear reads rows of 2D
        add_slot(P + "proprio_emb.0.weight", (int64_t)2 * D * c.proprio_dim, SLOT_TRANSPOSE, m->prop0_T, 2 * D, c.proprio_dim, 0);
        raw(m->prop0_b, P + "proprio_emb.0.bias", 2 * D);
        lin_begin(m->prop2, D, 2 * D, true);
        lin_part(m->prop2, P + "proprio_emb.2", D, 0, true);
    }
    lin_begin(m->sig1, 2 * D, D, true);
    lin_part(m->sig1, P + "sigma_emb.1", 2 * D, 0, true);
    lin_begin(m->sig3, D, 2 * D, true);
    lin_part(m->sig3, P + "sigma_emb.3", D, 0, true);
    m->Wa = b.take((size_t)D * A);  // stored transposed (A, D): lanes read 16 contiguous bytes per action component
    add_slot(P + "action_emb.weight", (int64_t)D * A, SLOT_TRANSPOSE, m->Wa, D, A, 0);
    raw(m->ba, P + "action_emb.bias", D);
    if (c.linear_output) {
        raw(m->Wp, P + "action_pred.weight", (int64_t)A * D);
        raw(m->bp, P + "action_pred.bias", A);
    } else {  // nn.Sequential(Linear(d, 100), GELU(), Linear(100, A))   (mdtv_transformer.py:181-185)
        m->HH = c.arch == MDT_ARCH_MDTV ? 100 : D;
        m->HP = (m->HH + 15) / 16 * 16;
        lin_begin(m->head0, m->HP, D, true);          // rows HH..HP-1 of the image and of the bias stay zero
        lin_part(m->head0, P + "action_pred.0", m->HH, 0, true);
        m->Wp = b.take((size_t)A * m->HP);             // (A, HP): columns HH.. stay zero
        add_slot(P + "action_pred.2.weight", (int64_t)A * m->HH, SLOT_PAD_COLS, m->Wp, A, m->HH, m->HP);
        raw(m->bp, P + "action_pred.2.bias", A);
    }
    // constant tables
    m->freqs = b.take(D / 2);
    m->rope_cos = b.take(16 * 16);
    m->rope_sin = b.take(16 * 16);
}

static bool is_ignored_param(const mdt_model* m, const std::string& name) {
    auto ends_with = [&](const char* suf) {
        const size_t n = strlen(suf);
        return name.size() >= n && name.compare(name.size() - n, n, suf) == 0;
    };
    if (!m->cfg.use_proprio && name.rfind("inner_model.proprio_emb.", 0) == 0) return true;
    if (ends_with(".rotary_pos_emb.freqs")) return true;
    if (name == "inner_model.pos_emb") return true;  // MDT-V (or MDT without abs pos emb) never reads it
    if (!m->cfg.use_modality_encoder && name.rfind("inner_model.lang_emb", 0) == 0) return false;
    return false;
}

extern "C" mdt_status mdt_create(const mdt_config* cfg, mdt_model** out) {
    if (!cfg || !out) return fail(MDT_ERR_INVALID_ARG, "mdt_create: null argument");
    const mdt_config& c = *cfg;
    if (c.arch != MDT_ARCH_MDTV && c.arch != MDT_ARCH_MDT) return fail(MDT_ERR_INVALID_ARG, "unknown arch %d", c.arch);
    if (c.goal_seq_len != 1) return fail(MDT_ERR_UNSUPPORTED, "goal_seq_len must be 1");
    if (c.embed_dim <= 0 || c.embed_dim % 16 || c.embed_dim > 512)
        return fail(MDT_ERR_UNSUPPORTED, "embed_dim %d: need a multiple of 16, <= 512", c.embed_dim);
    if (c.n_heads <= 0 || c.embed_dim % c.n_heads) return fail(MDT_ERR_INVALID_ARG, "embed_dim %% n_heads != 0");
    const int hd = c.embed_dim / c.n_heads;
    if (hd != 16 && hd != 32 && hd != 48 && hd != 64)
        return fail(MDT_ERR_UNSUPPORTED, "head dim %d: supported 16/32/48/64", hd);
    if (c.use_rot_embed && hd < 32)
        return fail(MDT_ERR_INVALID_ARG, "use_rot_embed needs head dim >= 32 (rotary dim is 32; the reference asserts)");
    if (c.obs_dim % 16 || c.goal_dim % 16 || c.obs_dim <= 0 || c.goal_dim <= 0)
        return fail(MDT_ERR_UNSUPPORTED, "obs_dim / goal_dim must be positive multiples of 16");
    if (c.goal_dim == 2 * c.obs_dim)
        return fail(MDT_ERR_UNSUPPORTED, "goal_dim == 2*obs_dim triggers the reference's goal truncation "
                                         "(mdtv_transformer.py:252-253); not implemented");
    if (c.action_dim < 1 || c.action_dim > 16) return fail(MDT_ERR_UNSUPPORTED, "action_dim must be 1..16");
    if (c.action_seq_len < 1 || c.action_seq_len > 16) return fail(MDT_ERR_UNSUPPORTED, "action_seq_len must be 1..16");
    const int n_tok = c.arch == MDT_ARCH_MDTV ? c.n_obs_token : 2;
    // use_noise_encoder only selects the block type of the adaLN-style decoder (TransformerFiLMDecoder); without
    // use_ada_conditioning the reference builds a plain TransformerDecoder and the flag is never read
    const int cond = !c.use_ada_conditioning ? COND_TOKEN : (c.use_noise_encoder ? COND_NOISE : COND_ADALN);
    const int sig_tok = cond == COND_TOKEN ? 1 : 0;
    const bool prop = c.use_proprio != 0;
    if (prop && c.arch != MDT_ARCH_MDTV)
        return fail(MDT_ERR_INVALID_ARG, "use_proprio: only MDTVTransformer reads state['state_obs'] "
                                         "(MDTTransformer.process_state_embeddings returns None, mdt_transformer.py:309-316)");
    if (prop && (c.proprio_dim < 1 || c.proprio_dim > 16)) return fail(MDT_ERR_UNSUPPORTED, "proprio_dim must be 1..16");
    if (n_tok < 1 || sig_tok + 1 + n_tok + (prop ? 1 : 0) > 16)
        return fail(MDT_ERR_UNSUPPORTED, "context length must be <= 16 tokens");
    const bool gc = c.no_goal_conditioning == 0;
    if (!gc && c.arch == MDT_ARCH_MDT && c.use_ada_conditioning)
        return fail(MDT_ERR_UNSUPPORTED, "MDTTransformer with goal_conditioned=False needs use_ada_conditioning=False "
                                         "(the reference concatenates the absent sigma token, mdt_transformer.py:334)");
    // MDT-V keeps the goal token of an un-conditioned model behind the state tokens -- unless the proprioceptive token
    // takes that place (concatenate_inputs, mdtv_transformer.py:291-294)
    const int has_goal = (gc || (c.arch == MDT_ARCH_MDTV && !prop)) ? 1 : 0;
    if (c.n_enc_layers < 0 || c.n_dec_layers < 1) return fail(MDT_ERR_INVALID_ARG, "bad layer counts");
    if (!(c.sigma_data > 0.f)) return fail(MDT_ERR_INVALID_ARG, "sigma_data must be > 0");

    mdt_model* m = new mdt_model();
    m->cfg = c;
    m->cond = cond; m->sig_tok = sig_tok;
    m->D = c.embed_dim; m->H = c.n_heads; m->hd = hd; m->n_tok = n_tok; m->Te = sig_tok + has_goal + n_tok + (prop ? 1 : 0); m->Ta = c.action_seq_len;
    m->Pd = prop ? c.proprio_dim : 0;
    m->p_row = prop ? m->Te - 1 : -1;
    m->g_row = !has_goal ? -1 : (gc ? sig_tok : sig_tok + n_tok);
    m->tok_row = gc ? sig_tok + 1 : sig_tok;
    m->A = c.action_dim; m->Le = c.n_enc_layers; m->Ld = c.n_dec_layers; m->G = c.goal_dim; m->O = c.obs_dim;

    Bump count;
    build_params(m, count, false);
    m->arena_floats = count.off;
    hipError_t e = hipMalloc((void**)&m->arena, m->arena_floats * sizeof(float));
    if (e != hipSuccess) { delete m; return fail(MDT_ERR_HIP, "hipMalloc(arena) failed: %s", hipGetErrorString(e)); }
    e = hipMemset(m->arena, 0, m->arena_floats * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(m->arena); delete m; return fail(MDT_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e)); }
    Bump real;
    real.base = m->arena;
    build_params(m, real, true);

    // constant tables (host fp32 math mirroring the reference's torch fp32 expressions)
    const int half = m->D / 2;
    std::vector<float> fr(half), rc(256), rs(256);
    const float step = (float)(-(std::log(10000.0) / (double)(half - 1)));  // mdtv_transformer.py:21-22
    for (int j = 0; j < half; ++j) fr[j] = expf((float)j * step);
    for (int i = 0; i < 16; ++i) {  // position_embeddings.py:104: theta ** -(arange(0,32,2)/32)
        const float f = 1.0f / powf(10000.0f, (float)(2 * i) / 32.0f);
        for (int p = 0; p < 16; ++p) {
            rc[p * 16 + i] = cosf((float)p * f);
            rs[p * 16 + i] = sinf((float)p * f);
        }
    }
    (void)hipMemcpy(m->freqs, fr.data(), half * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(m->rope_cos, rc.data(), 256 * sizeof(float), hipMemcpyHostToDevice);
    e = hipMemcpy(m->rope_sin, rs.data(), 256 * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(m->arena); delete m; return fail(MDT_ERR_HIP, "table upload failed: %s", hipGetErrorString(e)); }

    size_t mx = 0;
    for (const Slot& s : m->slots) mx = std::max(mx, (size_t)s.numel);
    m->staging_floats = mx;
    e = hipMalloc((void**)&m->staging, mx * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(m->arena); delete m; return fail(MDT_ERR_HIP, "hipMalloc(staging) failed: %s", hipGetErrorString(e)); }
    // the collapsed cross-attention needs a step-independent context and an unconditioned query input
    m->xfold = cond == COND_ADALN && !c.use_rot_embed && mdt_xattn_apply_supported(m->D, m->H, m->Te, m->Ta);
    if (const char* x = getenv("MDT_HIP_XFOLD")) m->xfold = m->xfold && atoi(x) != 0;
    if (const char* w = getenv("MDT_HIP_WAYS")) m->ways = std::max(1, std::min(MAX_WAYS, atoi(w)));
    // (the auxiliary streams of MDT_HIP_WAYS > 1 are created on first use: every stream a process creates takes a slot in the
    //  runtime's small pool of hardware queues, and three idle ones per handle pushed the training path's side stream onto a
    //  queue it shares -- the B = 1024 step read 9.75 ms inside bench.py, where two handles exist, against 9.19 alone)
    if (hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess) {
        mdt_destroy(m);
        return fail(MDT_ERR_HIP, "could not create the sampler's fork event");
    }
    *out = m;
    return MDT_OK;
}

extern "C" mdt_status mdt_destroy(mdt_model* m) {
    if (!m) return MDT_OK;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < MAX_WAYS - 1; ++i) {
        if (m->aux[i]) (void)hipStreamDestroy(m->aux[i]);
        if (m->ev_join[i]) (void)hipEventDestroy(m->ev_join[i]);
    }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    mdt_train_free(m);
    (void)hipFree(m->arena);
    (void)hipFree(m->staging);
    (void)mdt_dev_free(m->ws);
    if (m->tab_dev) (void)hipFree(m->tab_dev);
    for (int i = 0; i < 2; ++i) {
        if (m->tab_host[i]) (void)hipHostFree(m->tab_host[i]);
        if (m->ev_tab[i]) (void)hipEventDestroy(m->ev_tab[i]);
    }
    if (m->ev_tab_use) (void)hipEventDestroy(m->ev_tab_use);
    delete m;
    return MDT_OK;
}

extern "C" int64_t mdt_param_count(const mdt_model* m) { return m ? (int64_t)m->slots.size() : 0; }
extern "C" const char* mdt_param_name(const mdt_model* m, int64_t i) {
    return (m && i >= 0 && i < (int64_t)m->slots.size()) ? m->slots[i].name.c_str() : nullptr;
}
extern "C" int64_t mdt_param_numel(const mdt_model* m, int64_t i) {
    return (m && i >= 0 && i < (int64_t)m->slots.size()) ? m->slots[i].numel : -1;
}

// A captured HIP graph of a large-batch call replays the split launches WITHOUT the refresh that a stale image needs: the first
// load that makes the images stale also bumps the generation the graph owners compare (mdt_ws_generation), so they capture again --
// with the refresh inside, which a replay then repeats (correct; 24 small launches) until the next capture.
static inline void mark_split_stale(mdt_model* m) {
    if (!m->split_stale) { m->split_stale = true; ++m->ws_generation; }
}

extern "C" mdt_status mdt_load_param(mdt_model* m, const char* name, const float* src, int64_t numel, void* stream) {
    if (!m || !name || !src) return fail(MDT_ERR_INVALID_ARG, "mdt_load_param: null argument");
    hipStream_t s = (hipStream_t)stream;
    std::string nm(name);
    if (!m->cfg.use_modality_encoder && nm.rfind("inner_model.lang_emb", 0) == 0)
        nm.replace(0, strlen("inner_model.lang_emb"), "inner_model.goal_emb");  // same module in the reference
    Slot* slot = nullptr;
    for (Slot& c : m->slots)
        if (c.name == nm) { slot = &c; break; }
    if (!slot) {
        if (is_ignored_param(m, nm)) return MDT_OK;
        return fail(MDT_ERR_INVALID_ARG, "mdt_load_param: unknown parameter '%s'", name);
    }
    if (numel != slot->numel)
        return fail(MDT_ERR_INVALID_ARG, "mdt_load_param: '%s' has %lld elements, expected %lld", name, (long long)numel,
                    (long long)slot->numel);
    std::vector<Slot*> targets;
    targets.push_back(slot);
    for (Slot& e : m->extra)
        if (e.name == nm) targets.push_back(&e);
    const float* dev_src = nullptr;  // device image of the source, staged at most once
    for (Slot* t : targets) {
        if (t->kind == SLOT_RAW) {
            HIP_TRY(hipMemcpyAsync(t->dst, src, numel * sizeof(float), hipMemcpyDefault, s));
            continue;
        }
        if (t->kind == SLOT_PAD_COLS) {  // (rows, K) -> (rows, n_off): the pad columns keep the arena's zeros
            HIP_TRY(hipMemcpy2DAsync(t->dst, (size_t)t->n_off * sizeof(float), src, (size_t)t->K * sizeof(float),
                                     (size_t)t->K * sizeof(float), (size_t)t->rows, hipMemcpyDefault, s));
            continue;
        }
        if (dev_src == nullptr) {
            hipPointerAttribute_t attr;
            hipError_t pe = hipPointerGetAttributes(&attr, src);
            if (pe == hipSuccess && attr.type == hipMemoryTypeDevice) {
                dev_src = src;
            } else {
                (void)hipGetLastError();  // unregistered host memory reports an error: clear it
                HIP_TRY(hipMemcpyAsync(m->staging, src, numel * sizeof(float), hipMemcpyHostToDevice, s));
                dev_src = m->staging;
            }
        }
        if (t->kind == SLOT_PACK_SPLIT) {
            if (m->train) { mark_split_stale(m); continue; }   // training: re-made on demand (mdt_model_types.h)
            HIP_TRY(mdt_launch_pack_weight_split(dev_src, t->rows, t->K, t->dst, s, t->n_off));
            continue;
        }
        if (t->kind == SLOT_TRANSPOSE) HIP_TRY(mdt_launch_transpose(dev_src, t->dst, t->rows, t->K, s));
        else if (t->kind == SLOT_PACK_T) HIP_TRY(mdt_launch_pack_weight_t(dev_src, t->rows, t->K, t->K, t->dst, 0, t->rows / 16, s));  // image of the transpose
        else HIP_TRY(mdt_launch_pack_weight(dev_src, t->rows, t->K, t->dst, t->n_off, s));
        if (t->kind == SLOT_PACK && t->lin && t->lin->wt)  // training: image of W^T for dX = dY W
            HIP_TRY(mdt_launch_pack_weight_t(dev_src, t->rows, t->K, t->K, t->lin->wt, t->n_off, t->lin->N / 16, s));
    }
    slot->loaded = true;
    return MDT_OK;
}

// load_state_dict / the re-upload after an optimizer step as ONE launch: all `n` parameters (device pointers, reference
// layout) are moved into their packed / transposed / raw images by k_multi_load.  Host-resident sources, and handles whose
// table memory cannot be allocated, go through mdt_load_param one by one.
extern "C" mdt_status mdt_load_params(mdt_model* m, int32_t n, const char* const* names, const float* const* srcs,
                                      const int64_t* numels, void* stream) {
    if (!m || n < 0 || (n > 0 && (!names || !srcs || !numels))) return fail(MDT_ERR_INVALID_ARG, "mdt_load_params: bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<mdt_load_entry> tab;
    std::vector<int2> blocks;
    std::vector<Slot*> touched;
    auto add = [&](const float* src, float* dst, int kind, int rows, int K, int p0, int p1) {
        mdt_load_entry e;
        memset(&e, 0, sizeof e);
        e.src = src; e.dst = dst; e.kind = kind; e.rows = rows; e.K = K; e.p0 = p0; e.p1 = p1;
        const int64_t work = kind == MDT_LOAD_PACK_T ? (int64_t)((rows + 3) / 4) * K
                             : (kind == MDT_LOAD_PACK || kind == MDT_LOAD_PACK_SPLIT) ? (int64_t)rows * (K / 4) : (int64_t)rows * K;
        for (int64_t c = 0; c * 1024 < work; ++c) blocks.push_back(make_int2((int)tab.size(), (int)c));
        tab.push_back(e);
    };
    for (int i = 0; i < n; ++i) {
        if (!names[i] || !srcs[i]) return fail(MDT_ERR_INVALID_ARG, "mdt_load_params: null entry %d", i);
        std::string nm(names[i]);
        if (!m->cfg.use_modality_encoder && nm.rfind("inner_model.lang_emb", 0) == 0)
            nm.replace(0, strlen("inner_model.lang_emb"), "inner_model.goal_emb");
        Slot* slot = nullptr;
        for (Slot& c : m->slots)
            if (c.name == nm) { slot = &c; break; }
        if (!slot) {
            if (is_ignored_param(m, nm)) continue;
            return fail(MDT_ERR_INVALID_ARG, "mdt_load_params: unknown parameter '%s'", names[i]);
        }
        if (numels[i] != slot->numel)
            return fail(MDT_ERR_INVALID_ARG, "mdt_load_params: '%s' has %lld elements, expected %lld", names[i],
                        (long long)numels[i], (long long)slot->numel);
        hipPointerAttribute_t attr;
        const hipError_t pe = hipPointerGetAttributes(&attr, srcs[i]);
        if (pe != hipSuccess || attr.type != hipMemoryTypeDevice || ((uintptr_t)srcs[i] & 15)) {
            (void)hipGetLastError();
            MDT_TRY(mdt_load_param(m, names[i], srcs[i], numels[i], stream));  // host memory: the staged path
            continue;
        }
        std::vector<Slot*> targets;
        targets.push_back(slot);
        for (Slot& e : m->extra)
            if (e.name == nm) targets.push_back(&e);
        for (Slot* t : targets) {
            if (t->kind == SLOT_RAW) add(srcs[i], t->dst, MDT_LOAD_RAW, 1, (int)numels[i], 0, 0);
            else if (t->kind == SLOT_PAD_COLS) add(srcs[i], t->dst, MDT_LOAD_PAD_COLS, t->rows, t->K, t->n_off, 0);
            else if (t->kind == SLOT_TRANSPOSE) add(srcs[i], t->dst, MDT_LOAD_TRANSPOSE, t->rows, t->K, 0, 0);
            else if (t->kind == SLOT_PACK_T) add(srcs[i], t->dst, MDT_LOAD_PACK_T, t->rows, t->K, 0, t->rows / 16);
            else if (t->kind == SLOT_PACK_SPLIT) {
                if (m->train) mark_split_stale(m);   // training: re-made on demand (mdt_model_types.h)
                else add(srcs[i], t->dst, MDT_LOAD_PACK_SPLIT, t->rows, t->K, t->n_off, 0);
            }
            else {
                add(srcs[i], t->dst, MDT_LOAD_PACK, t->rows, t->K, t->n_off, 0);
                if (t->lin && t->lin->wt) add(srcs[i], t->lin->wt, MDT_LOAD_PACK_T, t->rows, t->K, t->n_off, t->lin->N / 16);
            }
        }
        touched.push_back(slot);
    }
    if (!tab.empty()) {
        const size_t tab_bytes = (tab.size() * sizeof(mdt_load_entry) + 255) & ~(size_t)255;
        const size_t total = tab_bytes + blocks.size() * sizeof(int2);
        std::vector<char> bytes(total, 0);
        memcpy(bytes.data(), tab.data(), tab.size() * sizeof(mdt_load_entry));
        memcpy(bytes.data() + tab_bytes, blocks.data(), blocks.size() * sizeof(int2));
        if (total > m->tab_cap) {
            HIP_TRY(hipStreamSynchronize(s));  // rare: the first upload, or a larger parameter set than before
            if (m->tab_dev) (void)hipFree(m->tab_dev);
            for (int i = 0; i < 2; ++i)
                if (m->tab_host[i]) (void)hipHostFree(m->tab_host[i]);
            m->tab_dev = nullptr; m->tab_host[0] = m->tab_host[1] = nullptr; m->tab_cap = 0; m->tab_last.clear();
            m->tab_used = false;
            HIP_TRY(hipMalloc(&m->tab_dev, total * 2));
            for (int i = 0; i < 2; ++i) {
                HIP_TRY(hipHostMalloc(&m->tab_host[i], total * 2, hipHostMallocDefault));
                if (!m->ev_tab[i]) HIP_TRY(hipEventCreateWithFlags(&m->ev_tab[i], hipEventDisableTiming));
            }
            m->tab_cap = total * 2;
        }
        // the table buffer is shared by every stream that calls in: a caller on another stream than the last one first orders
        // itself behind that stream's copy and kernel (a hit would read a table still in flight, a miss overwrite one in use)
        if (!m->ev_tab_use) HIP_TRY(hipEventCreateWithFlags(&m->ev_tab_use, hipEventDisableTiming));
        if (m->tab_used && m->tab_stream != s) HIP_TRY(hipStreamWaitEvent(s, m->ev_tab_use, 0));
        if (m->tab_last != bytes) {  // new table: through the pinned buffer whose previous copy has long completed
            const int t = m->tab_turn ^= 1;
            HIP_TRY(hipEventSynchronize(m->ev_tab[t]));
            memcpy(m->tab_host[t], bytes.data(), total);
            HIP_TRY(hipMemcpyAsync(m->tab_dev, m->tab_host[t], total, hipMemcpyHostToDevice, s));
            HIP_TRY(hipEventRecord(m->ev_tab[t], s));
            m->tab_last.swap(bytes);
        }
        LAUNCH(mdt_launch_multi_load((const mdt_load_entry*)m->tab_dev, (const int2*)((const char*)m->tab_dev + tab_bytes),
                                     (int)blocks.size(), s));
        HIP_TRY(hipEventRecord(m->ev_tab_use, s));
        m->tab_stream = s; m->tab_used = true;
    }
    for (Slot* t : touched) t->loaded = true;
    m->cached_batch = 0;
    return MDT_OK;
}

static mdt_status check_loaded(const mdt_model* m) {
    for (const Slot& s : m->slots)
        if (!s.loaded) return fail(MDT_ERR_NOT_LOADED, "parameter '%s' was never loaded", s.name.c_str());
    return MDT_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
static void carve_ws(mdt_model* m, Bump& b, int64_t B) {
    const int64_t Re = (int64_t)m->Te * B, Ra = (int64_t)m->Ta * B, Rx = std::max(Re, Ra);
    const int64_t Rm = std::max<int64_t>(B, MAX_STEPS);
    const int D = m->D;
    m->h_enc = b.take(Re * D);
    m->qkv = b.take(Rx * 3 * D);
    m->att = b.take(Rx * D);
    m->hid = b.take(Rx * 4 * D);
    m->ctx = b.take(Re * D);
    m->kvx = b.take(Re * m->Ld * 2 * D);
    m->y = b.take(Ra * D);
    m->qx = b.take(Ra * D);
    m->sig_e = b.take(Rm * D);
    m->sig_t = b.take(Rm * 2 * D);
    m->sig_c = b.take(Rm * D);
    m->mod = b.take(m->cond == COND_ADALN ? Rm * m->Ld * 6 * D : 0);
    m->cmod = b.take(m->cond == COND_NOISE ? Rm * 2 * D : 0);
    m->xbuf = b.take(Ra * m->A);
    m->noised = b.take(Ra * m->A);
    m->Fbuf = b.take(Ra * m->A);
    m->steps = b.take(MAX_STEPS * 4);
    m->sigs = b.take(MAX_STEPS + 1);
    m->loss_part = b.take(MDT_LOSS_PARTS);
    if (m->xfold) {
        const int64_t np = (int64_t)4 * m->H;  // every head padded to 4 context tokens (fragment order, k_xattn_fold)
        m->xU = b.take((size_t)m->Ld * B * np * D);
        m->xW = b.take((size_t)m->Ld * B * np * D);
        m->xc = b.take((size_t)m->Ld * B * np);
    }
}

extern "C" mdt_status mdt_reserve(mdt_model* m, int64_t max_batch) {
    if (!m || max_batch < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_reserve: bad argument");
    if (max_batch <= m->cap) return MDT_OK;
    if (max_batch * std::max(m->Te, m->Ta) > (int64_t)1 << 24) return fail(MDT_ERR_INVALID_ARG, "batch too large");
    if (m->ws) {
        HIP_TRY(hipDeviceSynchronize());  // previous work may still read the old workspace
        HIP_TRY(mdt_dev_free(m->ws));
        m->ws = nullptr;
        m->cap = 0;
        m->cached_batch = 0;
    }
    Bump count;
    carve_ws(m, count, max_batch);
    HIP_TRY(mdt_dev_malloc((void**)&m->ws, count.off * sizeof(float)));
    Bump real;
    real.base = m->ws;
    carve_ws(m, real, max_batch);
    if (m->cond == COND_NOISE)  // the "scale" half of every [c | ones] row; the c half is rewritten per call
        HIP_TRY(hipMemsetD32((hipDeviceptr_t)m->cmod, 0x3f800000u, (size_t)std::max<int64_t>(max_batch, MAX_STEPS) * 2 * m->D));
    m->cap = max_batch;
    ++m->ws_generation;
    return MDT_OK;
}

extern "C" int64_t mdt_ws_generation(const mdt_model* m) { return m ? m->ws_generation : -1; }

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
static bool misaligned(const void* p) { return ((uintptr_t)p & 15) != 0; }

// one transformer Block on the encoder tokens / the self-attention + MLP halves of a ConditionedBlock
struct ModRef {
    const float* mod = nullptr;  // conditioning row base for this decoder block (nullptr: unconditioned Block)
    int64_t stride = 0;          // floats between the rows of consecutive samples (0: one row for the whole batch)
    int shift = -1, scale = -1, gate = -1;  // offsets inside the row; -1 = absent
    ModRef() {}
    ModRef(const float* m_, int64_t st, int sh, int sc, int g) : mod(m_), stride(st), shift(sh), scale(sc), gate(g) {}
};

// largest batch whose self-attention runs fused into the projection (MDT_HIP_ATTN_PROJ_MAX overrides; measured crossover)
static int64_t g_attn_proj_max_batch() {
    static int64_t v = -1;
    // (round 2, B = 2 / 4 / 8: 1.87 -> 1.71, 2.02 -> 1.85, 2.24 -> 2.07 ms; B = 16 lost then, 2.76 -> 2.83.  Round 5, with the fused
    //  kernel's attention on the MFMA pipe, its rows requested in one batch and its LDS cut to the T real rows -- two workgroups
    //  per CU: B = 12 / 16 / 20 / 24 / 32 1.653 / 1.649 / 1.918 / 2.006 / 2.017 -> 1.49 / 1.50 / 1.74 / 1.93 / 1.95 ms; B = 48 loses,
    //  2.30 -> 2.42)
    if (v < 0) { const char* e = getenv("MDT_HIP_ATTN_PROJ_MAX"); v = e ? atoll(e) : 32; }
    return v;
}

// the residual stream as the previous sublayer left it: one array (parts <= 1: V.y) or the partial slabs of a fused MLP
// launch, to be summed by whoever reads them next
struct Stream {
    const float* base = nullptr;
    int parts = 1;
    int64_t stride = 0;
};

// row count from which the self-attention runs in the prologue of its output projection (MDT_HIP_ATTN_WIDE_MIN; 0 disables)
static int g_attn_wide_override = -1;  // mdt_op_set_attn_wide_min (tests / A-B runs)
static int g_attn_wide_min_rows() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MDT_HIP_ATTN_WIDE_MIN"); v = e ? atoi(e) : 1401; if (v == 0) v = 1 << 30; }
    return g_attn_wide_override >= 0 ? (g_attn_wide_override == 0 ? 1 << 30 : g_attn_wide_override) : v;
}
extern "C" void mdt_op_set_attn_wide_min(int32_t rows) { g_attn_wide_override = rows; }

// row count from which (and batch up to which: two rounds of one workgroup per CU) one workgroup per sample runs self-attention,
// projection AND the collapsed cross-attention (k_attn_xattn); MDT_HIP_ATTN_XATTN_MIN (0 disables) / MDT_HIP_ATTN_XATTN_MAX_B;
// mdt_op_set_attn_wide_min(0) switches it off too
static int g_attn_xattn_min_rows() {
    static int v = -1;
    // (round 3 started it where the fused MLP launch starts, 1401 rows; measured down the batch sizes in round 4 -- tools/latency.py,
    //  profiles/r04_lowbatch.txt -- it beats k_attn + projection + k_xattn_apply from 200 rows on: B = 20 2.03 -> 1.99 ms, 32 2.13 -> 2.09,
    //  40 2.48 -> 2.35, 80 3.12 -> 3.03, 100 3.26 -> 3.17, 128 3.62 -> 3.52, 140 4.33 -> 4.14; at 160 rows it loses, 1.71 -> 1.75)
    if (v < 0) { const char* e = getenv("MDT_HIP_ATTN_XATTN_MIN"); v = e ? atoi(e) : 200; if (v == 0) v = 1 << 30; }
    return g_attn_wide_override == 0 ? 1 << 30 : v;
}
static int g_attn_xattn_max_batch() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MDT_HIP_ATTN_XATTN_MAX_B"); v = e ? atoi(e) : 512; }
    return v;
}

// `fx` (optional): the collapsed cross-attention that follows on the same rows; when the one-sample-per-workgroup kernel takes
// both, *fused is set and the caller skips its own launch.
static mdt_status run_self_attn(mdt_model* m, const EncBlock& e, const View& V, int64_t B, int T, bool causal,
                                ModRef mr, hipStream_t s, Stream in = Stream(), const mdt_xapply_args* fx = nullptr,
                                bool* fused = nullptr) {
    float* x = V.y;
    const int D = m->D, M = (int)(B * T);
    mdt_gemm_args g = gemm_args(x, D, e.qkv, V.qkv, 3 * D, M);
    g.ln = 1; g.ln_w = e.ln1_w; g.ln_b = e.ln1_b;
    g.rows_per_sample = T;
    if (in.parts > 1) {  // rows = sum of the MLP slabs; the column-0 tiles also leave the sum in V.y for the residual below
        g.A = in.base; g.a_parts = in.parts; g.a_part_stride = in.stride; g.a_merged = x;
    }
    if (mr.mod && mr.shift >= 0) { g.mod = mr.mod; g.mod_stride = mr.stride; g.shift_off = mr.shift; g.scale_off = mr.scale; }
    LAUNCH(mdt_launch_gemm(g, s));
    mdt_gemm_args p = gemm_args(V.att, D, e.proj, x, D, M);
    p.residual = 1; p.rows_per_sample = T;
    if (mr.mod && mr.gate >= 0) { p.mod = mr.mod; p.mod_stride = mr.stride; p.gate_off = mr.gate; }
    if (B <= g_attn_proj_max_batch() && mdt_attn_proj_supported(p, m->H, m->hd, T, m->cfg.use_rot_embed)) {
        // rollout batch: attention and projection in one launch (the attention output never leaves the workgroup)
        LAUNCH(mdt_launch_attn_proj(p, V.qkv, 3 * D, m->H, m->hd, T, causal, s));
        return MDT_OK;
    }
    if (fx && fused && M >= g_attn_xattn_min_rows() && B <= g_attn_xattn_max_batch() &&
        mdt_attn_xattn_supported(p, *fx, m->H, m->hd, T, causal, m->cfg.use_rot_embed)) {
        LAUNCH(mdt_launch_attn_xattn(p, V.qkv, 3 * D, *fx, m->H, m->hd, T, s));
        *fused = true;
        return MDT_OK;
    }
    // (up to one workgroup per CU: its 156 KB of LDS allow no second one, so beyond 256 tiles -- B > 272 -- the workgroups
    //  queue up behind each other while the plain projection keeps three per CU in flight: B = 512 10.35 vs 10.13 ms)
    const int64_t wide_tiles = (int64_t)((M + 31) / 32) * ((D + 127) / 128);
    if (M >= g_attn_wide_min_rows() && (wide_tiles <= 256 || g_attn_wide_override > 0) &&
        mdt_attn_proj_wide_supported(p, m->H, m->hd, T, causal, m->cfg.use_rot_embed)) {
        // large batch: the causal attention of each 32-row tile in the projection's prologue (no attention launch, no
        // round trip of the attention output)
        LAUNCH(mdt_launch_attn_proj_wide(p, V.qkv, 3 * D, m->H, m->hd, T, s));
        return MDT_OK;
    }
    mdt_attn_args a;
    memset(&a, 0, sizeof a);
    a.q = V.qkv; a.ldq = 3 * D; a.k = V.qkv + D; a.v = V.qkv + 2 * D; a.ldkv = 3 * D;
    a.out = V.att; a.ldo = D; a.B = (int)B; a.H = m->H; a.hd = m->hd; a.Tq = T; a.Tk = T;
    a.causal = causal; a.rope = m->cfg.use_rot_embed;
    LAUNCH(mdt_launch_attention(a, m->rope_cos, m->rope_sin, s));
    LAUNCH(mdt_launch_gemm(p, s));
    return MDT_OK;
}

// row count from which the MLP sublayer runs as ONE launch (k_mlp) that leaves partial slabs instead of the residual
// stream (MDT_HIP_MLP_FUSE_MIN overrides; 0 disables): below it the wide tiles do not fill the chip
static int g_mlp_fuse_override = -1;  // mdt_op_set_mlp_fuse_min (tests / A-B runs)
// (`split`: the launch would run in its bf16 split form, which pays from fewer rows -- mdt_split_min_rows(), B = 128: 1280 rows
// per call 3.42 -> 2.96 ms with the qkv products split too (mdt_internal.h); an explicit setting, hook or environment, is taken as it is)
static int g_mlp_fuse_min_rows(bool split = false) {
    static int v = -2;
    if (v == -2) { const char* e = getenv("MDT_HIP_MLP_FUSE_MIN"); v = e ? atoi(e) : -1; if (v == 0) v = 1 << 30; }
    if (g_mlp_fuse_override >= 0) return g_mlp_fuse_override == 0 ? 1 << 30 : g_mlp_fuse_override;
    if (v >= 0) return v;
    return split ? std::min(1401, mdt_split_min_rows()) : 1401;
}
extern "C" void mdt_op_set_mlp_fuse_min(int32_t rows) { g_mlp_fuse_override = rows; }

// mdt_op_trace_mlp: every fused-MLP launch (k_mlp, the dominant kernel of the B = 256 sampler call) of the model-level calls
// that follow is bracketed by a pair of HIP events on its stream -- the kernel's duration INSIDE the launch chain (the
// preceding kernel's tail and the launch gap included, like a kernel-trace row), which 200 back-to-back launches of the
// kernel alone do not give (bench.py).  mdt_op_trace_mlp_read synchronises the events, returns the
// durations in microseconds (at most `cap`) and releases them.
static bool g_trace_mlp = false;
struct TraceEv { hipEvent_t e0, e1, e2; };   // e0 .. e1 bracket the launch; e1 .. e2 bracket NOTHING (the cost of a bracket itself)
static std::vector<TraceEv> g_trace_mlp_events;
static std::vector<float> g_trace_mlp_empty;
extern "C" void mdt_op_trace_mlp(int32_t enable) { g_trace_mlp = enable != 0; }
extern "C" int32_t mdt_op_trace_mlp_read(float* us, int32_t cap) {
    int32_t n = 0;
    g_trace_mlp_empty.clear();
    for (auto& ev : g_trace_mlp_events) {
        float ms = 0.f, ms0 = 0.f;
        if (hipEventSynchronize(ev.e2) == hipSuccess && hipEventElapsedTime(&ms, ev.e0, ev.e1) == hipSuccess &&
            hipEventElapsedTime(&ms0, ev.e1, ev.e2) == hipSuccess && us && n < cap) {
            us[n++] = ms * 1e3f;
            g_trace_mlp_empty.push_back(ms0 * 1e3f);
        }
        (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); (void)hipEventDestroy(ev.e2);
    }
    g_trace_mlp_events.clear();
    return n;
}
// the EMPTY brackets recorded behind each traced launch (same order, same count as the last mdt_op_trace_mlp_read): what a pair
// of events costs on the stream with nothing in between -- subtracted, the bracketed time is the kernel's own (a kernel trace's row)
extern "C" int32_t mdt_op_trace_mlp_read_empty(float* us, int32_t cap) {
    int32_t n = 0;
    for (float v : g_trace_mlp_empty)
        if (us && n < cap) us[n++] = v;
    return n;
}

// mdt_op_clock_stamp: the two clock counters of the wave that runs it, in stream order (include/mdt_hip_ops.h)
// The shader-clock counter is per XCD (eight of them, not synchronised with each other): 64 one-wave workgroups go out, each
// writes its pair into the slot of the XCD it landed on (the last writer of a slot wins: all within a microsecond), so that a
// caller can difference two stamps XCD by XCD whatever else shares the chip.
__global__ void k_clock_stamp(uint64_t* __restrict__ out) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        out[2 * xcc] = __builtin_amdgcn_s_memtime();
        out[2 * xcc + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
extern "C" mdt_status mdt_op_clock_stamp(uint64_t* out16, void* stream) {
    if (!out16) return fail(MDT_ERR_INVALID_ARG, "mdt_op_clock_stamp: null output");
    hipLaunchKernelGGL(k_clock_stamp, dim3(64), dim3(64), 0, (hipStream_t)stream, out16);
    HIP_TRY(hipGetLastError());
    return MDT_OK;
}

// the split images of every block, re-made from the fp32 fragment images (after parameter loads that skipped them: split_stale)
static mdt_status refresh_split(mdt_model* m, hipStream_t s) {
    auto one = [&](const Lin& l) -> mdt_status {
        if (l.ws) LAUNCH(mdt_launch_split_from_packed(l.wp, l.N, l.K, l.ws, s));
        return MDT_OK;
    };
    for (const EncBlock& e : m->enc) { MDT_TRY(one(e.qkv)); MDT_TRY(one(e.fc)); MDT_TRY(one(e.proj2)); }
    for (const DecBlock& d : m->dec) { MDT_TRY(one(d.qkv)); MDT_TRY(one(d.fc)); MDT_TRY(one(d.proj2)); }
    m->split_stale = false;
    return MDT_OK;
}
static inline mdt_status split_ready(mdt_model* m, int64_t rows, hipStream_t s) {
    if (m->split_stale && mdt_mlp_split_enabled() && rows >= mdt_split_min_rows()) return refresh_split(m, s);
    return MDT_OK;
}

// `out` (optional): when given and the fused launch applies, the sublayer's output is left as slabs in V.hid (described in
// *out) and V.y is NOT updated -- the caller hands *out to the next reader; otherwise V.y is updated in place.
// `pre_x` (optional; rollout batches): the collapsed cross-attention that is still to run on these rows -- it goes into the
// c_fc launch (k_xattn_gemm_smallm); the caller has checked mdt_xattn_gemm_supported.
static mdt_status run_mlp(mdt_model* m, const EncBlock& e, const View& V, int64_t B, int T, ModRef mr, hipStream_t s,
                          Stream* out = nullptr, const mdt_xapply_args* pre_x = nullptr) {
    float* x = V.y;
    const int D = m->D, M = (int)(B * T);
    mdt_gemm_args g = gemm_args(x, D, e.fc, V.hid, 4 * D, M);
    g.ln = 1; g.ln_w = e.ln2_w; g.ln_b = e.ln2_b; g.act = MDT_ACT_GELU;
    g.rows_per_sample = T;
    if (mr.mod && mr.shift >= 0) { g.mod = mr.mod; g.mod_stride = mr.stride; g.shift_off = mr.shift; g.scale_off = mr.scale; }
    mdt_gemm_args p = gemm_args(V.hid, 4 * D, e.proj2, pre_x ? pre_x->y_out : x, D, M);  // pre_x: the residual stream moves to its y_out
    p.residual = 1; p.rows_per_sample = T;
    if (mr.mod && mr.gate >= 0) { p.mod = mr.mod; p.mod_stride = mr.stride; p.gate_off = mr.gate; }
    if (out) *out = Stream();
    // (never with `pre_x`: the caller skipped its own cross-attention launch because the c_fc launch was to run it)
    // the three-way bf16 split form of the launch (mdt_mlp_split.h) wherever its images exist and it is not switched off; it pays
    // from fewer rows than the fp32 launch (g_mlp_fuse_min_rows)
    const bool split = e.fc.ws && e.proj2.ws && mdt_mlp_split_enabled() && mdt_mlp_split_supported(g, p);
    if (out && !pre_x && M >= g_mlp_fuse_min_rows(split) && mdt_mlp_slices(D) >= 2 && mdt_mlp_supported(g, p)) {
        // the hidden buffer (M x 4D) is free in this form: it holds the S <= 4 slabs of (M x D)
        p.ldo = D;
        const int64_t stride = (int64_t)M * D;
        if (g_trace_mlp) {  // measurement hook (mdt_op_trace_mlp): this launch between its own pair of HIP events, inside the chain
            TraceEv ev;
            HIP_TRY(hipEventCreate(&ev.e0)); HIP_TRY(hipEventCreate(&ev.e1)); HIP_TRY(hipEventCreate(&ev.e2));
            HIP_TRY(hipEventRecord(ev.e0, s));
            if (split) LAUNCH(mdt_launch_mlp_split(g, p, e.fc.ws, e.proj2.ws, V.hid, stride, s));
            else LAUNCH(mdt_launch_mlp(g, p, V.hid, stride, s));
            HIP_TRY(hipEventRecord(ev.e1, s));
            HIP_TRY(hipEventRecord(ev.e2, s));
            g_trace_mlp_events.push_back(ev);
        } else if (split) {
            LAUNCH(mdt_launch_mlp_split(g, p, e.fc.ws, e.proj2.ws, V.hid, stride, s));
        } else {
            LAUNCH(mdt_launch_mlp(g, p, V.hid, stride, s));
        }
        out->base = V.hid; out->parts = mdt_mlp_slices(D); out->stride = stride;
        return MDT_OK;
    }
    if (pre_x) LAUNCH(mdt_launch_xattn_gemm(*pre_x, g, s));
    else LAUNCH(mdt_launch_gemm(g, s));
    LAUNCH(mdt_launch_gemm(p, s));
    return MDT_OK;
}

// batch up to which the cross-attention of a decoder block runs inside the c_fc launch that follows it (rollout batches;
// MDT_HIP_XATTN_FC_MAX_B, 0 disables)
static int g_xattn_fc_max_batch() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MDT_HIP_XATTN_FC_MAX_B"); v = e ? atoi(e) : 2; }  // B = 4: 1.67 vs 1.58 ms per call
    return v;
}

// sigma_emb: sinusoidal(ln(sigma)/4) -> Linear -> Mish -> Linear for R sigmas (mdtv_transformer.py:105-110,282-288);
// the second Linear's output rows go to out (leading dimension ldo, row r -> row r*gout) after `act`.
// emb_done: m->sig_e already holds the R embeddings (the sampler's one-launch preparation wrote them)
// side: the GEMMs are queued as side jobs (mdt_gemm_side_push: they ride in the launches of the small-M products that follow;
// the caller flushes the queue before anything reads their outputs)
static mdt_status run_sigma_mlp(mdt_model* m, const float* sigma, int64_t sstride, int R, float* out, int64_t ldo,
                                int gout, int act, hipStream_t s, bool emb_done = false, bool side = false) {
    const int D = m->D;
    if (!emb_done) LAUNCH(mdt_launch_sigma_emb(sigma, sstride, m->freqs, m->sig_e, R, D, s));
    mdt_gemm_args a = gemm_args(m->sig_e, D, m->sig1, m->sig_t, 2 * D, R);
    a.act = MDT_ACT_MISH;
    LAUNCH(side ? mdt_gemm_side_push(a, s) : mdt_launch_gemm(a, s));
    mdt_gemm_args b = gemm_args(m->sig_t, 2 * D, m->sig3, out, ldo, R);
    b.act = act;
    b.gin = 1; b.gout = gout; b.goff = 0;
    LAUNCH(side ? mdt_gemm_side_push(b, s) : mdt_launch_gemm(b, s));
    return MDT_OK;
}

// argument checks of every entry point that encodes a context (also run by mdt_sample_ddim BEFORE it enqueues anything, so
// that an invalid-argument return leaves no device work behind)
static mdt_status check_encode_args(const mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                    const float* ctx_out) {
    if (!tokens || !goal) return fail(MDT_ERR_INVALID_ARG, "encode: null tokens/goal");
    if (m->cfg.arch == MDT_ARCH_MDT && !tokens2) return fail(MDT_ERR_INVALID_ARG, "encode: MDT needs the gripper tokens");
    if (m->p_row >= 0 && !tokens2)
        return fail(MDT_ERR_INVALID_ARG, "encode: this handle was created with use_proprio; state_obs (tokens2) is required");
    if (misaligned(tokens) || misaligned(goal) || misaligned(tokens2) || misaligned(ctx_out))
        return fail(MDT_ERR_INVALID_ARG, "encode: pointers must be 16-byte aligned");
    return MDT_OK;
}

// sigma / sstride: only read when the context starts with the sigma token (COND_TOKEN); sstride 0 = one sigma for
// the whole batch, 1 = one per sample.
static mdt_status run_encode(mdt_model* m, const float* tokens, const float* tokens2, const float* goal, int modality,
                             int honour_modality, int64_t B, const float* sigma, int64_t sstride, float* ctx_out,
                             hipStream_t s) {
    const mdt_config& c = m->cfg;
    const int D = m->D, Te = m->Te;
    const int t0 = m->sig_tok;  // context row of the goal token
    if (t0 && !sigma)
        return fail(MDT_ERR_INVALID_ARG, "encode: use_ada_conditioning=False puts sigma into the context; sigma is required");
    MDT_TRY(check_encode_args(m, tokens, tokens2, goal, ctx_out));
    MDT_TRY(check_loaded(m));
    MDT_TRY(mdt_reserve(m, B));
    MDT_TRY(split_ready(m, B * std::max(m->Te, m->Ta), s));
    m->cached_batch = 0;
    const bool lang = honour_modality && c.use_modality_encoder && modality == MDT_MODALITY_LANG;
    const Lin& g0 = lang ? m->lang0 : m->goal0;
    const Lin& g2 = lang ? m->lang2 : m->goal2;
    const float* pos0 = (c.arch == MDT_ARCH_MDT && c.use_abs_pos_emb) ? m->pos_emb : nullptr;
    const float* pos1 = pos0 ? m->pos_emb + (int64_t)c.goal_seq_len * D : nullptr;
    // sigma token -> row 0 of every sample's context           (concatenate_inputs, mdtv_transformer.py:296-297)
    if (t0) MDT_TRY(run_sigma_mlp(m, sigma, sstride, (int)B, m->h_enc, D, Te, MDT_ACT_NONE, s));
    // rollout batches: the state-token embedding does not depend on the goal embedding -- it rides in the goal GEMM's launch
    // (mdt_gemm_side_push_front; both are small-M products then, so the goal GEMM below takes it along)
    const bool tok_side = c.arch == MDT_ARCH_MDTV && m->g_row >= 0 && B * m->n_tok <= 15;
    const size_t side_before = mdt_gemm_side_pending();
    if (tok_side) {
        mdt_gemm_args a = gemm_args(tokens, m->O, m->tok, m->h_enc, D, (int)(B * m->n_tok));
        a.gin = m->n_tok; a.gout = Te; a.goff = m->tok_row;
        LAUNCH(mdt_gemm_side_push_front(a, s));
    }
    // goal token -> its row                                     (process_goal_embeddings, mdtv_transformer.py:268)
    if (m->g_row >= 0) {
        const float* gin = goal;
        int64_t ld = m->G;
        if (c.use_mlp_goal) {
            mdt_gemm_args a = gemm_args(goal, m->G, g0, m->hid, 2 * D, (int)B);
            a.act = MDT_ACT_GELU;
            LAUNCH(mdt_launch_gemm(a, s));
            gin = m->hid; ld = 2 * D;
        }
        mdt_gemm_args a = gemm_args(gin, ld, g2, m->h_enc, D, (int)B);
        a.gin = 1; a.gout = Te; a.goff = m->g_row; a.rowvec = pos0;
        LAUNCH(mdt_launch_gemm(a, s));
    }
    // The goal product above normally took the queued token embedding along.  Where it did not (a goal width or an override
    // -- MDT_HIP_SMALLM_MAX / _ROWS / _TILES -- that routes it away from the small-M kernel), the job must not stay at the head of
    // the queue: the next small-M launch would be the encoder's first LayerNorm product, which READS the rows the job writes.
    if (tok_side && mdt_gemm_side_pending() > side_before) LAUNCH(mdt_gemm_side_launch_front(s));
    // state tokens -> the rows after it                         (process_state_embeddings, :260 / mdt :300)
    if (c.arch == MDT_ARCH_MDTV) {
        if (!tok_side) {
            mdt_gemm_args a = gemm_args(tokens, m->O, m->tok, m->h_enc, D, (int)(B * m->n_tok));
            a.gin = m->n_tok; a.gout = Te; a.goff = m->tok_row;
            LAUNCH(mdt_launch_gemm(a, s));
        }
    } else {
        mdt_gemm_args a = gemm_args(tokens, m->O, m->tok, m->h_enc, D, (int)B);
        a.gin = 1; a.gout = Te; a.goff = m->tok_row; a.rowvec = pos1;
        LAUNCH(mdt_launch_gemm(a, s));
        mdt_gemm_args b2 = gemm_args(tokens2, m->O, m->incam, m->h_enc, D, (int)B);
        b2.gin = 1; b2.gout = Te; b2.goff = m->tok_row + 1; b2.rowvec = pos1;
        LAUNCH(mdt_launch_gemm(b2, s));
    }
    if (m->p_row >= 0) {  // proprioceptive token -> the last row          (process_state_embeddings, :260-266)
        LAUNCH(mdt_launch_narrow_linear(tokens2, m->prop0_T, m->prop0_b, nullptr, m->hid, (int)B, m->Pd, 2 * D, MDT_ACT_MISH, s));
        mdt_gemm_args a = gemm_args(m->hid, 2 * D, m->prop2, m->h_enc, D, (int)B);
        a.gin = 1; a.gout = Te; a.goff = m->p_row;
        LAUNCH(mdt_launch_gemm(a, s));
    }
    for (int l = 0; l < m->Le; ++l) {
        MDT_TRY(run_self_attn(m, m->enc[l], encoder_view(m), B, Te, false, ModRef(), s));
        MDT_TRY(run_mlp(m, m->enc[l], encoder_view(m), B, Te, ModRef(), s));
    }
    LAUNCH(mdt_launch_layernorm(m->h_enc, m->enc_ln_w, m->enc_ln_b, m->ctx, (int)(B * Te), D, s, ctx_out));  // both copies in one launch
    // cross-attention K|V of all decoder blocks in one GEMM (sigma independent: hoisted out of the step loop)
    {
        mdt_gemm_args a = gemm_args(m->ctx, D, m->kv_all, m->kvx, (int64_t)m->Ld * 2 * D, (int)(B * Te));
        LAUNCH(mdt_launch_gemm(a, s));
    }
    if (m->xfold) {  // fold K|V into the cross-attention projections, once per context (sigma independent)
        const int64_t np = (int64_t)4 * m->H;
        std::vector<mdt_xfold_args> sets(m->Ld);
        for (int l = 0; l < m->Ld; ++l) {
            mdt_xfold_args& f = sets[l];
            memset(&f, 0, sizeof f);
            f.kv = m->kvx + (int64_t)l * 2 * D; f.ldkv = (int64_t)m->Ld * 2 * D;
            f.WqT_p = m->dec[l].xq_pT; f.bq = m->dec[l].xq.bias; f.Wo_p = m->dec[l].xproj.wp;
            f.U = m->xU + (int64_t)l * m->cap * np * D; f.Wf = m->xW + (int64_t)l * m->cap * np * D;
            f.c = m->xc + (int64_t)l * m->cap * np;
            f.B = (int)B; f.H = m->H; f.hd = m->hd; f.D = D; f.Te = Te;
        }
        LAUNCH(mdt_launch_xattn_fold_n(sets.data(), m->Ld, s));  // every decoder block in one launch
    }
    m->cached_batch = B;
    return MDT_OK;
}

// The decoder's conditioning rows for R sigmas.
//   COND_ADALN: mod (R, Ld*6D) = adaLN_zero Linear(SiLU(c)) of every block, one stacked GEMM
//   COND_NOISE: cmod (R, 2D)   = [c | ones]
//   COND_TOKEN: nothing (sigma lives in the context)
static mdt_status run_modulation(mdt_model* m, const float* sigma, int64_t sstride, int R, hipStream_t s, bool emb_done = false,
                                 bool side = false) {
    const int D = m->D;
    if (m->cond == COND_TOKEN) return MDT_OK;
    if (m->cond == COND_NOISE) return run_sigma_mlp(m, sigma, sstride, R, m->cmod, 2 * D, 1, MDT_ACT_NONE, s, emb_done, side);
    // AdaLNZero applies SiLU to c before its Linear; c itself is used nowhere else
    MDT_TRY(run_sigma_mlp(m, sigma, sstride, R, m->sig_c, D, 1, MDT_ACT_SILU, s, emb_done, side));
    mdt_gemm_args c = gemm_args(m->sig_c, D, m->mod_all, m->mod, (int64_t)m->Ld * 6 * D, R);
    LAUNCH(side ? mdt_gemm_side_push(c, s) : mdt_launch_gemm(c, s));
    return MDT_OK;
}

// conditioning rows of step / sigma-row r (see run_modulation) and their stride across samples
static const float* cond_row(const mdt_model* m, int64_t r) {
    if (m->cond == COND_ADALN) return m->mod + r * (int64_t)m->Ld * 6 * m->D;
    if (m->cond == COND_NOISE) return m->cmod + r * 2 * (int64_t)m->D;
    return nullptr;
}
static int64_t cond_width(const mdt_model* m) {
    return m->cond == COND_ADALN ? (int64_t)m->Ld * 6 * m->D : (m->cond == COND_NOISE ? 2 * (int64_t)m->D : 0);
}

// the Ld decoder blocks on the residual stream V.y: ConditionedBlock (transformer_blocks.py:291-309), NoiseBlock
// (:335-341) or the plain cross-attending Block (:209-214), by m->cond
// `fin` (optional): the caller's next reader (the action head) can sum MLP slabs itself; then the last block may leave its
// output as slabs, described in *fin.  Without it the residual stream ends in V.y.
static mdt_status run_decoder_blocks(mdt_model* m, const View& V, int64_t B, const float* mod_row, int64_t mod_stride,
                                     hipStream_t s, Stream* fin = nullptr) {
    const int D = m->D, Ta = m->Ta, M = (int)(B * Ta);
    View W = V;  // W.y / W.att trade places whenever a block's cross-attention runs inside its c_fc launch (rollout batches)
    Stream cur;  // where the residual stream lives between blocks
    if (fin) *fin = Stream();
    MDT_TRY(split_ready(m, M, s));
    for (int l = 0; l < m->Ld; ++l) {
        const DecBlock& d = m->dec[l];
        ModRef ma, mx, mm;  // self-attention half, cross-attention query, MLP half
        if (m->cond == COND_ADALN) {
            const float* row = mod_row + (int64_t)l * 6 * D;
            ma = ModRef(row, mod_stride, 0, D, 2 * D);
            mm = ModRef(row, mod_stride, 3 * D, 4 * D, 5 * D);
        } else if (m->cond == COND_NOISE) {
            ma = mx = ModRef(mod_row, mod_stride, 0, D, -1);  // ln(x)*1 + c, residual ungated; MLP unconditioned
        }
        // cross attention: ln3 (biased LayerNorm) -> q ; K|V precomputed ; never gated
        mdt_xapply_args x;
        memset(&x, 0, sizeof x);
        if (m->xfold) {
            const int64_t np = (int64_t)4 * m->H;
            x.y = W.y; x.ln_w = d.ln3_w; x.ln_b = d.ln3_b; x.bo = d.xproj.bias;
            x.U = m->xU + ((int64_t)l * m->cap + V.b0) * np * D;
            x.Wf = m->xW + ((int64_t)l * m->cap + V.b0) * np * D;
            x.c = m->xc + ((int64_t)l * m->cap + V.b0) * np;
            x.B = (int)B; x.H = m->H; x.D = D; x.Te = m->Te; x.Ta = Ta;
        }
        // rollout batches: the cross-attention runs inside the c_fc launch (run_mlp below), if that launch is the small-M kernel
        // (the workgroups that repeat the cross-attention read W.y while one of them writes: the new rows go to W.att, which is
        //  free here, and the two trade places -- an even number of blocks ends in V.y again) ...
        bool x_in_fc = false;
        if (m->xfold) {
            mdt_gemm_args gf = gemm_args(W.y, D, d.fc, W.hid, 4 * D, M);
            gf.ln = 1; gf.rows_per_sample = Ta;
            x.y_out = W.att;
            x_in_fc = B <= g_xattn_fc_max_batch() && M <= 192 && m->Ld % 2 == 0 && mdt_xattn_gemm_supported(x, gf);
            if (!x_in_fc) x.y_out = nullptr;
        }
        bool xdone = false;
        MDT_TRY(run_self_attn(m, d, W, B, Ta, true, ma, s, cur, m->xfold ? &x : nullptr, &xdone));
        cur = Stream();
        if (xdone) {
            // self-attention, projection and cross-attention went as one launch (one workgroup per sample)
        } else if (m->xfold) {
            if (!x_in_fc) LAUNCH(mdt_launch_xattn_apply(x, s));
        } else {
            mdt_gemm_args q = gemm_args(W.y, D, d.xq, W.qx, D, M);
            q.ln = 1; q.ln_w = d.ln3_w; q.ln_b = d.ln3_b; q.rows_per_sample = Ta;
            if (mx.mod) { q.mod = mx.mod; q.mod_stride = mx.stride; q.shift_off = mx.shift; q.scale_off = mx.scale; }
            LAUNCH(mdt_launch_gemm(q, s));
            mdt_attn_args a;
            memset(&a, 0, sizeof a);
            a.q = W.qx; a.ldq = D;
            a.k = W.kvx + (int64_t)l * 2 * D; a.v = a.k + D; a.ldkv = (int64_t)m->Ld * 2 * D;
            a.out = W.att; a.ldo = D; a.B = (int)B; a.H = m->H; a.hd = m->hd; a.Tq = Ta; a.Tk = m->Te;
            a.causal = 1;  // SDPA is_causal on a Ta x Te matrix: top-left aligned (transformer_blocks.py:204,142)
            a.rope = m->cfg.use_rot_embed;
            LAUNCH(mdt_launch_attention(a, m->rope_cos, m->rope_sin, s));
            mdt_gemm_args p = gemm_args(W.att, D, d.xproj, W.y, D, M);
            p.residual = 1; p.rows_per_sample = Ta;
            LAUNCH(mdt_launch_gemm(p, s));
        }
        const bool last = l == m->Ld - 1;
        MDT_TRY(run_mlp(m, d, W, B, Ta, mm, s, (!last || fin) ? &cur : nullptr, x_in_fc ? &x : nullptr));
        if (x_in_fc) std::swap(W.y, W.att);
    }
    if (fin) *fin = cur;
    return MDT_OK;
}

static mdt_head_args head_args(mdt_model* m, const float* y, int64_t B, const float* x, const float* sigma,
                               int64_t sstride, float* out, int mode) {
    mdt_head_args h;
    memset(&h, 0, sizeof h);
    h.y = y; h.ln_w = m->dec_ln_w; h.ln_b = m->dec_ln_b; h.Wp = m->Wp; h.bp = m->bp;
    h.x = x; h.sigma = sigma; h.sigma_stride = sstride; h.out = out;
    h.M = (int)(B * m->Ta); h.D = m->D; h.A = m->A; h.rows_per_sample = m->Ta; h.mode = mode;
    h.sigma_data = m->cfg.sigma_data;
    return h;
}

// The action head on the rows of `h.y`.  Linear head: ONE launch (decoder LN, action_pred, EDM combine, DDIM update,
// next step's embedding).  MLP head (linear_output = 0): decoder LN -> action_pred.0 + GELU on the GEMM ->
// action_pred.2 and the rest in the head kernel reading the hidden layer as it is; `scratch` holds M * (D + HP)
// floats (the slice's MLP hidden buffer is free at this point); the next step's embedding is its own launch then.
static mdt_status run_head(mdt_model* m, mdt_head_args h, float* scratch, const float* sigma_next, hipStream_t s) {
    if (m->HP == 0) {
        LAUNCH(mdt_launch_head(h, s));
        return MDT_OK;
    }
    const int D = m->D, HP = m->HP;
    float* ln = scratch;
    float* hh = scratch + (int64_t)h.M * D;
    LAUNCH(mdt_launch_layernorm(h.y, m->dec_ln_w, m->dec_ln_b, ln, h.M, D, s));
    mdt_gemm_args g = gemm_args(ln, D, m->head0, hh, HP, h.M);
    g.act = MDT_ACT_GELU;
    LAUNCH(mdt_launch_gemm(g, s));
    float* y_next = h.y_next;
    h.y = hh; h.D = HP; h.no_ln = 1; h.y_next = nullptr;
    LAUNCH(mdt_launch_head(h, s));
    if (y_next)
        LAUNCH(mdt_launch_action_embed(h.out, sigma_next, 0, m->cfg.sigma_data, m->Wa, m->ba, y_next, h.M, m->A, D,
                                       m->Ta, s));
    return MDT_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI: model level
// ------------------------------------------------------------------------------------------------
extern "C" mdt_status mdt_encode(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                 int32_t modality, int32_t honour_modality, const float* sigma, int64_t batch,
                                 float* ctx_out, void* stream) {
    if (!m || batch < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_encode: bad argument");
    return run_encode(m, tokens, tokens2, goal, modality, honour_modality, batch, sigma, 1, ctx_out, (hipStream_t)stream);
}

extern "C" mdt_status mdt_denoise_cached(mdt_model* m, const float* x, const float* sigma, int64_t batch,
                                         int32_t flags, float* out, void* stream) {
    if (!m || !x || !sigma || !out || batch < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_denoise_cached: bad argument");
    if (m->cached_batch != batch)
        return fail(MDT_ERR_STATE, "mdt_denoise_cached: no cached context for batch %lld (last mdt_encode batch: %lld)",
                    (long long)batch, (long long)m->cached_batch);
    hipStream_t s = (hipStream_t)stream;
    const bool scalar = (flags & MDT_SIGMA_SCALAR) != 0;
    const int64_t modw = cond_width(m);
    const int64_t sst = scalar ? 0 : 1;  // stride of sigma / of the modulation rows across samples
    MDT_TRY(run_modulation(m, sigma, 1, scalar ? 1 : (int)batch, s));
    LAUNCH(mdt_launch_action_embed(x, (flags & MDT_RAW_INPUT) ? nullptr : sigma, sst, m->cfg.sigma_data, m->Wa, m->ba,
                                   m->y, (int)(batch * m->Ta), m->A, m->D, m->Ta, s));
    Stream fin;
    const bool head_sums = m->HP == 0 && m->A <= 8;  // the one-launch head adds MLP slabs itself
    MDT_TRY(run_decoder_blocks(m, decoder_view(m, 0), batch, cond_row(m, 0), scalar ? 0 : modw, s, head_sums ? &fin : nullptr));
    mdt_head_args h = head_args(m, m->y, batch, x, sigma, sst, out, (flags & MDT_RAW_OUTPUT) ? MDT_HEAD_RAW : MDT_HEAD_DENOISED);
    if (fin.parts > 1) { h.y = fin.base; h.y_parts = fin.parts; h.y_part_stride = fin.stride; }
    return run_head(m, h, m->hid, nullptr, s);
}

extern "C" mdt_status mdt_forward(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                  int32_t modality, const float* x, const float* sigma, int64_t batch, float* out,
                                  float* ctx_out, void* stream) {
    if (!m) return fail(MDT_ERR_INVALID_ARG, "mdt_forward: null handle");
    // MDTTransformer.forward -> enc_only_forward always embeds the goal with goal_emb (mdt_transformer.py:215)
    const int honour = m->cfg.arch == MDT_ARCH_MDTV;
    MDT_TRY(mdt_encode(m, tokens, tokens2, goal, modality, honour, sigma, batch, ctx_out, stream));
    return mdt_denoise_cached(m, x, sigma, batch, 0, out, stream);
}

// sigmas_host or sigmas_dev (exactly one non-null): the n_steps + 1 noise levels
static mdt_status sample_ddim_impl(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                   int32_t modality, const float* x_T, const float* sigmas, const float* sigmas_dev,
                                   int32_t n_steps, int64_t batch, float* out, float* ctx_out, void* stream) {
    if (!m || !x_T || (!sigmas && !sigmas_dev) || !out || batch < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_sample_ddim: bad argument");
    if (n_steps < 1 || n_steps > MAX_STEPS) return fail(MDT_ERR_INVALID_ARG, "n_steps must be 1..%d", MAX_STEPS);
    hipStream_t s = (hipStream_t)stream;
    const int honour = m->cfg.arch == MDT_ARCH_MDTV;
    const bool per_step_ctx = m->cond == COND_TOKEN;  // sigma is a context token: the encoder cannot be hoisted
    // ---- cut the batch into sample-aligned slices (multiples of 16 samples = 5 row tiles) on separate streams ----
    int ways = per_step_ctx ? 1 : m->ways;  // the encoder works on whole-batch buffers
    while (ways > 1 && batch / ways < 32) --ways;
    MDT_TRY(check_encode_args(m, tokens, tokens2, goal, ctx_out));  // before anything is enqueued
    MDT_TRY(check_loaded(m));
    MDT_TRY(mdt_reserve(m, batch));
    // per-step scalars, fp32 like the reference's 0-dim tensor math (gc_sampling.py:946-950):
    // t = -ln(sigma); ratio = exp(-t_next)/exp(-t); coef = -expm1(-(t_next - t)) -- by ONE routine (on the device)
    // whether the schedule arrives in host memory (the reference's CPU default) or on the device (mdtv_agent.py:660-667: no
    // copy, no synchronisation then), so that the eager call, the call with device sigmas and the graph replay of either give
    // the same bits (host libm and the device's expf / logf differ in the last place)
    if (ways == 1) {
        // ONE launch: per-step scalars, the sigma embeddings of all steps, the first action embedding (k_sample_prep); a host
        // schedule rides in the kernel arguments (no copy launch in front of it).  It goes FIRST: the three GEMMs of the
        // conditioning table (M = n_steps rows) depend on nothing else and are queued as side jobs -- they ride in the launches
        // of the encoder's first small products (rollout batches; at large batches whatever the encoder did not take along is
        // launched behind it)
        LAUNCH(mdt_launch_sample_prep(sigmas_dev, sigmas_dev ? nullptr : sigmas, n_steps, m->steps, m->freqs,
                                      m->cond == COND_TOKEN ? nullptr : m->sig_e, m->D, x_T, m->cfg.sigma_data, m->Wa, m->ba,
                                      decoder_view(m, 0).y, (int)(batch * m->Ta), m->A, s));
        mdt_status ms = run_modulation(m, m->steps + 3, 4, n_steps, s, true, !per_step_ctx);  // one row of conditioning vectors per step
        if (ms == MDT_OK && !per_step_ctx) ms = run_encode(m, tokens, tokens2, goal, modality, honour, batch, nullptr, 0, ctx_out, s);
        if (ms != MDT_OK) { mdt_gemm_side_drop(); return ms; }
        LAUNCH(mdt_gemm_side_flush(s));
    } else {
        MDT_TRY(run_encode(m, tokens, tokens2, goal, modality, honour, batch, nullptr, 0, ctx_out, s));
        if (!sigmas_dev) {
            HIP_TRY(hipMemcpyAsync(m->sigs, sigmas, (size_t)(n_steps + 1) * sizeof(float), hipMemcpyHostToDevice, s));
            sigmas_dev = m->sigs;
        }
        LAUNCH(mdt_launch_ddim_steps(sigmas_dev, n_steps, m->steps, s));
        MDT_TRY(run_modulation(m, m->steps + 3, 4, n_steps, s));
    }
    int64_t b0[MAX_WAYS + 1];
    b0[0] = 0;
    for (int w = 0; w < ways; ++w) {
        int64_t nb = (batch - b0[w]) / (ways - w);
        if (w + 1 < ways) nb = std::min<int64_t>(batch - b0[w], (nb + 15) / 16 * 16);
        b0[w + 1] = b0[w] + nb;
    }
    hipStream_t st[MAX_WAYS];
    st[0] = s;
    if (ways > 1) {
        for (int w = 1; w < ways; ++w) {
            if (!m->aux[w - 1]) HIP_TRY(hipStreamCreateWithFlags(&m->aux[w - 1], hipStreamNonBlocking));
            if (!m->ev_join[w - 1]) HIP_TRY(hipEventCreateWithFlags(&m->ev_join[w - 1], hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(m->ev_fork, s));
        for (int w = 1; w < ways; ++w) {
            st[w] = m->aux[w - 1];
            HIP_TRY(hipStreamWaitEvent(st[w], m->ev_fork, 0));
        }
    }
    const int64_t xs = (int64_t)m->Ta * m->A;  // floats of x per sample
    for (int w = 0; ways > 1 && w < ways; ++w) {
        const int64_t nb = b0[w + 1] - b0[w];
        const View V = decoder_view(m, b0[w]);
        LAUNCH(mdt_launch_action_embed(x_T + b0[w] * xs, m->steps + 3, 0, m->cfg.sigma_data, m->Wa, m->ba, V.y,
                                       (int)(nb * m->Ta), m->A, m->D, m->Ta, st[w]));
    }
    for (int i = 0; i < n_steps; ++i) {
        const bool last = i == n_steps - 1;
        if (per_step_ctx)  // the reference leaves the LAST step's context in latent_encoder_emb
            MDT_TRY(run_encode(m, tokens, tokens2, goal, modality, honour, batch, m->steps + 4 * i + 3, 0,
                               last ? ctx_out : nullptr, s));
        for (int w = 0; w < ways; ++w) {
            const int64_t nb = b0[w + 1] - b0[w];
            const View V = decoder_view(m, b0[w]);
            Stream fin;
            const bool head_sums = m->HP == 0 && m->A <= 8;  // the one-launch head adds MLP slabs itself
            MDT_TRY(run_decoder_blocks(m, V, nb, cond_row(m, i), 0, st[w], head_sums ? &fin : nullptr));
            const float* xin = (i == 0 ? x_T : m->xbuf) + b0[w] * xs;
            float* xout = (last ? out : m->xbuf) + b0[w] * xs;
            mdt_head_args h = head_args(m, V.y, nb, xin, m->steps + 4 * i + 3, 0, xout, MDT_HEAD_DDIM);
            if (fin.parts > 1) { h.y = fin.base; h.y_parts = fin.parts; h.y_part_stride = fin.stride; }
            h.step = m->steps + 4 * i;
            if (!last) { h.y_next = V.y; h.Wa = m->Wa; h.ba = m->ba; }
            MDT_TRY(run_head(m, h, V.hid, m->steps + 4 * (i + 1) + 3, st[w]));
        }
    }
    for (int w = 1; w < ways; ++w) {
        HIP_TRY(hipEventRecord(m->ev_join[w - 1], st[w]));
        HIP_TRY(hipStreamWaitEvent(s, m->ev_join[w - 1], 0));
    }
    return MDT_OK;
}

extern "C" mdt_status mdt_sample_ddim(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                      int32_t modality, const float* x_T, const float* sigmas, int32_t n_steps,
                                      int64_t batch, float* out, float* ctx_out, void* stream) {
    if (!sigmas) return fail(MDT_ERR_INVALID_ARG, "mdt_sample_ddim: null sigmas");
    return sample_ddim_impl(m, tokens, tokens2, goal, modality, x_T, sigmas, nullptr, n_steps, batch, out, ctx_out, stream);
}

extern "C" mdt_status mdt_sample_ddim_dev(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                          int32_t modality, const float* x_T, const float* sigmas_dev, int32_t n_steps,
                                          int64_t batch, float* out, float* ctx_out, void* stream) {
    if (!sigmas_dev) return fail(MDT_ERR_INVALID_ARG, "mdt_sample_ddim_dev: null sigmas");
    return sample_ddim_impl(m, tokens, tokens2, goal, modality, x_T, nullptr, sigmas_dev, n_steps, batch, out, ctx_out, stream);
}

extern "C" mdt_status mdt_loss_fwd(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                   int32_t modality, const float* action, const float* noise, const float* sigma,
                                   int64_t batch, float* loss_out, float* model_output, float* ctx_out, void* stream) {
    if (!m || !action || !noise || !sigma || !loss_out || batch < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_loss_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int honour = m->cfg.arch == MDT_ARCH_MDTV;
    MDT_TRY(run_encode(m, tokens, tokens2, goal, modality, honour, batch, sigma, 1, ctx_out, s));
    const int per = m->Ta * m->A;
    const int64_t n = batch * per;
    LAUNCH(mdt_launch_noise_input(action, noise, sigma, m->noised, n, per, s));
    float* F = model_output ? model_output : m->Fbuf;
    MDT_TRY(mdt_denoise_cached(m, m->noised, sigma, batch, MDT_RAW_OUTPUT, F, stream));
    LAUNCH(mdt_launch_loss_reduce(F, action, m->noised, sigma, m->cfg.sigma_data, n, per, loss_out, m->loss_part, s));
    return MDT_OK;
}

extern "C" double mdt_flops_per_chunk(const mdt_model* m, int32_t n_steps) {
    if (!m) return 0.0;
    const double D = m->D, Te = m->Te, Ta = m->Ta, A = m->A, G = m->G, O = m->O;
    auto attn = [&](double Tq, double Tk) { return 2.0 * (2.0 * Tq * Tk * D); };
    double goal = m->g_row < 0 ? 0.0 : (m->cfg.use_mlp_goal ? 2.0 * (G * 2 * D + 2 * D * D) : 2.0 * G * D);
    double enc = goal + 2.0 * m->n_tok * O * D +
                 m->Le * (Te * 2.0 * (4 * D * D + 8 * D * D) + attn(Te, Te));
    double kv = m->Ld * Te * 2.0 * 2 * D * D;
    double sig = 2.0 * (D * 2 * D + 2 * D * D);
    double blk = (m->cond == COND_ADALN ? 2.0 * D * 6 * D : 0.0) + Ta * 2.0 * (4 * D * D) + Ta * 2.0 * (2 * D * D) +
                 Ta * 2.0 * 8 * D * D + attn(Ta, Ta) + attn(Ta, Te);
    const double head = m->HP ? 2.0 * Ta * ((double)D * m->HH + (double)m->HH * A) : 2.0 * Ta * A * D;
    double step = sig + 2.0 * Ta * A * D + head + m->Ld * blk;
    if (m->cond == COND_TOKEN) return n_steps * (enc + kv + step);  // the sigma token re-runs the encoder every step
    return enc + kv + n_steps * step;
}

// ------------------------------------------------------------------------------------------------
// C ABI: kernel level (include/mdt_hip_ops.h)
// ------------------------------------------------------------------------------------------------
extern "C" int64_t mdt_op_packed_numel(int64_t N, int64_t K) { return N * K; }

extern "C" mdt_status mdt_op_pack_weight(const float* w, int64_t n_rows, int64_t K, float* packed, int64_t n_off,
                                         int64_t N_total, void* stream) {
    if (!w || !packed || K % 16 || N_total % 16 || n_off < 0 || n_off + n_rows > N_total)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight: bad argument (N, K must be multiples of 16)");
    LAUNCH(mdt_launch_pack_weight(w, (int)n_rows, (int)K, packed, (int)n_off, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_pack_weight_glu(const float* w, int64_t H2, int64_t K, float* packed, void* stream) {
    if (!w || !packed || H2 < 32 || (H2 % 32) || K < 16 || (K % 16) || misaligned(w) || misaligned(packed))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight_glu: (2H, K) with H and K multiples of 16, 16-byte aligned pointers");
    LAUNCH(mdt_launch_pack_weight_glu(w, (int)(H2 / 2), (int)K, packed, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_gemm(const mdt_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->Wp || !a->out) return fail(MDT_ERR_INVALID_ARG, "mdt_op_gemm: null pointer");
    if (a->N % 16 || a->K % 16 || a->lda % 4 || a->ldo % 4 || a->M < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_gemm: N, K multiples of 16 and lda, ldo multiples of 4 required");
    if (a->ln && a->K > 512) return fail(MDT_ERR_UNSUPPORTED, "mdt_op_gemm: LayerNorm prologue needs K <= 512");
    if (a->gin < 1 || a->rows_per_sample < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_op_gemm: gin/rows_per_sample < 1");
    if (misaligned(a->A) || misaligned(a->Wp) || misaligned(a->out) || misaligned(a->bias) || misaligned(a->mod))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_gemm: pointers must be 16-byte aligned");
    LAUNCH(mdt_launch_gemm(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_mlp(const mdt_gemm_args* fc, const mdt_gemm_args* proj, float* parts, int64_t part_stride,
                                 int32_t* n_parts, void* stream) {
    if (!fc || !proj || !fc->A || !fc->Wp || !proj->Wp || !parts || !fc->ln_w)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_mlp: null pointer");
    if (!mdt_mlp_supported(*fc, *proj))
        return fail(MDT_ERR_UNSUPPORTED, "mdt_op_mlp: needs D = fc.K a multiple of 128 (<= 512), fc.N = proj.K = 4 D, proj.N = D, "
                                         "a LayerNorm prologue and plain output rows");
    if (fc->lda % 4 || proj->ldo % 4 || proj->ldo < proj->N || part_stride < (int64_t)fc->M * proj->ldo)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_mlp: lda / ldo multiples of 4, part_stride >= M * ldo required");
    if (misaligned(fc->A) || misaligned(fc->Wp) || misaligned(proj->Wp) || misaligned(parts) || misaligned(fc->bias) ||
        misaligned(proj->bias) || misaligned(fc->mod) || misaligned(proj->mod) || (part_stride & 3))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_mlp: pointers must be 16-byte aligned");
    LAUNCH(mdt_launch_mlp(*fc, *proj, parts, part_stride, (hipStream_t)stream));
    if (n_parts) *n_parts = mdt_mlp_slices(fc->K);
    return MDT_OK;
}

extern "C" mdt_status mdt_op_pack_weight_split(const float* w, int64_t n_rows, int64_t K, void* image, void* stream) {
    if (!w || !image) return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight_split: null pointer");
    if (n_rows < 16 || n_rows % 16 || K < 32 || K % 32 || n_rows * K >= ((int64_t)1 << 30))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight_split: n_rows a multiple of 16, K a multiple of 32");
    if (misaligned(w) || misaligned(image)) return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight_split: pointers must be 16-byte aligned");
    LAUNCH(mdt_launch_pack_weight_split(w, (int)n_rows, (int)K, image, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_pack_weight_split_rows(const float* w, int64_t n_rows, int64_t K, void* image, int64_t n_off, void* stream) {
    if (!w || !image) return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight_split_rows: null pointer");
    if (n_rows < 1 || n_off < 0 || n_off % 16 || K < 32 || K % 32 || (n_rows + n_off) * K >= ((int64_t)1 << 30))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight_split_rows: n_off a multiple of 16, K a multiple of 32");
    if (misaligned(w) || misaligned(image)) return fail(MDT_ERR_INVALID_ARG, "mdt_op_pack_weight_split_rows: pointers must be 16-byte aligned");
    LAUNCH(mdt_launch_pack_weight_split(w, (int)n_rows, (int)K, image, (hipStream_t)stream, (int)n_off));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_mlp_split(const mdt_gemm_args* fc, const mdt_gemm_args* proj, const void* fc_split, const void* proj_split,
                                       float* parts, int64_t part_stride, int32_t* n_parts, void* stream) {
    if (!fc || !proj || !fc->A || !fc_split || !proj_split || !parts || !fc->ln_w)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_mlp_split: null pointer");
    if (!mdt_mlp_split_supported(*fc, *proj))
        return fail(MDT_ERR_UNSUPPORTED, "mdt_op_mlp_split: needs D = fc.K a multiple of 128 (<= 512), fc.N = proj.K = 4 D, proj.N = D, "
                                         "a LayerNorm prologue and plain output rows");
    if (fc->lda % 4 || proj->ldo % 4 || proj->ldo < proj->N || part_stride < (int64_t)fc->M * proj->ldo)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_mlp_split: lda / ldo multiples of 4, part_stride >= M * ldo required");
    if (misaligned(fc->A) || misaligned(fc_split) || misaligned(proj_split) || misaligned(parts) || misaligned(fc->bias) ||
        misaligned(proj->bias) || misaligned(fc->mod) || misaligned(proj->mod) || (part_stride & 3))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_mlp_split: pointers must be 16-byte aligned");
    LAUNCH(mdt_launch_mlp_split(*fc, *proj, fc_split, proj_split, parts, part_stride, (hipStream_t)stream));
    if (n_parts) *n_parts = mdt_mlp_slices(fc->K);
    return MDT_OK;
}

extern "C" mdt_status mdt_op_attention(const mdt_attn_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->v || !a->out) return fail(MDT_ERR_INVALID_ARG, "mdt_op_attention: null pointer");
    if (a->Tq < 1 || a->Tq > 16 || a->Tk < 1 || a->Tk > 16) return fail(MDT_ERR_UNSUPPORTED, "Tq, Tk must be 1..16");
    if (a->hd != 16 && a->hd != 32 && a->hd != 48 && a->hd != 64) return fail(MDT_ERR_UNSUPPORTED, "hd must be 16/32/48/64");
    if (a->rope) {
        if (a->hd < 32) return fail(MDT_ERR_INVALID_ARG, "rope needs hd >= 32");
        // stand-alone op calls build the rotary tables on the fly
        static float *cs = nullptr, *sn = nullptr;
        if (!cs) {
            std::vector<float> rc(256), rs(256);
            for (int i = 0; i < 16; ++i) {
                const float f = 1.0f / powf(10000.0f, (float)(2 * i) / 32.0f);
                for (int p = 0; p < 16; ++p) { rc[p * 16 + i] = cosf((float)p * f); rs[p * 16 + i] = sinf((float)p * f); }
            }
            HIP_TRY(hipMalloc((void**)&cs, 256 * sizeof(float)));
            HIP_TRY(hipMalloc((void**)&sn, 256 * sizeof(float)));
            HIP_TRY(hipMemcpy(cs, rc.data(), 256 * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(sn, rs.data(), 256 * sizeof(float), hipMemcpyHostToDevice));
        }
        LAUNCH(mdt_launch_attention(*a, cs, sn, (hipStream_t)stream));
        return MDT_OK;
    }
    LAUNCH(mdt_launch_attention(*a, nullptr, nullptr, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_xattn_fold(const mdt_xfold_args* a, void* stream) {
    if (!a || !a->kv || !a->WqT_p || !a->bq || !a->Wo_p || !a->U || !a->Wf || !a->c)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_xattn_fold: null pointer");
    if (a->H * a->hd != a->D || a->Te < 1 || a->Te > 4 || a->D > 512 || a->D % 128 || (a->H != 4 && a->H != 8) || a->hd % 16)
        return fail(MDT_ERR_UNSUPPORTED, "mdt_op_xattn_fold: need 4 or 8 heads of 16 / 32 / 48 / 64, H*hd == D <= 512 a multiple of 128 "
                                         "and 1 <= Te <= 4");
    if (misaligned(a->U) || misaligned(a->Wf) || misaligned(a->c) || misaligned(a->kv) || (a->ldkv & 3))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_xattn_fold: kv / U / Wf / c must be 16-byte aligned, ldkv a multiple of 4");
    LAUNCH(mdt_launch_xattn_fold(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_xattn_apply(const mdt_xapply_args* a, void* stream) {
    if (!a || !a->y || !a->ln_w || !a->U || !a->Wf || !a->c) return fail(MDT_ERR_INVALID_ARG, "mdt_op_xattn_apply: null pointer");
    if (!mdt_xattn_apply_supported(a->D, a->H, a->Te, a->Ta))
        return fail(MDT_ERR_UNSUPPORTED, "mdt_op_xattn_apply: unsupported (D, H, Te, Ta) = (%d, %d, %d, %d): needs 4 or 8 heads, D <= 512 a "
                                         "multiple of 128, 1 <= Te <= 4, Ta <= 16", a->D, a->H, a->Te, a->Ta);
    if (misaligned(a->y) || misaligned(a->y_out) || misaligned(a->U) || misaligned(a->Wf) || misaligned(a->c) || misaligned(a->ln_w) ||
        misaligned(a->ln_b) || misaligned(a->bo))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_xattn_apply: pointers must be 16-byte aligned");
    LAUNCH(mdt_launch_xattn_apply(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_attn_proj(const mdt_gemm_args* proj, const float* qkv, int64_t ldq, int32_t hd, int32_t T,
                                       int32_t causal, void* stream) {
    if (!proj || !qkv || !proj->Wp || !proj->out) return fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_proj: null argument");
    if (T >= 1 && proj->M / T > 64 && mdt_attn_proj_wide_supported(*proj, 8, hd, T, causal, 0)) {
        if (misaligned(qkv) || misaligned(proj->Wp) || misaligned(proj->out) || (ldq & 3))
            return fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_proj: pointers must be 16-byte aligned, ldq a multiple of 4");
        LAUNCH(mdt_launch_attn_proj_wide(*proj, qkv, ldq, 8, hd, T, (hipStream_t)stream));  // many samples: the tiled form
        return MDT_OK;
    }
    if (!mdt_attn_proj_supported(*proj, 8, hd, T, 0))
        return fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_proj: needs 8 heads of 16/32/48/64 (K = 8 * hd), T <= 16 rows per sample "
                                         "(M = samples * T, at most 64 samples), a plain projection");
    LAUNCH(mdt_launch_attn_proj(*proj, qkv, ldq, 8, hd, T, causal, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_xattn_gemm(const mdt_xapply_args* x, const mdt_gemm_args* g, void* stream) {
    if (!x || !g || !x->y || !x->ln_w || !x->U || !x->Wf || !x->c || !g->Wp || !g->out || !g->ln_w)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_xattn_gemm: null argument");
    if (misaligned(x->y) || misaligned(x->y_out) || misaligned(x->U) || misaligned(x->Wf) || misaligned(x->c) || misaligned(g->Wp) ||
        misaligned(g->out))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_xattn_gemm: pointers must be 16-byte aligned");
    if (!mdt_xattn_gemm_supported(*x, *g))
        return fail(MDT_ERR_UNSUPPORTED, "mdt_op_xattn_gemm: needs a cross-attention mdt_op_xattn_apply supports with its own output array "
                                         "(y_out != y) and a LayerNorm-prologue Linear on the same rows (A == x->y, lda = K = D, M = B * Ta, "
                                         "rows_per_sample = Ta, N a multiple of 16, no residual / row remap)");
    LAUNCH(mdt_launch_xattn_gemm(*x, *g, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_attn_xattn(const mdt_gemm_args* proj, const float* qkv, int64_t ldq, const mdt_xapply_args* x,
                                        int32_t hd, int32_t T, void* stream) {
    if (!proj || !qkv || !x || !proj->Wp || !proj->out || !x->ln_w || !x->U || !x->Wf || !x->c)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_xattn: null argument");
    if (misaligned(qkv) || misaligned(proj->Wp) || misaligned(proj->out) || misaligned(x->U) || misaligned(x->Wf) || (ldq & 3))
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_attn_xattn: pointers must be 16-byte aligned, ldq a multiple of 4");
    if (!mdt_attn_xattn_supported(*proj, *x, 8, hd, T, 1, 0) || ldq != 3 * (int64_t)proj->K)
        return fail(MDT_ERR_UNSUPPORTED, "mdt_op_attn_xattn: needs 8 heads of 48 (K = N = ldo = 384, ldq = 3 K), a gated / residual projection "
                                         "on M = x->B * T rows, T = x->Ta <= 16, x->y == proj->out and a (D, H, Te, Ta) "
                                         "mdt_op_xattn_apply supports");
    LAUNCH(mdt_launch_attn_xattn(*proj, qkv, ldq, *x, 8, hd, T, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_layernorm(const float* in, const float* w, const float* b, float* out, int64_t M, int32_t D,
                                       void* stream) {
    if (!in || !w || !out || D % 4 || D > 512 || M < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_op_layernorm: bad argument");
    LAUNCH(mdt_launch_layernorm(in, w, b, out, (int)M, D, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_head(const mdt_head_args* a, void* stream) {
    if (!a || !a->y || !a->out || a->A < 1 || a->A > 16 || a->D % 4 || a->D > 512)
        return fail(MDT_ERR_INVALID_ARG, "mdt_op_head: bad argument");
    LAUNCH(mdt_launch_head(*a, (hipStream_t)stream));
    return MDT_OK;
}

extern "C" mdt_status mdt_op_action_embed(const float* x, const float* sigma, int64_t sigma_stride, float sigma_data,
                                          const float* Wa, const float* ba, float* y, int64_t M, int32_t A, int32_t D,
                                          int32_t rows_per_sample, void* stream) {
    if (!x || !Wa || !ba || !y || D % 4) return fail(MDT_ERR_INVALID_ARG, "mdt_op_action_embed: bad argument");
    LAUNCH(mdt_launch_action_embed(x, sigma, sigma_stride, sigma_data, Wa, ba, y, (int)M, A, D, rows_per_sample,
                                   (hipStream_t)stream));
    return MDT_OK;
}
