// mdt_train.hip -- TRAINING path of the denoiser handle (include/mdt_hip_train.h, "model level"): the forward of
// GCDenoiser.loss with every activation its backward needs kept on a tape, and the backward that turns a tape into
// the gradient of every parameter (reference state_dict layout) and of the encoder inputs.
//
// Reference: torch.autograd through score_wrappers.py:45-63 (loss), mdtv_transformer.py:208-236 /
// mdt_transformer.py:207-242 (encoder + decoder), transformer_blocks.py:209-214 (Block), :291-309
// (ConditionedBlock), :335-341 (NoiseBlock); driven by MDTVAgent.training_step -> diffusion_loss
// (mdtv_agent.py:222-262,:508-521).  All three conditioning modes train: adaLN-Zero rows, NoiseBlock (the sigma
// embedding added to the attention inputs), and the sigma token at the head of the encoder context.
//
// Structure: the forward runs the SAME kernels as inference, un-fused where the backward needs the intermediate
// (LayerNorm output, pre-GELU, the un-gated branch outputs); every dense contraction of the backward runs on the
// forward's fp32-MFMA GEMM (mdt_linear_bwd, mdt_train_ops.hip).  Dropout (attention probabilities, attention /
// MLP branch outputs) uses counter-based masks (mdt_device.h: dropout_scale) that the backward regenerates.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "mdt_model_types.h"

#define fail mdt_fail

namespace {

struct BlockTape {  // one transformer block; rows M = B*T
    float *x_in, *st1, *h1, *qkv, *att, *a1, *x1;          // self-attention half (a1: branch output before gate/dropout)
    float *st3, *h3, *q, *att2, *a2, *x2;                   // cross-attention half (decoder only)
    float *st2, *h2, *u, *hid, *mo, *x3;                    // MLP half (mo: branch output before gate/dropout)
};

struct Tape {
    bool in_use = false;
    bool has_decoder = false;
    int64_t B = 0, cap = 0;
    int lang = 0;  // 1: the goal went through lang_emb
    mdt_dropout drop = {0.f, 0.f, 0.f, 0.f, 0};
    float* buf = nullptr;
    // stream of the last call that read or wrote the tape, and the event mdt_tape_release records on it: a later forward
    // that reuses the buffers from ANOTHER stream waits for it (the release only marks the tape free on the host)
    hipStream_t stream = nullptr;
    hipEvent_t freed = nullptr;
    bool freed_pending = false;
    // inputs
    float *tokens, *tokens2, *goal, *action, *noised, *sigma, *loss_part;
    float *p_pre = nullptr, *p_h = nullptr;  // proprio_emb: pre-activation and Mish output of its first layer (B, 2D)
    // encoder
    float *g_pre, *g_h;
    std::vector<BlockTape> enc;
    float *x_enc_out, *st_f, *ctx, *kvx;
    // sigma path
    float *sig_e, *sig_tpre, *sig_t, *sig_cpre, *sig_s, *mod;
    // decoder
    float *xin, *y0;
    std::vector<BlockTape> dec;
    float *st_h, *lnout, *F;
    float *hpre, *hh;  // MLP action head: hidden layer before / after GELU, (rows, HP)
};

}  // namespace

struct mdt_train_state {
    float* wt_arena = nullptr;
    std::vector<Tape> tapes;
    std::vector<int64_t> grad_off;  // per slot
    int64_t grad_numel = 0;
    // backward scratch (one backward at a time per handle)
    float* scratch = nullptr;
    int64_t scratch_cap = 0;  // batch capacity
    float *dx, *dxe, *t_d, *t_d2, *t_3d, *t_4d, *d_mod, *d_kvx, *pw, *pb, *narrow, *lin_scratch, *dF, *small;
    // Small column sums of one backward (LayerNorm weight / bias partials, per-slice bias partials of the Linears) are
    // collected here and run as ONE launch at the end (flush_deferred): their inputs live in defer_buf until then.
    float* defer_buf = nullptr;
    int64_t defer_cap = 0, defer_off = 0;  // floats
    // the scratch serves one backward at a time: a backward on another stream waits for the previous one's end
    hipStream_t scratch_stream = nullptr;
    hipEvent_t scratch_done = nullptr;
    bool scratch_pending = false;
    std::vector<mdt_colsum_entry> deferred;
    // Weight gradients beside the chain (round 6, MDT_HIP_DW_STREAM): the dW products of the blocks are leaves of the backward --
    // nothing reads them before the optimizer -- so they run on a second stream, each behind an event that says its dY exists,
    // while the chain (dX products, LayerNorm / attention backward) goes on.  Their dY operands then must outlive the chain's reuse
    // of its scratch: in this mode every dY of a block lives in its own piece of `dy_arena` (no reuse inside one backward;
    // ~1 GB at B = 1024, of 288).  The side stream has its own partial-product scratch.
    static const int MAX_SIDE = 3;
    hipStream_t side[MAX_SIDE] = {nullptr, nullptr, nullptr};
    hipEvent_t side_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t side_done[MAX_SIDE] = {nullptr, nullptr, nullptr};
    int side_ev_next = 0, side_rr = 0;
    bool side_used[MAX_SIDE] = {false, false, false};
    float *dy_arena = nullptr, *lin_scratch2[MAX_SIDE] = {nullptr, nullptr, nullptr};
    int64_t dy_cap = 0, dy_off = 0;
    float *narrow2 = nullptr, *small2 = nullptr;  // the decoder tail's own scratch when it runs beside the encoder's backward
    hipEvent_t fwd_fork = nullptr, fwd_join = nullptr;  // forward: the sigma path / action embedding beside the encoder
    // a backward in stages (mdt_train_loss_bwd_stage): the stage expected next, and the MLP-merged gradient of the block about to run
    int bwd_next = 0;
    mdt_tape_id bwd_tape = -1;
    const float* gm = nullptr;
};

// MDT_HIP_DW_STREAM = number of side streams the blocks' weight gradients rotate over (0: everything in the chain's stream, as in
// rounds 1-5; default 1; measured at B = 1024, train mode: 0 -> 9.43 ms per step, 1 -> 9.15)
static int dw_stream_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MDT_HIP_DW_STREAM"); v = e ? atoi(e) : 1; v = v < 0 ? 0 : (v > mdt_train_state::MAX_SIDE ? mdt_train_state::MAX_SIDE : v); }
    return v;
}
static mdt_status side_fork(mdt_train_state* ts, int q, hipStream_t s, hipStream_t* out);  // (defined with the backward)
// a dY buffer of n floats: its own piece of the arena when the weight gradients run beside the chain, else `shared`
static float* dy_take(mdt_train_state* ts, int64_t n, float* shared) {
    if (!ts->dy_arena) return shared;
    n = (n + 63) & ~(int64_t)63;
    if (ts->dy_off + n > ts->dy_cap) return shared;  // (sized for one whole backward: not reached)
    float* p = ts->dy_arena + ts->dy_off;
    ts->dy_off += n;
    return p;
}

static float* defer_take(mdt_train_state* ts, int64_t n) {
    n = (n + 3) & ~(int64_t)3;
    if (ts->defer_off + n > ts->defer_cap) return nullptr;
    float* p = ts->defer_buf + ts->defer_off;
    ts->defer_off += n;
    return p;
}

// entry of every backward: order this stream behind the last user of the shared scratch
static mdt_status scratch_enter(mdt_model* m, hipStream_t s) {
    mdt_train_state* ts = m->train;
    if (ts->scratch_pending && ts->scratch_stream != s) HIP_TRY(hipStreamWaitEvent(s, ts->scratch_done, 0));
    ts->scratch_stream = s;
    ts->scratch_pending = false;
    return MDT_OK;
}
static mdt_status scratch_leave(mdt_model* m, hipStream_t s) {
    mdt_train_state* ts = m->train;
    if (!ts->scratch_done) HIP_TRY(hipEventCreateWithFlags(&ts->scratch_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ts->scratch_done, s));
    ts->scratch_pending = true;
    return MDT_OK;
}

// End of a stage of a backward: the side streams' weight gradients joined into `s`, the small column sums collected so far
// launched -- every gradient the stage completes is complete in the order of `s` behind this.  last: the scratch is handed on.
static mdt_status stage_finish(mdt_model* m, hipStream_t s, bool last) {
    mdt_train_state* ts = m->train;
    for (int i = 0; i < mdt_train_state::MAX_SIDE; ++i) {
        if (!ts->side_used[i]) continue;  // the weight gradients that ran beside the chain (and their bias partials) before anything reads them
        if (!ts->side_done[i]) HIP_TRY(hipEventCreateWithFlags(&ts->side_done[i], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ts->side_done[i], ts->side[i]));
        HIP_TRY(hipStreamWaitEvent(s, ts->side_done[i], 0));
        ts->side_used[i] = false;
    }
    // entries of one launch run concurrently: two sums into the same gradient (a parameter read twice) go to successive launches
    std::vector<mdt_colsum_entry>& d = ts->deferred;
    size_t lo = 0;
    for (size_t i = 0; i <= d.size(); ++i) {
        bool cut = i == d.size();
        for (size_t j = lo; !cut && j < i; ++j) cut = d[j].dst == d[i].dst;
        if (cut && i > lo) { LAUNCH(mdt_launch_colsum_batched(d.data() + lo, (int)(i - lo), s)); lo = i; }
    }
    ts->deferred.clear();
    if (!last) return MDT_OK;   // (the partial tables stay where they are: defer_off only grows inside one backward)
    ts->defer_off = 0;
    return scratch_leave(m, s);
}
static mdt_status flush_deferred(mdt_model* m, hipStream_t s) { return stage_finish(m, s, true); }

static const int NARROW_SLICES = 512;  // row slices of the narrow (A x D) weight gradients: 512 x 2 workgroups of 20 rows at B = 1024 (128 slices: 80 dependent row trips per thread, 42 us per launch)

// ------------------------------------------------------------------------------------------------
// setup
// ------------------------------------------------------------------------------------------------
extern "C" mdt_status mdt_train_prepare(mdt_model* m) {
    if (!m) return fail(MDT_ERR_INVALID_ARG, "mdt_train_prepare: null handle");
    if (m->train) return MDT_OK;
    mdt_train_state* t = new mdt_train_state();
    std::vector<Lin*> lins;
    for (const LinPart& p : m->parts)
        if (std::find(lins.begin(), lins.end(), p.lin) == lins.end()) lins.push_back(p.lin);
    Bump count;
    for (Lin* l : lins) count.take((size_t)l->N * l->K);
    hipError_t e = hipMalloc((void**)&t->wt_arena, count.off * sizeof(float));
    if (e != hipSuccess) { delete t; return fail(MDT_ERR_HIP, "hipMalloc(transposed weights) failed: %s", hipGetErrorString(e)); }
    (void)hipMemset(t->wt_arena, 0, count.off * sizeof(float));
    Bump real;
    real.base = t->wt_arena;
    for (Lin* l : lins) l->wt = real.take((size_t)l->N * l->K);
    // Gradient layout: the parts of a stacked Linear (q|k|v, the K|V of all decoder blocks, all adaLN projections)
    // are contiguous in packed-row order -- weights (N, K), then biases (N) -- so ONE dW GEMM / ONE column sum
    // serves the whole stack; every other parameter follows in slot order.  Offsets are multiples of 4 floats.
    t->grad_off.assign(m->slots.size(), -1);
    int64_t off = 0;
    for (size_t i = 0; i < m->slots.size(); ++i) {
        if (t->grad_off[i] >= 0) continue;
        const Lin* l = m->slots[i].kind == SLOT_PACK ? m->slots[i].lin : nullptr;
        if (!l) { t->grad_off[i] = off; off += (m->slots[i].numel + 3) & ~(int64_t)3; continue; }
        const int64_t wbase = off, bbase = off + (int64_t)l->N * l->K;
        for (const LinPart& p : m->parts) {
            if (p.lin != l) continue;
            t->grad_off[p.w_slot] = wbase + (int64_t)p.n_off * l->K;
            if (p.b_slot >= 0) t->grad_off[p.b_slot] = bbase + p.n_off;
        }
        off = bbase + (l->bias ? l->N : 0);
    }
    t->grad_numel = off;
    for (Slot& s : m->slots) s.loaded = false;  // every weight needs its transposed image: upload again
    m->cached_batch = 0;
    m->train = t;
    return MDT_OK;
}

void mdt_train_free(mdt_model* m) {
    if (!m || !m->train) return;
    mdt_train_state* t = m->train;
    for (Tape& tp : t->tapes) {
        (void)mdt_dev_free(tp.buf);
        if (tp.freed) (void)hipEventDestroy(tp.freed);
    }
    (void)mdt_dev_free(t->scratch);
    if (t->scratch_done) (void)hipEventDestroy(t->scratch_done);
    for (hipEvent_t e : t->side_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : t->side_done) if (e) (void)hipEventDestroy(e);
    if (t->fwd_fork) (void)hipEventDestroy(t->fwd_fork);
    if (t->fwd_join) (void)hipEventDestroy(t->fwd_join);
    for (hipStream_t q : t->side) if (q) (void)hipStreamDestroy(q);
    (void)hipFree(t->wt_arena);
    for (const LinPart& p : m->parts) p.lin->wt = nullptr;
    delete t;
    m->train = nullptr;
}

extern "C" int64_t mdt_grad_numel(const mdt_model* m) { return (m && m->train) ? m->train->grad_numel : -1; }

extern "C" int64_t mdt_grad_offset(const mdt_model* m, int64_t i) {
    if (!m || !m->train || i < 0 || i >= (int64_t)m->slots.size()) return -1;
    return m->train->grad_off[i];
}

// ------------------------------------------------------------------------------------------------
// tapes
// ------------------------------------------------------------------------------------------------
static void carve_block(const mdt_model* m, Bump& b, BlockTape& t, int64_t M, bool dec) {
    const int D = m->D;
    t.x_in = b.take(M * D); t.st1 = b.take(M * 2); t.h1 = b.take(M * D); t.qkv = b.take(M * 3 * D); t.att = b.take(M * D);
    t.a1 = b.take(M * D);
    t.x1 = b.take(M * D);
    if (dec) {
        t.st3 = b.take(M * 2); t.h3 = b.take(M * D); t.q = b.take(M * D); t.att2 = b.take(M * D); t.a2 = b.take(M * D);
        t.x2 = b.take(M * D);
    } else {
        t.st3 = t.h3 = t.q = t.att2 = t.a2 = nullptr;
        t.x2 = t.x1;
    }
    t.st2 = b.take(M * 2); t.h2 = b.take(M * D); t.u = b.take(M * 4 * D); t.hid = b.take(M * 4 * D);
    t.mo = b.take(M * D);
    t.x3 = b.take(M * D);
}

static void carve_tape(const mdt_model* m, Bump& b, Tape& t, int64_t B) {
    const int D = m->D;
    const int64_t Me = B * m->Te, Ma = B * m->Ta;
    t.tokens = b.take(B * (m->cfg.arch == MDT_ARCH_MDTV ? m->n_tok : 1) * m->O);
    t.tokens2 = b.take(m->cfg.arch == MDT_ARCH_MDT ? B * m->O : B * m->Pd);  // MDT-V: state_obs (B, Pd) when use_proprio
    t.p_pre = b.take(m->p_row >= 0 ? B * 2 * D : 0); t.p_h = b.take(m->p_row >= 0 ? B * 2 * D : 0);
    t.goal = b.take(B * m->G);
    t.action = b.take(Ma * m->A); t.noised = b.take(Ma * m->A); t.sigma = b.take(B); t.loss_part = b.take(MDT_LOSS_PARTS);
    t.g_pre = b.take(B * 2 * D); t.g_h = b.take(B * 2 * D);
    t.enc.resize(m->Le);
    for (int l = 0; l < m->Le; ++l) carve_block(m, b, t.enc[l], Me, false);
    t.x_enc_out = b.take(Me * D);  // input of the first block / output chain starts here when Le == 0
    t.st_f = b.take(Me * 2); t.ctx = b.take(Me * D); t.kvx = b.take(Me * m->Ld * 2 * D);
    t.sig_e = b.take(B * D); t.sig_tpre = b.take(B * 2 * D); t.sig_t = b.take(B * 2 * D); t.sig_cpre = b.take(B * D);
    t.sig_s = b.take(B * D); t.mod = b.take(B * m->Ld * 6 * D);
    t.xin = b.take(Ma * m->A); t.y0 = b.take(Ma * D);
    t.dec.resize(m->Ld);
    for (int l = 0; l < m->Ld; ++l) carve_block(m, b, t.dec[l], Ma, true);
    t.st_h = b.take(Ma * 2); t.lnout = b.take(Ma * D); t.F = b.take(Ma * m->A);
    t.hpre = b.take(Ma * m->HP); t.hh = b.take(Ma * m->HP);
}

static mdt_status acquire_tape(mdt_model* m, int64_t B, mdt_tape_id* id, hipStream_t s) {
    mdt_train_state* ts = m->train;
    int pick = -1;
    for (size_t i = 0; i < ts->tapes.size(); ++i)
        if (!ts->tapes[i].in_use && ts->tapes[i].cap >= B) { pick = (int)i; break; }
    if (pick < 0)
        for (size_t i = 0; i < ts->tapes.size(); ++i)
            if (!ts->tapes[i].in_use) { pick = (int)i; break; }
    if (pick < 0) {
        if (ts->tapes.size() >= 16) return fail(MDT_ERR_STATE, "more than 16 tapes alive: release tapes after their backward");
        ts->tapes.emplace_back();
        pick = (int)ts->tapes.size() - 1;
    }
    Tape& t = ts->tapes[pick];
    if (t.cap < B) {
        if (t.buf) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(mdt_dev_free(t.buf)); t.buf = nullptr; t.cap = 0; }
        Bump count;
        carve_tape(m, count, t, B);
        HIP_TRY(mdt_dev_malloc((void**)&t.buf, count.off * sizeof(float)));
        t.cap = B;
    }
    Bump real;
    real.base = t.buf;
    carve_tape(m, real, t, t.cap);
    if (t.freed_pending && t.stream != s) HIP_TRY(hipStreamWaitEvent(s, t.freed, 0));  // its last reader ran elsewhere
    t.freed_pending = false;
    t.stream = s;
    t.in_use = true;
    t.B = B;
    t.has_decoder = false;
    *id = pick;
    return MDT_OK;
}

static mdt_status get_tape(mdt_model* m, mdt_tape_id id, Tape** out) {
    if (!m || !m->train) return fail(MDT_ERR_STATE, "training was not prepared (mdt_train_prepare)");
    if (id < 0 || id >= (int)m->train->tapes.size() || !m->train->tapes[id].in_use)
        return fail(MDT_ERR_INVALID_ARG, "invalid or released tape %d", id);
    *out = &m->train->tapes[id];
    return MDT_OK;
}

extern "C" mdt_status mdt_tape_release(mdt_model* m, mdt_tape_id id) {
    Tape* t;
    MDT_TRY(get_tape(m, id, &t));
    // the backward that last read the tape may still be in flight on its stream: leave a marker there for the next user
    if (!t->freed) HIP_TRY(hipEventCreateWithFlags(&t->freed, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(t->freed, t->stream));
    t->freed_pending = true;
    t->in_use = false;
    return MDT_OK;
}

static void carve_scratch(const mdt_model* m, Bump& b, mdt_train_state* ts, int64_t B) {
    const int D = m->D;
    const int64_t Me = B * m->Te, Ma = B * m->Ta, Mx = std::max(Me, Ma);
    ts->dx = b.take(Ma * D); ts->dxe = b.take(Me * D);
    ts->t_d = b.take(Mx * D); ts->t_d2 = b.take(Mx * D); ts->t_3d = b.take(Mx * 3 * D); ts->t_4d = b.take(Mx * 4 * D);
    ts->d_mod = b.take(B * m->Ld * 6 * D); ts->d_kvx = b.take(Me * m->Ld * 2 * D);
    ts->pw = b.take(B * D); ts->pb = b.take(B * D);
    {   // deferred column sums: two partial tables per LayerNorm call, 64 slices of bias partials per Linear
        int64_t widest = 4 * D;
        for (const LinPart& p : m->parts) widest = std::max<int64_t>(widest, p.lin->N);
        ts->defer_cap = (int64_t)(2 * m->Le + 3 * m->Ld + 2) * 2 * B * D + (int64_t)m->parts.size() * 256 * widest + 1024;
        ts->defer_buf = b.take(ts->defer_cap);
    }
    ts->narrow = b.take((size_t)NARROW_SLICES * 16 * std::max({D, m->HP, m->p_row >= 0 ? 2 * D : 0}));
    // Linear backward scratch: the largest need over every (rows, N, K) this model's backward runs
    int64_t need = 0;
    for (int64_t rows : {Ma, Me}) {
        for (auto nk : {std::pair<int, int>(4 * D, D), std::pair<int, int>(D, 4 * D), std::pair<int, int>(3 * D, D),
                        std::pair<int, int>(D, D), std::pair<int, int>(m->Ld * 2 * D, D)})
            need = std::max(need, mdt_linear_bwd_scratch(rows, nk.first, nk.second));
    }
    for (auto nk : {std::pair<int, int>(m->Ld * 6 * D, D), std::pair<int, int>(D, 2 * D), std::pair<int, int>(2 * D, D),
                    std::pair<int, int>(2 * D, m->G), std::pair<int, int>(D, m->G), std::pair<int, int>(D, m->O)})
        need = std::max(need, mdt_linear_bwd_scratch(B, nk.first, nk.second));
    need = std::max(need, mdt_linear_bwd_scratch(B * m->n_tok, D, m->O));
    if (m->HP) need = std::max(need, mdt_linear_bwd_scratch(Ma, m->HP, D));
    ts->lin_scratch = b.take(need);
    if (dw_stream_mode()) {
        for (int i = 0; i < mdt_train_state::MAX_SIDE; ++i) ts->lin_scratch2[i] = i < dw_stream_mode() ? b.take(need) : nullptr;
        // every dY of every block once: (3 + dec) x (M, D) merged gradients / dq, (M, 4D), (M, 3D) per block
        ts->dy_cap = (int64_t)m->Le * Me * (3 + 4 + 3) * D + (int64_t)m->Ld * Ma * (4 + 4 + 3) * D + 64 * (int64_t)(m->Le + m->Ld) * 8 + 1024;
        ts->dy_arena = b.take(ts->dy_cap);
        ts->narrow2 = b.take((size_t)NARROW_SLICES * 16 * std::max({D, m->HP, m->p_row >= 0 ? 2 * D : 0}));
        ts->small2 = b.take(B * 3 * D + 64);
    } else {
        ts->narrow2 = ts->small2 = nullptr;
        for (float*& q : ts->lin_scratch2) q = nullptr;
        ts->dy_arena = nullptr; ts->dy_cap = 0;
    }
    ts->dF = b.take(Ma * m->A);
    ts->small = b.take(std::max<int64_t>({B * 2 * D, Mx * (int64_t)std::max(m->O, m->G), (int64_t)16 * m->HP}));
}

static mdt_status reserve_scratch(mdt_model* m, int64_t B) {
    mdt_train_state* ts = m->train;
    if (B > ts->scratch_cap) {
        if (ts->scratch) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(mdt_dev_free(ts->scratch)); ts->scratch = nullptr; }
        Bump count;
        carve_scratch(m, count, ts, B);
        HIP_TRY(mdt_dev_malloc((void**)&ts->scratch, count.off * sizeof(float)));
        ts->scratch_cap = B;
    }
    Bump real;
    real.base = ts->scratch;
    carve_scratch(m, real, ts, ts->scratch_cap);
    return MDT_OK;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static mdt_ln_train_args ln_args(const float* x, const float* w, const float* b, float* out, float* stats, int M, int D) {
    mdt_ln_train_args a;
    memset(&a, 0, sizeof a);
    a.x = x; a.w = w; a.b = b; a.out = out; a.stats = stats; a.M = M; a.D = D; a.rows_per_sample = 1;
    a.shift_off = a.scale_off = -1;
    return a;
}

// dropout sites: one id per (block, place); the element index inside a site is the place's own flat index
enum { SITE_ATTN = 0, SITE_RESID = 1, SITE_MLP = 2, SITE_XATTN = 3, SITE_XRESID = 4 };
enum { SITE_EMBED_CTX = 0, SITE_EMBED_ACTION = 1 };  // places of the pseudo block Le + Ld: the embedding dropouts
static uint32_t site_id(int block, int place) { return (uint32_t)(block * 8 + place + 1); }

static mdt_merge_args merge_args(const float* x, const float* a, const float* gate, int64_t gstride, float* out, int64_t B,
                                 int T, int D, float p, uint32_t site, uint64_t seed) {
    mdt_merge_args g;
    memset(&g, 0, sizeof g);
    g.x = x; g.a = a; g.gate = gate; g.gate_stride = gstride; g.out = out; g.B = (int)B; g.rows_per_sample = T; g.D = D;
    g.p = p; g.site = site; g.seed = seed;
    return g;
}

static mdt_status attn_fwd(mdt_model* m, const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out,
                           int64_t B, int Tq, int Tk, bool causal, const mdt_dropout& dr, uint32_t site, hipStream_t s) {
    // (without dropout too where the MFMA form of the training kernel applies: it is the faster of the two at training batches)
    if ((dr.seed != 0 && dr.attn_p > 0.f) || mdt_attn_train_mfma_supported(m->hd, m->H, m->cfg.use_rot_embed)) {
        mdt_attn_train_args a;
        memset(&a, 0, sizeof a);
        a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ldkv = ldkv; a.out = out; a.ldo = m->D;
        a.B = (int)B; a.H = m->H; a.hd = m->hd; a.Tq = Tq; a.Tk = Tk; a.causal = causal;
        a.p = dr.seed ? dr.attn_p : 0.f; a.site = site; a.seed = dr.seed;
        a.rope = m->cfg.use_rot_embed; a.rope_cos = m->rope_cos; a.rope_sin = m->rope_sin;
        LAUNCH(mdt_launch_attn_fwd_train(a, s));
        return MDT_OK;
    }
    mdt_attn_args a;
    memset(&a, 0, sizeof a);
    a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ldkv = ldkv; a.out = out; a.ldo = m->D;
    a.B = (int)B; a.H = m->H; a.hd = m->hd; a.Tq = Tq; a.Tk = Tk; a.causal = causal;
    a.rope = m->cfg.use_rot_embed;
    LAUNCH(mdt_launch_attention(a, m->rope_cos, m->rope_sin, s));
    return MDT_OK;
}

// Where a block's LayerNorms and branch gates find their conditioning in a per-sample row of `mod`:
//   adaLN-Zero (transformer_blocks.py:291-309): [shift_msa | scale_msa | gate_msa | shift_mlp | scale_mlp | gate_mlp]
//   NoiseBlock (:335-341): the row IS c, a shift on ln_1 and ln3; no scales, no gates, the MLP unconditioned
//   none: plain Block (encoder; decoder of the sigma-token variant)
struct CondLayout { int sh1, sc1, g1, sh3, sh2, sc2, g2; };
static CondLayout cond_layout(int cond, int D) {
    if (cond == COND_ADALN) return {0, D, 2 * D, -1, 3 * D, 4 * D, 5 * D};
    if (cond == COND_NOISE) return {0, -1, -1, 0, -1, -1, -1};
    return {-1, -1, -1, -1, -1, -1, -1};
}
static void ln_cond(mdt_ln_train_args& l, const float* mod, int64_t modw, int sh, int sc, int T) {
    if (!mod || (sh < 0 && sc < 0)) return;
    l.mod = mod; l.mod_stride = modw; l.shift_off = sh; l.scale_off = sc; l.rows_per_sample = T;
}

// MDT_HIP_TRAIN_FUSE (A/B runs; default 3): bit 0 = every branch merge of the forward in one launch with the LayerNorm that
// reads its result (k_merge_ln_fwd4), bit 1 = every LayerNorm backward in one launch with the merge backward behind it
// (k_ln_bwd4<true>).  0 restores the separate launches of rounds 1-5; both forms give the same bits except the gate gradient's
// summation order over a sample's rows.
static int train_fuse() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MDT_HIP_TRAIN_FUSE"); v = e ? atoi(e) : 3; }
    return v;
}

// the first LayerNorm of a block (ln_1 on its input, conditioned as the block is)
static mdt_ln_train_args block_ln1_args(mdt_model* m, const EncBlock& e, BlockTape& t, int64_t B, int T, int cond, const float* mod,
                                        int64_t modw) {
    const CondLayout c = cond_layout(mod ? cond : COND_TOKEN, m->D);
    mdt_ln_train_args l1 = ln_args(t.x_in, e.ln1_w, e.ln1_b, t.h1, t.st1, (int)(B * T), m->D);
    ln_cond(l1, mod, modw, c.sh1, c.sc1, T);
    return l1;
}

// merge g, then LayerNorm l on its result: one launch or two (train_fuse)
static mdt_status merge_then_ln(const mdt_merge_args& g, const mdt_ln_train_args& l, hipStream_t s) {
    if (train_fuse() & 1) { LAUNCH(mdt_launch_merge_ln_fwd(g, l, s)); return MDT_OK; }
    LAUNCH(mdt_launch_merge_fwd(g, s));
    LAUNCH(mdt_launch_ln_fwd_train(l, s));
    return MDT_OK;
}

// one block forward; mod == nullptr: plain Block, else conditioned by the rows of `mod` (stride modw) as `cond` lays
// them out.  `blk` numbers the block for the dropout sites (encoder blocks first, then decoder blocks).
// ln1_done: the block's first LayerNorm already ran (in the launch of the previous block's last merge); tail_ln: the LayerNorm
// that reads this block's output (the next block's ln_1 or the stack's final LayerNorm) -- it rides in the last merge's launch.
static mdt_status block_fwd(mdt_model* m, const EncBlock& e, const DecBlock* d, BlockTape& t, int64_t B, int T, bool causal,
                            int cond, const float* mod, int64_t modw, const float* kv, const mdt_dropout& dr, int blk,
                            hipStream_t s, bool ln1_done = false, const mdt_ln_train_args* tail_ln = nullptr, int half = 0) {
    // half (decoder blocks): 1 = only the self-attention half, which needs no context (through the merge + ln3 launch and the
    // cross-attention's query product); 2 = the rest; 0 = the whole block
    const int D = m->D, M = (int)(B * T);
    const CondLayout c = cond_layout(mod ? cond : COND_TOKEN, D);
    const mdt_merge_args g1 = merge_args(t.x_in, t.a1, c.g1 >= 0 ? mod + c.g1 : nullptr, modw, t.x1, B, T, D, dr.resid_p,
                                         site_id(blk, SITE_RESID), dr.seed);
    mdt_ln_train_args l2 = ln_args(t.x2, e.ln2_w, e.ln2_b, t.h2, t.st2, M, D);
    ln_cond(l2, mod, modw, c.sh2, c.sc2, T);
    if (half != 2) {
        if (!ln1_done) LAUNCH(mdt_launch_ln_fwd_train(block_ln1_args(m, e, t, B, T, cond, mod, modw), s));
        LAUNCH(mdt_launch_gemm(gemm_args(t.h1, D, e.qkv, t.qkv, 3 * D, M), s));
        MDT_TRY(attn_fwd(m, t.qkv, 3 * D, t.qkv + D, t.qkv + 2 * D, 3 * D, t.att, B, T, T, causal, dr, site_id(blk, SITE_ATTN), s));
        LAUNCH(mdt_launch_gemm(gemm_args(t.att, D, e.proj, t.a1, D, M), s));
    }
    if (d) {
        if (half != 2) {
            mdt_ln_train_args l3 = ln_args(t.x1, d->ln3_w, d->ln3_b, t.h3, t.st3, M, D);
            ln_cond(l3, mod, modw, c.sh3, -1, T);
            MDT_TRY(merge_then_ln(g1, l3, s));
            LAUNCH(mdt_launch_gemm(gemm_args(t.h3, D, d->xq, t.q, D, M), s));
        }
        if (half == 1) return MDT_OK;
        // SDPA is_causal on a Ta x Te matrix: top-left aligned (transformer_blocks.py:204,142)
        MDT_TRY(attn_fwd(m, t.q, D, kv, kv + D, (int64_t)m->Ld * 2 * D, t.att2, B, T, m->Te, true, dr, site_id(blk, SITE_XATTN), s));
        LAUNCH(mdt_launch_gemm(gemm_args(t.att2, D, d->xproj, t.a2, D, M), s));
        MDT_TRY(merge_then_ln(merge_args(t.x1, t.a2, nullptr, 0, t.x2, B, T, D, dr.resid_p, site_id(blk, SITE_XRESID), dr.seed), l2, s));
    } else {
        MDT_TRY(merge_then_ln(g1, l2, s));  // t.x2 == t.x1 in an encoder block
    }
    {   // c_fc and its GELU in one launch: the epilogue leaves the pre-activation u (the backward's operand) beside gelu(u)
        mdt_gemm_args g = gemm_args(t.h2, D, e.fc, t.hid, 4 * D, M);
        g.act = MDT_ACT_GELU; g.aux = t.u; g.aux_mode = 1;
        LAUNCH(mdt_launch_gemm(g, s));
    }
    LAUNCH(mdt_launch_gemm(gemm_args(t.hid, 4 * D, e.proj2, t.mo, D, M), s));
    const mdt_merge_args g2 = merge_args(t.x2, t.mo, c.g2 >= 0 ? mod + c.g2 : nullptr, modw, t.x3, B, T, D, dr.mlp_p,
                                         site_id(blk, SITE_MLP), dr.seed);
    if (tail_ln) MDT_TRY(merge_then_ln(g2, *tail_ln, s));
    else LAUNCH(mdt_launch_merge_fwd(g2, s));
    return MDT_OK;
}

// c = sigma_emb(sigma): sinusoidal -> Linear -> Mish -> Linear, kept on the tape (mdtv_transformer.py:169-174,238-244).
// The last Linear writes row r of its output to row r * gout of `out` (leading dimension D).
static mdt_status sigma_fwd(mdt_model* m, Tape& t, const float* sigma, float* out, int gout, hipStream_t s) {
    const int D = m->D;
    const int64_t B = t.B;
    HIP_TRY(hipMemcpyAsync(t.sigma, sigma, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
    LAUNCH(mdt_launch_sigma_emb(t.sigma, 1, m->freqs, t.sig_e, (int)B, D, s));
    LAUNCH(mdt_launch_gemm(gemm_args(t.sig_e, D, m->sig1, t.sig_tpre, 2 * D, (int)B), s));
    LAUNCH(mdt_launch_act_fwd(t.sig_tpre, t.sig_t, B * 2 * D, MDT_ACT_MISH, s));
    mdt_gemm_args a = gemm_args(t.sig_t, 2 * D, m->sig3, out, D, (int)B);
    a.gin = 1; a.gout = gout; a.goff = 0;
    LAUNCH(mdt_launch_gemm(a, s));
    return MDT_OK;
}

// first context row the embedding dropout applies to (rows below it are kept), or -1: no dropout on the context
static int embed_drop_from(const mdt_model* m) {
    if (m->cfg.arch == MDT_ARCH_MDT) return m->sig_tok;
    return m->cfg.no_goal_conditioning ? m->g_row : -1;
}

static float* enc_first_input(const mdt_model* m, Tape& t) { return m->Le > 0 ? t.enc[0].x_in : t.x_enc_out; }
static float* enc_last_output(const mdt_model* m, Tape& t) { return m->Le > 0 ? t.enc[m->Le - 1].x3 : t.x_enc_out; }

static mdt_status enc_fwd(mdt_model* m, Tape& t, const float* tokens, const float* tokens2, const float* goal, int modality,
                          int honour, const float* sigma, float* ctx_out, hipStream_t s) {
    const mdt_config& c = m->cfg;
    const int D = m->D, Te = m->Te;
    const int t0 = m->sig_tok;  // context row of the goal token (1 when the sigma token leads the context)
    const int64_t B = t.B;
    if (t0 && !sigma)
        return fail(MDT_ERR_INVALID_ARG, "use_ada_conditioning=False puts sigma into the context: sigma is required");
    const int ntok_rows = c.arch == MDT_ARCH_MDTV ? m->n_tok : 1;
    HIP_TRY(hipMemcpyAsync(t.tokens, tokens, (size_t)B * ntok_rows * m->O * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (c.arch == MDT_ARCH_MDT)
        HIP_TRY(hipMemcpyAsync(t.tokens2, tokens2, (size_t)B * m->O * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (m->p_row >= 0) {
        if (!tokens2) return fail(MDT_ERR_INVALID_ARG, "this handle was created with use_proprio: state_obs (tokens2) is required");
        HIP_TRY(hipMemcpyAsync(t.tokens2, tokens2, (size_t)B * m->Pd * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(t.goal, goal, (size_t)B * m->G * sizeof(float), hipMemcpyDeviceToDevice, s));
    t.lang = honour && c.use_modality_encoder && modality == MDT_MODALITY_LANG;
    const Lin& g0 = t.lang ? m->lang0 : m->goal0;
    const Lin& g2 = t.lang ? m->lang2 : m->goal2;
    const float* pos0 = (c.arch == MDT_ARCH_MDT && c.use_abs_pos_emb) ? m->pos_emb : nullptr;
    const float* pos1 = pos0 ? m->pos_emb + (int64_t)c.goal_seq_len * D : nullptr;
    float* x0 = enc_first_input(m, t);
    if (t0) MDT_TRY(sigma_fwd(m, t, sigma, x0, Te, s));  // concatenate_inputs (mdtv_transformer.py:296-297)
    if (m->g_row >= 0) {
        const float* gin = t.goal;
        int64_t ld = m->G;
        if (c.use_mlp_goal) {
            LAUNCH(mdt_launch_gemm(gemm_args(t.goal, m->G, g0, t.g_pre, 2 * D, (int)B), s));
            LAUNCH(mdt_launch_act_fwd(t.g_pre, t.g_h, B * 2 * D, MDT_ACT_GELU, s));
            gin = t.g_h; ld = 2 * D;
        }
        mdt_gemm_args a = gemm_args(gin, ld, g2, x0, D, (int)B);
        a.gin = 1; a.gout = Te; a.goff = m->g_row; a.rowvec = pos0;
        LAUNCH(mdt_launch_gemm(a, s));
    }
    if (c.arch == MDT_ARCH_MDTV) {
        mdt_gemm_args a = gemm_args(t.tokens, m->O, m->tok, x0, D, (int)(B * m->n_tok));
        a.gin = m->n_tok; a.gout = Te; a.goff = m->tok_row;
        LAUNCH(mdt_launch_gemm(a, s));
    } else {
        mdt_gemm_args a = gemm_args(t.tokens, m->O, m->tok, x0, D, (int)B);
        a.gin = 1; a.gout = Te; a.goff = m->tok_row; a.rowvec = pos1;
        LAUNCH(mdt_launch_gemm(a, s));
        mdt_gemm_args b2 = gemm_args(t.tokens2, m->O, m->incam, x0, D, (int)B);
        b2.gin = 1; b2.gout = Te; b2.goff = m->tok_row + 1; b2.rowvec = pos1;
        LAUNCH(mdt_launch_gemm(b2, s));
    }
    if (m->p_row >= 0) {  // proprioceptive token: proprio_emb(state_obs) -> the last context row (mdtv_transformer.py:260-266)
        LAUNCH(mdt_launch_narrow_linear(t.tokens2, m->prop0_T, m->prop0_b, t.p_pre, t.p_h, (int)B, m->Pd, 2 * D, MDT_ACT_MISH, s));
        mdt_gemm_args a = gemm_args(t.p_h, 2 * D, m->prop2, x0, D, (int)B);
        a.gin = 1; a.gout = Te; a.goff = m->p_row;
        LAUNCH(mdt_launch_gemm(a, s));
    }
    // embedding dropout (self.drop): MDTTransformer drops every embedded goal / state token (mdt_transformer.py:220-227),
    // MDTVTransformer only the goal token it appends when goal_conditioned=False (mdtv_transformer.py:293-294)
    const int drop_lo = embed_drop_from(m);
    if (drop_lo >= 0)
        LAUNCH(mdt_launch_dropout_rows(x0, B * Te, D, Te, drop_lo, t.drop.embed_p, site_id(m->Le + m->Ld, SITE_EMBED_CTX),
                                       t.drop.seed, s));
    for (int l = 1; l < m->Le; ++l) t.enc[l].x_in = t.enc[l - 1].x3;  // chain: a block's input is its predecessor's output buffer
    const mdt_ln_train_args lf = ln_args(enc_last_output(m, t), m->enc_ln_w, m->enc_ln_b, t.ctx, t.st_f, (int)(B * Te), D);
    for (int l = 0; l < m->Le; ++l) {
        // the LayerNorm behind this block (the next block's ln_1, at the end the encoder's final one) rides in its last merge
        const mdt_ln_train_args tail = l + 1 < m->Le ? block_ln1_args(m, m->enc[l + 1], t.enc[l + 1], B, Te, COND_TOKEN, nullptr, 0) : lf;
        MDT_TRY(block_fwd(m, m->enc[l], nullptr, t.enc[l], B, Te, false, COND_TOKEN, nullptr, 0, nullptr, t.drop, l, s, l > 0, &tail));
    }
    if (m->Le == 0) LAUNCH(mdt_launch_ln_fwd_train(lf, s));
    if (ctx_out) HIP_TRY(hipMemcpyAsync(ctx_out, t.ctx, (size_t)B * Te * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    return MDT_OK;
}

// What the decoder needs that does NOT depend on the context: the conditioning rows (sigma MLP, stacked adaLN Linear), the noised /
// preconditioned actions and their embedding.  With the sigma token (COND_TOKEN) the encoder has run the sigma MLP itself (and
// filled t.sigma), so this runs behind it; otherwise it is independent of the encoder and may run beside it (mdt_train_loss_fwd).
static mdt_status dec_fwd_pre(mdt_model* m, Tape& t, const float* action, const float* noise, const float* sigma, hipStream_t s) {
    const int D = m->D, Ta = m->Ta, A = m->A;
    const int64_t B = t.B, Ma = B * Ta;
    const int64_t modw = m->cond == COND_ADALN ? (int64_t)m->Ld * 6 * D : D;
    HIP_TRY(hipMemcpyAsync(t.action, action, (size_t)Ma * A * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (m->cond == COND_ADALN) {  // sigma embedding -> SiLU -> the stacked modulation Linear of every block
        MDT_TRY(sigma_fwd(m, t, sigma, t.sig_cpre, 1, s));
        LAUNCH(mdt_launch_act_fwd(t.sig_cpre, t.sig_s, B * D, MDT_ACT_SILU, s));
        LAUNCH(mdt_launch_gemm(gemm_args(t.sig_s, D, m->mod_all, t.mod, modw, (int)B), s));
    } else if (m->cond == COND_NOISE) {
        MDT_TRY(sigma_fwd(m, t, sigma, t.mod, 1, s));
    }  // COND_TOKEN: enc_fwd already ran the sigma MLP (and filled t.sigma)
    // noised actions, preconditioned input, action embedding
    const int per = Ta * A;
    LAUNCH(mdt_launch_noise_input(t.action, noise, t.sigma, t.noised, Ma * A, per, s));
    LAUNCH(mdt_launch_scaled_input(t.noised, t.sigma, m->cfg.sigma_data, Ma * A, per, t.xin, s));
    LAUNCH(mdt_launch_action_embed(t.xin, nullptr, 0, m->cfg.sigma_data, m->Wa, m->ba, t.y0, (int)Ma, A, D, Ta, s));
    LAUNCH(mdt_launch_dropout_rows(t.y0, Ma, D, Ta, 0, t.drop.embed_p, site_id(m->Le + m->Ld, SITE_EMBED_ACTION), t.drop.seed, s));
    return MDT_OK;
}

static mdt_status dec_fwd(mdt_model* m, Tape& t, const float* action, const float* noise, const float* sigma, float* loss_out,
                          float* model_output, hipStream_t s, bool pre_done = false) {
    const int D = m->D, Ta = m->Ta, A = m->A;
    const int64_t B = t.B, Ma = B * Ta;
    // conditioning rows of the decoder blocks: adaLN-Zero (B, Ld*6D), NoiseBlock (B, D) = c, sigma token: none
    const int64_t modw = m->cond == COND_ADALN ? (int64_t)m->Ld * 6 * D : D;
    const int64_t mod_blk = m->cond == COND_ADALN ? 6 * D : 0;  // a block's offset in the row
    if (!pre_done) MDT_TRY(dec_fwd_pre(m, t, action, noise, sigma, s));
    const int per = Ta * A;
    // cross-attention K|V of all decoder blocks
    LAUNCH(mdt_launch_gemm(gemm_args(t.ctx, D, m->kv_all, t.kvx, (int64_t)m->Ld * 2 * D, (int)(B * m->Te)), s));
    for (int l = 0; l < m->Ld; ++l) t.dec[l].x_in = l == 0 ? t.y0 : t.dec[l - 1].x3;
    float* xl = t.dec[m->Ld - 1].x3;
    const mdt_ln_train_args lh = ln_args(xl, m->dec_ln_w, m->dec_ln_b, t.lnout, t.st_h, (int)Ma, D);
    auto mod_of = [&](int l) { return m->cond == COND_TOKEN ? (const float*)nullptr : t.mod + l * mod_blk; };
    for (int l = 0; l < m->Ld; ++l) {
        const mdt_ln_train_args tail = l + 1 < m->Ld ? block_ln1_args(m, m->dec[l + 1], t.dec[l + 1], B, Ta, m->cond, mod_of(l + 1), modw) : lh;
        MDT_TRY(block_fwd(m, m->dec[l], &m->dec[l], t.dec[l], B, Ta, true, m->cond, mod_of(l), modw, t.kvx + (int64_t)l * 2 * D,
                          t.drop, m->Le + l, s, l > 0, &tail, (l == 0 && pre_done) ? 2 : 0));
    }
    mdt_head_args h;
    memset(&h, 0, sizeof h);
    h.y = xl; h.ln_w = m->dec_ln_w; h.ln_b = m->dec_ln_b; h.Wp = m->Wp; h.bp = m->bp;
    h.x = t.noised; h.sigma = t.sigma; h.sigma_stride = 1; h.out = t.F;
    h.M = (int)Ma; h.D = D; h.A = A; h.rows_per_sample = Ta; h.mode = MDT_HEAD_RAW; h.sigma_data = m->cfg.sigma_data;
    if (m->HP) {  // MLP head: action_pred.0 on the normalised rows, GELU, action_pred.2 in the head kernel
        LAUNCH(mdt_launch_gemm(gemm_args(t.lnout, D, m->head0, t.hpre, m->HP, (int)Ma), s));
        LAUNCH(mdt_launch_act_fwd(t.hpre, t.hh, Ma * m->HP, MDT_ACT_GELU, s));
        h.y = t.hh; h.D = m->HP; h.no_ln = 1;
    }
    LAUNCH(mdt_launch_head(h, s));
    if (model_output) HIP_TRY(hipMemcpyAsync(model_output, t.F, (size_t)Ma * A * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (loss_out) LAUNCH(mdt_launch_loss_reduce(t.F, t.action, t.noised, t.sigma, m->cfg.sigma_data, Ma * A, per, loss_out, t.loss_part, s));
    t.has_decoder = true;
    return MDT_OK;
}

static mdt_status check_dropout(const mdt_dropout* d) {
    if (!d) return MDT_OK;
    for (float p : {d->attn_p, d->resid_p, d->mlp_p, d->embed_p})
        if (!(p >= 0.f && p < 1.f)) return fail(MDT_ERR_INVALID_ARG, "dropout probabilities must be in [0, 1)");
    return MDT_OK;
}

static mdt_dropout effective_dropout(const mdt_dropout* d) {
    mdt_dropout z = {0.f, 0.f, 0.f, 0.f, 0};
    if (!d || d->seed == 0 || (d->attn_p <= 0.f && d->resid_p <= 0.f && d->mlp_p <= 0.f && d->embed_p <= 0.f)) return z;
    return *d;
}

static mdt_status check_ready(mdt_model* m) {
    if (!m) return fail(MDT_ERR_INVALID_ARG, "null handle");
    if (!m->train) return fail(MDT_ERR_STATE, "training was not prepared: call mdt_train_prepare() and upload the parameters");
    for (const Slot& sl : m->slots)
        if (!sl.loaded) return fail(MDT_ERR_NOT_LOADED, "parameter '%s' was not loaded after mdt_train_prepare()", sl.name.c_str());
    return MDT_OK;
}

extern "C" mdt_status mdt_train_encode_fwd(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                           int32_t modality, int32_t honour_modality, const float* sigma, int64_t batch,
                                           const mdt_dropout* drop, float* ctx_out, mdt_tape_id* tape, void* stream) {
    MDT_TRY(check_ready(m));
    if (!tokens || !goal || !tape || batch < 1) return fail(MDT_ERR_INVALID_ARG, "mdt_train_encode_fwd: bad argument");
    if (m->cfg.arch == MDT_ARCH_MDT && !tokens2) return fail(MDT_ERR_INVALID_ARG, "MDT needs the gripper tokens");
    MDT_TRY(check_dropout(drop));
    MDT_TRY(acquire_tape(m, batch, tape, (hipStream_t)stream));
    m->train->tapes[*tape].drop = effective_dropout(drop);
    mdt_status st = enc_fwd(m, m->train->tapes[*tape], tokens, tokens2, goal, modality, honour_modality, sigma, ctx_out,
                            (hipStream_t)stream);
    if (st != MDT_OK) m->train->tapes[*tape].in_use = false;
    return st;
}

extern "C" mdt_status mdt_train_loss_fwd(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                         int32_t modality, const float* action, const float* noise, const float* sigma,
                                         int64_t batch, const mdt_dropout* drop, float* loss_out, float* model_output,
                                         float* ctx_out, mdt_tape_id* tape, void* stream) {
    MDT_TRY(check_ready(m));
    if (!tokens || !goal || !action || !noise || !sigma || !loss_out || !tape || batch < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_train_loss_fwd: bad argument");
    if (m->cfg.arch == MDT_ARCH_MDT && !tokens2) return fail(MDT_ERR_INVALID_ARG, "MDT needs the gripper tokens");
    hipStream_t s = (hipStream_t)stream;
    MDT_TRY(check_dropout(drop));
    MDT_TRY(acquire_tape(m, batch, tape, s));
    Tape& t = m->train->tapes[*tape];
    t.drop = effective_dropout(drop);
    const int honour = m->cfg.arch == MDT_ARCH_MDTV;  // MDTTransformer.forward always uses goal_emb (mdt_transformer.py:215)
    // The decoder's context-independent preparation (sigma MLP, stacked adaLN Linear, action embedding: ~10 small launches) on the
    // side stream beside the encoder (round 6, MDT_HIP_DW_STREAM); the chain waits for it in front of the first decoder kernel
    mdt_train_state* ts = m->train;
    const bool pre_beside = dw_stream_mode() > 0 && m->cond != COND_TOKEN;
    mdt_status st = MDT_OK;
    if (pre_beside) {
        hipStream_t sq = nullptr;
        st = side_fork(ts, 0, s, &sq);
        if (st == MDT_OK) st = dec_fwd_pre(m, t, action, noise, sigma, sq);
        if (st == MDT_OK) {  // ... and the first decoder block's self-attention half (it reads the action embedding, not the context)
            const int64_t modw = m->cond == COND_ADALN ? (int64_t)m->Ld * 6 * m->D : m->D;
            t.dec[0].x_in = t.y0;
            st = block_fwd(m, m->dec[0], &m->dec[0], t.dec[0], batch, m->Ta, true, m->cond, t.mod, modw, nullptr, t.drop, m->Le, sq,
                           false, nullptr, 1);
        }
        if (st == MDT_OK && !ts->fwd_join && hipEventCreateWithFlags(&ts->fwd_join, hipEventDisableTiming) != hipSuccess)
            st = fail(MDT_ERR_HIP, "could not create the forward join event");
        if (st == MDT_OK && hipEventRecord(ts->fwd_join, sq) != hipSuccess) st = fail(MDT_ERR_HIP, "event record failed");
    }
    if (st == MDT_OK) st = enc_fwd(m, t, tokens, tokens2, goal, modality, honour, sigma, ctx_out, s);
    if (st == MDT_OK && pre_beside && hipStreamWaitEvent(s, ts->fwd_join, 0) != hipSuccess) st = fail(MDT_ERR_HIP, "stream wait failed");
    if (st == MDT_OK) st = dec_fwd(m, t, action, noise, sigma, loss_out, model_output, s, pre_beside);
    if (st != MDT_OK) t.in_use = false;
    return st;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// side stream q of the handle (created on first use)
static mdt_status side_stream(mdt_train_state* ts, int q, hipStream_t* out) {
    if (!ts->side[q]) {
        // MDT_HIP_DW_PRIO (A/B runs): 0 = default priority, 1 = the LOWEST the device offers (the chain's kernels first, the
        // weight gradients in what is left: measured 9.17 against 9.08 ms), 2 = the highest (9.08)
        static int prio = -1;
        if (prio < 0) { const char* e = getenv("MDT_HIP_DW_PRIO"); prio = e ? atoi(e) : 0; }
        int least = 0, greatest = 0;
        if (prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
            HIP_TRY(hipStreamCreateWithPriority(&ts->side[q], hipStreamNonBlocking, prio == 1 ? least : greatest));
        else
            HIP_TRY(hipStreamCreateWithFlags(&ts->side[q], hipStreamNonBlocking));
    }
    *out = ts->side[q];
    return MDT_OK;
}
// side stream q ordered behind everything enqueued on `s` so far
static mdt_status side_fork(mdt_train_state* ts, int q, hipStream_t s, hipStream_t* out) {
    MDT_TRY(side_stream(ts, q, out));
    hipEvent_t& ev = ts->side_ev[ts->side_ev_next++ & 7];
    if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, s));
    HIP_TRY(hipStreamWaitEvent(*out, ev, 0));
    return MDT_OK;
}

// gradient slots of a (possibly stacked) Linear: every reference Linear inside `l`
// beside: this product's dY lives in the arena (dy_take) -- its weight gradient may run on the side stream
static mdt_status lin_bwd(mdt_model* m, float* grads, const Lin& l, const float* X, int64_t ldx, const float* dY, int64_t ldy,
                          int M, float* dX, int64_t ldxo, int acc_dx, hipStream_t s, const float* act_u = nullptr, int act = 0,
                          bool beside = false, float* scratch_override = nullptr) {
    mdt_train_state* ts = m->train;
    const LinPart* first = nullptr;  // the part at row 0: the stack's gradient region starts at its slot
    for (const LinPart& p : m->parts)
        if (p.lin == &l && p.n_off == 0) { first = &p; break; }
    if (!first) return fail(MDT_ERR_STATE, "lin_bwd: Linear without a registered parameter");
    mdt_linear_bwd_args a;
    memset(&a, 0, sizeof a);
    a.X = X; a.ldx = ldx; a.dY = dY; a.ldy = ldy;
    // grads == nullptr: an input-gradient-only backward (mdt_denoise_vjp) -- no parameter gradient is formed
    a.dW = grads ? grads + ts->grad_off[first->w_slot] : nullptr;
    a.dbias = (grads && first->b_slot >= 0) ? grads + ts->grad_off[first->b_slot] : nullptr;
    a.accumulate_dw = 1;
    a.Wt = l.wt; a.dX = dX; a.ldxo = ldxo; a.accumulate_dx = acc_dx;
    a.M = M; a.N = l.N; a.K = l.K; a.scratch = scratch_override ? scratch_override : ts->lin_scratch;
    a.dx_act_u = act_u; a.dx_act = act;
    mdt_colsum_entry be;
    float* space = a.dbias ? defer_take(ts, (int64_t)256 * l.N) : nullptr;
    if (beside && a.dW && ts->dy_arena) {
        // dW (+ bias partials) on the side stream behind "dY exists"; dX stays in the chain
        const int q = ts->side_rr++ % dw_stream_mode();
        hipStream_t sq;
        MDT_TRY(side_fork(ts, q, s, &sq));
        mdt_linear_bwd_args w = a;
        w.dX = nullptr; w.scratch = ts->lin_scratch2[q];
        MDT_TRY(mdt_linear_bwd(w, sq, space ? &be : nullptr, space));
        if (space && be.src) ts->deferred.push_back(be);
        ts->side_used[q] = true;
        if (a.dX) {
            mdt_linear_bwd_args x = a;
            x.dW = nullptr; x.dbias = nullptr;
            MDT_TRY(mdt_linear_bwd(x, s));
        }
        return MDT_OK;
    }
    MDT_TRY(mdt_linear_bwd(a, s, space ? &be : nullptr, space));
    if (space && be.src) ts->deferred.push_back(be);
    return MDT_OK;
}

static int slot_of(const mdt_model* m, const float* dst) {
    for (size_t i = 0; i < m->slots.size(); ++i)
        if (m->slots[i].dst == dst) return (int)i;
    return -1;
}

static float* grad_of(mdt_model* m, float* grads, const float* param) {
    const int i = param ? slot_of(m, param) : -1;
    return i < 0 ? nullptr : grads + m->train->grad_off[i];
}

// LayerNorm backward + reduction of the per-sample weight/bias partials into the gradient slots
// mg: the backward of the branch merge that follows in the backward order, on the gradient this call leaves in dx (its
// `x`): rides in the same launch (train_fuse bit 1) or runs behind it
static mdt_status ln_bwd(mdt_model* m, float* grads, const float* x, const float* stats, const float* w, const float* b,
                         const float* mod, int64_t modw, int shift_off, int scale_off, const float* dh, float* dx, int acc,
                         float* d_mod, int64_t B, int T, hipStream_t s, int acc_dmod = 0, const mdt_merge_args* mg = nullptr) {
    mdt_train_state* ts = m->train;
    mdt_ln_bwd_args a;
    memset(&a, 0, sizeof a);
    a.x = x; a.stats = stats; a.w = w; a.b = b; a.mod = mod; a.mod_stride = modw; a.shift_off = shift_off; a.scale_off = scale_off;
    a.dh = dh; a.ld_dh = m->D; a.dx = dx; a.accumulate = acc; a.d_mod = d_mod; a.d_mod_stride = modw;
    a.accumulate_dmod = acc_dmod;
    if (shift_off < 0 && scale_off < 0) { a.mod = nullptr; a.d_mod = nullptr; }
    // per-sample partials of the weight / bias gradient: into the deferred area when their sums run at the end of the backward
    float* pw = grads ? defer_take(ts, B * m->D) : nullptr;
    float* pb = (grads && b) ? defer_take(ts, B * m->D) : nullptr;
    const bool deferred = pw && (!b || pb);
    a.pw = deferred ? pw : ts->pw; a.pb = b ? (deferred ? pb : ts->pb) : nullptr;
    a.B = (int)B; a.rows_per_sample = T; a.D = m->D;
    if (mg && (train_fuse() & 2)) {
        LAUNCH(mdt_launch_ln_bwd_merge(a, *mg, s));
    } else {
        LAUNCH(mdt_launch_ln_bwd(a, s));
        if (mg) LAUNCH(mdt_launch_merge_bwd(*mg, s));
    }
    if (!grads) return MDT_OK;
    if (deferred) {
        ts->deferred.push_back(mdt_colsum_entry{pw, grad_of(m, grads, w), (int64_t)m->D, (int)B, m->D, 1});
        if (b) ts->deferred.push_back(mdt_colsum_entry{pb, grad_of(m, grads, b), (int64_t)m->D, (int)B, m->D, 1});
        return MDT_OK;
    }
    if (b) LAUNCH(mdt_launch_colsum2(ts->pw, ts->pb, m->D, (int)B, m->D, grad_of(m, grads, w), grad_of(m, grads, b), 1, s));
    else LAUNCH(mdt_launch_colsum(ts->pw, m->D, (int)B, m->D, grad_of(m, grads, w), 1, s));
    return MDT_OK;
}

static mdt_attn_bwd_args attn_bwd_args(mdt_model* m, const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                       const float* d_out, float* dq, int64_t ld_dq, float* dk, float* dv, int64_t ld_dkv,
                                       int64_t B, int Tq, int Tk, bool causal, const mdt_dropout& dr, uint32_t site) {
    mdt_attn_bwd_args a;
    memset(&a, 0, sizeof a);
    a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ldkv = ldkv; a.d_out = d_out; a.ld_do = m->D;
    a.dq = dq; a.ld_dq = ld_dq; a.dk = dk; a.dv = dv; a.ld_dkv = ld_dkv; a.accumulate_kv = 0;
    a.B = (int)B; a.H = m->H; a.hd = m->hd; a.Tq = Tq; a.Tk = Tk; a.causal = causal;
    a.p = dr.seed ? dr.attn_p : 0.f; a.site = site; a.seed = dr.seed;
    a.rope = m->cfg.use_rot_embed; a.rope_cos = m->rope_cos; a.rope_sin = m->rope_sin;
    return a;
}

// backward of a block's LAST merge (the MLP branch): d_mo = mask/(1-p) * gate_mlp * dx -> ts->t_d2, d_gate_mlp -> d_mod
static mdt_merge_args block_mlp_merge_bwd(mdt_model* m, BlockTape& t, int64_t B, int T, int cond, const float* mod, float* d_mod,
                                          int64_t modw, float* dx, const mdt_dropout& dr, int blk) {
    const CondLayout c = cond_layout(mod ? cond : COND_TOKEN, m->D);
    mdt_merge_args g = merge_args(dx, t.mo, c.g2 >= 0 ? mod + c.g2 : nullptr, modw,
                                  dy_take(m->train, B * T * m->D, m->train->t_d2), B, T, m->D, dr.mlp_p, site_id(blk, SITE_MLP), dr.seed);
    if (c.g2 >= 0) { g.dgate = d_mod + c.g2; g.dgate_stride = modw; }
    return g;
}

// dx: gradient wrt the block's output on entry, wrt its input on return (in place)
// d_mod mirrors mod (same stride and offsets); NoiseBlock: every LayerNorm ADDS its d_shift into the one d_c row.
// Buffers: every sublayer finds its merged branch gradient in a dY buffer (ts->t_d2, or its own piece of the arena when the
// weight gradients run beside the chain) and leaves the gradient of its LayerNorm output in ts->t_d; the LayerNorm backward that
// consumes it also runs the NEXT sublayer's merge backward.
// gm: this block's MLP-merged gradient -- the caller's preceding LayerNorm backward ran block_mlp_merge_bwd into it;
// tail_merge: the merge backward that follows this block in the backward order (the previous block's MLP merge).
static mdt_status block_bwd(mdt_model* m, float* grads, const EncBlock& e, const DecBlock* d, BlockTape& t, int64_t B, int T,
                            bool causal, int cond, const float* mod, float* d_mod, int64_t modw, const float* kv, float* d_kv,
                            float* dx, const mdt_dropout& dr, int blk, hipStream_t s, const float* gm,
                            const mdt_merge_args* tail_merge = nullptr) {
    mdt_train_state* ts = m->train;
    const int D = m->D, M = (int)(B * T);
    const int64_t MD = (int64_t)M * D;
    const CondLayout c = cond_layout(mod ? cond : COND_TOKEN, D);
    const int acc_dmod = cond == COND_NOISE;
    const bool bs = true;  // dY buffers below come from dy_take
    // the self-attention branch's merge: x1 = x_in + g1 * drop(a1)
    float* g_self = dy_take(ts, MD, ts->t_d2);
    mdt_merge_args gs = merge_args(dx, t.a1, c.g1 >= 0 ? mod + c.g1 : nullptr, modw, g_self, B, T, D, dr.resid_p,
                                   site_id(blk, SITE_RESID), dr.seed);
    if (c.g1 >= 0) { gs.dgate = d_mod + c.g1; gs.dgate_stride = modw; }
    // ---- MLP half: x3 = x2 + g2 * drop(c_proj(gelu(c_fc(h2))))
    // d_u = (d_mo W_proj) * gelu'(u): the GELU backward is the epilogue of c_proj's input-gradient product
    float* d_u = dy_take(ts, 4 * MD, ts->t_4d);
    MDT_TRY(lin_bwd(m, grads, e.proj2, t.hid, 4 * D, gm, D, M, d_u, 4 * D, 0, s, t.u, MDT_ACT_GELU, bs));
    MDT_TRY(lin_bwd(m, grads, e.fc, t.h2, D, d_u, 4 * D, M, ts->t_d, D, 0, s, nullptr, 0, bs));
    if (d) {
        // ---- cross-attention half: x2 = x1 + drop(c_proj(attn(q(ln3(x1)), K, V)))
        float* g_x = dy_take(ts, MD, ts->t_d2);
        const mdt_merge_args gx = merge_args(dx, t.a2, nullptr, 0, g_x, B, T, D, dr.resid_p, site_id(blk, SITE_XRESID), dr.seed);
        MDT_TRY(ln_bwd(m, grads, t.x2, t.st2, e.ln2_w, e.ln2_b, mod, modw, c.sh2, c.sc2, ts->t_d, dx, 1, d_mod, B, T, s, acc_dmod, &gx));
        MDT_TRY(lin_bwd(m, grads, d->xproj, t.att2, D, g_x, D, M, ts->t_d, D, 0, s, nullptr, 0, bs));
        float* d_q = dy_take(ts, MD, ts->t_d2);
        LAUNCH(mdt_launch_attn_bwd(attn_bwd_args(m, t.q, D, kv, kv + D, (int64_t)m->Ld * 2 * D, ts->t_d, d_q, D, d_kv,
                                                 d_kv + D, (int64_t)m->Ld * 2 * D, B, T, m->Te, true, dr,
                                                 site_id(blk, SITE_XATTN)), s));
        MDT_TRY(lin_bwd(m, grads, d->xq, t.h3, D, d_q, D, M, ts->t_d, D, 0, s, nullptr, 0, bs));
        MDT_TRY(ln_bwd(m, grads, t.x1, t.st3, d->ln3_w, d->ln3_b, mod, modw, c.sh3, -1, ts->t_d, dx, 1, d_mod, B, T, s, acc_dmod, &gs));
    } else {
        MDT_TRY(ln_bwd(m, grads, t.x2, t.st2, e.ln2_w, e.ln2_b, mod, modw, c.sh2, c.sc2, ts->t_d, dx, 1, d_mod, B, T, s, acc_dmod, &gs));
    }
    // ---- self-attention half: x1 = x_in + g1 * drop(c_proj(attn(qkv(h1))))
    MDT_TRY(lin_bwd(m, grads, e.proj, t.att, D, g_self, D, M, ts->t_d, D, 0, s, nullptr, 0, bs));
    float* d_qkv = dy_take(ts, 3 * MD, ts->t_3d);
    LAUNCH(mdt_launch_attn_bwd(attn_bwd_args(m, t.qkv, 3 * D, t.qkv + D, t.qkv + 2 * D, 3 * D, ts->t_d, d_qkv, 3 * D,
                                             d_qkv + D, d_qkv + 2 * D, 3 * D, B, T, T, causal, dr, site_id(blk, SITE_ATTN)),
                               s));
    MDT_TRY(lin_bwd(m, grads, e.qkv, t.h1, D, d_qkv, 3 * D, M, ts->t_d, D, 0, s, nullptr, 0, bs));
    MDT_TRY(ln_bwd(m, grads, t.x_in, t.st1, e.ln1_w, e.ln1_b, mod, modw, c.sh1, c.sc1, ts->t_d, dx, 1, d_mod, B, T, s, acc_dmod, tail_merge));
    return MDT_OK;
}

// backward of sigma_fwd given d(c) (B, D) in `dc` (consumed); sigma itself takes no gradient
// tmp (B, 2D) / scratch: the caller's own buffers when this runs beside the chain (else the shared ones)
static mdt_status sigma_bwd(mdt_model* m, Tape& t, float* grads, float* dc, hipStream_t s, float* tmp = nullptr,
                            float* scratch = nullptr) {
    mdt_train_state* ts = m->train;
    const int D = m->D;
    const int64_t B = t.B;
    if (!tmp) tmp = ts->t_d2;
    MDT_TRY(lin_bwd(m, grads, m->sig3, t.sig_t, 2 * D, dc, D, (int)B, tmp, 2 * D, 0, s, nullptr, 0, false, scratch));
    LAUNCH(mdt_launch_act_bwd(t.sig_tpre, tmp, tmp, B * 2 * D, MDT_ACT_MISH, s));
    MDT_TRY(lin_bwd(m, grads, m->sig1, t.sig_e, D, tmp, 2 * D, (int)B, nullptr, 0, 0, s, nullptr, 0, false, scratch));
    return MDT_OK;
}

// encoder backward: dxe holds d(ctx) on entry
// The encoder backward in the pieces a staged backward runs one by one (ts->gm: the MLP-merged gradient of the block about to run):
// head = the final LayerNorm, one piece per block, tail = the token embeddings.  ts->dxe holds d(ctx) on entry.
static mdt_merge_args enc_merge_bwd(mdt_model* m, Tape& t, int l) {
    return block_mlp_merge_bwd(m, t.enc[l], t.B, m->Te, COND_TOKEN, nullptr, nullptr, 0, m->train->dxe, t.drop, l);
}
static mdt_status enc_bwd_head(mdt_model* m, Tape& t, float* grads, hipStream_t s) {
    mdt_train_state* ts = m->train;
    const int D = m->D, Te = m->Te;
    const int64_t B = t.B, Me = B * Te;
    // final LayerNorm: ctx = ln(x_L)
    HIP_TRY(hipMemcpyAsync(ts->t_d, ts->dxe, (size_t)Me * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    const mdt_merge_args g = m->Le > 0 ? enc_merge_bwd(m, t, m->Le - 1) : mdt_merge_args{};
    MDT_TRY(ln_bwd(m, grads, enc_last_output(m, t), t.st_f, m->enc_ln_w, m->enc_ln_b, nullptr, 0, -1, -1, ts->t_d, ts->dxe, 0,
                   nullptr, B, Te, s, 0, m->Le > 0 ? &g : nullptr));
    ts->gm = g.out;
    return MDT_OK;
}
static mdt_status enc_bwd_block(mdt_model* m, Tape& t, float* grads, int l, hipStream_t s) {
    mdt_train_state* ts = m->train;
    const mdt_merge_args g = l > 0 ? enc_merge_bwd(m, t, l - 1) : mdt_merge_args{};
    MDT_TRY(block_bwd(m, grads, m->enc[l], nullptr, t.enc[l], t.B, m->Te, false, COND_TOKEN, nullptr, nullptr, 0, nullptr, nullptr,
                      ts->dxe, t.drop, l, s, ts->gm, l > 0 ? &g : nullptr));
    ts->gm = g.out;
    return MDT_OK;
}
static mdt_status enc_bwd_tail(mdt_model* m, Tape& t, float* grads, float* d_tokens, float* d_tokens2, float* d_goal,
                               hipStream_t s) {
    mdt_train_state* ts = m->train;
    const mdt_config& c = m->cfg;
    const int D = m->D, Te = m->Te;
    const int64_t B = t.B, Me = B * Te;
    const int t0 = m->sig_tok;
    const int drop_lo = embed_drop_from(m);
    if (drop_lo >= 0)
        LAUNCH(mdt_launch_dropout_rows(ts->dxe, Me, D, Te, drop_lo, t.drop.embed_p, site_id(m->Le + m->Ld, SITE_EMBED_CTX),
                                       t.drop.seed, s));
    if (t0) {  // the sigma token is row 0 of every sample's context
        LAUNCH(mdt_launch_gather_rows(ts->dxe, ts->small, (int)B, D, 1, Te, 0, s));
        MDT_TRY(sigma_bwd(m, t, grads, ts->small, s));
    }
    // token embeddings: the forward scattered their rows into the context
    const Lin& g0 = t.lang ? m->lang0 : m->goal0;
    const Lin& g2 = t.lang ? m->lang2 : m->goal2;
    float* dg = ts->t_d;  // (B, D)
    const bool has_goal = m->g_row >= 0;
    if (has_goal) LAUNCH(mdt_launch_gather_rows(ts->dxe, dg, (int)B, D, 1, Te, m->g_row, s));
    else if (d_goal) HIP_TRY(hipMemsetAsync(d_goal, 0, (size_t)B * m->G * sizeof(float), s));
    const bool pos = c.arch == MDT_ARCH_MDT && c.use_abs_pos_emb;
    float* g_pos = pos ? grad_of(m, grads, m->pos_emb) : nullptr;
    if (pos && has_goal) LAUNCH(mdt_launch_colsum(dg, D, (int)B, D, g_pos, 1, s));
    if (!has_goal) {
        // goal_conditioned=False in MDTTransformer: the goal never enters the context
    } else if (c.use_mlp_goal) {
        MDT_TRY(lin_bwd(m, grads, g2, t.g_h, 2 * D, dg, D, (int)B, ts->small, 2 * D, 0, s));
        LAUNCH(mdt_launch_act_bwd(t.g_pre, ts->small, ts->small, B * 2 * D, MDT_ACT_GELU, s));
        MDT_TRY(lin_bwd(m, grads, g0, t.goal, m->G, ts->small, 2 * D, (int)B, d_goal, m->G, 0, s));
    } else {
        MDT_TRY(lin_bwd(m, grads, g2, t.goal, m->G, dg, D, (int)B, d_goal, m->G, 0, s));
    }
    if (m->p_row >= 0) {
        // proprio_emb: token = Linear2(mish(Linear0(state_obs))); the narrow first layer's weight gradient lands in the
        // reference's (2D, Pd) layout, summed over the row slices like action_emb's
        float* dp = ts->t_d2;  // (B, D)
        LAUNCH(mdt_launch_gather_rows(ts->dxe, dp, (int)B, D, 1, Te, m->p_row, s));
        MDT_TRY(lin_bwd(m, grads, m->prop2, t.p_h, 2 * D, dp, D, (int)B, ts->small, 2 * D, 0, s));
        LAUNCH(mdt_launch_act_bwd(t.p_pre, ts->small, ts->small, B * 2 * D, MDT_ACT_MISH, s));
        LAUNCH(mdt_launch_colsum(ts->small, 2 * D, (int)B, 2 * D, grad_of(m, grads, m->prop0_b), 1, s));
        LAUNCH(mdt_launch_narrow_dw(t.tokens2, ts->small, 2 * D, ts->narrow, NARROW_SLICES, (int)B, m->Pd, 2 * D, 1, s));
        LAUNCH(mdt_launch_colsum(ts->narrow, (int64_t)m->Pd * 2 * D, NARROW_SLICES, m->Pd * 2 * D, grad_of(m, grads, m->prop0_T), 1, s));
        if (d_tokens2) LAUNCH(mdt_launch_narrow_out(ts->small, 2 * D, m->prop0_T, d_tokens2, (int)B, m->Pd, 2 * D, s));
    }
    if (c.arch == MDT_ARCH_MDTV) {
        float* dt = ts->t_d2;  // (B*n_tok, D)
        LAUNCH(mdt_launch_gather_rows(ts->dxe, dt, (int)(B * m->n_tok), D, m->n_tok, Te, m->tok_row, s));
        MDT_TRY(lin_bwd(m, grads, m->tok, t.tokens, m->O, dt, D, (int)(B * m->n_tok), d_tokens, m->O, 0, s));
    } else {
        float* d1 = ts->t_d2;
        LAUNCH(mdt_launch_gather_rows(ts->dxe, d1, (int)B, D, 1, Te, m->tok_row, s));
        if (pos) LAUNCH(mdt_launch_colsum(d1, D, (int)B, D, g_pos + (int64_t)c.goal_seq_len * D, 1, s));
        MDT_TRY(lin_bwd(m, grads, m->tok, t.tokens, m->O, d1, D, (int)B, d_tokens, m->O, 0, s));
        LAUNCH(mdt_launch_gather_rows(ts->dxe, d1, (int)B, D, 1, Te, m->tok_row + 1, s));
        if (pos) LAUNCH(mdt_launch_colsum(d1, D, (int)B, D, g_pos + (int64_t)c.goal_seq_len * D, 1, s));
        MDT_TRY(lin_bwd(m, grads, m->incam, t.tokens2, m->O, d1, D, (int)B, d_tokens2, m->O, 0, s));
    }
    return MDT_OK;
}
static mdt_status enc_bwd(mdt_model* m, Tape& t, float* grads, float* d_tokens, float* d_tokens2, float* d_goal, hipStream_t s) {
    MDT_TRY(enc_bwd_head(m, t, grads, s));
    for (int l = m->Le - 1; l >= 0; --l) MDT_TRY(enc_bwd_block(m, t, grads, l, s));
    return enc_bwd_tail(m, t, grads, d_tokens, d_tokens2, d_goal, s);
}

extern "C" mdt_status mdt_train_encode_bwd(mdt_model* m, mdt_tape_id tape, const float* g_ctx, float* grads, float* d_tokens,
                                           float* d_tokens2, float* d_goal, void* stream) {
    Tape* t;
    MDT_TRY(get_tape(m, tape, &t));
    if (!g_ctx || !grads) return fail(MDT_ERR_INVALID_ARG, "mdt_train_encode_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    t->stream = s;
    MDT_TRY(scratch_enter(m, s));
    MDT_TRY(reserve_scratch(m, t->B));
    m->train->deferred.clear(); m->train->defer_off = 0; m->train->dy_off = 0;
    HIP_TRY(hipMemcpyAsync(m->train->dxe, g_ctx, (size_t)t->B * m->Te * m->D * sizeof(float), hipMemcpyDeviceToDevice, s));
    MDT_TRY(enc_bwd(m, *t, grads, d_tokens, d_tokens2, d_goal, s));
    return flush_deferred(m, s);
}

// Decoder backward from ts->dF = d(raw network output F) down to ts->dx = d(action embedding rows y0); the K|V gradient of
// every block is left in ts->d_kvx.  grads == nullptr: input gradients only (no parameter gradient, no sigma path).
// In the pieces a staged backward runs one by one: head = action head + final LayerNorm, one piece per block, tail = action
// embedding + sigma path.
struct DecCond {
    int64_t modw, mod_blk;
    bool rows;
};
static DecCond dec_cond(const mdt_model* m) {
    return {m->cond == COND_ADALN ? (int64_t)m->Ld * 6 * m->D : m->D, m->cond == COND_ADALN ? 6 * (int64_t)m->D : 0, m->cond != COND_TOKEN};
}
static mdt_merge_args dec_merge_bwd(mdt_model* m, Tape& t, int l) {
    const DecCond dc = dec_cond(m);
    return block_mlp_merge_bwd(m, t.dec[l], t.B, m->Ta, m->cond, dc.rows ? t.mod + l * dc.mod_blk : nullptr,
                               dc.rows ? m->train->d_mod + l * dc.mod_blk : nullptr, dc.modw, m->train->dx, t.drop, m->Le + l);
}
static mdt_status dec_bwd_head(mdt_model* m, Tape& t, float* grads, hipStream_t s) {
    mdt_train_state* ts = m->train;
    const int D = m->D, Ta = m->Ta, A = m->A;
    const int64_t B = t.B, Ma = B * Ta;
    if (grads) LAUNCH(mdt_launch_colsum(ts->dF, A, (int)Ma, A, grad_of(m, grads, m->bp), 1, s));
    if (m->HP) {
        // F = action_pred.2(gelu(action_pred.0(ln))): the narrow layer on the (rows, HP) hidden rows, whose gradient
        // lands padded (A, HP) and is added into the (A, HH) slot; then GELU and the d x HP Linear on the GEMM
        const int HP = m->HP;
        if (grads) {
            LAUNCH(mdt_launch_narrow_dw(ts->dF, t.hh, HP, ts->narrow, NARROW_SLICES, (int)Ma, A, HP, 0, s));
            LAUNCH(mdt_launch_colsum(ts->narrow, (int64_t)A * HP, NARROW_SLICES, A * HP, ts->small, 0, s));
            LAUNCH(mdt_launch_add_2d(ts->small, HP, grad_of(m, grads, m->Wp), m->HH, A, m->HH, s));
        }
        LAUNCH(mdt_launch_narrow_dx(ts->dF, m->Wp, ts->t_4d, (int)Ma, A, HP, s));
        LAUNCH(mdt_launch_act_bwd(t.hpre, ts->t_4d, ts->t_4d, Ma * HP, MDT_ACT_GELU, s));
        MDT_TRY(lin_bwd(m, grads, m->head0, t.lnout, D, ts->t_4d, HP, (int)Ma, ts->t_d, D, 0, s));
    } else {
        if (grads) {
            LAUNCH(mdt_launch_narrow_dw(ts->dF, t.lnout, D, ts->narrow, NARROW_SLICES, (int)Ma, A, D, 0, s));
            LAUNCH(mdt_launch_colsum(ts->narrow, (int64_t)A * D, NARROW_SLICES, A * D, grad_of(m, grads, m->Wp), 1, s));
        }
        LAUNCH(mdt_launch_narrow_dx(ts->dF, m->Wp, ts->t_d, (int)Ma, A, D, s));
    }
    if (m->cond == COND_NOISE) HIP_TRY(hipMemsetAsync(ts->d_mod, 0, (size_t)B * D * sizeof(float), s));  // d_c accumulates
    // the final LayerNorm's backward also runs the last block's MLP merge backward
    const mdt_merge_args g = dec_merge_bwd(m, t, m->Ld - 1);
    MDT_TRY(ln_bwd(m, grads, t.dec[m->Ld - 1].x3, t.st_h, m->dec_ln_w, m->dec_ln_b, nullptr, 0, -1, -1, ts->t_d, ts->dx, 0, nullptr,
                   B, Ta, s, 0, &g));
    ts->gm = g.out;
    return MDT_OK;
}
static mdt_status dec_bwd_block(mdt_model* m, Tape& t, float* grads, int l, hipStream_t s) {
    mdt_train_state* ts = m->train;
    const int D = m->D;
    const DecCond dc = dec_cond(m);
    const mdt_merge_args g = l > 0 ? dec_merge_bwd(m, t, l - 1) : mdt_merge_args{};
    MDT_TRY(block_bwd(m, grads, m->dec[l], &m->dec[l], t.dec[l], t.B, m->Ta, true, m->cond, dc.rows ? t.mod + l * dc.mod_blk : nullptr,
                      dc.rows ? ts->d_mod + l * dc.mod_blk : nullptr, dc.modw, t.kvx + (int64_t)l * 2 * D,
                      ts->d_kvx + (int64_t)l * 2 * D, ts->dx, t.drop, m->Le + l, s, ts->gm, l > 0 ? &g : nullptr));
    ts->gm = g.out;
    return MDT_OK;
}
// The tail's gradients (action_emb, the sigma MLP, the stacked adaLN Linear) are leaves: nothing in the rest of the backward --
// the cross K|V Linear and the encoder -- reads them or the buffers they come from (ts->dx, ts->d_mod are final).  With the
// weight gradients beside the chain (MDT_HIP_DW_STREAM) the whole tail therefore runs on the side stream, with its own scratch,
// while the chain goes on into the encoder (~0.25 ms of small launches at B = 1024).
static mdt_status dec_bwd_tail(mdt_model* m, Tape& t, float* grads, hipStream_t s) {
    mdt_train_state* ts = m->train;
    const int D = m->D, Ta = m->Ta, A = m->A;
    const int64_t B = t.B, Ma = B * Ta;
    const DecCond dc = dec_cond(m);
    const bool beside = grads && ts->dy_arena && ts->small2;
    float *narrow = ts->narrow, *small = ts->small, *tmp = nullptr, *scratch = nullptr;
    if (beside) {
        MDT_TRY(side_fork(ts, 0, s, &s));  // from here on `s` is the side stream
        ts->side_used[0] = true;
        narrow = ts->narrow2; small = ts->small2; tmp = ts->small2 + B * D; scratch = ts->lin_scratch2[0];
    }
    // ---- action embedding: y0 = drop(action_emb(xin)); no gradient flows to the noisy actions
    LAUNCH(mdt_launch_dropout_rows(ts->dx, Ma, D, Ta, 0, t.drop.embed_p, site_id(m->Le + m->Ld, SITE_EMBED_ACTION), t.drop.seed,
                                   s));
    if (!grads) return MDT_OK;
    LAUNCH(mdt_launch_colsum(ts->dx, D, (int)Ma, D, grad_of(m, grads, m->ba), 1, s));
    LAUNCH(mdt_launch_narrow_dw(t.xin, ts->dx, D, narrow, NARROW_SLICES, (int)Ma, A, D, 1, s));
    LAUNCH(mdt_launch_colsum(narrow, (int64_t)A * D, NARROW_SLICES, A * D, grad_of(m, grads, m->Wa), 1, s));
    // ---- sigma path.  adaLN: mod = modulation(silu(c)), c = sigma_emb(sigma); NoiseBlock: the rows are c itself;
    //      sigma token: its gradient arrives with the context's (enc_bwd)
    if (m->cond == COND_ADALN) {
        MDT_TRY(lin_bwd(m, grads, m->mod_all, t.sig_s, D, ts->d_mod, dc.modw, (int)B, small, D, 0, s, nullptr, 0, false, scratch));
        LAUNCH(mdt_launch_act_bwd(t.sig_cpre, small, small, B * D, MDT_ACT_SILU, s));
        MDT_TRY(sigma_bwd(m, t, grads, small, s, tmp, scratch));
    } else if (m->cond == COND_NOISE) {
        MDT_TRY(sigma_bwd(m, t, grads, ts->d_mod, s, tmp, scratch));
    }
    return MDT_OK;
}
static mdt_status dec_bwd(mdt_model* m, Tape& t, float* grads, hipStream_t s) {
    MDT_TRY(dec_bwd_head(m, t, grads, s));
    for (int l = m->Ld - 1; l >= 0; --l) MDT_TRY(dec_bwd_block(m, t, grads, l, s));
    return dec_bwd_tail(m, t, grads, s);
}

// The backward of mdt_train_loss_fwd in STAGES (round 6): stage k completes the gradients of a known set of parameters
// (mdt_train_param_stage) in the order of `stream`, so that a caller -- torch's DistributedDataParallel behind one autograd node
// per stage -- can start reducing them while the later stages still run.
//   0               : loss, action head, final LayerNorm, decoder block Ld - 1
//   1 .. Ld - 1     : decoder blocks Ld - 2 .. 0
//   Ld              : action embedding, sigma path (sigma MLP, the stacked adaLN Linear), the stacked cross K|V Linear, the
//                     encoder's final LayerNorm
//   Ld + 1 .. Ld + Le : encoder blocks Le - 1 .. 0
//   Ld + Le + 1     : token / goal embeddings (and the gradients of the encoder inputs)
extern "C" int32_t mdt_train_loss_bwd_stages(const mdt_model* m) { return m ? m->Ld + m->Le + 2 : -1; }

// sync: make the stage's gradients complete in the order of `stream` before returning (what the staged entry point promises);
// the one-call backward only needs that at its end -- its side-stream work then overlaps across the stages
static mdt_status loss_bwd_stage_impl(mdt_model* m, mdt_tape_id tape, int32_t stage, const float* g_loss, const float* g_ctx,
                                      float* grads, float* d_tokens, float* d_tokens2, float* d_goal, void* stream, bool sync) {
    Tape* tp;
    MDT_TRY(get_tape(m, tape, &tp));
    Tape& t = *tp;
    if (!grads) return fail(MDT_ERR_INVALID_ARG, "mdt_train_loss_bwd: null gradient buffer");
    if (!t.has_decoder) return fail(MDT_ERR_STATE, "tape %d holds an encoder-only forward: use mdt_train_encode_bwd", tape);
    hipStream_t s = (hipStream_t)stream;
    mdt_train_state* ts = m->train;
    const int Ld = m->Ld, Le = m->Le, n = Ld + Le + 2;
    if (stage < 0 || stage >= n) return fail(MDT_ERR_INVALID_ARG, "mdt_train_loss_bwd_stage: stage %d of %d", stage, n);
    if (stage == 0) {
        t.stream = s;
        MDT_TRY(scratch_enter(m, s));
        MDT_TRY(reserve_scratch(m, t.B));
        ts->deferred.clear(); ts->defer_off = 0; ts->dy_off = 0;
        ts->bwd_tape = tape;
    } else if (ts->bwd_next != stage || ts->bwd_tape != tape) {
        return fail(MDT_ERR_STATE, "mdt_train_loss_bwd_stage: stage %d of tape %d out of order (expected stage %d of tape %d)", stage,
                    tape, ts->bwd_next, ts->bwd_tape);
    }
    ts->bwd_next = -1;  // (an error below leaves the run unusable)
    const int D = m->D, Ta = m->Ta, A = m->A;
    const int64_t B = t.B, Ma = B * Ta, Me = B * m->Te;
    if (stage == 0) {
        // ---- loss and action head: F = action_pred(ln(x_L))
        LAUNCH(mdt_launch_loss_grad(t.F, t.action, t.noised, t.sigma, m->cfg.sigma_data, Ma * A, Ta * A, g_loss, ts->dF, s));
        MDT_TRY(dec_bwd_head(m, t, grads, s));
        MDT_TRY(dec_bwd_block(m, t, grads, Ld - 1, s));
    } else if (stage < Ld) {
        MDT_TRY(dec_bwd_block(m, t, grads, Ld - 1 - stage, s));
    } else if (stage == Ld) {
        MDT_TRY(dec_bwd_tail(m, t, grads, s));
        // ---- context: K|V projections of all blocks, plus whatever other losses hung onto latent_encoder_emb
        if (g_ctx) HIP_TRY(hipMemcpyAsync(ts->dxe, g_ctx, (size_t)Me * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        MDT_TRY(lin_bwd(m, grads, m->kv_all, t.ctx, D, ts->d_kvx, (int64_t)Ld * 2 * D, (int)Me, ts->dxe, D, g_ctx ? 1 : 0, s));
        MDT_TRY(enc_bwd_head(m, t, grads, s));
    } else if (stage <= Ld + Le) {
        MDT_TRY(enc_bwd_block(m, t, grads, Le - 1 - (stage - Ld - 1), s));
    } else {
        MDT_TRY(enc_bwd_tail(m, t, grads, d_tokens, d_tokens2, d_goal, s));
    }
    if (sync || stage == n - 1) MDT_TRY(stage_finish(m, s, stage == n - 1));
    ts->bwd_next = stage == n - 1 ? 0 : stage + 1;
    return MDT_OK;
}

extern "C" mdt_status mdt_train_loss_bwd_stage(mdt_model* m, mdt_tape_id tape, int32_t stage, const float* g_loss, const float* g_ctx,
                                               float* grads, float* d_tokens, float* d_tokens2, float* d_goal, void* stream) {
    return loss_bwd_stage_impl(m, tape, stage, g_loss, g_ctx, grads, d_tokens, d_tokens2, d_goal, stream, true);
}

extern "C" mdt_status mdt_train_loss_bwd(mdt_model* m, mdt_tape_id tape, const float* g_loss, const float* g_ctx, float* grads,
                                         float* d_tokens, float* d_tokens2, float* d_goal, void* stream) {
    if (!m || !m->train) return fail(MDT_ERR_STATE, "training was not prepared (mdt_train_prepare)");
    const int n = m->Ld + m->Le + 2;
    for (int k = 0; k < n; ++k)
        MDT_TRY(loss_bwd_stage_impl(m, tape, k, g_loss, g_ctx, grads, d_tokens, d_tokens2, d_goal, stream, false));
    return MDT_OK;
}

// The stage of mdt_train_loss_bwd_stage behind which parameter slot i's gradient is complete (by its reference name).
extern "C" int32_t mdt_train_param_stage(const mdt_model* m, int64_t i) {
    if (!m || i < 0 || i >= (int64_t)m->slots.size()) return -1;
    const std::string& nm = m->slots[i].name;
    const int Ld = m->Ld, Le = m->Le, last = Ld + Le + 1;
    auto block_of = [&](const char* key) -> int {
        const size_t p = nm.find(key);
        return p == std::string::npos ? -1 : atoi(nm.c_str() + p + strlen(key));
    };
    const int dl = block_of("decoder.blocks."), el = block_of("encoder.blocks.");
    if (dl >= 0 && dl < Ld) {
        // the stacked Linears of all blocks (adaLN modulation, cross-attention K | V) get their gradient in ONE product at stage Ld
        if (nm.find("adaLN_zero") != std::string::npos || nm.find("cross_att.key") != std::string::npos ||
            nm.find("cross_att.value") != std::string::npos)
            return Ld;
        return Ld - 1 - dl;
    }
    if (el >= 0 && el < Le) return Ld + 1 + (Le - 1 - el);
    if (nm.find("action_pred") != std::string::npos || nm.find("decoder.ln") != std::string::npos) return 0;
    if (nm.find("action_emb") != std::string::npos || nm.find("encoder.ln") != std::string::npos) return Ld;
    if (nm.find("sigma_emb") != std::string::npos) return m->cond == COND_TOKEN ? last : Ld;  // sigma token: with the embeddings
    return last;  // tok_emb, goal_emb / lang_emb, pos_emb, incam_embed, proprio_emb
}

// ------------------------------------------------------------------------------------------------
// vector-Jacobian product of the denoiser w.r.t. its noisy-action input (what log_likelihood differentiates,
// gc_sampling.py:469-487: torch.autograd.grad((d * v).sum(), action) with d = (action - D(action; sigma)) / sigma)
// ------------------------------------------------------------------------------------------------
extern "C" mdt_status mdt_denoise_vjp(mdt_model* m, const float* tokens, const float* tokens2, const float* goal,
                                      int32_t modality, const float* x, const float* sigma, const float* v, int64_t batch,
                                      float* denoised, float* vjp, void* stream) {
    MDT_TRY(check_ready(m));
    if (!tokens || !goal || !x || !sigma || !v || !denoised || !vjp || batch < 1)
        return fail(MDT_ERR_INVALID_ARG, "mdt_denoise_vjp: bad argument");
    if (m->cfg.arch == MDT_ARCH_MDT && !tokens2) return fail(MDT_ERR_INVALID_ARG, "MDT needs the gripper tokens");
    hipStream_t s = (hipStream_t)stream;
    mdt_tape_id id;
    MDT_TRY(acquire_tape(m, batch, &id, s));
    Tape& t = m->train->tapes[id];
    t.drop = effective_dropout(nullptr);  // eval-mode forward: D(x; sigma) itself
    const int honour = m->cfg.arch == MDT_ARCH_MDTV;
    mdt_status st = scratch_enter(m, s);
    if (st == MDT_OK) st = reserve_scratch(m, batch);
    if (st == MDT_OK) m->train->dy_off = 0;
    if (st == MDT_OK) st = enc_fwd(m, t, tokens, tokens2, goal, modality, honour, sigma, nullptr, s);
    // action := x, no noise: the tape's "noised" rows are x itself, F the raw network output
    if (st == MDT_OK) st = dec_fwd(m, t, x, nullptr, sigma, nullptr, nullptr, s);
    if (st == MDT_OK) {
        mdt_train_state* ts = m->train;
        const int64_t n = batch * m->Ta * m->A;
        const int per = m->Ta * m->A;
        // D = c_skip x + c_out F;  dF = c_out v
        hipError_t e = mdt_launch_vjp_seed(t.F, t.noised, t.sigma, v, m->cfg.sigma_data, n, per, denoised, ts->dF, s);
        if (e != hipSuccess) st = fail(MDT_ERR_HIP, "vjp seed launch failed: %s", hipGetErrorString(e));
        if (st == MDT_OK) st = dec_bwd(m, t, nullptr, s);
        if (st == MDT_OK) {
            // y0 = action_emb(c_in x): d x = c_in (d y0 . Wa) + c_skip v
            e = mdt_launch_narrow_out(ts->dx, m->D, m->Wa, ts->small, (int)(batch * m->Ta), m->A, m->D, s);
            if (e == hipSuccess) e = mdt_launch_vjp_finish(ts->small, t.sigma, v, m->cfg.sigma_data, n, per, vjp, s);
            if (e != hipSuccess) st = fail(MDT_ERR_HIP, "vjp finish launch failed: %s", hipGetErrorString(e));
        }
    }
    (void)mdt_tape_release(m, id);
    if (st == MDT_OK) st = scratch_leave(m, s);
    return st;
}
