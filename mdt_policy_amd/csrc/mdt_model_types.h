// mdt_model_types.h -- the denoiser handle's data structures, shared by mdt_model.hip (inference launch sequences)
// and mdt_train.hip (training forward / backward).
#pragma once
#include <string>
#include <vector>

#include "mdt_internal.h"

// ------------------------------------------------------------------------------------------------
// model description
// ------------------------------------------------------------------------------------------------
// TRANSPOSE: (rows, K) -> (K, rows);  PAD_COLS: (rows, K) -> (rows, n_off) row-major, columns K.. stay zero
enum SlotKind { SLOT_PACK = 0, SLOT_RAW = 1, SLOT_TRANSPOSE = 2, SLOT_PAD_COLS = 3, SLOT_PACK_T = 4,  // PACK_T: fragment image of the TRANSPOSE of a (rows, K) matrix
                SLOT_PACK_SPLIT = 5 };  // three-way bf16 split fragment image of a (rows, K) matrix (mdt_mlp_split.h): 6 rows K bytes at dst

struct Slot {
    std::string name;
    int64_t numel = 0;
    int kind = SLOT_RAW;
    float* dst = nullptr;  // packed image base (SLOT_PACK) or raw destination (SLOT_RAW)
    int rows = 0, K = 0, n_off = 0;
    bool loaded = false;
    Lin* lin = nullptr;    // SLOT_PACK: the Linear this part belongs to (training keeps a transposed image too)
};

// one reference nn.Linear inside a (possibly stacked) Lin: rows [n_off, n_off + rows) of its packed image
struct LinPart {
    Lin* lin = nullptr;
    int w_slot = -1, b_slot = -1;  // indices into mdt_model::slots
    int rows = 0, n_off = 0;
};

struct mdt_train_state;  // mdt_train.hip

struct EncBlock {
    float *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr, *ln2_b = nullptr;
    Lin qkv, proj, fc, proj2;
};

struct DecBlock : EncBlock {
    float *ln3_w = nullptr, *ln3_b = nullptr;
    Lin xq, xproj;
    float *xq_pT = nullptr;   // fragment image of cross_att.query.weight TRANSPOSED (collapsed cross-attention fold; its other
                              // weight operand is xproj.wp, the forward image of cross_att.c_proj.weight)
};

// how sigma conditions the decoder (reference constructor flags use_ada_conditioning / use_noise_encoder)
enum CondMode {
    COND_ADALN = 0,  // ConditionedBlock + AdaLNZero           (transformer_blocks.py:264-309) -- the shipped configs
    COND_NOISE = 1,  // NoiseBlock: ln(x) + c before both attentions (transformer_blocks.py:312-341)
    COND_TOKEN = 2,  // plain Block decoder; sigma embedding is the FIRST encoder token (mdtv_transformer.py:296-297)
};

struct mdt_model {
    mdt_config cfg;
    int D, H, hd, Te, Ta, A, Le, Ld, G, O, n_tok;
    int cond = COND_ADALN;
    int sig_tok = 0;  // 1 when the context starts with the sigma token (COND_TOKEN): Te = sig_tok + 1 + n_tok
    // context rows: [sigma token] [goal] state tokens  (goal_conditioned, the default);
    // goal_conditioned=False: MDTV [sigma] state tokens, goal ; MDT [sigma] state tokens (g_row = -1: no goal token)
    int g_row = 0, tok_row = 1;
    // parameters
    float* arena = nullptr;
    size_t arena_floats = 0;
    std::vector<Slot> slots;   // one per parameter the path reads (enumerated by mdt_param_*)
    std::vector<Slot> extra;   // additional images of an already listed parameter (same name)
    std::vector<LinPart> parts;
    // the split images (Lin.ws) are NOT rewritten by parameter loads while a training state exists (an optimizer step re-packs every
    // fp32 image; the split ones are only read by large-batch sampling): marked stale instead, re-made from the fp32 images by the next
    // model-level call that would read them (refresh_split, mdt_model.hip)
    bool split_stale = false;
    mdt_train_state* train = nullptr;  // non-null after mdt_train_prepare()
    Lin tok, incam, goal0, goal2, lang0, lang2, sig1, sig3, kv_all, mod_all;
    // proprioceptive token (cfg.use_proprio, MDT-V): proprio_emb = Linear(Pd, 2D) -> Mish -> Linear(2D, D)
    // (mdtv_transformer.py:160-164, 260-266).  The first layer is Pd <= 16 wide: held transposed (Pd, 2D) and applied
    // by k_narrow_linear; the second is an ordinary packed Linear.  p_row: the token's context row (the last), or -1.
    Lin prop2;
    float *prop0_T = nullptr, *prop0_b = nullptr;
    int Pd = 0, p_row = -1;
    // MLP action head (linear_output = 0): action_pred.0 = Linear(d, HH) padded to HP = ceil16(HH) rows, action_pred.2
    // = Linear(HH, A) held as (A, HP) with zero pad columns.  HH = 100 in MDTVTransformer (mdtv_transformer.py:182),
    // embed_dim in MDTTransformer (mdt_transformer.py:174).  HP = 0 with the plain Linear head.
    Lin head0;
    int HH = 0, HP = 0;
    std::vector<EncBlock> enc;
    std::vector<DecBlock> dec;
    float *enc_ln_w = nullptr, *enc_ln_b = nullptr, *dec_ln_w = nullptr, *dec_ln_b = nullptr;
    float *Wa = nullptr, *ba = nullptr, *Wp = nullptr, *bp = nullptr, *pos_emb = nullptr;
    float *freqs = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;  // constant tables
    float* staging = nullptr;  // host->device parameter staging
    size_t staging_floats = 0;
    // batched upload (mdt_load_params): move table + block list on the device, two pinned staging copies used in turn
    // (ev_tab[i] marks the last asynchronous copy out of host buffer i), the bytes last uploaded (an optimizer keeps its
    // parameters' storage, so step after step the table is identical and is not copied again)
    void* tab_dev = nullptr;
    void* tab_host[2] = {nullptr, nullptr};
    hipEvent_t ev_tab[2] = {nullptr, nullptr};
    size_t tab_cap = 0;
    int tab_turn = 0;
    hipEvent_t ev_tab_use = nullptr;   // behind the last k_multi_load that read tab_dev ...
    hipStream_t tab_stream = nullptr;  // ... on this stream; a caller on another stream waits for it first
    bool tab_used = false;
    std::vector<char> tab_last;
    // workspace
    float* ws = nullptr;
    int64_t cap = 0;
    int64_t ws_generation = 0;  // bumped whenever the workspace is (re)allocated: captured HIP graphs hold its addresses
    float *h_enc, *qkv, *att, *hid, *ctx, *kvx, *y, *qx, *sig_e, *sig_t, *sig_c, *mod, *xbuf, *noised, *Fbuf, *steps, *sigs, *loss_part;
    float* cmod = nullptr;  // COND_NOISE: rows of [c | ones(D)], read as (shift, scale) by the LayerNorm prologue
    int64_t cached_batch = 0;  // batch of the context currently cached by mdt_encode (0 = none)
    // sampler pipelining: the batch is cut into `ways` sample-aligned slices whose launch chains run on separate
    // HIP streams, so one slice's prologue / epilogue / launch gaps overlap another slice's MFMA main loops
    int ways = 1;
    // collapsed cross-attention (k_xattn_fold / k_xattn_apply): folded projections per sample and decoder block
    bool xfold = false;
    float *xU = nullptr, *xW = nullptr, *xc = nullptr;  // [Ld][cap][4 H * D] weight images (fragment order), same, [Ld][cap][4 H]
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
};

