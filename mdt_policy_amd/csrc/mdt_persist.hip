// mdt_persist.hip -- the PERSISTENT decoder kernel: all n_steps x (Ld decoder blocks + action head) of one
// mdt_sample_ddim call (gc_sampling.py:922-951 around mdtv_transformer.py:224-236) in ONE launch.
//
// Why: a 10-step call is 250 dependent launches; at B = 256 every GEMM is one tile per CU, so a
// launch costs a dispatch ramp, a cold L2 (the kernel-start invalidate drops the XCD's copy of activations AND weights)
// and an end-of-kernel write-back per operation; at B = 1 each of them is a 5-10 us kernel around < 1 us of work.
//
// How:
//   * grid = one 512-thread workgroup per CU (the LDS request admits exactly one).  A workgroup reads the XCC it
//     actually runs on (s_getreg HW_REG_XCC_ID -- not blockIdx % 8, that mapping is a speed-only observation) and takes a
//     slot 0..npx-1 on that XCD from an atomic counter.  Placement that is not exactly npx workgroups per XCD, or a
//     barrier that does not complete within a bounded number of polls, sets an error word and every workgroup leaves
//     (the host then reports the call as failed and stops using the kernel) -- the kernel cannot hang.
//   * XCD x owns samples [x S, x S + S): no activation ever crosses an XCD.  The phases of a block (qkv GEMM, self
//     attention, projection, collapsed cross attention, fc GEMM, c_proj GEMM) are the tile bodies of mdt_tiles.h with
//     COH = true; between phases the XCD's workgroups meet at a per-XCD counter: `s_waitcnt vmcnt(0)` (this thread's
//     stores have reached the XCD's L2) -> workgroup barrier -> one relaxed agent-scope atomic add -> relaxed polls.
//     No cache-wide fence: producers store plain (the lines stay in the XCD's L2), consumers load `sc1` (L1 bypass).
//     Measured 1.2 us per phase (profiles/r01_xcd_barrier_probe.txt); 0 stale words (profiles/r02_persist_probe.txt).
//   * workgroups without a tile in a phase pull the NEXT phase's weights (constants) into the XCD's L2, so that the
//     weight ring of the next phase starts from L2 hits instead of Infinity-Cache misses.
//   * two geometries: WIDE (>= 16 samples per XCD; the 32 x 512 / 384 / 128 MFMA tiles of k_gemm) and SMALL (one sample
//     per XCD, B <= 8: the split-K 16-column tiles of k_gemm_smallm, self-attention fused into its projection).
//
// BUILD: opt-in since round 4 (`MDT_BUILD_PERSIST=1 python -m mdt_policy_amd.build`, i.e. -DMDT_WITH_PERSIST).  The kernel has
// never beaten the launch sequence (6.74 vs 5.73 ms at B = 256, 3.32 vs 1.65 ms at B = 1 when it was measured; the launch
// sequence has since dropped to 4.68 / 1.31 ms), its small-batch geometry walks kernels the default dispatcher no longer
// picks for every product, and it was a tenth of the library's build time and size: the shipped library carries the entry
// points below as stubs (never supported, status 0, `mdt_persist_built() == 0`) and tests/test_gpu_persist.py skips the
// comparisons that need the real thing.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "mdt_model_types.h"
#include "mdt_persist.h"

#ifndef MDT_WITH_PERSIST
// ---- the shipped build: the switch exists, the kernel does not ----
extern "C" void mdt_op_set_persist(int32_t) {}
extern "C" int32_t mdt_persist_built(void) { return 0; }
extern "C" int32_t mdt_persist_status(mdt_model*) { return 0; }
extern "C" int64_t mdt_persist_launches(mdt_model*) { return 0; }
bool mdt_persist_supported(mdt_model*, int64_t) { return false; }
mdt_status mdt_persist_sample(mdt_model*, int64_t, int, const float*, float*, hipStream_t) {
    return mdt_fail(MDT_ERR_INVALID_ARG, "the persistent decoder kernel is not part of this build (MDT_BUILD_PERSIST=1)");
}
void mdt_persist_free(mdt_model*) {}
#else
extern "C" int32_t mdt_persist_built(void) { return 1; }
#include "mdt_tiles.h"

enum {
    PH_GEMM_QKV = 0,     // WIDE  <2,3,8, LN+mod(bcast)>            N = 3D
    PH_GEMM_FC = 1,      // WIDE  <2,4,8, LN+mod(bcast)>  GELU      N = 4D
    PH_GEMM_PROJ = 2,    // WIDE  <2,1,8, plain, residual>          N = D (K = D or 4D in LDS chunks)
    PH_GEMM_SMALL = 3,   // SMALL split-K 16-column tiles
    PH_ATTN = 4,         // WIDE  one sample per workgroup, two head groups
    PH_ATTN_PROJ = 5,    // SMALL attention fused into the projection
    PH_XATTN = 6,        // collapsed cross attention, one sample per workgroup
    PH_HEAD = 7,         // decoder LN + action_pred + EDM combine + DDIM update + next step's embedding
};

struct mdt_pphase {
    int32_t kind, kchunk, grid_n, ncompute;
    const float* pf[2];        // regions worth having in L2 when the NEXT phase starts (nullptr: none)
    int64_t pf_stride[2];      // floats per sample (per-sample operands of the XCD's own samples) or 0 (weights)
    int64_t pf_floats[2];      // floats (per sample when pf_stride != 0)
    const float* qkv;          // PH_ATTN_PROJ
    int64_t ldq;
    int32_t T, causal, hd, np;
    float scale;
    int32_t pad_;
    union U {
        mdt_gemm_args g;
        mdt_attn_args at;
        mdt_xapply_args xa;
        mdt_head_args h;
    } u;
};

struct mdt_pctl {
    unsigned arrive[8 * 32];  // per-XCD barrier counters, 128 bytes apart
    unsigned slots[8 * 32];   // per-XCD slot allocators
    unsigned err[32];         // [0] != 0: somebody gave up
};

static constexpr int LDS_BODY_FLOATS = 25344;           // 99 KiB: 32 x 772 floats (the chunked-K projection tile)
static constexpr int LDS_RED_FLOATS = 8 * 64 * 4 + 32;  // split-K partial tiles + row statistics
static constexpr size_t PERSIST_LDS_BYTES = (size_t)(LDS_BODY_FLOATS + LDS_RED_FLOATS) * sizeof(float);
static constexpr int SPIN_LIMIT = 600000;               // polls (~64 cycles of sleep + an L2 round trip each): ~0.2 s

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// XCD-local barrier.  Returns true when the kernel has to be abandoned.
__device__ __forceinline__ bool xcd_barrier(unsigned* ctr, unsigned target, unsigned* err, unsigned* err_host, int* s_bail,
                                            int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores sit in the XCD's L2 now
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (ld_relaxed(ctr) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                *s_bail = 1;
                break;
            }
            if ((spins & 1023) == 0 && ld_relaxed(err)) { *s_bail = 1; break; }
        }
    }
    __syncthreads();
    return *s_bail != 0;
}

// touch one dword of every 128-byte line of [p, p + floats): the lines land in the XCD's L2.  The destination register
// is tied through every load and released only behind the final s_waitcnt, so the compiler cannot hand it to another
// value while a load is still in flight.
__device__ __forceinline__ void l2_touch(const float* p, int64_t floats, int part, int nparts, int tid) {
    if (p == nullptr || nparts <= 0) return;
    const int64_t lines = (floats + 31) >> 5;
    float sink = 0.f;
    for (int64_t l = (int64_t)part * 512 + tid; l < lines; l += (int64_t)nparts * 512)
        asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p + (l << 5)) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
}

// ---- phase bodies ------------------------------------------------------------------------------------------------
// One function per phase kind, inlined into the kernel (as real functions every call would save / restore ~110
// callee-saved VGPRs around a 250-register body).  The kernel is instantiated per (head dim, key bound, H*Te) so that it
// holds ONE variant of each body: with all variants inlined hipcc spills several hundred registers.
#define PH_INLINE __forceinline__
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
template <class T>
__device__ __forceinline__ T* uni_ptr(T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = UNI((uint32_t)v), hi = UNI((uint32_t)(v >> 32));
    return (T*)(((uint64_t)hi << 32) | lo);
}
// A pointer that arrives as a function argument or is read out of a descriptor in memory is a GENERIC pointer to the
// compiler: it emits flat_load (which also counts on lgkmcnt) and treats every value loaded through it as divergent (the
// "uniform" descriptor fields then occupy VGPRs and the bodies spill).  glob() re-types such a pointer as global memory
// and pins it in an SGPR pair (the empty asm is opaque, so the cast pair is not folded back to the generic pointer);
// cglob() does the same for read-only data (constant address space: always scalar loads).  Wave-uniform pointers only.
template <class T>
__device__ __forceinline__ T* glob(T* p) {
    __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)uni_ptr(p);
    asm volatile("" : "+s"(g));
    return (T*)g;
}
template <class T>
__device__ __forceinline__ const T* cglob(const T* p) {
    __attribute__((address_space(4))) const T* g = (__attribute__((address_space(4))) const T*)uni_ptr(p);
    asm volatile("" : "+s"(g));
    return (const T*)g;
}
__device__ __forceinline__ void glob_args(mdt_gemm_args& a) {
    a.A = glob(a.A); a.Wp = glob(a.Wp); a.bias = glob(a.bias); a.out = glob(a.out); a.ln_w = glob(a.ln_w); a.ln_b = glob(a.ln_b);
    a.mod = glob(a.mod); a.rowvec = glob(a.rowvec);
}
__device__ __forceinline__ float* lds_at(unsigned byte_off) {
    return (float*)(__attribute__((address_space(3))) float*)(uintptr_t)byte_off;
}

struct PhaseCtx {  // wave-uniform values of the running workgroup
    const mdt_pphase* P;
    const float* zeros;
    int r0, nrows, s0, ns, slot, npx;
    unsigned lds_off;
};
#define PHASE_ARGS const mdt_pphase* P_, const float* zeros_, int r0_, int nrows_, int s0_, int ns_, int slot_, int npx_, unsigned lds_off_
#define PHASE_UNIFORM                                                                                   \
    const mdt_pphase& P = *cglob(P_);                                                                   \
    const float* zeros = glob(zeros_);                                                                      \
    const int r0 = UNI(r0_), nrows = UNI(nrows_), s0 = UNI(s0_), ns = UNI(ns_), slot = UNI(slot_), npx = UNI(npx_); \
    float* lds = lds_at(UNI(lds_off_));                                                                 \
    const int tid = threadIdx.x;                                                                        \
    (void)r0; (void)nrows; (void)s0; (void)ns; (void)slot; (void)npx; (void)lds; (void)zeros;

// this XCD's rows of a GEMM: A, out advanced to row r0, M = nrows
__device__ __forceinline__ mdt_gemm_args xcd_rows(const mdt_gemm_args& g, int r0, int nrows) {
    mdt_gemm_args a = g;
    glob_args(a);
    a.A += (int64_t)r0 * a.lda;
    a.out += (int64_t)r0 * a.ldo;
    a.M = nrows;
    return a;
}

template <int NTW, int PRO, bool RES>
__device__ PH_INLINE void ph_gemm_wide(PHASE_ARGS) {
    PHASE_UNIFORM
    const mdt_gemm_args a = xcd_rows(P.u.g, r0, nrows);
    const int gn = P.grid_n, nt = ((nrows + 31) >> 5) * gn, kchunk = P.kchunk;
    for (int t = slot; t < nt; t += npx) {
        if (t != slot) __syncthreads();
        const int by = t / gn, bx = t - by * gn;
        gemm_tile<2, NTW, 8, PRO, RES, true>(a, kchunk, by, bx, lds, zeros, tid);
    }
}

__device__ PH_INLINE void ph_gemm_small(PHASE_ARGS) {
    PHASE_UNIFORM
    const mdt_gemm_args a = xcd_rows(P.u.g, r0, nrows);
    const int ncomp = min(npx, P.ncompute), nt = a.N >> 4;
    float* red = lds + LDS_BODY_FLOATS;
    float* s_stat = red + 8 * 64 * 4;
    if (slot < ncomp)
        for (int t = slot; t < nt; t += ncomp) {
            if (t != slot) __syncthreads();
            gemm_smallm_tile<true>(a, t, 0, s_stat, red, zeros, tid);
        }
}

template <int HD>
__device__ PH_INLINE void ph_attn_proj(PHASE_ARGS) {
    PHASE_UNIFORM
    mdt_gemm_args a = P.u.g;
    glob_args(a);
    const float* qkv = glob(P.qkv);
    const int ncomp = min(npx, P.ncompute), nt = a.N >> 4;
    float* red = lds + LDS_BODY_FLOATS;
    if (slot < ncomp)
        for (int t = slot; t < nt; t += ncomp) {
            if (t != slot) __syncthreads();
            attn_proj_tile<HD, true>(a, qkv, P.ldq, P.T, P.causal, P.scale, t, s0, lds, red, zeros, tid);
        }
}

template <int HD, int TKC>
__device__ PH_INLINE void ph_attn(PHASE_ARGS) {
    PHASE_UNIFORM
    mdt_attn_args a = P.u.at;
    a.q = glob(a.q); a.k = glob(a.k); a.v = glob(a.v); a.out = glob(a.out);
    const float scale = P.scale;
    for (int t = slot; t < ns; t += npx) {
        if (t != slot) __syncthreads();
        attn_tile<HD, TKC, false, true>(a, nullptr, nullptr, scale, s0 + t, tid >> 8, 2, lds + (tid >> 8) * (LDS_BODY_FLOATS / 2),
                                        tid & 255);
    }
}

template <int NP>
__device__ PH_INLINE void ph_xattn(PHASE_ARGS) {
    PHASE_UNIFORM
    mdt_xapply_args a = P.u.xa;
    a.y = glob(a.y); a.ln_w = glob(a.ln_w); a.ln_b = glob(a.ln_b); a.U = glob(a.U); a.Wf = glob(a.Wf); a.c = glob(a.c); a.bo = glob(a.bo);
    for (int t = slot; t < ns; t += npx) {
        if (t != slot) __syncthreads();
        xattn_tile<NP, true>(a, s0 + t, lds, zeros, tid);
    }
}

template <int AMAX>
__device__ PH_INLINE void ph_head(PHASE_ARGS) {
    PHASE_UNIFORM
    mdt_head_args h = P.u.h;
    h.y = glob(h.y); h.ln_w = glob(h.ln_w); h.ln_b = glob(h.ln_b); h.Wp = glob(h.Wp); h.bp = glob(h.bp); h.x = glob(h.x);
    h.sigma = glob(h.sigma); h.out = glob(h.out); h.step = glob(h.step); h.y_next = glob(h.y_next); h.Wa = glob(h.Wa); h.ba = glob(h.ba);
    h.M = r0 + nrows;  // rows past this XCD's range belong to somebody else
    const int nt = (nrows + 15) >> 4, lane = tid & 63, wave = tid >> 6;
    for (int t = slot; t < nt; t += npx) {
        const int base = r0 + t * 16 + wave * 2;
        if (base < h.M) head_rows<AMAX, true>(h, base, lane, zeros);
    }
}

__device__ PH_INLINE void ph_prefetch(PHASE_ARGS, int nbusy_) {
    PHASE_UNIFORM
    const int nbusy = UNI(nbusy_);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* p = glob(P.pf[i]);
        if (p != nullptr) {
            const int64_t st = P.pf_stride[i];
            l2_touch(p + (int64_t)s0 * st, st ? (int64_t)ns * P.pf_floats[i] : P.pf_floats[i], slot - nbusy, npx - nbusy, tid);
        }
    }
}

#define PHASE_CALL P, zeros, r0, nrows, s0, ns, slot, npx, lds_off

template <int MODE, int HD, int TKC, int NP>  // MODE 0 = WIDE, 1 = SMALL; head dim, key bound (10 / 16), 4 H
__global__ __launch_bounds__(512) void k_decoder_persist(const mdt_pphase* __restrict__ prog, int nphases, mdt_pctl* ctl, int S,
                                                        int B, int Ta, const float* __restrict__ zeros, unsigned* err_host,
                                                        unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ int s_info[4];
    const int tid = threadIdx.x;
    const int npx = gridDim.x >> 3;  // workgroups per XCD
    if (tid == 0) {
        int x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 7;
        const unsigned slot = __hip_atomic_fetch_add(&ctl->slots[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_info[0] = x;
        s_info[1] = (int)slot;
        s_info[2] = 0;  // bail flag
        s_info[3] = slot >= (unsigned)npx;
        if (slot >= (unsigned)npx) {  // uneven placement: this launch cannot use per-XCD ownership
            __hip_atomic_store(&ctl->err[0], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(err_host, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    const int x = UNI(s_info[0]), slot = UNI(s_info[1]);
    if (s_info[3]) return;
    const int s0 = x * S, ns = min(B, s0 + S) - s0;  // this XCD's samples
    if (ns <= 0) return;
    const int r0 = s0 * Ta;
    const int nrows = ns * Ta;
    unsigned* ctr = &ctl->arrive[x * 32];
    const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds_dyn;

    for (int ph = 0; ph < nphases; ++ph) {
        const mdt_pphase* P = prog + ph;
        const int kind = P->kind;
        int nbusy = npx;  // workgroups that hold a tile in this phase
        // tuning hook (mdt_persist_set_debug): shader-clock stamps of [phase start, body end] per workgroup and phase
        if (dbg != nullptr && tid == 0) dbg[((size_t)blockIdx.x * nphases + ph) * 2] = __builtin_readcyclecounter();
        if (MODE == 0) {
            switch (kind) {
                case PH_GEMM_QKV:
                    nbusy = min(npx, ((nrows + 31) >> 5) * P->grid_n);
                    ph_gemm_wide<3, PRO_LN_MOD_BCAST, false>(PHASE_CALL);
                    break;
                case PH_GEMM_FC:
                    nbusy = min(npx, ((nrows + 31) >> 5) * P->grid_n);
                    ph_gemm_wide<4, PRO_LN_MOD_BCAST, false>(PHASE_CALL);
                    break;
                case PH_GEMM_PROJ:
                    nbusy = min(npx, ((nrows + 31) >> 5) * P->grid_n);
                    ph_gemm_wide<1, PRO_PLAIN, true>(PHASE_CALL);
                    break;
                case PH_ATTN:
                    nbusy = min(npx, ns);
                    ph_attn<HD, TKC>(PHASE_CALL);
                    break;
                default: break;
            }
        } else {
            switch (kind) {
                case PH_GEMM_SMALL:
                    nbusy = min(min(npx, P->ncompute), P->u.g.N >> 4);
                    ph_gemm_small(PHASE_CALL);
                    break;
                case PH_ATTN_PROJ:
                    nbusy = min(min(npx, P->ncompute), P->u.g.N >> 4);
                    ph_attn_proj<HD>(PHASE_CALL);
                    break;
                default: break;
            }
        }
        if (kind == PH_XATTN) {
            nbusy = min(npx, ns);
            ph_xattn<NP>(PHASE_CALL);
        } else if (kind == PH_HEAD) {
            nbusy = min(npx, (nrows + 15) >> 4);
            ph_head<8>(PHASE_CALL);
        }
        if (slot >= nbusy) ph_prefetch(PHASE_CALL, nbusy);  // no tile in this phase: warm the XCD's L2 for the next one
        if (dbg != nullptr && tid == 0)
            dbg[((size_t)blockIdx.x * nphases + ph) * 2 + 1] = __builtin_readcyclecounter() | ((unsigned long long)(slot < nbusy) << 63);
        if (xcd_barrier(ctr, (unsigned)npx * (unsigned)(ph + 1), &ctl->err[0], err_host, &s_info[2], tid)) return;
    }
    if (dbg != nullptr && tid == 0) dbg[(size_t)gridDim.x * nphases * 2 + blockIdx.x] = ((unsigned long long)x << 32) | (unsigned)slot;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
typedef void (*persist_kernel_t)(const mdt_pphase*, int, mdt_pctl*, int, int, int, const float*, unsigned*, unsigned long long*);
struct PersistVariant { int mode, hd, tkc, np; persist_kernel_t fn; };
// the instantiated configurations: MDT-V default (d = 384: 8 heads of 48, Te = 4), MDT default (d = 512: heads of 64,
// Te = 3) and the two d = 128 test-size models; Ta <= 10.  Anything else keeps the launch sequence.
#define PV(MODE, HD, TKC, NP) {MODE, HD, TKC, NP, k_decoder_persist<MODE, HD, TKC, NP>}
static const PersistVariant g_variants[] = {
    PV(0, 48, 10, 32), PV(1, 48, 10, 32), PV(0, 64, 10, 32), PV(1, 64, 10, 32),
    PV(0, 16, 10, 32), PV(1, 16, 10, 32),
};
static persist_kernel_t find_variant(int mode, int hd, int Ta, int np) {
    const int tkc = Ta <= 10 ? 10 : 16;
    for (const PersistVariant& v : g_variants)
        if (v.mode == mode && v.hd == hd && v.tkc == tkc && v.np == np) return v.fn;
    return nullptr;
}

struct mdt_persist_state {
    int device = 0;
    int n_cu = 0;
    bool usable = false;
    bool failed = false;       // a launch reported an error: never used again on this handle
    persist_kernel_t fn[2] = {nullptr, nullptr};  // WIDE, SMALL kernels of this handle's configuration
    mdt_pctl* ctl = nullptr;
    unsigned* err_host = nullptr;      // pinned, device-mapped
    unsigned* err_host_dev = nullptr;  // its device alias
    mdt_pphase* prog_dev = nullptr;
    size_t prog_cap = 0;               // phases
    mdt_pphase* stage[2] = {nullptr, nullptr};  // pinned staging, alternated
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int stage_next = 0;
    // cache key of the program currently in prog_dev
    const void* k_ws = nullptr;
    const void* k_x = nullptr;
    const void* k_out = nullptr;
    int64_t k_B = 0;
    int k_steps = 0, k_mode = -1, k_nph = 0;
    int64_t launches = 0;
    unsigned long long* dbg = nullptr;  // tuning hook: per-phase stamps of the next launches (caller-owned device buffer)
};

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int g_persist_override = -1;  // mdt_op_set_persist: -1 = environment (MDT_HIP_PERSIST, default off), 0 = off, 1 = on
static int persist_enabled() {
    if (g_persist_override >= 0) return g_persist_override;
    static int v = -1;
    if (v < 0) v = env_int("MDT_HIP_PERSIST", 0);  // measured slower than the launch sequence (profiles/r02_persist_phases.txt): opt-in
    return v;
}
extern "C" void mdt_op_set_persist(int32_t mode) { g_persist_override = mode < 0 ? -1 : (mode ? 1 : 0); }
static int persist_wide_min() {
    static int v = -1;
    if (v < 0) v = env_int("MDT_HIP_PERSIST_WIDE_MIN", 128);
    return v;
}
static int persist_small_max() {
    static int v = -1;
    if (v < 0) v = env_int("MDT_HIP_PERSIST_SMALL_MAX", 8);
    return v;
}

void mdt_persist_free(mdt_model* m) {
    mdt_persist_state* p = (mdt_persist_state*)m->persist;
    if (!p) return;
    if (p->ctl) (void)hipFree(p->ctl);
    if (p->prog_dev) (void)hipFree(p->prog_dev);
    if (p->err_host) (void)hipHostFree(p->err_host);
    for (int i = 0; i < 2; ++i) {
        if (p->stage[i]) (void)hipHostFree(p->stage[i]);
        if (p->stage_ev[i]) (void)hipEventDestroy(p->stage_ev[i]);
    }
    delete p;
    m->persist = nullptr;
}

static mdt_status persist_init(mdt_model* m) {
    if (m->persist) return MDT_OK;
    mdt_persist_state* p = new mdt_persist_state();
    m->persist = p;
    HIP_TRY(hipGetDevice(&p->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, p->device));
    p->n_cu = prop.multiProcessorCount;
    // one workgroup per CU, the same number on each of the 8 XCDs, at most 32 per XCD (counter spacing)
    if (p->n_cu % 8 != 0 || p->n_cu / 8 > 32 || p->n_cu / 8 < 8) return MDT_OK;  // usable stays false
    for (int mode = 0; mode < 2; ++mode) {
        persist_kernel_t fn = find_variant(mode, m->hd, m->Ta, 4 * m->H);
        if (!fn) return MDT_OK;
        HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PERSIST_LDS_BYTES));
        int nb = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)fn, 512, PERSIST_LDS_BYTES));
        if (nb < 1) return MDT_OK;
        p->fn[mode] = fn;
    }
    HIP_TRY(hipMalloc((void**)&p->ctl, sizeof(mdt_pctl)));
    HIP_TRY(hipHostMalloc((void**)&p->err_host, 64, hipHostMallocMapped));
    p->err_host[0] = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&p->err_host_dev, p->err_host, 0));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&p->stage_ev[i], hipEventDisableTiming));
    p->usable = true;
    return MDT_OK;
}

// Can the step loop of this call run as the persistent kernel?  (Otherwise: the launch sequence of mdt_model.hip.)
bool mdt_persist_supported(mdt_model* m, int64_t B) {
    if (!persist_enabled()) return false;
    if (m->cond != COND_ADALN || !m->xfold || m->cfg.use_rot_embed || m->HP != 0 || m->ways > 1) return false;
    if (m->H != 8 && B <= persist_small_max()) return false;  // attn_proj_tile: 8 waves = 8 heads
    if (m->H % 2) return false;
    if (m->hd != 16 && m->hd != 32 && m->hd != 48 && m->hd != 64) return false;
    if (m->D > 512 || m->D % 32 || m->Ta > 16 || m->A > 8) return false;
    if (m->H != 8 || !find_variant(0, m->hd, m->Ta, 4 * m->H)) return false;  // the collapsed cross-attention's image: 4 H rows
    if (!mdt_xattn_apply_supported(m->D, m->H, m->Te, m->Ta)) return false;
    const bool small = B <= persist_small_max(), wide = B >= persist_wide_min();
    if (!small && !wide) return false;
    // LDS budget of the bodies
    const int lp = m->hd == 48 ? 3 : (m->hd >= 32 ? 2 : 1);
    const int64_t attn_half = (int64_t)3 * m->Ta * (m->H / 2) * m->hd + (int64_t)(m->H / 2) * m->Ta * 16 * lp;
    if (attn_half > LDS_BODY_FLOATS / 2) return false;
    if ((int64_t)8 * (3 * 16 * (m->hd + 4) + 16 * 17) > LDS_BODY_FLOATS) return false;
    if ((int64_t)mdt_xattn_lds_floats(m->D, m->H) > LDS_BODY_FLOATS) return false;
    if ((int64_t)32 * (m->D + 4) > LDS_BODY_FLOATS) return false;
    // buffer-resource offsets are 32-bit byte offsets
    if ((int64_t)B * m->Ta * 4 * m->D * 4 >= ((int64_t)1 << 32)) return false;
    if (persist_init(m) != MDT_OK) return false;
    mdt_persist_state* p = (mdt_persist_state*)m->persist;
    return p->usable && !p->failed;
}

static void set_pf(mdt_pphase& ph, int i, const float* p, int64_t floats, int64_t stride = 0) {
    ph.pf[i] = p; ph.pf_floats[i] = floats; ph.pf_stride[i] = stride;
}

// Build the phase list of `n_steps` decoder evaluations (same tensors and argument structs as run_decoder_blocks /
// run_head in mdt_model.hip build for the launch path).
static void build_program(mdt_model* m, int64_t B, int n_steps, const float* x_T, float* out, bool small,
                          std::vector<mdt_pphase>& prog) {
    const int D = m->D, Ta = m->Ta, M = (int)(B * Ta);
    const int64_t np = (int64_t)4 * m->H;  // rows of the folded cross-attention images (heads padded to 4 context tokens)
    prog.clear();
    auto lin_floats = [](const Lin& l) { return (int64_t)l.N * l.K; };
    for (int i = 0; i < n_steps; ++i) {
        const bool last = i == n_steps - 1;
        const float* mod_row = m->mod + (int64_t)i * m->Ld * 6 * D;
        for (int l = 0; l < m->Ld; ++l) {
            const DecBlock& d = m->dec[l];
            const float* row = mod_row + (int64_t)l * 6 * D;
            // ---- qkv ----
            mdt_pphase q;
            memset(&q, 0, sizeof q);
            q.u.g = gemm_args(m->y, D, d.qkv, m->qkv, 3 * D, M);
            q.u.g.ln = 1; q.u.g.ln_w = d.ln1_w; q.u.g.ln_b = d.ln1_b; q.u.g.rows_per_sample = Ta;
            q.u.g.mod = row; q.u.g.mod_stride = 0; q.u.g.shift_off = 0; q.u.g.scale_off = D;
            q.kind = small ? PH_GEMM_SMALL : PH_GEMM_QKV;
            q.kchunk = D; q.grid_n = (3 * D + 383) / 384; q.ncompute = 24;
            set_pf(q, 0, d.proj.wp, lin_floats(d.proj));
            prog.push_back(q);
            // ---- self attention (+ projection) ----
            mdt_gemm_args pg = gemm_args(m->att, D, d.proj, m->y, D, M);
            pg.residual = 1; pg.rows_per_sample = Ta; pg.mod = row; pg.mod_stride = 0; pg.gate_off = 2 * D;
            if (small) {
                mdt_pphase ap;
                memset(&ap, 0, sizeof ap);
                ap.kind = PH_ATTN_PROJ; ap.u.g = pg; ap.qkv = m->qkv; ap.ldq = 3 * D; ap.T = Ta; ap.causal = 1; ap.hd = m->hd;
                ap.scale = 1.0f / sqrtf((float)m->hd); ap.ncompute = 24;
                set_pf(ap, 0, m->xU + (int64_t)l * m->cap * np * D, np * D, np * D);
                set_pf(ap, 1, m->xW + (int64_t)l * m->cap * np * D, np * D, np * D);
                prog.push_back(ap);
            } else {
                mdt_pphase at;
                memset(&at, 0, sizeof at);
                at.kind = PH_ATTN; at.hd = m->hd; at.scale = 1.0f / sqrtf((float)m->hd);
                mdt_attn_args& a = at.u.at;
                a.q = m->qkv; a.ldq = 3 * D; a.k = m->qkv + D; a.v = m->qkv + 2 * D; a.ldkv = 3 * D;
                a.out = m->att; a.ldo = D; a.B = (int)B; a.H = m->H; a.hd = m->hd; a.Tq = Ta; a.Tk = Ta; a.causal = 1; a.rope = 0;
                prog.push_back(at);
                mdt_pphase pj;
                memset(&pj, 0, sizeof pj);
                pj.kind = PH_GEMM_PROJ; pj.u.g = pg; pj.kchunk = D; pj.grid_n = (D + 127) / 128;
                set_pf(pj, 0, m->xU + (int64_t)l * m->cap * np * D, np * D, np * D);
                set_pf(pj, 1, m->xW + (int64_t)l * m->cap * np * D, np * D, np * D);
                prog.push_back(pj);
            }
            // ---- collapsed cross attention ----
            mdt_pphase xa;
            memset(&xa, 0, sizeof xa);
            xa.kind = PH_XATTN; xa.np = (int)np;
            mdt_xapply_args& x = xa.u.xa;
            x.y = m->y; x.ln_w = d.ln3_w; x.ln_b = d.ln3_b; x.bo = d.xproj.bias;
            x.U = m->xU + (int64_t)l * m->cap * np * D; x.Wf = m->xW + (int64_t)l * m->cap * np * D; x.c = m->xc + (int64_t)l * m->cap * np;
            x.B = (int)B; x.H = m->H; x.D = D; x.Te = m->Te; x.Ta = Ta;
            set_pf(xa, 0, d.fc.wp, lin_floats(d.fc));
            prog.push_back(xa);
            // ---- MLP ----
            mdt_pphase fc;
            memset(&fc, 0, sizeof fc);
            fc.u.g = gemm_args(m->y, D, d.fc, m->hid, 4 * D, M);
            fc.u.g.ln = 1; fc.u.g.ln_w = d.ln2_w; fc.u.g.ln_b = d.ln2_b; fc.u.g.act = MDT_ACT_GELU; fc.u.g.rows_per_sample = Ta;
            fc.u.g.mod = row; fc.u.g.mod_stride = 0; fc.u.g.shift_off = 3 * D; fc.u.g.scale_off = 4 * D;
            fc.kind = small ? PH_GEMM_SMALL : PH_GEMM_FC;
            fc.kchunk = D; fc.grid_n = (4 * D + 511) / 512; fc.ncompute = 24;
            set_pf(fc, 0, d.proj2.wp, lin_floats(d.proj2));
            prog.push_back(fc);
            mdt_pphase p2;
            memset(&p2, 0, sizeof p2);
            p2.u.g = gemm_args(m->hid, 4 * D, d.proj2, m->y, D, M);
            p2.u.g.residual = 1; p2.u.g.rows_per_sample = Ta; p2.u.g.mod = row; p2.u.g.mod_stride = 0; p2.u.g.gate_off = 5 * D;
            p2.kind = small ? PH_GEMM_SMALL : PH_GEMM_PROJ;
            p2.kchunk = mdt_gemm_kchunk(4 * D, 0, 768); p2.grid_n = (D + 127) / 128; p2.ncompute = 24;
            {   // next: the following block's qkv, or (through the head phase) the next step's first block
                const DecBlock& nx = m->dec[(l + 1) % m->Ld];
                if (l + 1 < m->Ld || !last) set_pf(p2, 0, nx.qkv.wp, lin_floats(nx.qkv));
            }
            prog.push_back(p2);
        }
        // ---- head ----
        mdt_pphase hd;
        memset(&hd, 0, sizeof hd);
        hd.kind = PH_HEAD;
        mdt_head_args& h = hd.u.h;
        h.y = m->y; h.ln_w = m->dec_ln_w; h.ln_b = m->dec_ln_b; h.Wp = m->Wp; h.bp = m->bp;
        h.x = i == 0 ? x_T : m->xbuf; h.sigma = m->steps + 4 * i + 3; h.sigma_stride = 0;
        h.out = last ? out : m->xbuf;
        h.M = M; h.D = D; h.A = m->A; h.rows_per_sample = Ta; h.mode = MDT_HEAD_DDIM; h.sigma_data = m->cfg.sigma_data;
        h.step = m->steps + 4 * i;
        if (!last) { h.y_next = m->y; h.Wa = m->Wa; h.ba = m->ba; }
        prog.push_back(hd);
    }
}

// Error word of the last persistent launches of this handle (0 = fine).  The word lives in host-mapped memory, so this
// never synchronises; call it after the stream has been synchronised to learn about the latest launch.
extern "C" int32_t mdt_persist_status(mdt_model* m) {
    if (!m || !m->persist) return 0;
    mdt_persist_state* p = (mdt_persist_state*)m->persist;
    if (!p->err_host) return 0;
    const unsigned e = *(volatile unsigned*)p->err_host;
    if (e) p->failed = true;
    return (int32_t)e;
}

// Tuning hook: the next persistent launches of this handle write, per workgroup and phase, the shader-clock stamps of
// phase start and body end (bit 63 of the second: the workgroup held a tile) to `buf` (device memory,
// (n_cu * nphases * 2 + n_cu) 8-byte words; the tail holds xcc << 32 | slot per workgroup).  nullptr switches it off.
extern "C" int32_t mdt_persist_set_debug(mdt_model* m, unsigned long long* buf) {
    if (!m) return MDT_ERR_INVALID_ARG;
    if (!m->persist && persist_init(m) != MDT_OK) return MDT_ERR_HIP;
    ((mdt_persist_state*)m->persist)->dbg = buf;
    return MDT_OK;
}
extern "C" int32_t mdt_persist_phase_count(mdt_model* m) {
    if (!m || !m->persist) return 0;
    return ((mdt_persist_state*)m->persist)->k_nph;
}
extern "C" int32_t mdt_persist_phase_kind(int32_t steps_unused, int32_t idx, int32_t n_dec_layers, int32_t small) {
    (void)steps_unused;
    const int per = n_dec_layers * (small ? 5 : 6) + 1, i = idx % per;
    if (i == per - 1) return PH_HEAD;
    const int j = i % (small ? 5 : 6);
    if (small) { const int k[5] = {PH_GEMM_SMALL, PH_ATTN_PROJ, PH_XATTN, PH_GEMM_SMALL + 100, PH_GEMM_SMALL + 200}; return k[j]; }
    const int k[6] = {PH_GEMM_QKV, PH_ATTN, PH_GEMM_PROJ, PH_XATTN, PH_GEMM_FC, PH_GEMM_PROJ + 100};
    return k[j];
}

extern "C" int64_t mdt_persist_launches(mdt_model* m) {
    if (!m || !m->persist) return 0;
    return ((mdt_persist_state*)m->persist)->launches;
}

// The step loop of mdt_sample_ddim as one launch.  Preconditions (the caller -- mdt_sample_ddim -- has enqueued them on
// `s`): context encoded and folded, modulation table of the n_steps sigmas in m->mod, m->steps filled, the first
// action embedding in m->y.
mdt_status mdt_persist_sample(mdt_model* m, int64_t B, int n_steps, const float* x_T, float* out, hipStream_t s) {
    mdt_persist_state* p = (mdt_persist_state*)m->persist;
    if (!p || !p->usable) return mdt_fail(MDT_ERR_STATE, "persistent decoder kernel not initialised");
    if (mdt_persist_status(m) != 0) {
        return mdt_fail(MDT_ERR_HIP, "an earlier persistent decoder launch was abandoned (status %d: 1 = XCD barrier timed out, 2 = "
                                    "workgroups not spread evenly over the XCDs); its output is invalid. Set MDT_HIP_PERSIST=0 to use the "
                                    "launch sequence", (int)*p->err_host);
    }
    const bool small = B <= persist_small_max();
    const int S = (int)((B + 7) / 8);
    const int nph = n_steps * (m->Ld * (small ? 5 : 6) + 1);
    if (p->k_ws != m->ws || p->k_x != x_T || p->k_out != out || p->k_B != B || p->k_steps != n_steps || p->k_mode != (int)small) {
        if ((size_t)nph > p->prog_cap) {
            HIP_TRY(hipStreamSynchronize(s));
            if (p->prog_dev) HIP_TRY(hipFree(p->prog_dev));
            for (int i = 0; i < 2; ++i)
                if (p->stage[i]) { HIP_TRY(hipHostFree(p->stage[i])); p->stage[i] = nullptr; }
            p->prog_cap = (size_t)nph + 64;
            HIP_TRY(hipMalloc((void**)&p->prog_dev, p->prog_cap * sizeof(mdt_pphase)));
            for (int i = 0; i < 2; ++i) HIP_TRY(hipHostMalloc((void**)&p->stage[i], p->prog_cap * sizeof(mdt_pphase), hipHostMallocDefault));
        }
        std::vector<mdt_pphase> prog;
        build_program(m, B, n_steps, x_T, out, small, prog);
        if ((int)prog.size() != nph) return mdt_fail(MDT_ERR_STATE, "persistent program size mismatch");
        const int si = p->stage_next;
        p->stage_next ^= 1;
        HIP_TRY(hipEventSynchronize(p->stage_ev[si]));  // the copy that last used this staging buffer is long done
        memcpy(p->stage[si], prog.data(), (size_t)nph * sizeof(mdt_pphase));
        HIP_TRY(hipMemcpyAsync(p->prog_dev, p->stage[si], (size_t)nph * sizeof(mdt_pphase), hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(p->stage_ev[si], s));
        p->k_ws = m->ws; p->k_x = x_T; p->k_out = out; p->k_B = B; p->k_steps = n_steps; p->k_mode = (int)small; p->k_nph = nph;
    }
    HIP_TRY(hipMemsetAsync(p->ctl, 0, sizeof(mdt_pctl), s));
    const float* zeros = mdt_zeros();
    if (!zeros) return mdt_fail(MDT_ERR_HIP, "zeros buffer unavailable");
    hipLaunchKernelGGL(p->fn[small ? 1 : 0], dim3(p->n_cu), dim3(512), PERSIST_LDS_BYTES, s, (const mdt_pphase*)p->prog_dev, nph, p->ctl, S,
                       (int)B, m->Ta, zeros, p->err_host_dev, p->dbg);
    HIP_TRY(hipGetLastError());
    p->launches++;
    return MDT_OK;
}

#endif  // MDT_WITH_PERSIST
