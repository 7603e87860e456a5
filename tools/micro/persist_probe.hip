// Probes behind the persistent decoder kernel.  Every spin is bounded.
//   A. how fast can the 32 CUs of ONE XCD stream a weight set that lives in the Infinity Cache (the rollout batch
//      B <= 8 runs one sample per XCD: 54 MB of decoder weights per step through each active XCD), and what do all
//      eight XCDs reach together when each streams the whole set (the B = 256 pattern)?
//   B. hand-off inside an XCD without fences: plain 16-byte stores -> s_waitcnt vmcnt(0) -> per-XCD counter barrier ->
//      `buffer_load_dwordx4 ... sc1` (L1 bypass) by every other workgroup of the XCD, under uneven load, consumer
//      re-reading the same addresses every phase (L1-warm), every word checked.
//   C. the same across XCDs with write-through (sc0 sc1) stores and one chip-wide counter (no cache-wide fences).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/persist_probe.hip -o tools/micro/bin/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}
__device__ __forceinline__ unsigned load_sc1(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
}

// ---- A: weight streaming ----
// mode 0: only XCD `only` works; mode 1: every XCD streams the whole buffer.  Each workgroup of an XCD takes an
// interleaved share (16 KiB pieces); 8 x 16-byte loads in flight per lane.
__global__ __launch_bounds__(512) void k_stream(const f32x4* w, size_t n16, int iters, int only, int nt,
                                                unsigned* slot_ctr, unsigned long long* ticks, float* sink) {
    extern __shared__ float pad[];
    __shared__ int s_slot, s_x;
    if (threadIdx.x == 0) {
        s_x = xcc_id();
        s_slot = (only < 0 || s_x == only) ? (int)atomicAdd(&slot_ctr[s_x * 32], 1u) : -1;
    }
    __syncthreads();
    const int slot = s_slot;
    if (slot < 0) return;
    const size_t per = 1024;  // f32x4 per piece = 16 KiB
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        for (size_t p = (size_t)slot * per; p + per <= n16; p += 32 * per) {
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4* a = w + p + u * 512 + threadIdx.x;
                v[u] = nt ? __builtin_nontemporal_load(a) : *a;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) acc += v[u];
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (acc.x == 12345.678f) sink[0] = acc.y;
}

// ---- B / C: hand-off ----
// PAY f32x4 per workgroup per phase.  mode 0: XCD-local (plain stores, per-XCD counter, sc1 buffer loads of ALL other
// members' payload); mode 1: chip-wide (sc0 sc1 stores, one counter, reads the payload of 8 workgroups on other XCDs).
template <int PAY>
__global__ __launch_bounds__(512) void k_handoff(unsigned* cnt, unsigned* slot_ctr, int* slot_block, float* buf, int iters, int mode,
                                                 unsigned long long* ticks, unsigned* err) {
    extern __shared__ float pad[];
    __shared__ int s_slot, s_x, bail;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_x = xcc_id();
        s_slot = (int)atomicAdd(&slot_ctr[s_x * 32], 1u);
        if (s_slot < 32) slot_block[s_x * 32 + s_slot] = blockIdx.x;
        bail = s_slot >= 32;
        if (bail) err[0] = 2;
    }
    __syncthreads();
    if (bail) return;
    const int x = s_x, slot = s_slot;
    const int me = x * 32 + slot;            // payload row of this workgroup
    unsigned* my = mode == 0 ? cnt + 32 * x : cnt + 32 * 8;
    const unsigned members = mode == 0 ? 32u : 256u;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(buf);
    unsigned bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        // uneven load: some workgroups dawdle before publishing
        if (((slot * 7 + it) & 7) == 0) __builtin_amdgcn_s_sleep(64);
        for (int i = tid; i < PAY; i += 512) {
            const float val = (float)(it * 4096 + (me * 16 + (i & 15)));
            f32x4 v = (f32x4){val, val + 0.25f, val + 0.5f, val + 0.75f};
            float* p = buf + ((size_t)me * PAY + i) * 4;
            if (mode == 0) *(f32x4*)p = v;
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = members * (unsigned)it;
            int spins = 0;
            while (load_sc1(my) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 400000) { err[0] = 1; bail = 1; break; }
                if ((spins & 1023) == 0 && load_sc1(err)) { bail = 1; break; }
            }
        }
        __syncthreads();
        if (bail) break;
        // read: mode 0 every member of my XCD; mode 1 the same slot on the 7 other XCDs + my own
        const int nsrc = mode == 0 ? 32 : 8;
        for (int s = 0; s < nsrc; ++s) {
            const int src = mode == 0 ? x * 32 + s : s * 32 + slot;
            for (int i = tid; i < PAY; i += 512) {
                const u32x4 raw = ld_sc1(rb, (unsigned)(((size_t)src * PAY + i) * 16));
                const f32x4 v = __builtin_bit_cast(f32x4, raw);
                const float want = (float)(it * 4096 + (src * 16 + (i & 15)));
                // the producer may already have written phase it + 1 (never more: it waits at the next barrier)
                const float want2 = (float)((it + 1) * 4096 + (src * 16 + (i & 15)));
                const bool ok = (v.x == want && v.w == want + 0.75f) || (v.x == want2 && v.w == want2 + 0.75f);
                bad += !ok;
            }
        }
        // all reads of this phase must be done before anybody overwrites: second barrier (same counter)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    if (bad) atomicAdd(&err[1], bad);
}

int main() {
    const int G = 256;
    unsigned *d_cnt, *d_slot, *d_err;
    int* d_sb;
    unsigned long long* d_ticks;
    float *d_w, *d_sink, *d_buf;
    const size_t wbytes = 56u << 20;  // ~ the decoder's weights
    CHECK(hipMalloc(&d_cnt, 32 * 16 * 4)); CHECK(hipMalloc(&d_slot, 32 * 16 * 4)); CHECK(hipMalloc(&d_err, 8));
    CHECK(hipMalloc(&d_sb, 256 * 4)); CHECK(hipMalloc(&d_ticks, G * 8)); CHECK(hipMalloc(&d_w, wbytes)); CHECK(hipMalloc(&d_sink, 16));
    CHECK(hipMemset(d_w, 0, wbytes));
    const int lds = 96 * 1024;  // one workgroup per CU
    CHECK(hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)k_handoff<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)k_handoff<64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int nb = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_handoff<1024>, 512, lds));
    printf("occupancy query (512 threads, 96 KiB LDS): %d block(s) per CU\n", nb);
    std::vector<unsigned long long> t(G);
    for (int mode = 0; mode < 2; ++mode)
        for (int nt = 0; nt < 2; ++nt) {
            const int iters = 12;
            CHECK(hipMemset(d_slot, 0, 32 * 16 * 4)); CHECK(hipMemset(d_ticks, 0, G * 8));
            hipLaunchKernelGGL(k_stream, dim3(G), dim3(512), lds, 0, (const f32x4*)d_w, wbytes / 16, 2, mode == 0 ? 0 : -1, nt, d_slot, d_ticks, d_sink);  // warm the MALL
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(d_slot, 0, 32 * 16 * 4)); CHECK(hipMemset(d_ticks, 0, G * 8));
            hipLaunchKernelGGL(k_stream, dim3(G), dim3(512), lds, 0, (const f32x4*)d_w, wbytes / 16, iters, mode == 0 ? 0 : -1, nt, d_slot, d_ticks, d_sink);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(t.data(), d_ticks, G * 8, hipMemcpyDeviceToHost));
            unsigned long long mx = 0;
            for (auto v : t) mx = std::max(mx, v);
            const double us = (double)mx / 100.0;
            const double per_xcd = (double)wbytes * iters / us / 1e6;  // TB/s
            printf("A stream %-10s %s: %.1f us per %zu MB pass -> %.2f TB/s per XCD%s\n", mode == 0 ? "one XCD" : "all 8 XCDs", nt ? "nt   " : "plain",
                   us / iters, wbytes >> 20, per_xcd, mode ? "" : "");
            if (mode) printf("         aggregate L2-fill rate %.2f TB/s\n", per_xcd * 8);
        }
    for (int mode = 0; mode < 2; ++mode)
        for (int big = 0; big < 2; ++big) {
            const int iters = 1500;
            const int pay = big ? 1024 : 64;  // 16 KiB or 1 KiB per workgroup and phase
            CHECK(hipMalloc(&d_buf, (size_t)256 * pay * 16));
            CHECK(hipMemset(d_buf, 0, (size_t)256 * pay * 16));
            CHECK(hipMemset(d_cnt, 0, 32 * 16 * 4)); CHECK(hipMemset(d_slot, 0, 32 * 16 * 4)); CHECK(hipMemset(d_err, 0, 8)); CHECK(hipMemset(d_ticks, 0, G * 8));
            if (big) hipLaunchKernelGGL(k_handoff<1024>, dim3(G), dim3(512), lds, 0, d_cnt, d_slot, d_sb, d_buf, iters, mode, d_ticks, d_err);
            else hipLaunchKernelGGL(k_handoff<64>, dim3(G), dim3(512), lds, 0, d_cnt, d_slot, d_sb, d_buf, iters, mode, d_ticks, d_err);
            CHECK(hipDeviceSynchronize());
            unsigned err[2];
            CHECK(hipMemcpy(t.data(), d_ticks, G * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(err, d_err, 8, hipMemcpyDeviceToHost));
            unsigned long long mx = 0;
            for (auto v : t) mx = std::max(mx, v);
            printf("%s hand-off, %5d B per WG: %7.3f us per phase (publish + barrier + read %d peers + drain), bad words %u%s\n",
                   mode == 0 ? "B XCD-local (plain st, sc1 buffer ld)" : "C chip-wide (sc0sc1 st, sc1 buffer ld)", pay * 16, (double)mx / iters / 100.0,
                   mode == 0 ? 32 : 8, err[1], err[0] == 1 ? "  [BARRIER TIMED OUT]" : (err[0] == 2 ? "  [UNEVEN PLACEMENT]" : ""));
            CHECK(hipFree(d_buf));
        }
    return 0;
}
