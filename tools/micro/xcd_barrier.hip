// Probe for a persistent decoder-step kernel (DESIGN.md "performance next"): what does a barrier cost when the
// workgroups that must meet all sit on ONE XCD (activations of a sample slice never leave their XCD's L2), against a
// chip-wide barrier, and are another CU's plain stores visible behind it without fences when read with sc1 loads?
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_barrier.hip -o tools/micro/bin/xcd_barrier
// Every spin is bounded: a barrier that cannot complete sets an error flag and the kernel exits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}
__device__ __forceinline__ unsigned load_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__global__ void k_census(int* xcc_of_block) { if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = xcc_id(); }

// mode 0: per-XCD counter, no fences, sc1 polls            (data read with sc1 loads)
// mode 1: per-XCD counter, release fence before / acquire fence after (data read with plain loads)
// mode 2: ONE chip-wide counter with release / acquire fences (the boundary a kernel launch replaces)
__global__ __launch_bounds__(256) void k_barrier(unsigned* cnt, const int* n_in_xcd, unsigned* buf, const int* peer,
                                                 int iters, int mode, unsigned long long* ticks, unsigned* err) {
    const int b = blockIdx.x, x = xcc_id();
    unsigned* my = mode == 2 ? cnt : cnt + 32 * x;  // counters 128 bytes apart
    const unsigned members = mode == 2 ? gridDim.x : n_in_xcd[x];
    unsigned stale = 0;
    __shared__ int bail;
    if (threadIdx.x == 0) bail = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        buf[(size_t)b * 256 + threadIdx.x] = (unsigned)it;  // this phase's output (1 KiB per workgroup)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (mode != 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = members * (unsigned)it;
            int spins = 0;
            while (load_sc1(my) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 200000) { err[0] = 1; bail = 1; break; }          // ~0.3 s: this barrier cannot complete
                if ((spins & 1023) == 0 && load_sc1(err)) { bail = 1; break; }   // somebody else gave up
            }
            if (mode != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (bail) break;
        // read what the peer workgroup (same XCD in modes 0 / 1, any in mode 2) wrote in this phase
        const unsigned* src = buf + (size_t)peer[b] * 256 + threadIdx.x;
        const unsigned v = mode == 0 ? load_sc1(src) : *(volatile const unsigned*)src;
        stale += v < (unsigned)it;  // (the peer may already have stored phase it + 1: that is not stale)
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[b] = t1 - t0;
    if (stale) atomicAdd(&err[1], stale);
}

int main() {
    const int G = 256, iters = 2000;
    int *d_xcc, *d_n, *d_peer;
    unsigned *d_cnt, *d_buf, *d_err;
    unsigned long long* d_ticks;
    CHECK(hipMalloc(&d_xcc, G * 4)); CHECK(hipMalloc(&d_n, 16 * 4)); CHECK(hipMalloc(&d_peer, G * 4));
    CHECK(hipMalloc(&d_cnt, 32 * 16 * 4)); CHECK(hipMalloc(&d_buf, (size_t)G * 256 * 4)); CHECK(hipMalloc(&d_err, 8));
    CHECK(hipMalloc(&d_ticks, G * 8));
    std::vector<int> xcc(G);
    bool same = true;
    for (int rep = 0; rep < 3; ++rep) {  // placement must repeat from launch to launch for the per-XCD counts to hold
        hipLaunchKernelGGL(k_census, dim3(G), dim3(256), 0, 0, d_xcc);
        std::vector<int> now(G);
        CHECK(hipMemcpy(now.data(), d_xcc, G * 4, hipMemcpyDeviceToHost));
        if (rep && now != xcc) same = false;
        xcc = now;
    }
    int n[16] = {0};
    for (int b = 0; b < G; ++b) n[xcc[b]]++;
    printf("census: blocks per XCD =");
    for (int x = 0; x < 8; ++x) printf(" %d", n[x]);
    printf("  (block b on XCD b %% 8: %s; repeatable: %s)\n", [&] { for (int b = 0; b < G; ++b) if (xcc[b] != b % 8) return "no"; return "yes"; }(),
           same ? "yes" : "no");
    std::vector<int> peer(G);
    for (int b = 0; b < G; ++b) {  // next block on the same XCD (cyclic)
        int p = b;
        do { p = (p + 1) % G; } while (xcc[p] != xcc[b]);
        peer[b] = p;
    }
    CHECK(hipMemcpy(d_n, n, 64, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_peer, peer.data(), G * 4, hipMemcpyHostToDevice));
    const char* names[3] = {"per-XCD counter, no fences, sc1 reads", "per-XCD counter, release/acquire fences, plain reads",
                            "chip-wide counter, release/acquire fences"};
    for (int mode = 0; mode < 3; ++mode) {
        CHECK(hipMemset(d_cnt, 0, 32 * 16 * 4)); CHECK(hipMemset(d_err, 0, 8)); CHECK(hipMemset(d_buf, 0, (size_t)G * 256 * 4));
        hipLaunchKernelGGL(k_barrier, dim3(G), dim3(256), 0, 0, d_cnt, d_n, d_buf, d_peer, iters, mode, d_ticks, d_err);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(G);
        unsigned err[2];
        CHECK(hipMemcpy(t.data(), d_ticks, G * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(err, d_err, 8, hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for (auto v : t) mx = v > mx ? v : mx;
        printf("%-56s %7.3f us per phase (write 1 KiB + barrier + read peer), stale reads %u%s\n", names[mode],
               (double)mx / iters / 100.0, err[1], err[0] ? "  [BARRIER TIMED OUT]" : "");
    }
    return 0;
}
