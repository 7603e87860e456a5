// What does a kernel pay before its first global load can be issued?  One workgroup-0 thread stamps the shader clock at
// entry, after its kernel arguments have arrived in SGPRs (they are an s_load from the kernarg segment the command
// processor wrote just before the dispatch -- unless they are PRELOADED into user SGPRs), and after a dependent cold
// global load.  Three argument forms: a 200-byte struct by value (what the decoder kernels take), four scalars
// (preloadable with -mllvm -amdgpu-kernarg-preload-count=N), a pointer to the same struct in device memory.
// build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=16] tools/micro/kernarg_probe.hip -o tools/micro/bin/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Big { const float* a; long lda; const float* w; float* out; unsigned long long* ts; long pad[20]; int m, n, k, idx; };

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

__global__ void k_struct(Big g) {
    const unsigned long long t0 = now();
    unsigned long long* ts = g.ts;             // needs the kernarg load
    asm volatile("" ::"s"(ts));
    const unsigned long long t1 = now();
    const float v = g.a[(size_t)blockIdx.x * 4096 + threadIdx.x * 64 + g.idx];   // cold, dependent on the arguments
    asm volatile("" ::"v"(v));
    const unsigned long long t2 = now();
    if (threadIdx.x == 0) { ts[blockIdx.x * 4 + 0] = t1 - t0; ts[blockIdx.x * 4 + 1] = t2 - t1; ts[blockIdx.x * 4 + 2] = t0; }
    if (v == 123.456f) g.out[0] = v;
}
__global__ void k_scalar(const float* a, float* out, unsigned long long* ts, int idx) {
    const unsigned long long t0 = now();
    asm volatile("" ::"s"(ts));
    const unsigned long long t1 = now();
    const float v = a[(size_t)blockIdx.x * 4096 + threadIdx.x * 64 + idx];
    asm volatile("" ::"v"(v));
    const unsigned long long t2 = now();
    if (threadIdx.x == 0) { ts[blockIdx.x * 4 + 0] = t1 - t0; ts[blockIdx.x * 4 + 1] = t2 - t1; ts[blockIdx.x * 4 + 2] = t0; }
    if (v == 123.456f) out[0] = v;
}
__global__ void k_ptr(const Big* gp) {
    const unsigned long long t0 = now();
    unsigned long long* ts = gp->ts;           // kernarg load, then a scalar load from device memory
    asm volatile("" ::"s"(ts));
    const unsigned long long t1 = now();
    const float v = gp->a[(size_t)blockIdx.x * 4096 + threadIdx.x * 64 + gp->idx];
    asm volatile("" ::"v"(v));
    const unsigned long long t2 = now();
    if (threadIdx.x == 0) { ts[blockIdx.x * 4 + 0] = t1 - t0; ts[blockIdx.x * 4 + 1] = t2 - t1; ts[blockIdx.x * 4 + 2] = t0; }
    if (v == 123.456f) gp->out[0] = v;
}

int main() {
    const int G = 256;
    float *a, *out; unsigned long long* ts; Big* gd; float* flush;
    CHECK(hipMalloc(&a, (size_t)G * 4096 * 4 * 8)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&ts, G * 4 * 8)); CHECK(hipMalloc(&gd, sizeof(Big)));
    CHECK(hipMalloc(&flush, 512u << 20));
    Big g{}; g.a = a; g.out = out; g.ts = ts; g.idx = 0;
    std::vector<unsigned long long> h(G * 4);
    auto report = [&](const char* name) {
        CHECK(hipMemcpy(h.data(), ts, G * 4 * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> d0, d1;
        for (int i = 0; i < G; ++i) { d0.push_back(h[i * 4]); d1.push_back(h[i * 4 + 1]); }
        std::sort(d0.begin(), d0.end()); std::sort(d1.begin(), d1.end());
        printf("%-34s arguments usable after p10 %5llu  p50 %5llu  p90 %5llu clk | dependent cold load after p10 %5llu  p50 %5llu  p90 %5llu clk\n", name,
               d0[G / 10], d0[G / 2], d0[G * 9 / 10], d1[G / 10], d1[G / 2], d1[G * 9 / 10]);
    };
    for (int rep = 0; rep < 3; ++rep) {
        g.idx = rep;  // another cache line of `a` every round; the flush below evicts L2 / MALL
        CHECK(hipMemcpy(gd, &g, sizeof g, hipMemcpyHostToDevice));
        CHECK(hipMemset(flush, rep, 512u << 20)); CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_struct, dim3(G), dim3(64), 0, 0, g); CHECK(hipDeviceSynchronize()); report("struct by value (200 B)");
        CHECK(hipMemset(flush, rep + 7, 512u << 20)); CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_scalar, dim3(G), dim3(64), 0, 0, (const float*)a, out, ts, rep); CHECK(hipDeviceSynchronize()); report("four scalars");
        CHECK(hipMemset(flush, rep + 9, 512u << 20)); CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_ptr, dim3(G), dim3(64), 0, 0, (const Big*)gd); CHECK(hipDeviceSynchronize()); report("pointer to the struct (device)");
        // back to back without a flush: what a launch inside a stream of launches sees
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_struct, dim3(G), dim3(64), 0, 0, g);
        CHECK(hipDeviceSynchronize()); report("struct by value, 4th in a row");
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_scalar, dim3(G), dim3(64), 0, 0, (const float*)a, out, ts, rep);
        CHECK(hipDeviceSynchronize()); report("four scalars, 4th in a row");
    }
    return 0;
}
