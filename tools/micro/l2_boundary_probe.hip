// Does an XCD's L2 keep READ-ONLY lines across a kernel boundary?  Kernel A (one workgroup per CU) reads buffer X: workgroup b
// takes the 64-KiB slice b.  Kernel B, launched behind it on the same stream, has workgroup b read slice b again (same XCD by
// the b % 8 placement) -- or slice b of a buffer Y nobody touched for 100 MB of other traffic (served by the Infinity Cache / HBM).
// If the boundary's acquire dropped the L2, both take the same time.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/l2_boundary_probe.hip -o tools/micro/bin/l2_boundary_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void touch(const float* __restrict__ X, float* out, unsigned long long* ts, int slice_floats) {
    const float* p = X + (size_t)blockIdx.x * slice_floats;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = threadIdx.x * 4; i < slice_floats; i += 256 * 4 * 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(p + min(i + u * 1024, slice_floats - 4));
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) ts[blockIdx.x] = t1 - t0;
}
// producer kernel: workgroup b WRITES slice b (plain stores, or write-through sc0 sc1 stores)
template <bool WT>
__global__ __launch_bounds__(256) void fill(float* __restrict__ X, int slice_floats, float val) {
    float* p = X + (size_t)blockIdx.x * slice_floats;
    for (int i = threadIdx.x * 4; i < slice_floats; i += 256 * 4) {
        const f32x4 v = (f32x4){val, val + 1.f, val + 2.f, val + 3.f};
        if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p + i), "v"(v) : "memory");
        else *(f32x4*)(p + i) = v;
    }
}
int main() {
    const int wgs = 256, slice = 64 * 1024 / 4;   // 64 KiB per workgroup, 16 MiB per buffer (2 MiB per XCD L2)
    float *X, *Y, *Z, *out; unsigned long long* ts;
    (void)hipMalloc(&X, (size_t)wgs * slice * 4); (void)hipMalloc(&Y, (size_t)wgs * slice * 4); (void)hipMalloc(&Z, (size_t)512 << 20);
    (void)hipMalloc(&out, wgs * 256 * 4); (void)hipMalloc(&ts, wgs * 8);
    (void)hipMemset(X, 0, (size_t)wgs * slice * 4); (void)hipMemset(Y, 0, (size_t)wgs * slice * 4); (void)hipMemset(Z, 1, (size_t)512 << 20);
    auto mean = [&]() { std::vector<unsigned long long> h(wgs); (void)hipMemcpy(h.data(), ts, wgs * 8, hipMemcpyDeviceToHost); double m = 0; for (auto v : h) m += v; return m / wgs; };
    for (int rep = 0; rep < 3; ++rep) {
        // evict everything: stream 512 MB through the chip
        hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, Z, out, ts, (512 << 20) / 4 / 2048);
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(256), 0, 0, X, out, ts, slice);   // kernel A: X -> L2 (cold read)
        (void)hipDeviceSynchronize(); const double cold = mean();
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(256), 0, 0, X, out, ts, slice);   // kernel B: X again, next kernel
        (void)hipDeviceSynchronize(); const double again = mean();
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(256), 0, 0, Y, out, ts, slice);   // Y: cold (evicted by the 512 MB stream)
        (void)hipDeviceSynchronize(); const double coldy = mean();
        // back to back without a host sync in between
        hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, Z, out, ts, (512 << 20) / 4 / 2048);
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(256), 0, 0, X, out, ts, slice);
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(256), 0, 0, X, out, ts, slice);
        (void)hipDeviceSynchronize(); const double b2b = mean();
        // X read, then 64 MB of other reads (every L2 is 4 MB: X is gone from them, the 256-MB Infinity Cache still has it), X again
        hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, Z, out, ts, (512 << 20) / 4 / 2048);
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(256), 0, 0, X, out, ts, slice);
        hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, 0, Z, out, ts, (64 << 20) / 4 / 1024);
        hipLaunchKernelGGL(touch, dim3(wgs), dim3(256), 0, 0, X, out, ts, slice);
        (void)hipDeviceSynchronize(); const double mall = mean();
        printf("   the same slice after 64 MB of other traffic (Infinity Cache hit, L2 miss): %7.0f clk\n", mall);
        // the producer / consumer case of the decoder's launch chain: kernel A WRITES slice b on XCD b % 8, kernel B reads it there
        double wr[2];
        for (int wt = 0; wt < 2; ++wt) {
            hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, Z, out, ts, (512 << 20) / 4 / 2048);
            if (wt) hipLaunchKernelGGL(fill<true>, dim3(wgs), dim3(256), 0, 0, X, slice, (float)rep);
            else hipLaunchKernelGGL(fill<false>, dim3(wgs), dim3(256), 0, 0, X, slice, (float)rep);
            hipLaunchKernelGGL(touch, dim3(wgs), dim3(256), 0, 0, X, out, ts, slice);
            (void)hipDeviceSynchronize(); wr[wt] = mean();
        }
        printf("   slice WRITTEN by the previous kernel's workgroup b, read by workgroup b: plain stores %7.0f clk | write-through (sc0 sc1) stores %7.0f clk\n", wr[0], wr[1]);
        printf("64 KiB per workgroup: first read (HBM) %7.0f clk | same slice, next kernel %7.0f clk (back to back on the stream: %7.0f) | untouched buffer %7.0f clk\n", cold, again, b2b, coldy);
    }
    return 0;
}
