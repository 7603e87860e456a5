// Peak-rate probe: how many v_mfma_f32_16x16x4_f32 per second does this chip sustain (no memory traffic)?
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long*)out)[1 << 20] = t1 - t0;
}
template <int NACC>
void run(const char* name, int wgs, float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, d, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc; hipMemcpy(&cyc, (char*)d + (size_t)(1 << 20) * 8, 8, hipMemcpyDeviceToHost);
    double mfma = (double)wgs * 4 * iters * 8 * NACC;
    double tf = mfma * 2048 / (ms * 1e-3) / 1e12;
    printf("%-28s wgs=%4d  %8.3f ms  %7.1f TFLOP/s   readcyclecounter: %llu ticks -> %.3f ticks/us... %.1f ticks per MFMA per wave\n",
           name, wgs, ms, tf, cyc, cyc / (ms * 1e3), (double)cyc / (iters * 8.0 * NACC));
}
int main() {
    float* d; hipMalloc(&d, (size_t)(1 << 20) * 8 + 64);
    run<4>("4 acc, 1 wave/SIMD", 256, d, 20000);
    run<2>("2 acc, 1 wave/SIMD", 256, d, 40000);
    run<4>("4 acc, 2 waves/SIMD", 512, d, 20000);
    run<2>("2 acc, 2 waves/SIMD", 512, d, 40000);
    run<8>("8 acc, 2 waves/SIMD", 512, d, 10000);
    run<4>("4 acc, long (200k iters)", 512, d, 200000);
    return 0;
}
