// How fast does the FP32 matrix pipe run on REAL data?  The same back-to-back v_mfma_f32_16x16x4_f32 stream as mfma_peak.hip
// (4 accumulators, 2 waves per SIMD, no memory traffic) with (a) constant operands, (b) pseudo-random operands in [-1, 1) that
// change every instruction (8 register pairs, rotated), accumulators kept bounded.  The clock the chip sustains depends on
// the toggling of the operands (power): read TFLOP/s and ticks/us.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o tools/micro/bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;
}
template <int MODE, int TPB = 256>
__global__ __launch_bounds__(TPB) void k(float* out, int iters) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        const unsigned id = (blockIdx.x * TPB + threadIdx.x) * 16 + i;
        a[i] = MODE == 0 ? 1.0001f : (MODE == 1 ? rnd(id) : 0.f);
        b[i] = MODE == 0 ? 0.9999f : (MODE == 1 ? rnd(id + 8) * 0.05f : 0.f);
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + i) & 7], b[(u * 3 + i) & 7], acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * TPB + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long*)out)[1 << 20] = t1 - t0;
}
template <int MODE, int TPB = 256>
void run(const char* name, float* d, int iters, int launches, int wgs = 512) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, TPB>), dim3(wgs), dim3(TPB), 0, 0, d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<MODE, TPB>), dim3(wgs), dim3(TPB), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc; (void)hipMemcpy(&cyc, (char*)d + (size_t)(1 << 20) * 8, 8, hipMemcpyDeviceToHost);
    const double mfma = (double)wgs * (TPB / 64) * iters * 32 * launches;
    printf("%-44s %8.3f ms  %6.1f TFLOP/s  %7.1f ticks/us  %.2f ticks per MFMA per wave\n", name, ms, mfma * 2048 / (ms * 1e-3) / 1e12,
           cyc * (double)launches / (ms * 1e3), (double)cyc / (iters * 32.0));
}
int main() {
    float* d; (void)hipMalloc(&d, (size_t)(1 << 20) * 8 + 64);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("constant operands, one 17 ms launch", d, 20000, 1);
        run<1>("random operands, one 17 ms launch", d, 20000, 1);
        run<2>("zero operands, one 17 ms launch", d, 20000, 1);
        run<1>("random operands, 200 launches of 85 us", d, 100, 200);
        run<0>("constant operands, 200 launches of 85 us", d, 100, 200);
        run<1, 512>("random, 200 launches, 256 WGs x 512 thr", d, 100, 200, 256);
        run<1, 512>("random, 200 launches, 240 WGs x 512 thr", d, 100, 200, 240);
    }
    return 0;
}
