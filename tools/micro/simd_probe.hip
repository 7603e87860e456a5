// Which SIMD does wave w of a 512-thread workgroup land on?  (mlp_tile's skew assumes waves w and w + 4 share one.)
// hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/simd_probe tools/micro/simd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    extern __shared__ float lds[];  // 115 KB: one workgroup per CU, as k_mlp
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
    if (threadIdx.x == 9999) lds[0] = 1.f;
}
int main() {
    unsigned* d; hipMalloc(&d, 240 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 115776);
    hipLaunchKernelGGL(k, dim3(240), dim3(512), 115776, 0, d);
    unsigned h[240 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0, rr = 0;
    for (int b = 0; b < 240; ++b) {
        bool s = true, r = true;
        for (int w = 0; w < 4; ++w) { s = s && (((h[b*8+w] >> 4) & 3) == ((h[b*8+w+4] >> 4) & 3)); }
        for (int w = 0; w < 8; ++w) { r = r && (((h[b*8+w] >> 4) & 3) == ((h[b*8] >> 4) + w) % 4u % 4u); }
        same += s; rr += r;
    }
    printf("workgroups whose waves w and w+4 share a SIMD: %d / 240 (round-robin from wave 0's SIMD: %d)\n", same, rr);
    for (int b = 0; b < 3; ++b) { printf("wg %d: simd of waves 0..7 =", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b*8+w] >> 4) & 3); printf("  (wave slots:"); for (int w = 0; w < 8; ++w) printf(" %u", h[b*8+w] & 15); printf(")\n"); }
    return 0;
}
