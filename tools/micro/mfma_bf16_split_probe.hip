// What would a THREE-way bf16 split of fp32 operands buy on the matrix pipe?  
// fp32 path of this library: K = 32 of a 16 x 16 tile = 8 x v_mfma_f32_16x16x4_f32.
// split path: a = a1 + a2 + a3, b = b1 + b2 + b3 (bf16 each): K = 32 = 6 x v_mfma_f32_16x16x32_bf16 (a1 b1, a1 b2, a2 b1, a1 b3, a2 b2,
// a3 b1; fp32 accumulation) -- or 3 products for a two-way split (16 mantissa bits).
// Pure issue-rate probe, operands in registers, no memory traffic; (a) fp32, (b) six-product bf16, (c) three-product bf16, each with
// the SAME number of independent accumulators per wave, and (d) the six-product form with a ds_read_b128 of every operand in front
// of each k-step (the fragments of a real kernel come from LDS: 3 + 3 reads per k32 instead of 8 x 2 for fp32's 8 k4 steps).
// build + run: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16_split_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 6];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = a0 + threadIdx.x * 1e-6f, b = 0.9999f;
    bf16x8 a1, a2, a3, b1, b2, b3;
    for (int e = 0; e < 8; ++e) {
        a1[e] = (__bf16)(a + e); a2[e] = (__bf16)(a * 1e-3f); a3[e] = (__bf16)(a * 1e-6f);
        b1[e] = (__bf16)(b + e); b2[e] = (__bf16)(b * 1e-3f); b3[e] = (__bf16)(b * 1e-6f);
    }
    for (int i = threadIdx.x; i < 256 * 4 * 6; i += 256) lds[i] = a + i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // fp32: 8 k4 steps per accumulator
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + u, b, acc[i], 0, 0, 0);
        } else {
            if constexpr (MODE == 3) {      // operands of this k32 step from LDS: six 16-byte reads per lane
                const f32x4* p = (const f32x4*)lds + threadIdx.x;
                a1 = __builtin_bit_cast(bf16x8, p[0]); a2 = __builtin_bit_cast(bf16x8, p[256]); a3 = __builtin_bit_cast(bf16x8, p[512]);
                b1 = __builtin_bit_cast(bf16x8, p[768]); b2 = __builtin_bit_cast(bf16x8, p[1024]); b3 = __builtin_bit_cast(bf16x8, p[1280]);
            }
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc[i], 0, 0, 0);
                if constexpr (MODE != 2) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc[i], 0, 0, 0);
                }
            }
        }
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int MODE, int NACC>
void run(const char* name, int wgs, float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, d, 10, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // fp32-EQUIVALENT work: every iteration covers K = 32 of a 16 x 16 tile per accumulator = 16384 FLOP
    const double flop = (double)wgs * 4 * iters * NACC * 16384.0;
    printf("%-58s wgs=%4d %8.3f ms  %8.1f TFLOP/s fp32-equivalent  (%.2f x the 157.3 TFLOP/s fp32 matrix peak)\n", name, wgs, ms,
           flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
}
int main() {
    float* d; hipMalloc(&d, (size_t)(1 << 22));
    run<0, 4>("(a) fp32 16x16x4, 8 per k32, 4 acc, 2 waves/SIMD", 512, d, 20000);
    run<1, 4>("(b) bf16 3-way split, 6 x 16x16x32 per k32, 4 acc", 512, d, 20000);
    run<2, 4>("(c) bf16 2-way split, 3 x 16x16x32 per k32, 4 acc", 512, d, 20000);
    run<3, 4>("(d) as (b) with six ds_read_b128 per k32 step", 512, d, 20000);
    run<1, 8>("(b) 8 acc", 512, d, 10000);
    run<3, 8>("(d) 8 acc", 512, d, 10000);
    return 0;
}
