// Do vector-ALU instructions of OTHER waves run beside a SIMD's MFMA stream, or do they take its time?
// One 512-thread workgroup per CU (2 waves per SIMD).  Wave pair (w, w + 4) shares a SIMD: waves 0..3 issue a back-to-back
// v_mfma_f32_16x16x4_f32 stream (4 accumulators), waves 4..7 issue, per MODE,
//   0 nothing (exit)            1 the same MFMA stream            2 independent v_fma_f32          3 v_exp_f32
//   4 v_pk_fma_f32              5 ds_read_b128 (conflict free)    6 v_max3_f32
// and, MODE 7..9: ONE wave per SIMD that interleaves K vector instructions (fma / exp / ds_read) after every MFMA itself.
// Reported: clocks per MFMA of wave 0 (32 = the pipe alone) and what the partner wave got done per MFMA of wave 0.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_mfma_probe.hip -o tools/micro/bin/valu_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4000;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ts) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    float r = 0.f;
    unsigned long long t0 = 0, t1 = 0, work = 0;
    if (wave < 4 || MODE == 1) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float a = 1.0001f + lane * 1e-7f, b = 0.9999f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        const f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
        r = s.x + s.y + s.z + s.w;
        work = (unsigned long long)ITERS * 8;
    } else if (MODE >= 2 && MODE <= 6) {
        // run until wave (wave - 4) is done: a flag in LDS would perturb; run a fixed count sized to outlast the MFMA stream
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = 0.5f + lane * 1e-3f + i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_readcyclecounter();
        const int n = ITERS * 8;   // groups of 8 instructions
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 2) x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f);
                if (MODE == 3) x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.f + x[i];   // exp + fma: counted as the pair
                if (MODE == 4) { f32x2 p = {x[i], x[(i + 1) & 7]}; p = p * 0.999f + 0.001f; x[i] = p.x; }
                if (MODE == 5) { v += *(const volatile f32x4*)(lds + ((lane * 4 + i * 256 + it * 4) & 8188)); }
                if (MODE == 6) x[i] = fmaxf(fmaxf(x[i], x[(i + 1) & 7]), 0.25f);
            }
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 8; ++i) r += x[i];
        r += v.x + v.y + v.z + v.w;
        work = (unsigned long long)n * 8;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0 && blockIdx.x == 0) { ts[wave * 2] = t1 - t0; ts[wave * 2 + 1] = work; }
}

// one wave per SIMD (256 threads), K vector instructions of kind KIND after every MFMA, in the same wave
template <int KIND, int K>
__global__ __launch_bounds__(256) void k_self(float* out, unsigned long long* ts) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = 1.0001f + lane * 1e-7f, b = 0.9999f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.5f + lane * 1e-3f + i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int i = (u * K + j) & 7;
                if (KIND == 0) x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f);
                if (KIND == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);
                if (KIND == 2) v += *(const volatile f32x4*)(lds + ((lane * 4 + i * 256 + it * 4) & 8188));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const f32x4 s = acc[0] + acc[1] + acc[2] + acc[3] + v;
    float r = s.x + s.y + s.z + s.w;
    for (int i = 0; i < 8; ++i) r += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ts[0] = t1 - t0; ts[1] = (unsigned long long)ITERS * 8; }
}

template <int MODE>
void run(const char* name, float* d, unsigned long long* ts) {
    (void)hipMemset(ts, 0, 16 * 8);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, ts);
    (void)hipDeviceSynchronize();
    unsigned long long h[16];
    (void)hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
    const double per_mfma = (double)h[0] / (double)h[1];
    printf("%-46s wave 0: %6.2f clocks per MFMA", name, per_mfma);
    if (h[9]) printf("   partner wave 4: %8.2f clocks per instruction (%5.2f per MFMA slot of 32)", (double)h[8] / (double)h[9], 32.0 * h[9] / (double)h[8]);
    printf("\n");
}
template <int KIND, int K>
void run_self(const char* name, float* d, unsigned long long* ts) {
    hipLaunchKernelGGL((k_self<KIND, K>), dim3(256), dim3(256), 0, 0, d, ts);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s %6.2f clocks per MFMA (+ %d instructions each)\n", name, (double)h[0] / (double)h[1], K);
}
int main() {
    float* d; unsigned long long* ts;
    (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&ts, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("MFMA wave alone on its SIMD", d, ts);
        run<1>("partner: the same MFMA stream", d, ts);
        run<2>("partner: v_fma_f32", d, ts);
        run<3>("partner: v_exp_f32 + v_fma_f32 pairs", d, ts);
        run<4>("partner: v_pk_fma_f32", d, ts);
        run<5>("partner: ds_read_b128", d, ts);
        run<6>("partner: 2 x v_max_f32", d, ts);
        run_self<0, 1>("same wave: 1 v_fma per MFMA", d, ts);
        run_self<0, 4>("same wave: 4 v_fma per MFMA", d, ts);
        run_self<0, 8>("same wave: 8 v_fma per MFMA", d, ts);
        run_self<1, 1>("same wave: 1 v_exp per MFMA", d, ts);
        run_self<1, 2>("same wave: 2 v_exp per MFMA", d, ts);
        run_self<2, 1>("same wave: 1 ds_read_b128 per MFMA", d, ts);
    }
    return 0;
}
