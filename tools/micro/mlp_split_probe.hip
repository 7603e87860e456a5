// Would the fused MLP launch (k_mlp: 32-row tile x 512-column hidden slice per workgroup, weights streamed L2 -> registers) gain from
// the three-way bf16 split?    Issue / bandwidth probe of the split form's two MFMA phases with the real
// launch shape -- 240 workgroups x 8 waves, B = 256 (2560 rows), d = 384, hidden 1536 -- and the real operand traffic: per k32 step
// a wave requests 3 parts x NT column tiles of PRE-SPLIT weight fragments (16 B per lane each) from a 2.36 MB-per-slice image all the
// workgroups of a slice share, and reads its activation fragments (3 parts x 2 row tiles) from LDS.  No LayerNorm staging, garbage
// data; the GELU + re-split epilogue between the phases is there (VALU + LDS stores).  Compare with k_mlp's 51-55 us.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/micro/mlp_split_probe.hip -o tools/micro/bin/mlp_split_probe && tools/micro/bin/mlp_split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define PIN __builtin_amdgcn_sched_barrier(0x6);

__device__ __forceinline__ bf16x8 ldw(const char* p) { return __builtin_bit_cast(bf16x8, __builtin_nontemporal_load((const f32x4*)p)); }
__device__ __forceinline__ bf16x8 ldw_plain(const char* p) { return __builtin_bit_cast(bf16x8, *(const f32x4*)p); }

template <int NT, int K32, int R, int MT = 2>
__device__ __forceinline__ void phase(const char* __restrict__ wimg, int64_t tile_stride, const char* lds_a, int rowb, int part, int lane,
                                      f32x4 (&acc)[MT][NT]) {
    // fragment (column tile j, step kk, part p): 1 KiB at wimg + j * tile_stride + (kk * 3 + p) * 1024, lane's 16 bytes at lane * 16
    bf16x8 w[R][NT][3];
#pragma unroll
    for (int u = 0; u < R - 1; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) w[u][j][p] = ldw_plain(wimg + j * tile_stride + (u * 3 + p) * 1024 + lane * 16);
    const int aoff = (lane & 15) * rowb + (lane >> 4) * 16;
    bf16x8 x1[MT], x2[MT], x3[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const char* q = lds_a + aoff + i * 16 * rowb;
        x1[i] = *(const bf16x8*)q; x2[i] = *(const bf16x8*)(q + part); x3[i] = *(const bf16x8*)(q + 2 * part);
    }
#pragma unroll
    for (int kk = 0; kk < K32; ++kk) {
        const int u = kk % R, un = (kk + R - 1) % R;
        const bool nx = kk + 1 < K32;
        if (kk + R - 1 < K32) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) w[un][j][p] = ldw_plain(wimg + j * tile_stride + ((kk + R - 1) * 3 + p) * 1024 + lane * 16);
        }
        const char* p0 = lds_a + aoff + (kk + 1) * 64;
        PIN
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][j][0], x3[i], acc[i][j], 0, 0, 0);
        }
        PIN
        if (nx) {
#pragma unroll
            for (int i = 0; i < MT; ++i) x3[i] = *(const bf16x8*)(p0 + i * 16 * rowb + 2 * part);
        }
        PIN
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][j][1], x2[i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][j][0], x2[i], acc[i][j], 0, 0, 0);
        }
        PIN
        if (nx) {
#pragma unroll
            for (int i = 0; i < MT; ++i) x2[i] = *(const bf16x8*)(p0 + i * 16 * rowb + part);
        }
        PIN
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][j][2], x1[i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][j][1], x1[i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][j][0], x1[i], acc[i][j], 0, 0, 0);
        }
        PIN
        if (nx) {
#pragma unroll
            for (int i = 0; i < MT; ++i) x1[i] = *(const bf16x8*)(p0 + i * 16 * rowb);
        }
        PIN
    }
}

__device__ __forceinline__ float gelu(float x) {   // stand-in of the library's A&S form: one rcp, one exp, a Horner chain
    const float z = fabsf(x) * 0.70710678f, t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    const float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float h = 0.5f * p * __expf(-z * z);
    return fmaxf(x, 0.f) - fabsf(x) * h;
}

template <int R1, int R2, int XCD>
__global__ __launch_bounds__(512) void k_probe(const char* __restrict__ w1, const char* __restrict__ w2, float* __restrict__ out, int phases) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int ROWB1 = 2 * 384 + 32, PART1 = 32 * ROWB1, ROWB2 = 2 * 512 + 32, PART2 = 32 * ROWB2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    if (XCD) b = (b & 7) * 30 + (b >> 3);           // XCD x: 30 consecutive (slice-major) workgroups
    const int s = b / 80, by = b - s * 80;
    for (int i = tid; i < (3 * PART2) / 16; i += 512) ((f32x4*)lds)[i] = (f32x4){1.f + i * 1e-6f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---- phase 1: hidden columns [512 s + 64 wave, + 64): column tiles (s * 8 + wave) * 4 + j of W1 (96 tiles x 12 steps x 3 KiB) ----
    f32x4 acc1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = zero4;
    if (phases & 1) phase<4, 12, R1>(w1 + (int64_t)((s * 8 + wave) * 4) * 12 * 3072, 12 * 3072, lds, ROWB1, PART1, lane, acc1);
    __syncthreads();                                 // everybody has read the x tile: the hidden slice may overwrite it
    // ---- activation + re-split -> LDS (lane holds hidden[16 i + lane % 16][(4 wave + j) * 16 + 4 (lane / 16) .. + 3]) ----
    if (phases & 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 p1, p2, p3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = gelu(acc1[i][j][e] + 0.1f);
                    const __bf16 a = (__bf16)v; const float r = v - (float)a; const __bf16 bb = (__bf16)r;
                    p1[e] = a; p2[e] = bb; p3[e] = (__bf16)(r - (float)bb);
                }
                const int c = (4 * wave + j) * 16 + 4 * (lane >> 4);
                char* q = lds + (16 * i + (lane & 15)) * ROWB2 + (c >> 5) * 64 + ((c & 15) >> 2) * 16 + ((c & 31) >> 4) * 8;
                *(bf16x4*)q = p1; *(bf16x4*)(q + PART2) = p2; *(bf16x4*)(q + 2 * PART2) = p3;
            }
    }
    __syncthreads();
    // ---- phase 2: output column tiles 3 wave + j over the K slice [512 s, + 512) of W2 (24 tiles x 48 steps x 3 KiB) ----
    f32x4 acc2[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc2[i][j] = zero4;
    if (phases & 2) phase<3, 16, R2>(w2 + ((int64_t)(3 * wave) * 48 + 16 * s) * 3072, 48 * 3072, lds, ROWB2, PART2, lane, acc2);
    float* o = out + ((int64_t)s * 2560 + by * 32) * 384;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            *(f32x4*)(o + (16 * i + (lane & 15)) * 384 + (3 * wave + j) * 16 + 4 * (lane >> 4)) = acc2[i][j] + acc1[i][j];
}


// variant B: workgroup = (64 rows, 256-column hidden slice): 40 row tiles x 6 slices = 240 workgroups, HALF the weight bytes per workgroup
// (1.18 MB), the same FLOPs; split x tile 64 x 800 B x 3 = 153.6 KB of LDS, hidden slice 64 x 544 B x 3 overlays it
template <int R1, int R2>
__global__ __launch_bounds__(512) void k_probe64(const char* __restrict__ w1, const char* __restrict__ w2, float* __restrict__ out, int phases) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int ROWB1 = 2 * 384 + 32, PART1 = 64 * ROWB1, ROWB2 = 2 * 256 + 32, PART2 = 64 * ROWB2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    b = (b & 7) * 30 + (b >> 3);
    const int s = b / 40, by = b - s * 40;
    for (int i = tid; i < (3 * PART1) / 16; i += 512) ((f32x4*)lds)[i] = (f32x4){1.f + i * 1e-6f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc1[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = zero4;
    if (phases & 1) phase<2, 12, R1, 4>(w1 + (int64_t)((s * 8 + wave) * 2) * 12 * 3072, 12 * 3072, lds, ROWB1, PART1, lane, acc1);
    __syncthreads();
    if (phases & 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x4 p1, p2, p3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = gelu(acc1[i][j][e] + 0.1f);
                    const __bf16 a = (__bf16)v; const float r = v - (float)a; const __bf16 bb = (__bf16)r;
                    p1[e] = a; p2[e] = bb; p3[e] = (__bf16)(r - (float)bb);
                }
                const int c = (2 * wave + j) * 16 + 4 * (lane >> 4);
                char* q = lds + (16 * i + (lane & 15)) * ROWB2 + (c >> 5) * 64 + ((c & 15) >> 2) * 16 + ((c & 31) >> 4) * 8;
                *(bf16x4*)q = p1; *(bf16x4*)(q + PART2) = p2; *(bf16x4*)(q + 2 * PART2) = p3;
            }
    }
    __syncthreads();
    f32x4 acc2[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc2[i][j] = zero4;
    if (phases & 2) phase<3, 8, R2, 4>(w2 + ((int64_t)(3 * wave) * 48 + 8 * s) * 3072, 48 * 3072, lds, ROWB2, PART2, lane, acc2);
    float* o = out + ((int64_t)(s % 3) * 2560 + by * 64) * 384;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            *(f32x4*)(o + (16 * i + (lane & 15)) * 384 + (3 * wave + j) * 16 + 4 * (lane >> 4)) = acc2[i][j] + acc1[i][j % 2];
}
template <int R1, int R2>
void run64(const char* name, const char* w1, const char* w2, float* out, int phases) {
    const size_t ldsb = 3 * 64 * (2 * 384 + 32);
    hipFuncSetAttribute((const void*)k_probe64<R1, R2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_probe64<R1, R2>), dim3(240), dim3(512), ldsb, 0, w1, w2, out, phases);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int n = 50;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_probe64<R1, R2>), dim3(240), dim3(512), ldsb, 0, w1, w2, out, phases);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %7.2f us per launch (back to back)\n", name, ms * 1e3 / n);
}


// variant C: the same products on v_mfma_f32_32x32x16_bf16 (32 weight columns x 32 rows x 16 k per instruction: half the MFMA
// instructions for the same matrix-pipe time).  Per k32 step and 32-column group: 2 k16 halves x 6 products; operands: weights
// 3 parts x 2 halves x NG groups of 16-byte loads (same bytes), activations 3 parts x 2 halves ds_read_b128 (row-major bf16 rows).
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NG, int K32, int R>
__device__ __forceinline__ void phase32(const char* __restrict__ wimg, int64_t grp_stride, const char* lds_a, int rowb, int part, int lane,
                                        f32x16 (&acc)[NG]) {
    // fragment (group g, step kk, half h, part p): 1 KiB at wimg + g * grp_stride + ((kk * 2 + h) * 3 + p) * 1024
    bf16x8 w[R][NG][2][3];
#pragma unroll
    for (int u = 0; u < R - 1; ++u)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int p = 0; p < 3; ++p) w[u][g][h][p] = ldw_plain(wimg + g * grp_stride + ((u * 2 + h) * 3 + p) * 1024 + lane * 16);
    const int aoff = (lane & 31) * rowb + (lane >> 5) * 16;
    bf16x8 x1[2], x2[2], x3[2];   // [k16 half]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const char* q = lds_a + aoff + h * 32;
        x1[h] = *(const bf16x8*)q; x2[h] = *(const bf16x8*)(q + part); x3[h] = *(const bf16x8*)(q + 2 * part);
    }
#pragma unroll
    for (int kk = 0; kk < K32; ++kk) {
        const int u = kk % R, un = (kk + R - 1) % R;
        const bool nx = kk + 1 < K32;
        if (kk + R - 1 < K32) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        w[un][g][h][p] = ldw_plain(wimg + g * grp_stride + (((kk + R - 1) * 2 + h) * 3 + p) * 1024 + lane * 16);
        }
        const char* p0 = lds_a + aoff + (kk + 1) * 64;
        PIN
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[u][g][h][0], x3[h], acc[g], 0, 0, 0);
        PIN
        if (nx) { x3[0] = *(const bf16x8*)(p0 + 2 * part); x3[1] = *(const bf16x8*)(p0 + 32 + 2 * part); }
        PIN
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[u][g][h][1], x2[h], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[u][g][h][0], x2[h], acc[g], 0, 0, 0);
        PIN
        if (nx) { x2[0] = *(const bf16x8*)(p0 + part); x2[1] = *(const bf16x8*)(p0 + 32 + part); }
        PIN
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[u][g][h][2], x1[h], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[u][g][h][1], x1[h], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[u][g][h][0], x1[h], acc[g], 0, 0, 0);
        PIN
        if (nx) { x1[0] = *(const bf16x8*)p0; x1[1] = *(const bf16x8*)(p0 + 32); }
        PIN
    }
}
template <int R1>
__global__ __launch_bounds__(512) void k_probe32(const char* __restrict__ w1, float* __restrict__ out, int phases) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int ROWB1 = 2 * 384 + 16, PART1 = 32 * ROWB1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    b = (b & 7) * 30 + (b >> 3);
    const int s = b / 80, by = b - s * 80;
    for (int i = tid; i < (3 * PART1) / 16; i += 512) ((f32x4*)lds)[i] = (f32x4){1.f + i * 1e-6f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    if (phases & 1) phase32<2, 12, R1>(w1 + (int64_t)((s * 8 + wave) * 2) * 12 * 6144, 12 * 6144, lds, ROWB1, PART1, lane, acc);
    float* o = out + ((int64_t)s * 2560 + by * 32) * 384;
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[g][e];
    o[tid] = sum;
}
template <int R1>
void run32(const char* name, const char* w1, float* out, int phases) {
    const size_t ldsb = 3 * 32 * (2 * 384 + 16);
    hipFuncSetAttribute((const void*)k_probe32<R1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_probe32<R1>), dim3(240), dim3(512), ldsb, 0, w1, out, phases);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int n = 50;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_probe32<R1>), dim3(240), dim3(512), ldsb, 0, w1, out, phases);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %7.2f us per launch (back to back)\n", name, ms * 1e3 / n);
}

template <int R1, int R2, int XCD>
void run(const char* name, const char* w1, const char* w2, float* out, int phases) {
    const size_t ldsb = 3 * 32 * (2 * 512 + 32);
    hipFuncSetAttribute((const void*)k_probe<R1, R2, XCD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_probe<R1, R2, XCD>), dim3(240), dim3(512), ldsb, 0, w1, w2, out, phases);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int n = 50;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_probe<R1, R2, XCD>), dim3(240), dim3(512), ldsb, 0, w1, w2, out, phases);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %7.2f us per launch (back to back)\n", name, ms * 1e3 / n);
}
int main() {
    char *w1, *w2; float* out;
    const size_t b1 = (size_t)96 * 12 * 3072, b2 = (size_t)24 * 48 * 3072;   // 3.5 MB each: the two split images of one block
    hipMalloc(&w1, b1); hipMalloc(&w2, b2); hipMalloc(&out, (size_t)3 * 2560 * 384 * 4);
    hipMemset(w1, 0x3c, b1); hipMemset(w2, 0x3c, b2);
    run<2, 2, 1>("both phases + activation, ring 2 / 2, slices on their own XCDs", w1, w2, out, 7);
    run<2, 2, 0>("both phases + activation, ring 2 / 2, launch order", w1, w2, out, 7);
    run<3, 3, 1>("both phases + activation, ring 3 / 3, slices on their own XCDs", w1, w2, out, 7);
    run<2, 2, 1>("phase 1 only", w1, w2, out, 1);
    run<2, 2, 1>("phase 2 only", w1, w2, out, 2);
    run<2, 2, 1>("activation + re-split only", w1, w2, out, 4);
    run<2, 2, 1>("nothing (fill, barriers, stores)", w1, w2, out, 0);
    run64<2, 2>("B: 64 rows x 256 hidden per workgroup, both phases + activation", w1, w2, out, 7);
    run64<3, 3>("B: ... ring 3 / 3", w1, w2, out, 7);
    run64<2, 2>("B: phase 1 only", w1, w2, out, 1);
    run64<2, 2>("B: phase 2 only", w1, w2, out, 2);
    run64<2, 2>("B: nothing", w1, w2, out, 0);
    run32<2>("C: first product only on v_mfma_f32_32x32x16_bf16", w1, out, 1);
    run32<3>("C: ... ring 3", w1, out, 1);
    run32<2>("C: nothing", w1, out, 0);
    return 0;
}
