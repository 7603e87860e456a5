// Weight-stream probe: does the ORDER of the packed weight image decide how fast 240 workgroups can stream their fragments?
// Mimics k_mlp's two products at B = 256 (240 workgroups x 8 waves, one per CU; wave = NTW column tiles, a ring of 3 k-steps of
// 1-KiB fragment loads, 8 * NTW MFMAs per k-step) with the image in
//   tile-major order  (tile nt, k16 step k) at ((nt * K16) + k) KiB   -- the shipped k_pack_weight layout: at one k-step the
//                     streams of a workgroup's 32 / 24 column tiles sit K16 KiB apart (24 KiB for W1, 96 KiB for W2)
//   k-major order     at ((k * N16) + nt) KiB                          -- one k-step of all tiles is one contiguous run
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/wstream_probe.hip -o tools/micro/bin/wstream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// PHASE 1: W1 (N16 = 96 tiles, K16 = 24): slice s, wave w -> tiles (s * 8 + w) * 4 + j, k = 0..23
// PHASE 2: W2 (N16 = 24 tiles, K16 = 96): wave w -> tiles w * 3 + j, k16 = 32 s + k, k = 0..31
template <int PHASE, bool KMAJOR, int MFMAS, int NKDIV = 1, bool STAMP = false, int RR = 3, int SPREAD = 0>
__global__ __launch_bounds__(512) void k(const float* __restrict__ W, float* out, unsigned long long* ts, int reps) {
    constexpr int NTW = PHASE == 1 ? 4 : 3, N16 = PHASE == 1 ? 96 : 24, K16 = PHASE == 1 ? 24 : 96, NK = (PHASE == 1 ? 24 : 32) / NKDIV, R = RR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = xcd_remap(blockIdx.x, gridDim.x), s = lb % 3;
    const float* wp[NTW];
    int64_t kst;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = PHASE == 1 ? (s * 8 + wave) * 4 + j : wave * 3 + j;
        const int kb = PHASE == 1 ? 0 : 32 * s;
        wp[j] = W + (KMAJOR ? ((int64_t)kb * N16 + nt) : ((int64_t)nt * K16 + kb)) * 256 + lane * 4;
    }
    kst = KMAJOR ? (int64_t)N16 * 256 : 256;
    f32x4 acc[2][NTW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float b0 = 1.0f + lane * 1e-6f, b1 = 0.5f;
    unsigned long long stall = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        f32x4 ring[R][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(wp[j]));  // opaque: the loads of a pass are not loop invariant
#pragma unroll
        for (int u = 0; u < R - 1; ++u)
#pragma unroll
            for (int j = 0; j < NTW; ++j) ring[u][j] = *(const f32x4*)(wp[j] + u * kst);
        for (int kc = 0; kc + R <= NK + (R - NK % R) % R; kc += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int kpf = min(kc + u + R - 1, NK - 1);
                if constexpr (SPREAD == 0) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) ring[(u + R - 1) % R][j] = *(const f32x4*)(wp[j] + kpf * kst);
                }
                __builtin_amdgcn_sched_barrier(0x6);
                if constexpr (STAMP) {  // how long does the wave sit at the wait for the fragments of THIS step (requested R - 1 steps ago)?
                    const unsigned long long ta = __builtin_readcyclecounter();
                    if constexpr (R == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NTW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 1) * NTW) : "memory");
                    const unsigned long long tb = __builtin_readcyclecounter();
                    stall += tb - ta;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kc + u < NK) {
#pragma unroll
                    for (int e = 0; e < MFMAS; ++e) {
                        if constexpr (SPREAD == 1) {  // one fragment request in front of each quarter of the step's MFMAs
                            if (e < NTW) {
                                ring[(u + R - 1) % R][e] = *(const f32x4*)(wp[e] + kpf * kst);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < NTW; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u][j][e & 3], i ? b1 : b0, acc[i][j], 0, 0, 0);
                        if constexpr (SPREAD == 1) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (MFMAS == 0) {
#pragma unroll
                        for (int j = 0; j < NTW; ++j) acc[0][j] += ring[u][j];
                    }
                }
                __builtin_amdgcn_sched_barrier(0x6);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) sum += acc[i][j];
    out[(size_t)blockIdx.x * 512 + tid] = sum.x + sum.y + sum.z + sum.w;
    if (tid == 0) ts[blockIdx.x] = t1 - t0;
    if (tid == 256) ts[512 + blockIdx.x] = STAMP ? stall : t1 - t0;  // wave 4: the SIMD partner of wave 0 (STAMP: its cycles at the fragment wait)
    if (STAMP && tid == 0) ts[blockIdx.x] = stall;
    extern __shared__ float lds_pad[];                // 100 KiB of dynamic LDS: ONE workgroup per CU, as in k_mlp
    if (reps < 0) out[0] = lds_pad[tid];
}

template <int PHASE, bool KMAJOR, int MFMAS, int NKDIV = 1, bool STAMP = false, int RR = 3, int SPREAD = 0>
void run(const char* name, const float* W, float* out, unsigned long long* ts, int wgs, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<PHASE, KMAJOR, MFMAS, NKDIV, STAMP, RR, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PHASE, KMAJOR, MFMAS, NKDIV, STAMP, RR, SPREAD>), dim3(wgs), dim3(512), 102400, 0, W, out, ts, reps);
    hipDeviceSynchronize();
    const int L = 20;
    hipEventRecord(e0);
    for (int i = 0; i < L; ++i) hipLaunchKernelGGL((k<PHASE, KMAJOR, MFMAS, NKDIV, STAMP, RR, SPREAD>), dim3(wgs), dim3(512), 102400, 0, W, out, ts, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs), h4(wgs);
    hipMemcpy(h.data(), ts, wgs * 8, hipMemcpyDeviceToHost);
    hipMemcpy(h4.data(), ts + 512, wgs * 8, hipMemcpyDeviceToHost);
    double mean = 0, mean4 = 0; unsigned long long mx = 0;
    for (auto v : h) { mean += v; mx = v > mx ? v : mx; }
    for (auto v : h4) { mean4 += v; mx = v > mx ? v : mx; }
    mean /= wgs; mean4 /= wgs;
    const int NK = (PHASE == 1 ? 24 : 32) / NKDIV, NTW = PHASE == 1 ? 4 : 3;
    const double bytes = (double)wgs * 8 * NTW * NK * 1024.0 * reps, floor_cyc = (double)NK * MFMAS * 2 * NTW * 32 * 2 * reps;
    printf("%-44s %7.2f us/launch  wave 0 mean %8.0f wave 4 mean %8.0f max %8llu cyc (MFMA floor %7.0f)  %6.2f TB/s L2->CU  %5.1f B/clk/CU\n", name,
           ms * 1e3 / L, mean, mean4, mx, floor_cyc, bytes / (ms * 1e-3 / L) / 1e12, bytes / wgs / mean4);
}


// MODE 0: every workgroup streams the SAME 768 KiB (its slice of W1, as k_mlp's first product: 10 row-tile workgroups per XCD
//         read identical lines at about the same time); MODE 1: every workgroup streams a PRIVATE 96 KiB region over and over
//         (30 x 96 KiB per XCD: L2 resident, no two CUs ever ask for the same line).  DMA: global_load_lds_dwordx4 into a 64 KiB
//         LDS ring (never read) instead of VGPR loads.  Loads only; reports bytes / clk / CU.
template <int MODE, bool DMA>
__global__ __launch_bounds__(512) void kstream(const float* __restrict__ W, float* out, unsigned long long* ts, int reps) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = xcd_remap(blockIdx.x, gridDim.x), s = lb % 3;
    // 768 fragments of 1 KiB per pass and workgroup, 96 per wave
    const float* base = MODE == 0 ? W + (int64_t)s * 768 * 256 : W + (int64_t)lb * 96 * 256;
    const int nfrag = MODE == 0 ? 768 : 96, passes = MODE == 0 ? 1 : 8;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps * passes; ++rep) {
        asm volatile("" : "+v"(base));
        for (int f = wave; f < nfrag; f += 64) {  // 8 fragments per wave in flight
            if constexpr (DMA) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int fi = min(f + 8 * u, nfrag - 1);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (int64_t)fi * 256 + lane * 4),
                                                     (__attribute__((address_space(3))) void*)(lds + ((wave * 8 + u) & 63) * 256), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(base + (int64_t)min(f + 8 * u, nfrag - 1) * 256 + lane * 4);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (DMA) acc.x += lds[tid];
    out[(size_t)blockIdx.x * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
    if (tid == 0) ts[blockIdx.x] = t1 - t0;
}
template <int MODE, bool DMA>
void run_stream(const char* name, const float* W, float* out, unsigned long long* ts, int wgs, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kstream<MODE, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kstream<MODE, DMA>), dim3(wgs), dim3(512), 102400, 0, W, out, ts, reps);
    hipDeviceSynchronize();
    const int L = 20;
    hipEventRecord(e0);
    for (int i = 0; i < L; ++i) hipLaunchKernelGGL((kstream<MODE, DMA>), dim3(wgs), dim3(512), 102400, 0, W, out, ts, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), ts, wgs * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += v;
    mean /= wgs;
    const double bytes = (double)wgs * 768 * 1024.0 * reps;
    printf("%-52s %7.2f us/launch  in-kernel mean %8.0f cyc  %6.2f TB/s L2->CU  %5.1f B/clk/CU\n", name, ms * 1e3 / L, mean,
           bytes / (ms * 1e-3 / L) / 1e12, bytes / wgs / mean);
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 240, reps = argc > 2 ? atoi(argv[2]) : 4;
    float* W; hipMalloc(&W, (size_t)256 * 96 * 1024 + 4096);  // 24 MiB: room for 256 private 96-KiB regions
    std::vector<float> h(96 * 24 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-5f - 0.3f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, (size_t)wgs * 512 * 4);
    unsigned long long* ts; hipMalloc(&ts, 1030 * 8);
    printf("wgs = %d, reps = %d (each rep = one pass over the workgroup's slice)\n", wgs, reps);
    run<1, false, 4>("W1 (K16 = 24), tile-major, 32 MFMA/step", W, out, ts, wgs, reps);
    run<1, true, 4>("W1 (K16 = 24), k-major,    32 MFMA/step", W, out, ts, wgs, reps);
    run<2, false, 4>("W2 (K16 = 96), tile-major, 24 MFMA/step", W, out, ts, wgs, reps);
    run<2, true, 4>("W2 (K16 = 96), k-major,    24 MFMA/step", W, out, ts, wgs, reps);
    run<1, false, 4, 1, false, 3, 1>("W1 tile-major, requests SPREAD over the step", W, out, ts, wgs, reps);
    run<2, false, 4, 1, false, 3, 1>("W2 tile-major, requests spread over the step", W, out, ts, wgs, reps);
    run<1, true, 4, 1, false, 3, 1>("W1 k-major, requests spread over the step", W, out, ts, wgs, reps);
    run<1, false, 4, 1, true>("W1 tile-major: CYCLES AT THE FRAGMENT WAIT (w0, w4)", W, out, ts, wgs, reps);
    run<2, false, 4, 1, true>("W2 tile-major: cycles at the fragment wait", W, out, ts, wgs, reps);
    run<1, false, 4, 1, false, 5>("W1 tile-major, ring of 5", W, out, ts, wgs, reps);
    run<1, false, 4, 1, true, 5>("W1 tile-major, ring of 5: cycles at the wait", W, out, ts, wgs, reps);
    run<1, false, 8, 2>("W1 tile-major, HALF the bytes (64 MFMA/step)", W, out, ts, wgs, reps);
    run<1, false, 16, 4>("W1 tile-major, QUARTER the bytes (128/step)", W, out, ts, wgs, reps);
    run<1, false, 96, 24>("W1 tile-major, 1/24 of the bytes", W, out, ts, wgs, reps);
    run<1, false, 0>("W1, tile-major, loads only", W, out, ts, wgs, reps);
    run<1, true, 0>("W1, k-major,    loads only", W, out, ts, wgs, reps);
    run<2, false, 0>("W2, tile-major, loads only", W, out, ts, wgs, reps);
    run<2, true, 0>("W2, k-major,    loads only", W, out, ts, wgs, reps);
    run_stream<0, false>("stream, shared slice (768 KiB x 3), VGPR loads", W, out, ts, wgs, reps);
    run_stream<1, false>("stream, private 96 KiB per workgroup, VGPR loads", W, out, ts, wgs, reps);
    run_stream<0, true>("stream, shared slice, global_load_lds", W, out, ts, wgs, reps);
    run_stream<1, true>("stream, private 96 KiB, global_load_lds", W, out, ts, wgs, reps);
    return 0;
}
