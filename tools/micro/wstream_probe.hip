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
                if constexpr (SPREAD == 0 || SPREAD == 2) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) ring[(u + R - 1) % R][j] = *(const f32x4*)(wp[j] + kpf * kst);
                }
                if constexpr (SPREAD != 2) __builtin_amdgcn_sched_barrier(0x6);
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
                if constexpr (SPREAD == 2) {
                    // the scheduler's own interleave for this step: one MFMA, then at most one memory / vector instruction, ...
#pragma unroll
                    for (int q = 0; q < MFMAS * 2 * NTW; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                        if ((q % 8) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a VMEM read every 8th slot
                        __builtin_amdgcn_sched_group_barrier(0x006, 1, 0);   // 1 VALU / SALU
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else
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

// 16 waves per workgroup (1024 threads, 4 per SIMD), 2 column tiles each: the same 32 tiles x 24 k-steps of W1 per workgroup
template <int NW>
__global__ __launch_bounds__(64 * NW) void kmany(const float* __restrict__ W, float* out, unsigned long long* ts, int reps) {
    constexpr int NTW = 32 / NW, K16 = 24, NK = 24, R = 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = xcd_remap(blockIdx.x, gridDim.x), s = lb % 3;
    const float* wp[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) wp[j] = W + ((int64_t)((s * NW + wave) * NTW + j) * K16) * 256 + lane * 4;
    f32x4 acc[2][NTW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float b0 = 1.0f + lane * 1e-6f, b1 = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        f32x4 ring[R][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(wp[j]));
#pragma unroll
        for (int u = 0; u < R - 1; ++u)
#pragma unroll
            for (int j = 0; j < NTW; ++j) ring[u][j] = *(const f32x4*)(wp[j] + u * 256);
        for (int kc = 0; kc < NK; kc += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int kpf = min(kc + u + R - 1, NK - 1);
#pragma unroll
                for (int j = 0; j < NTW; ++j) ring[(u + R - 1) % R][j] = *(const f32x4*)(wp[j] + kpf * 256);
                __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u][j][e], i ? b1 : b0, acc[i][j], 0, 0, 0);
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
    out[(size_t)blockIdx.x * 64 * NW + tid] = sum.x + sum.y + sum.z + sum.w;
    if (lane == 0) atomicMax(ts + blockIdx.x, t1 - t0);
    extern __shared__ float lds_pad[];
    if (reps < 0) out[0] = lds_pad[tid];
}
template <int NW>
void run_many(const char* name, const float* W, float* out, unsigned long long* ts, int wgs, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kmany<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kmany<NW>), dim3(wgs), dim3(64 * NW), 102400, 0, W, out, ts, reps);
    hipDeviceSynchronize();
    hipMemset(ts, 0, 1030 * 8);
    const int L = 20;
    hipEventRecord(e0);
    for (int i = 0; i < L; ++i) hipLaunchKernelGGL((kmany<NW>), dim3(wgs), dim3(64 * NW), 102400, 0, W, out, ts, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), ts, wgs * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += v;
    printf("%-44s %7.2f us/launch  slowest wave of a workgroup: mean %8.0f cyc (MFMA floor 196608)\n", name, ms * 1e3 / L, mean / wgs);
}

// Does MEMORY TRAFFIC serialise with the matrix pipe?  8 waves x 768 MFMAs each per pass (no operand loads: registers), while
//   SRC 0: nothing else happens (the floor);
//   SRC 1: every wave also reads 4 KiB per 32 MFMAs from LDS (ds_read_b128 into registers it then folds into an operand);
//   SRC 2: every wave also DMAs 4 KiB per 32 MFMAs from L2 into LDS (global_load_lds, never read);
//   SRC 3: every wave also loads 4 KiB per 32 MFMAs from L2 into VGPRs and folds them in (= the k-loop of gemm_tile).
template <int SRC>
__global__ __launch_bounds__(512) void kserial(const float* __restrict__ W, float* out, unsigned long long* ts, int reps) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = xcd_remap(blockIdx.x, gridDim.x), s = lb % 3;
    const float* wp = W + ((int64_t)(s * 8 + wave) * 4 * 24) * 256 + lane * 4;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a = (f32x4){1.0f + lane * 1e-6f, 0.5f, 0.25f, 0.75f};
    const float b0 = 0.3f;
    const __amdgpu_buffer_rsrc_t rs_tid = __builtin_amdgcn_make_buffer_rsrc((void*)W, 16, 0x7fffffff, 0x00800000);  // stride 16, ADD_TID_ENABLE (DATA_FORMAT bits = stride[17:14] then: zero)
    const __amdgpu_buffer_rsrc_t rs_lin = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0xffffffffu, 0x00020000);
    float chk = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        asm volatile("" : "+v"(wp));
        for (int kc = 0; kc < 24; ++kc) {
            f32x4 v[4];
            if constexpr (SRC == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *(const f32x4*)(lds + ((wave * 4 + j + kc) & 31) * 256 + lane * 4);
            }
            if constexpr (SRC == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + (j * 24 + kc) * 256),
                                                     (__attribute__((address_space(3))) void*)(lds + ((wave * 4 + j) & 31) * 256 + 8192), 16, 0, 0);
            }
            if constexpr (SRC == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *(const f32x4*)(wp + (j * 24 + kc) * 256);
            }
            if constexpr (SRC == 4 || SRC == 5) {
                // buffer loads WITHOUT an address VGPR: the resource's ADD_TID_ENABLE makes the hardware add lane * stride (16 B);
                // the fragment's base travels in the scalar offset.  SRC 5: the same loads with a VGPR offset (offen) for comparison
                const unsigned sbase = (unsigned)(((s * 8 + wave) * 4 * 24) * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned so = sbase + (unsigned)((j * 24 + kc) * 1024);
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    u32x4_t r;
                    if constexpr (SRC == 4) r = __builtin_amdgcn_raw_buffer_load_b128(rs_tid, 0, __builtin_amdgcn_readfirstlane(so), 0);
                    else r = __builtin_amdgcn_raw_buffer_load_b128(rs_lin, lane * 16, __builtin_amdgcn_readfirstlane(so), 0);
                    v[j] = __builtin_bit_cast(f32x4, r);
                }
            }
            __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x6);
            if constexpr (SRC == 1 || SRC >= 3) {   // the loaded data is "used" (waited for) without vector work of its own
                a = v[0];
                asm volatile("" ::"v"(v[1]), "v"(v[2]), "v"(v[3]));
                if (rep == 0 && kc == 5) chk = v[2].y;   // correctness of the addressing: compared with the plain load below
            }
        }
        if constexpr (SRC == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 sum = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) sum += acc[i];
    out[(size_t)blockIdx.x * 512 + tid] = sum.x + sum.y + sum.z + sum.w + (reps < 0 ? lds[tid] : 0.f);
    if (lane == 0) atomicMax(ts + blockIdx.x, t1 - t0);
    if (SRC >= 3) {   // every lane: did the load at (rep 0, k-step 5, tile 2) return W[...]?
        const float want = (W + ((int64_t)(s * 8 + wave) * 4 * 24) * 256 + lane * 4 + (2 * 24 + 5) * 256)[1];
        if (chk != want) atomicAdd((unsigned*)(ts + 1029), 1u);
    }
}
template <int SRC>
void run_serial(const char* name, const float* W, float* out, unsigned long long* ts, int wgs, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kserial<SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kserial<SRC>), dim3(wgs), dim3(512), 102400, 0, W, out, ts, reps);
    hipDeviceSynchronize();
    hipMemset(ts, 0, 1030 * 8);
    const int L = 20;
    hipEventRecord(e0);
    for (int i = 0; i < L; ++i) hipLaunchKernelGGL((kserial<SRC>), dim3(wgs), dim3(512), 102400, 0, W, out, ts, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), ts, wgs * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += v;
    unsigned long long bad = 0;
    hipMemcpy(&bad, ts + 1029, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.2f us/launch  slowest wave of a workgroup: mean %8.0f cyc (MFMA floor %d)  wrong lanes %llu\n", name, ms * 1e3 / L, mean / wgs, 196608 / 4 * reps, bad);
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 240, reps = argc > 2 ? atoi(argv[2]) : 4;
    float* W; hipMalloc(&W, (size_t)256 * 96 * 1024 + 4096);  // 24 MiB: room for 256 private 96-KiB regions
    std::vector<float> h(96 * 24 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-5f - 0.3f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, (size_t)wgs * 1024 * 4);
    unsigned long long* ts; hipMalloc(&ts, 1030 * 8);
    printf("wgs = %d, reps = %d (each rep = one pass over the workgroup's slice)\n", wgs, reps);
    run<1, false, 4>("W1 (K16 = 24), tile-major, 32 MFMA/step", W, out, ts, wgs, reps);
    run<1, true, 4>("W1 (K16 = 24), k-major,    32 MFMA/step", W, out, ts, wgs, reps);
    run<2, false, 4>("W2 (K16 = 96), tile-major, 24 MFMA/step", W, out, ts, wgs, reps);
    run<2, true, 4>("W2 (K16 = 96), k-major,    24 MFMA/step", W, out, ts, wgs, reps);
    run_serial<0>("MFMA only (operands in registers)", W, out, ts, wgs, reps);
    run_serial<1>("MFMA + 4 KiB per 32 MFMAs per wave LDS -> VGPR", W, out, ts, wgs, reps);
    run_serial<2>("MFMA + 4 KiB per 32 MFMAs per wave L2 -> LDS (DMA)", W, out, ts, wgs, reps);
    run_serial<3>("MFMA + 4 KiB per 32 MFMAs per wave L2 -> VGPR", W, out, ts, wgs, reps);
    run_serial<5>("   ... as buffer loads with a 32-bit VGPR offset", W, out, ts, wgs, reps);
    run_serial<4>("   ... as buffer loads with NO address VGPR (ADD_TID)", W, out, ts, wgs, reps);
    run_many<8>("W1, 8 waves x 4 tiles (2 per SIMD)", W, out, ts, wgs, reps);
    run_many<16>("W1, 16 waves x 2 tiles (4 per SIMD)", W, out, ts, wgs, reps);
    run<1, false, 4, 1, false, 3, 2>("W1 tile-major, sched_group_barrier interleave", W, out, ts, wgs, reps);
    run<2, false, 4, 1, false, 3, 2>("W2 tile-major, sched_group_barrier interleave", W, out, ts, wgs, reps);
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
