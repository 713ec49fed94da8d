// Micro-benchmark behind spline_r8.h and the question "what would an N-split / continuous-stream version of the 4-chain
// RealNVP kernel buy": every wave of a workgroup streams ITS OWN 1-KiB weight tiles L2 -> AGPR through a ring of RD tiles
// filled by inline-asm loads with hand-counted s_waitcnt (the ring is NEVER drained: no compiler-inserted vmcnt(0) at loop
// back edges or barriers), multiplies every tile into RB row blocks of 4 chains (4 RB v_mfma_f32_4x4x1 per tile, k mod 4 on
// separate accumulators) and meets the other waves at one LDS-only barrier per stage of NQS tiles (an epilogue writes 4 RB
// values per lane to LDS first).  NW waves per workgroup: 4 = one per SIMD, 5 / 8 / 10 = up to 3 per SIMD.
// Prints cycles per KiB-tile of one wave, B/clk per CU, and cycles per "20 KiB" (the unit of ubench_wxw.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int N> struct IC { static constexpr int value = N; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int IMM> __device__ __forceinline__ void aload(f32x4& d, unsigned voff, const float4* sb) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(d) : "v"(voff), "s"(sb), "n"(IMM));
}
template <int N> __device__ __forceinline__ void await(f32x4& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+a"(r) : "n"(N)); }

// MODE 0: stream + MFMA, 1: pure stream (one v_add per tile), 2: MFMA only (no refills)
template <int NW, int RD, int RB, int MODE, int NQS, int BAR>
__global__ __launch_bounds__(64 * NW) void k_ns(const float4* __restrict__ src, int n_stages, size_t wave_stride,
                                                float* __restrict__ sink, long long* __restrict__ cycles) {
    static_assert(NQS % RD == 0, "static slots");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WS = 4 * NQS + 4;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, arow = lane & 3;
    for (int e = tid; e < RB * 4 * WS; e += 64 * NW) lds[e] = 0.001f * (float)(e % 97);
    float* out = lds + RB * 4 * WS;
    __syncthreads();
    const float4* sp = src + (size_t)wave * wave_stride;
    unsigned voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) voff[j] = (unsigned)(lane * 16 + 4096 * j);
    f32x4 ring[RD];
    static_for<0, RD - 1>([&](auto dc) { constexpr int d = decltype(dc)::value; aload<(d % 4) * 1024>(ring[d], voff[(d / 4) % 8], sp + (size_t)(d / 32) * 2048); });
    sp += (size_t)(RD - 1) * 64;
    f32x4 acc[4][RB];
    float tot = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < n_stages; ++st) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[k][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* ap = lds + arow * WS;
        float4 an[RB];                                   // A operand of the next tile: one step ahead, as in spline_r8.h
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ap + rb * 4 * WS);
#pragma unroll 1
        for (int q0 = 0; q0 < NQS; q0 += RD) {
            static_for<0, RD>([&](auto dc) {
                constexpr int d = decltype(dc)::value;
                if constexpr (MODE != 2) await<RD - 2>(ring[d]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == 1) {
                    acc[0][0] += ring[d];
                } else {
                    float4 a[RB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a[rb] = an[rb];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ap + rb * 4 * WS + 4 * ((d + 1) % RD));
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[0][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].x, ring[d].x, acc[0][rb], 0, 0, 0);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[1][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].y, ring[d].y, acc[1][rb], 0, 0, 0);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[2][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].z, ring[d].z, acc[2][rb], 0, 0, 0);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[3][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].w, ring[d].w, acc[3][rb], 0, 0, 0);
                }
                if constexpr (MODE != 2) {
                    constexpr int dp = (d + RD - 1) % RD;
                    aload<(d % 4) * 1024>(ring[dp], voff[(d / 4) % 8], sp + (size_t)(d / 32) * 2048);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            sp += (size_t)RD * 64;
            ap += 4 * RD;
        }
        if constexpr (BAR) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const f32x4 o = (acc[0][rb] + acc[1][rb]) + (acc[2][rb] + acc[3][rb]);
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(4 * rb + r) * (64 * NW + 4) + 64 * wave + lane] = o[r] > 0.f ? o[r] : 0.f;
            }
            lds_barrier();
        } else {
            tot += acc[0][0][0] + acc[1][0][1] + acc[2][0][2] + acc[3][0][3];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    static_for<0, RD>([&](auto dc) { await<0>(ring[decltype(dc)::value]); tot += ring[decltype(dc)::value].x; });
    sink[blockIdx.x * 64 * NW + tid] = tot + out[tid];
    if (lane == 0) cycles[blockIdx.x * NW + wave] = t1 - t0;             // per wave: the workgroup's time is its slowest wave's
}

template <int NW, int RD, int RB, int MODE, int NQS, int BAR>
static void run(const char* name, const float4* src, size_t region_bytes, int n_wg, float* sink, long long* cyc) {
    const size_t wave_bytes = region_bytes / NW;
    const int n_stages = (int)(wave_bytes / 1024 / NQS) - 2;
    const size_t lds = (size_t)(RB * 4 * (4 * NQS + 4) + RB * 4 * (64 * NW + 4) + 64 * NW) * 4;
    auto kern = k_ns<NW, RD, RB, MODE, NQS, BAR>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * NW), lds, 0, src, n_stages, wave_bytes / 16, sink, cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<long long> h((size_t)n_wg * NW);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0, mean0 = 0;                          // mean over workgroups of the SLOWEST wave / of wave 0
    for (int g = 0; g < n_wg; ++g) {
        long long mx = 0;
        for (int w = 0; w < NW; ++w) mx = h[(size_t)g * NW + w] > mx ? h[(size_t)g * NW + w] : mx;
        mean += (double)mx; mean0 += (double)h[(size_t)g * NW];
    }
    mean /= n_wg; mean0 /= n_wg;
    const double tiles = (double)n_stages * NQS;                       // per wave
    printf("%-30s NW=%2d RD=%2d RB=%d NQS=%3d bar=%d %3d WGs: %6.1f cycles per tile and wave  %5.1f B/clk/CU  %6.1f cycles per 20 KiB  (by wave 0's clock alone: %5.1f B/clk/CU)\n",
           name, NW, RD, RB, NQS, BAR, n_wg, mean / tiles, tiles * NW * 1024.0 / mean, mean / (tiles * NW) * 20.0,
           tiles * NW * 1024.0 / mean0);
}

int main() {
    const size_t region = 10u << 20;
    float4* src; float* sink; long long* cyc;
    (void)hipMalloc((void**)&src, region + (2u << 20)); (void)hipMemset(src, 0, region + (2u << 20));
    (void)hipMalloc((void**)&sink, 1024 * 1024 * 4); (void)hipMalloc((void**)&cyc, 1024 * 16 * 8);
    const int n_wg = 256;
    run<4, 32, 1, 1, 64, 0>("pure stream", src, region, n_wg, sink, cyc);
    run<4, 32, 1, 1, 64, 1>("pure stream + barrier", src, region, n_wg, sink, cyc);
    run<4, 16, 1, 1, 64, 1>("pure stream + barrier", src, region, n_wg, sink, cyc);
    run<8, 32, 1, 1, 64, 0>("pure stream", src, region, n_wg, sink, cyc);
    run<8, 32, 1, 1, 64, 1>("pure stream + barrier", src, region, n_wg, sink, cyc);
    run<8, 16, 1, 1, 64, 1>("pure stream + barrier", src, region, n_wg, sink, cyc);
    run<4, 32, 1, 2, 64, 0>("MFMA only", src, region, n_wg, sink, cyc);
    run<8, 32, 1, 2, 64, 0>("MFMA only", src, region, n_wg, sink, cyc);
    run<4, 32, 2, 2, 64, 0>("MFMA only", src, region, n_wg, sink, cyc);
    run<8, 32, 2, 2, 64, 0>("MFMA only", src, region, n_wg, sink, cyc);
    run<4, 32, 1, 0, 64, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<5, 32, 1, 0, 64, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<8, 32, 1, 0, 64, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<8, 16, 1, 0, 64, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<10, 16, 1, 0, 48, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<10, 16, 1, 0, 32, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<12, 16, 1, 0, 32, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<16, 16, 1, 0, 32, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<4, 32, 2, 0, 64, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<8, 32, 2, 0, 64, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<8, 16, 2, 0, 64, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<8, 16, 2, 0, 32, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    run<16, 16, 2, 0, 32, 1>("stream + MFMA + barrier", src, region, n_wg, sink, cyc);
    return 0;
}
