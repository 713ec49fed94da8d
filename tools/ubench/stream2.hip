// What request PATTERN gets the most bytes per clock out of the L2 -> CU path while the matrix cores work?  (r3's wxw.hip saw
// 87.6 B/clk/CU as a pure stream from 8 waves and 69.6 with MFMAs - bursts of 5 compiler-scheduled loads into VGPRs - while the
// product's stream (stream_r8.h: one asm request per tile, AGPR destination, scalar base + lane offset) delivers ~50.)
// One kernel, every combination: NW waves per workgroup (4 = one per SIMD, 8 = two), ring of RD 1-KiB tiles per wave, requests
// issued in bursts of B after B tiles were multiplied (4 RB v_mfma_f32_4x4x1 per tile, A from LDS), destination registers
// "v" or "a", address = scalar base + lane offset or a 64-bit lane address, wave streams apart (each wave its own contiguous
// region) or interleaved tile by tile, 0 or 2 workgroup barriers per stage of NT tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int N> struct IC { static constexpr int value = N; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int AREG, int VADDR>
__device__ __forceinline__ void req(f32x4& dst, unsigned voff, const char* sbase, const char* vaddr) {
    if constexpr (VADDR) {
        if constexpr (AREG) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(dst) : "v"(vaddr));
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(vaddr));
    } else {
        if constexpr (AREG) asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(dst) : "v"(voff), "s"(sbase));
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
    }
}
template <int AREG, int N> __device__ __forceinline__ void wait_for(f32x4& r) {
    if constexpr (AREG) asm volatile("s_waitcnt vmcnt(%1)" : "+a"(r) : "n"(N));
    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N));
}

template <int NW, int RD, int B, int RB, int AREG, int VADDR, int ILV, int SYNC, int NT, int LAYOUT>
__global__ __launch_bounds__(64 * NW) void k_stream2(const char* __restrict__ src, size_t wave_off, int n_stages,
                                                      float* __restrict__ sink, long long* __restrict__ cycles) {
    static_assert(RD % B == 0 && NT % RD == 0 && RD <= 63, "static slots");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WS = 4 * NT + 4;
    constexpr size_t tile_stride = LAYOUT ? (size_t)NW * 1024 : 1024;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int e = tid; e < 8 * WS; e += 64 * NW) lds[e] = 0.001f * (float)(e % 97);
    __syncthreads();
    const char* sb = src + (size_t)wave * wave_off;                 // tile t of this wave: sb + t * tile_stride (+ lane * 16)
    const unsigned voff = lane * 16;
    f32x4 ring[RD];
    static_for<0, RD>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        req<AREG, VADDR>(ring[i], voff, sb + (size_t)i * tile_stride, sb + (size_t)i * tile_stride + voff);
    });
    sb += (size_t)RD * tile_stride;
    f32x4 acc[4][RB];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[k][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ap = lds + (lane & 3) * WS;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < n_stages; ++st) {
        float4 an[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ap + rb * 4 * WS);
        static_for<0, NT / B>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            static_for<0, B>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int t = g * B + j, slot = t % RD;
                wait_for<AREG, ILV ? RD - B - 1 : RD - 1 - j>(ring[slot]);
                __builtin_amdgcn_sched_barrier(0);
                float4 a[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a[rb] = an[rb];
                if constexpr (t + 1 < NT) {                           // the next tile's A operand is read behind this tile's MFMAs
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ap + rb * 4 * WS + 4 * (t + 1));
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[0][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].x, ring[slot].x, acc[0][rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[1][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].y, ring[slot].y, acc[1][rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[2][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].z, ring[slot].z, acc[2][rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[3][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].w, ring[slot].w, acc[3][rb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ILV) {                                  // one request per tile, for the slot used B tiles ago:
                    constexpr int ps = (t + RD - B) % RD;             // the ring runs RD - B tiles ahead, RD - B - 1 requests are newer
                    req<AREG, VADDR>(ring[ps], voff, sb + (size_t)(t - B) * tile_stride, sb + (size_t)(t - B) * tile_stride + voff);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if constexpr (!ILV) {
                static_for<0, B>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int t = g * B + j, slot = t % RD;
                    req<AREG, VADDR>(ring[slot], voff, sb + (size_t)t * tile_stride, sb + (size_t)t * tile_stride + voff);
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        sb += (size_t)NT * tile_stride;
        if constexpr (SYNC == 2) { lds_barrier(); lds_barrier(); }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    static_for<0, RD>([&](auto ic) { constexpr int i = decltype(ic)::value; wait_for<AREG, 0>(ring[i]); s += ring[i].x; });
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) s += acc[k][rb][0] + acc[k][rb][1] + acc[k][rb][2] + acc[k][rb][3];
    sink[blockIdx.x * 64 * NW + tid] = s;
    if (lane == 0) cycles[blockIdx.x * NW + wave] = t1 - t0;
}

// LAYOUT 0: every wave its own contiguous region; 1: the waves' tiles interleaved (tile t of wave w at (t NW + w) KiB)
template <int NW, int RD, int B, int RB, int AREG, int VADDR, int ILV, int SYNC, int NT, int LAYOUT>
static void run(const char* src, size_t region, int n_wg, float* sink, long long* cyc) {
    const size_t wave_bytes = region / NW;
    const int n_stages = (int)(wave_bytes / 1024 / NT) - 1;
    auto kern = k_stream2<NW, RD, B, RB, AREG, VADDR, ILV, SYNC, NT, LAYOUT>;
    const size_t lds = (size_t)8 * (4 * NT + 4) * 4;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t wave_off = LAYOUT ? 1024 : wave_bytes;
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * NW), lds, 0, src, wave_off, n_stages, sink, cyc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("failed: %s\n", hipGetErrorString(e)); return; }
    std::vector<long long> h((size_t)n_wg * NW);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int g = 0; g < n_wg; ++g) {
        long long mx = 0;
        for (int w = 0; w < NW; ++w) mx = h[(size_t)g * NW + w] > mx ? h[(size_t)g * NW + w] : mx;
        mean += (double)mx;
    }
    mean /= n_wg;
    const double tiles_simd = (double)n_stages * NT * (NW / 4);
    if (getenv("PER_WAVE")) {            // round 5: is the L2 -> CU service of the waves of a workgroup systematically uneven?
        printf("   per-wave cycles / 1000 (mean over workgroups; min .. max over workgroups):");
        for (int w = 0; w < NW; ++w) {
            double m = 0; long long lo = h[w], hi = h[w];
            for (int g = 0; g < n_wg; ++g) { const long long v = h[(size_t)g * NW + w]; m += (double)v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
            printf("  w%d %.0f (%.0f..%.0f)", w, m / n_wg / 1e3, lo / 1e3, hi / 1e3);
        }
        printf("\n");
    }
    printf("NW=%d RD=%2d burst=%d%s RB=%d dst=%s addr=%s layout=%s sync=%d NT=%3d: %6.1f cycles per tile and SIMD  %5.1f B/clk/CU  (MFMA floor %d)\n",
           NW, RD, B, ILV ? " (1/tile)" : "", RB, AREG ? "a" : "v", VADDR ? "vaddr" : "saddr", LAYOUT ? "interleaved" : "apart", SYNC, NT,
           mean / tiles_simd, tiles_simd * 4 * 1024 / mean, 32 * RB);
    fflush(stdout);
}

int main() {
    const size_t region = 10u << 20;
    char* src; float* sink; long long* cyc;
    (void)hipMalloc((void**)&src, region + (4u << 20)); (void)hipMemset(src, 0, region + (4u << 20));
    (void)hipMalloc((void**)&sink, 1024 * 512 * 4); (void)hipMalloc((void**)&cyc, 1024 * 8 * 8);
    const int W = 256;
    //  NW RD  B RB  A  VA ILV SYNC NT LAYOUT
    printf("-- the product's pattern and its neighbours (4 waves, 4 chains = RB 1)\n");
    run<4, 32, 1, 1, 1, 0, 0, 0, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 0, 0, 0, 0, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 1, 0, 0, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 0, 0, 0, 96, 1>(src, region, W, sink, cyc);
    run<4, 32, 4, 1, 1, 0, 0, 0, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 8, 1, 1, 0, 0, 0, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 8, 1, 0, 1, 0, 0, 96, 1>(src, region, W, sink, cyc);
    run<4, 32, 8, 1, 1, 0, 1, 0, 96, 0>(src, region, W, sink, cyc);
    run<4, 48, 8, 1, 1, 0, 0, 0, 96, 0>(src, region, W, sink, cyc);
    printf("-- 8 waves (two per SIMD), 4 chains\n");
    run<8, 24, 1, 1, 1, 0, 0, 0, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 0, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 1, 0, 0, 0, 48, 1>(src, region, W, sink, cyc);
    run<8, 24, 4, 1, 1, 0, 0, 0, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 4, 1, 1, 0, 0, 0, 48, 1>(src, region, W, sink, cyc);
    run<8, 24, 4, 1, 0, 1, 0, 0, 48, 1>(src, region, W, sink, cyc);
    run<8, 24, 8, 1, 1, 0, 0, 0, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 8, 1, 1, 0, 0, 0, 48, 1>(src, region, W, sink, cyc);
    run<8, 24, 8, 1, 1, 0, 0, 2, 48, 1>(src, region, W, sink, cyc);
    run<8, 24, 8, 1, 1, 0, 0, 2, 48, 0>(src, region, W, sink, cyc);
    run<8, 16, 8, 1, 1, 0, 0, 0, 48, 1>(src, region, W, sink, cyc);
    printf("-- round 5: the free-running winners WITH two workgroup barriers per stage (a product stage ends in a partial-sum exchange)\n");
    run<4, 32, 1, 1, 1, 0, 0, 2, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 0, 0, 0, 2, 96, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 1, 0, 0, 2, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 2, 48, 0>(src, region, W, sink, cyc);
    run<8, 16, 1, 1, 0, 0, 0, 2, 48, 0>(src, region, W, sink, cyc);
    run<8, 12, 1, 1, 0, 0, 0, 2, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 2, 24, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 0, 48, 0>(src, region, W, sink, cyc);
    printf("-- 8 chains (RB 2)\n");
    run<4, 32, 1, 2, 1, 0, 0, 0, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 8, 2, 1, 0, 0, 0, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 2, 1, 0, 0, 0, 96, 1>(src, region, W, sink, cyc);
    run<8, 24, 1, 2, 1, 0, 0, 0, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 8, 2, 1, 0, 0, 0, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 8, 2, 1, 0, 0, 0, 48, 1>(src, region, W, sink, cyc);
    run<8, 24, 8, 2, 1, 0, 0, 2, 48, 1>(src, region, W, sink, cyc);
    return 0;
}
