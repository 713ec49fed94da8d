// Micro-benchmark of the W x W stage of the 4-chain-tile flow kernels (flow_r4.h): every workgroup (4 waves, one per
// SIMD) streams the SAME weight image L2 -> VGPR through a ring of RD items (G tiles of 1 KiB per wave and item) and
// multiplies every item into RB row blocks of 4 chains on v_mfma_f32_4x4x1_16b (4 G RB MFMAs per item and wave).
// Question: how many cycles per item does the stream + MFMA mix cost under different issue orders, ring depths and
// row-block counts, against the pure stream of the same bytes?   Prints cycles per item and bytes / clock / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma44(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N> struct IC { static constexpr int value = N; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}

// ILV 0: the item's MFMAs, then its slot is refilled (flow_r4.h today).  1: the slot of the PREVIOUS item is refilled one
// tile behind every 4 RB MFMAs of this item (sched_group_barrier pipeline).  2: no MFMAs at all (pure stream through
// the same ring, each tile touched by one v_add).  3: MFMAs only (the ring is loaded once and never refilled).
// SYNC 1: after every NQ items the partial products go through LDS with two LDS-only barriers (a stage boundary).
template <int G, int RD, int RB, int ILV, int SYNC, int NQ>
__global__ __launch_bounds__(256) void k_wxw(const float4* __restrict__ src, int n_stages, float* __restrict__ sink,
                                             long long* __restrict__ cycles) {
    static_assert(NQ % RD == 0, "slot indices must be compile-time constants");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WS = 64 * G + 4;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, arow = lane & 3;
    for (int e = tid; e < RB * 4 * WS; e += 256) lds[e] = 0.001f * (float)(e % 97);
    float* part = lds + RB * 4 * WS;
    __syncthreads();
    constexpr int IS = 4 * G * 64;                                     // float4 between consecutive items of one wave
    const float4* sp = src + ((size_t)wave * G) * 64 + lane;
    float4 ring[RD][G];
#pragma unroll
    for (int i = 0; i < RD; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) ring[i][g] = sp[(size_t)i * IS + g * 64];
    f32x4 acc[RB][G];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[rb][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < n_stages; ++st) {
        const float* a0 = lds + arow * WS + 4 * NQ / 4 * 0;
        static_for<0, NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int slot = q % RD, prev = (q + RD - 1) % RD;
            if constexpr (ILV == 2) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc[0][g] += (f32x4){ring[slot][g].x, ring[slot][g].y, ring[slot][g].z, ring[slot][g].w};
                    ring[slot][g] = sp[(size_t)(q + RD) * IS + g * 64];
                }
            } else {
                float4 a[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const float4*>(a0 + rb * 4 * WS + 4 * (q % 16));
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[rb][g] = mfma44(a[rb].x, ring[slot][g].x, acc[rb][g]);
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[rb][g] = mfma44(a[rb].y, ring[slot][g].y, acc[rb][g]);
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[rb][g] = mfma44(a[rb].z, ring[slot][g].z, acc[rb][g]);
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[rb][g] = mfma44(a[rb].w, ring[slot][g].w, acc[rb][g]);
                }
                if constexpr (ILV == 0) {
#pragma unroll
                    for (int g = 0; g < G; ++g) ring[slot][g] = sp[(size_t)(q + RD) * IS + g * 64];
                } else if constexpr (ILV == 1) {
#pragma unroll
                    for (int g = 0; g < G; ++g) ring[prev][g] = sp[(size_t)(q + RD - 1) * IS + g * 64];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4 * RB, 0);      // 4 RB MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // one VMEM read
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        sp += (size_t)NQ * IS;
        if constexpr (SYNC == 1) {
            float* pw = part + (size_t)wave * 4 * 64 * G + lane;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[r * 64 * G + 64 * g] = acc[0][g][r];
            lds_barrier();
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const float* p = part + (256 * i + tid);
                v += (p[0] + p[4 * 64 * G]) + (p[2 * 4 * 64 * G] + p[3 * 4 * 64 * G]);
            }
            lds[(tid % (4 * WS))] = v * 1e-9f;
            lds_barrier();
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < G; ++g) s += acc[rb][g][0] + acc[rb][g][1] + acc[rb][g][2] + acc[rb][g][3];
#pragma unroll
    for (int i = 0; i < RD; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) s += ring[i][g].x;
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int G, int RD, int RB, int ILV, int SYNC, int NQ>
static void run(const char* name, const float4* src, size_t region_bytes, int n_wg, float* sink, long long* cyc) {
    const size_t item_round = (size_t)4 * G * 1024;                         // bytes per item of all four waves
    const int n_stages = (int)(region_bytes / item_round / NQ) - 1;          // (the ring runs RD items ahead)
    const size_t lds = (size_t)(RB * 4 * (64 * G + 4) + 4 * 4 * 64 * G) * 4;
    auto kern = k_wxw<G, RD, RB, ILV, SYNC, NQ>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(n_wg), dim3(256), lds, 0, src, n_stages, sink, cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<long long> h(n_wg);
    (void)hipMemcpy(h.data(), cyc, n_wg * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= n_wg;
    const double items = (double)n_stages * NQ;
    printf("%-34s G=%d RD=%2d RB=%d NQ=%2d %3d WGs: %7.1f cycles/item  %5.1f B/clk/CU  (MFMA floor %d cycles/item)\n", name, G, RD,
           RB, NQ, n_wg, mean / items, item_round * items / mean, ILV == 2 ? 0 : 32 * G * RB);
}

int main() {
    const size_t region = 10u << 20;
    float4* src; float* sink; long long* cyc;
    (void)hipMalloc((void**)&src, region + (1u << 20)); (void)hipMemset(src, 0, region + (1u << 20));
    (void)hipMalloc((void**)&sink, 1024 * 256 * 4); (void)hipMalloc((void**)&cyc, 1024 * 8);
    for (int n_wg : {1, 256}) {
        run<5, 6, 1, 2, 0, 24>("pure stream", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 3, 0, 24>("MFMA only", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 0, 0, 24>("MFMAs then refill", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 1, 0, 24>("refill interleaved", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 0, 1, 24>("MFMAs then refill + stage sync", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 1, 1, 24>("interleaved + stage sync", src, region, n_wg, sink, cyc);
        run<5, 8, 1, 1, 0, 24>("interleaved, ring 8", src, region, n_wg, sink, cyc);
        run<5, 12, 1, 1, 0, 24>("interleaved, ring 12", src, region, n_wg, sink, cyc);
        run<5, 6, 2, 3, 0, 24>("MFMA only, 2 row blocks", src, region, n_wg, sink, cyc);
        run<5, 6, 2, 0, 0, 24>("MFMAs then refill, 2 row blocks", src, region, n_wg, sink, cyc);
        run<5, 6, 2, 1, 0, 24>("interleaved, 2 row blocks", src, region, n_wg, sink, cyc);
        run<5, 6, 2, 1, 1, 24>("interleaved, 2 row blocks + sync", src, region, n_wg, sink, cyc);
        run<5, 8, 2, 1, 0, 24>("interleaved, 2 row blocks, ring 8", src, region, n_wg, sink, cyc);
    }
    return 0;
}
