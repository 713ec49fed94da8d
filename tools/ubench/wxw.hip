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
template <int G, int RD, int RB, int ILV, int SYNC, int NQ, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_wxw(const float4* __restrict__ src, int n_stages, float* __restrict__ sink,
                                             long long* __restrict__ cycles) {
    static_assert(NQ % RD == 0, "slot indices must be compile-time constants");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WS = 64 * G + 4;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, arow = lane & 3;
    for (int e = tid; e < RB * 4 * WS; e += 64 * NW) lds[e] = 0.001f * (float)(e % 97);
    float* part = lds + RB * 4 * WS;
    if (tid == 0) reinterpret_cast<int*>(part + NW * 4 * 64 * G - 4)[0] = 0;
    __syncthreads();
    constexpr int IS = NW * G * 64;                                    // float4 between consecutive items of one wave
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
                const float* p = part + ((256 * i + tid) % (4 * 64 * G));
#pragma unroll
                for (int w = 0; w < NW; ++w) v += p[w * 4 * 64 * G];
            }
            lds[(tid % (4 * WS))] = v * 1e-9f;
            lds_barrier();
        } else if constexpr (SYNC == 2) {                 // partial store + ONE barrier, no reduction
            float* pw = part + (size_t)wave * 4 * 64 * G + lane;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[r * 64 * G + 64 * g] = acc[0][g][r];
            lds_barrier();
        } else if constexpr (SYNC == 3) {                 // two bare barriers
            lds_barrier();
            lds_barrier();
        } else if constexpr (SYNC == 6) {                 // software barrier: an LDS counter every wave bumps and polls
            volatile int* ctr = reinterpret_cast<volatile int*>(part + NW * 4 * 64 * G - 4);
            if (lane == 0) atomicAdd(const_cast<int*>(ctr), 1);
            const int want = NW * (st + 1);
            while (*ctr < want) __builtin_amdgcn_s_sleep(1);
        } else if constexpr (SYNC >= 50) {                // two bare barriers, then the second wave of every SIMD waits
            lds_barrier();
            lds_barrier();
            if (wave >= 4) __builtin_amdgcn_s_sleep(SYNC - 50);          // (64 cycles per unit)
        } else if constexpr (SYNC == 4) {                 // one barrier; every wave reduces only the slice it consumes next
            float* pw = part + (size_t)wave * 4 * 64 * G + lane;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[r * 64 * G + 64 * g] = acc[0][g][r];
            lds_barrier();
            float v = 0.f;
            constexpr int SL = 64 * G / NW;               // columns of this wave's slice
            if (lane < SL) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int w = 0; w < NW; ++w) v += part[(size_t)w * 4 * 64 * G + r * 64 * G + SL * wave + lane];
                lds[arow * WS + SL * wave + lane] = v * 1e-9f;
            }
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

template <int G, int RD, int RB, int ILV, int SYNC, int NQ, int NW = 4>
static void run(const char* name, const float4* src, size_t region_bytes, int n_wg, float* sink, long long* cyc) {
    const size_t item_round = (size_t)NW * G * 1024;                        // bytes per item of all the waves
    const int n_stages = (int)(region_bytes / item_round / NQ) - 1;          // (the ring runs RD items ahead)
    const size_t lds = (size_t)(RB * 4 * (64 * G + 4) + NW * 4 * 64 * G) * 4;
    auto kern = k_wxw<G, RD, RB, ILV, SYNC, NQ, NW>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * NW), lds, 0, src, n_stages, sink, cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<long long> h(n_wg);
    (void)hipMemcpy(h.data(), cyc, n_wg * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= n_wg;
    const double items = (double)n_stages * NQ;
    printf("%-34s G=%d RD=%2d RB=%d NQ=%2d NW=%d %3d WGs: %7.1f cycles per 20 KiB  %5.1f B/clk/CU\n", name, G, RD,
           RB, NQ, NW, n_wg, mean / items * (20480.0 / item_round), item_round * items / mean);
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
        // two waves per SIMD (8 per workgroup, each half the K range): does the second wave fill the request-issue stalls?
        run<5, 3, 1, 2, 0, 12, 8>("8 waves: pure stream, ring 3", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 0, 12, 8>("8 waves: MFMAs then refill, ring 3", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 1, 12, 8>("8 waves: ... + stage sync", src, region, n_wg, sink, cyc);
        run<5, 4, 1, 0, 1, 12, 8>("8 waves: ring 4 + stage sync", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 0, 1, 12, 8>("8 waves: ring 6 + stage sync", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 2, 12, 8>("8 waves: store + 1 barrier", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 3, 12, 8>("8 waves: 2 bare barriers", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 4, 12, 8>("8 waves: 1 barrier + slice reduce", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 0, 4, 12, 8>("8 waves: same, ring 6", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 6, 12, 8>("8 waves: software barrier (LDS counter)", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 2, 6, 12, 8>("8 waves: pure stream + software barrier", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 0, 6, 24, 4>("4 waves: software barrier", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 51, 12, 8>("8 waves: 2 barriers + skew 64", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 53, 12, 8>("8 waves: 2 barriers + skew 192", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 56, 12, 8>("8 waves: 2 barriers + skew 384", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 0, 3, 24, 8>("8 waves: 2 barriers per 24 items", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 3, 3, 12, 8>("8 waves: MFMA only + 2 barriers", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 3, 0, 12, 8>("8 waves: MFMA only", src, region, n_wg, sink, cyc);
        run<5, 3, 1, 2, 3, 12, 8>("8 waves: pure stream + 2 barriers", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 0, 3, 24, 4>("4 waves: 2 bare barriers", src, region, n_wg, sink, cyc);
        run<5, 6, 1, 0, 4, 24, 4>("4 waves: 1 barrier + slice reduce", src, region, n_wg, sink, cyc);
    }
    return 0;
}
