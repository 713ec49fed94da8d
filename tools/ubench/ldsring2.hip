// LDS-ring weight stream, second form (see ldsring.hip for the first): ONE LOADER WAVE PER CONSUMER WAVE (8 waves, two per
// SIMD; the loader issues no MFMA, so the SIMD's matrix pipe belongs to its consumer).  Loader w streams consumer w's tiles
// into that consumer's own ring of R 1-KiB slots (R a power of two), KF requests in flight:
//   loader:   wait until the consumer's progress word says slot (t - R) is free -> m0 = slot address, global_load_lds_dwordx4
//             -> s_waitcnt vmcnt(KF - 1) -> publish "tiles landed" = t - KF + 2 every PUB tiles
//   consumer: wait until landed > t (cached) -> ds_read_b128 of the slot -> 4 RB MFMAs; publish its progress every PUB tiles
// Workgroup barriers (stage ends) are joined by the loaders R - 1 tiles late (ldsring.hip).
// ldsring.hip's loader (one wave, four streams, 64-bit address arithmetic and a modulo per row) was ISSUE-bound at 305 cycles
// per row whatever was in flight; ldsdma.hip shows the DMA path itself delivers what register loads do (44 B/clk from one wave
// with 32 requests in flight, 50 from four).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int N> struct IC { static constexpr int value = N; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}

// m0 = lds_dst; 1 KiB from sbase + voff + IMM to LDS
template <int IMM>
__device__ __forceinline__ void glds16s(unsigned voff, const float4* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" :: "v"(voff), "s"(sbase), "s"(lds_dst), "n"(IMM) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int flag_read(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void flag_write(unsigned addr, int val) {       // every lane writes the same word: no exec games
    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(val) : "memory");
}

// MODE 0: stream + MFMA, 1: stream only, 2: MFMA only (consumers do not wait for the loader)
template <int R, int KF, int PUB, int RB, int MODE, int NQS, int EPI>
__global__ __launch_bounds__(512) void k_lr2(const float4* __restrict__ src, int n_stages, size_t wave_stride,
                                             float* __restrict__ sink, long long* __restrict__ cycles) {
    static_assert((R & (R - 1)) == 0 && KF <= R && KF < 64 && R % PUB == 0 && NQS % PUB == 0, "ring");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WS = 4 * NQS + 4;
    // LDS: rings [4][R][256 floats] | flags: landed[4], consumed[4] | A tile [RB*4][WS] | out [RB*4][260]
    float* ring = lds;
    int* flags = reinterpret_cast<int*>(lds + 4 * R * 256);
    float* act = lds + 4 * R * 256 + 16;
    float* out = act + RB * 4 * WS;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, arow = lane & 3;
    for (int e = tid; e < RB * 4 * WS; e += 512) act[e] = 0.001f * (float)(e % 97);
    if (tid < 16) flags[tid] = 0;
    __syncthreads();
    const int total = n_stages * NQS;
    const int cw = wave & 3;                                               // the consumer this wave is / serves
    const unsigned ring_b = (unsigned)(size_t)ring + cw * R * 1024;
    const unsigned fl_landed = (unsigned)(size_t)flags + 4 * cw, fl_cons = fl_landed + 16;
    if (wave >= 4) {
        // ---------------- loader of consumer cw ----------------
        const float4* sp = src + (size_t)cw * wave_stride;
        const unsigned voff = lane * 16;
        int cons = 0, bars = 0, next_bar = NQS + R - 1;                    // tile index at which barrier `bars` is due
        const long long lt0 = __builtin_amdgcn_s_memtime();
        for (int t0 = 0; t0 < total; t0 += R) {
            static_for<0, R>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int t = t0 + j;
                if (t >= next_bar) { lds_barrier(); ++bars; next_bar += NQS; }
                if constexpr (MODE != 2 && MODE != 3 && MODE != 4) {
                    if (j % PUB == 0 && t >= R) {                          // slots t .. t + PUB - 1 free once tiles < t - R + PUB are consumed
                        while (cons < t - R + PUB) { asm volatile("" ::: "memory"); cons = __builtin_amdgcn_readfirstlane(flags[4 + cw]); }
                    }
                }
                if constexpr (MODE != 4) glds16s<(j % 4) * 1024>(voff, sp + (size_t)(j / 4) * 256, ring_b + (j / 4) * 4096);   // (the offset field moves the LDS address too)
                if (j % PUB == PUB - 1) {
                    vm_wait<KF - 1>();
                    flags[cw] = t - KF + 2;
                }
            });
            sp += (size_t)R * 64;
        }
        vm_wait<0>();
        flags[cw] = total;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long lt1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && cw == 0) cycles[gridDim.x * 4 + blockIdx.x] = lt1 - lt0;
        while (bars < n_stages) { lds_barrier(); ++bars; }
        return;
    }
    // ---------------- consumer cw ----------------
    // All LDS reads are plain loads the compiler tracks (exact lgkmcnt waits); a "memory" clobber per tile makes it re-read what
    // the DMA changed behind its back.  Tiles t .. t + LA - 1 are in flight into registers; the `landed` word is polled WITHOUT
    // blocking (read issued one tile before its value is looked at) - only a consumer that has run dry spins.
    constexpr int LA = 3;
    static_assert(NQS % LA == 0 || true, "");
    f32x4 acc[4][RB];
    float tot = 0.f;
    int landed = 0;
    const float* tile_p = ring + cw * R * 256 + lane * 4;
    const int* fl_l = flags + cw;
    int* fl_c = flags + 4 + cw;
    auto wait_landed = [&](int upto) {
        while (landed < upto) { asm volatile("" ::: "memory"); landed = __builtin_amdgcn_readfirstlane(*fl_l); }
    };
    const long long t0 = __builtin_amdgcn_s_memtime();
    int t = 0;
    f32x4 b[LA];
    int flv = 0;
    if constexpr (MODE != 2 && MODE != 4) wait_landed(LA);
#pragma unroll
    for (int u = 0; u < LA; ++u) b[u] = *reinterpret_cast<const f32x4*>(tile_p + ((u & (R - 1)) * 256));
    for (int st = 0; st < n_stages; ++st) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[k][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* ap = act + arow * WS;
        float4 an[RB];                                                     // A operand of the next tile, one step ahead
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ap + rb * 4 * WS);
#pragma unroll 1
        for (int q0 = 0; q0 < (MODE == 3 ? 0 : NQS); q0 += LA) {
            static_for<0, LA>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (q0 + u < NQS) {
                    asm volatile("" ::: "memory");
                    // tile t + LA: look at the flag value requested one tile ago, spin only when dry
                    if constexpr (MODE != 2 && MODE != 4) {
                        if (landed < t + LA + 1) { landed = __builtin_amdgcn_readfirstlane(flv); if (landed < t + LA + 1 && t + LA < total) wait_landed(t + LA + 1); }
                    }
                    const f32x4 bn = *reinterpret_cast<const f32x4*>(tile_p + (((t + LA) & (R - 1)) * 256));
                    if constexpr (MODE != 2 && MODE != 4) flv = *fl_l;
                    float4 a[RB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a[rb] = an[rb];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ap + rb * 4 * WS + 4 * ((q0 + u + 1) % NQS));
                    const f32x4 bc = b[u];
                    if constexpr (MODE == 1) {
                        acc[0][0] += bc;
                    } else {
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) acc[0][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].x, bc.x, acc[0][rb], 0, 0, 0);
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) acc[1][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].y, bc.y, acc[1][rb], 0, 0, 0);
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) acc[2][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].z, bc.z, acc[2][rb], 0, 0, 0);
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) acc[3][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].w, bc.w, acc[3][rb], 0, 0, 0);
                    }
                    b[u] = bn;
                    ++t;
                    if ((t & (PUB - 1)) == 0) *fl_c = t;                   // tiles < t have been read (LDS serves a wave's accesses in order)
                }
            });
        }
        if constexpr (EPI) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const f32x4 o = (acc[0][rb] + acc[1][rb]) + (acc[2][rb] + acc[3][rb]);
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(4 * rb + r) * 260 + 64 * wave + lane] = o[r] > 0.f ? o[r] : 0.f;
            }
        } else {
            tot += acc[0][0][0] + acc[1][0][1] + acc[2][0][2] + acc[3][0][3];
        }
        lds_barrier();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 256 + tid] = tot + out[tid];
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int R, int KF, int PUB, int RB, int MODE, int NQS, int EPI>
static void run(const char* name, const float4* src, size_t region_bytes, int n_wg, float* sink, long long* cyc) {
    const size_t wave_bytes = region_bytes / 4;
    const int n_stages = ((int)(wave_bytes / 1024 / NQS) - 1);
    const size_t lds = (size_t)(4 * R * 256 + 16 + RB * 4 * (4 * NQS + 4) + RB * 4 * 260 + 256) * 4;
    auto kern = k_lr2<R, KF, PUB, RB, MODE, NQS, EPI>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipMemset(cyc, 0, 1024 * 16 * 8);
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(512), lds, 0, src, n_stages, wave_bytes / 16, sink, cyc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: failed: %s\n", name, hipGetErrorString(e)); return; }
    std::vector<long long> h((size_t)n_wg * 13);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) if (h[(size_t)n_wg * 5 + w]) printf("   wg 0 wave %d stuck: %lld\n", w, h[(size_t)n_wg * 5 + w]);
    double mean = 0, lmean = 0;
    for (int g = 0; g < n_wg; ++g) {
        long long mx = 0;
        for (int w = 0; w < 4; ++w) mx = h[(size_t)g * 4 + w] > mx ? h[(size_t)g * 4 + w] : mx;
        mean += (double)mx;
        lmean += (double)h[(size_t)n_wg * 4 + g];
    }
    mean /= n_wg; lmean /= n_wg;
    const double tiles = (double)n_stages * NQS;
    printf("%-28s R=%2d KF=%2d PUB=%d RB=%d NQS=%3d epi=%d %3d WGs: %6.1f cycles per tile and wave  %5.1f B/clk/CU  (loader %6.1f)\n", name, R,
           KF, PUB, RB, NQS, EPI, n_wg, mean / tiles, tiles * 4 * 1024.0 / mean, lmean / tiles);
    fflush(stdout);
}

int main() {
    const size_t region = 10u << 20;
    float4* src; float* sink; long long* cyc;
    (void)hipMalloc((void**)&src, region + (2u << 20)); (void)hipMemset(src, 0, region + (2u << 20));
    (void)hipMalloc((void**)&sink, 1024 * 1024 * 4); (void)hipMalloc((void**)&cyc, 1024 * 16 * 8);
    const int n_wg = 256;
    run<16, 8, 2, 2, 3, 64, 1>("idle consumers", src, region, n_wg, sink, cyc);
    run<16, 8, 2, 2, 4, 64, 1>("MFMA only, idle loaders", src, region, n_wg, sink, cyc);
    run<16, 8, 2, 2, 2, 64, 1>("MFMA only (no waits)", src, region, n_wg, sink, cyc);
    run<16, 8, 2, 1, 1, 64, 0>("stream only", src, region, n_wg, sink, cyc);
    run<16, 8, 2, 2, 0, 64, 1>("stream + MFMA + stage sync", src, region, n_wg, sink, cyc);
    run<16, 12, 2, 2, 0, 64, 1>("stream + MFMA + stage sync", src, region, n_wg, sink, cyc);
    run<16, 8, 4, 2, 0, 64, 1>("stream + MFMA + stage sync", src, region, n_wg, sink, cyc);
    run<16, 8, 2, 2, 0, 16, 1>("short stages", src, region, n_wg, sink, cyc);
    run<16, 8, 2, 4, 0, 64, 1>("16 chains", src, region, n_wg, sink, cyc);
    return 0;
}
