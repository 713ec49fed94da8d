// LDS-ring weight stream, THIRD form: ldsring2.hip's loaders (one loader wave per consumer wave) with a HAND-SCHEDULED consumer
// (one asm statement per tile: explicit lgkmcnt, B tiles two ahead, A operand one ahead, the `landed` word read without
// blocking, progress published every tile).  Second form (see ldsring.hip for the first): ONE LOADER WAVE PER CONSUMER WAVE (8 waves, two per
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
__global__ __launch_bounds__(512) void k_lr3(const float4* __restrict__ src, int n_stages, size_t wave_stride,
                                             float* __restrict__ sink, long long* __restrict__ cycles) {
    static_assert((R & (R - 1)) == 0 && KF <= R && KF < 64 && R % PUB == 0 && NQS % PUB == 0 && NQS % 6 == 0 && RB == 2, "ring");
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
    // ---------------- consumer cw: one asm statement per tile ----------------
    // LDS operations of tile t, in issue order: A(t+1) x 2, B(t+2), landed word, progress word.  At the top of tile t
    // `s_waitcnt lgkmcnt(3)` leaves B(t+1), the flag read and the progress write of tile t-1 in flight and guarantees A(t)
    // (and B(t), requested in tile t-2).  Registers: B ring b0 / b1 / b2 (tile mod 3), A sets e / o (tile mod 2) - fixed
    // roles, no copies; the flag value read in tile t-1 is moved to an SGPR inside tile t's statement, after its wait.
    f32x4 acc[4][2];
    float tot = 0.f;
    const unsigned tile_v = ring_b + lane * 16;
    const unsigned a_base = (unsigned)(size_t)act + arow * WS * 4;
    int landed = 0;
    auto wait_landed = [&](int upto) { while (landed < upto) landed = flag_read(fl_landed); };
    const long long t0 = __builtin_amdgcn_s_memtime();
    int t = 0;
    f32x4 b0, b1, b2, ae0, ae1, ao0, ao1;
    int fla = 0, flb = 0;
    if constexpr (MODE != 2 && MODE != 4) wait_landed(2);
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(ae0), "=&v"(ae1), "=&v"(b0), "=&v"(b1) : "v"(a_base), "v"(tile_v), "n"(4 * WS * 4) : "memory");
#define LR3_TILE(BC, BN, AC0, AC1, AN0, AN1, FLP, FLN, Q)                                                                       \
    do {                                                                                                                      \
        if constexpr (MODE != 2 && MODE != 4) { if (landed < t + 3 && t + 2 < total) wait_landed(t + 3); }                      \
        const unsigned baddr = tile_v + (unsigned)(((t + 2) & (R - 1)) << 10);                                                  \
        int sf;                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(3)\n\t"                                                                                \
                     "v_readfirstlane_b32 %[sf], %[flp]\n\t"                                                                  \
                     "ds_read_b128 %[an0], %[aa] offset:%[oa0]\n\t"                                                           \
                     "ds_read_b128 %[an1], %[aa] offset:%[oa1]\n\t"                                                           \
                     "ds_read_b128 %[bn], %[ba]\n\t"                                                                          \
                     "ds_read_b32 %[fln], %[fa]\n\t"                                                                          \
                     "ds_write_b32 %[fa], %[tv] offset:16\n\t"                                                                \
                     "v_mfma_f32_4x4x1_16b_f32 %[c00], %[a0x], %[bx], %[c00]\n\t"                                             \
                     "v_mfma_f32_4x4x1_16b_f32 %[c01], %[a1x], %[bx], %[c01]\n\t"                                             \
                     "v_mfma_f32_4x4x1_16b_f32 %[c10], %[a0y], %[by], %[c10]\n\t"                                             \
                     "v_mfma_f32_4x4x1_16b_f32 %[c11], %[a1y], %[by], %[c11]\n\t"                                             \
                     "v_mfma_f32_4x4x1_16b_f32 %[c20], %[a0z], %[bz], %[c20]\n\t"                                             \
                     "v_mfma_f32_4x4x1_16b_f32 %[c21], %[a1z], %[bz], %[c21]\n\t"                                             \
                     "v_mfma_f32_4x4x1_16b_f32 %[c30], %[a0w], %[bw], %[c30]\n\t"                                             \
                     "v_mfma_f32_4x4x1_16b_f32 %[c31], %[a1w], %[bw], %[c31]"                                                  \
                     : [sf] "=&s"(sf), [an0] "=&v"(AN0), [an1] "=&v"(AN1), [bn] "=&v"(BN), [fln] "=&v"(FLN),                    \
                       [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),              \
                       [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1])               \
                     : [a0x] "v"(AC0.x), [a0y] "v"(AC0.y), [a0z] "v"(AC0.z), [a0w] "v"(AC0.w), [a1x] "v"(AC1.x),                \
                       [a1y] "v"(AC1.y), [a1z] "v"(AC1.z), [a1w] "v"(AC1.w), [bx] "v"(BC.x), [by] "v"(BC.y), [bz] "v"(BC.z),    \
                       [bw] "v"(BC.w), [aa] "v"(a_base), [ba] "v"(baddr), [fa] "v"(fl_landed), [tv] "v"(t + 1), [flp] "v"(FLP),  \
                       [oa0] "n"(16 * (((Q) + 1) % NQS)), [oa1] "n"(4 * WS * 4 + 16 * (((Q) + 1) % NQS))                         \
                     : "memory");                                                                                             \
        if constexpr (MODE != 2 && MODE != 4) landed = sf > landed ? sf : landed;                                             \
        ++t;                                                                                                                  \
    } while (0)
    for (int st = 0; st < n_stages; ++st) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[k][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE != 3) {
            static_for<0, NQS / 6>([&](auto gc) {
                constexpr int q = 6 * decltype(gc)::value;
                LR3_TILE(b0, b2, ae0, ae1, ao0, ao1, fla, flb, q);
                LR3_TILE(b1, b0, ao0, ao1, ae0, ae1, flb, fla, q + 1);
                LR3_TILE(b2, b1, ae0, ae1, ao0, ao1, fla, flb, q + 2);
                LR3_TILE(b0, b2, ao0, ao1, ae0, ae1, flb, fla, q + 3);
                LR3_TILE(b1, b0, ae0, ae1, ao0, ao1, fla, flb, q + 4);
                LR3_TILE(b2, b1, ao0, ao1, ae0, ae1, flb, fla, q + 5);
            });
        }
        if constexpr (EPI) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
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
    auto kern = k_lr3<R, KF, PUB, RB, MODE, NQS, EPI>;
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
    run<16, 8, 2, 2, 3, 48, 1>("idle consumers", src, region, n_wg, sink, cyc);
    run<16, 8, 2, 2, 4, 48, 1>("MFMA only, idle loaders", src, region, n_wg, sink, cyc);
    run<16, 8, 2, 2, 0, 48, 1>("stream + MFMA + stage sync", src, region, n_wg, sink, cyc);
    run<16, 12, 2, 2, 0, 48, 1>("stream + MFMA + stage sync", src, region, n_wg, sink, cyc);
    return 0;
}
