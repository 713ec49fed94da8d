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
__global__ __launch_bounds__(64 * (NW + (SYNC >= 20 ? 1 : 0))) void k_stream2(const char* __restrict__ src, size_t wave_off, int n_stages,
                                                      float* __restrict__ sink, long long* __restrict__ cycles, int getwait = 0, int wrap_stages = 0, int pf_ahead = 1, int pf_on = 1) {
    static_assert(RD % B == 0 && NT % RD == 0 && RD <= 63, "static slots");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WS = 4 * NT + 4;
    constexpr size_t tile_stride = LAYOUT ? (size_t)NW * 1024 : 1024;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int e = tid; e < 8 * WS; e += 64 * NW) lds[e] = 0.001f * (float)(e % 97);
    if (tid < 8) reinterpret_cast<int*>(lds + 8 * WS)[tid] = 0;
    __syncthreads();
    // SYNC 20 / 21 / 22 (round 5): two barriers per stage AND one more wave per workgroup that only PREFETCHES - it touches the
    // 128-byte lines the compute waves will request one stage later (one dword per line and lane: 8 KiB of lines per
    // instruction; nothing is waited for), 1 / 32, 1 / 8 or all of them (the 32 CUs of an XCD share its L2 and run in step),
    // and joins the stage's barriers, which is all the pacing it needs
    if constexpr (SYNC >= 20) {
        if (wave == NW) {
            constexpr int S = SYNC == 20 ? 32 : (SYNC == 21 ? 8 : 1);
            constexpr int CH = NW * NT / 8;                         // chunks of 64 lines per stage
            const int share = (blockIdx.x >> 3) % S;
            constexpr int CW = (CH + S - 1) / S;                    // chunks of this workgroup per stage
            float pend[CW], keep = 0.f;                             // compiler-tracked loads, consumed one stage later (long landed)
#pragma unroll
            for (int i = 0; i < CW; ++i) pend[i] = 0.f;
            for (int st = 0; st < n_stages; ++st) {
#pragma unroll
                for (int i = 0; i < CW; ++i) keep += pend[i];
#pragma unroll
                for (int i = 0; i < CW; ++i) {
                    const int c = share + i * S;
                    const int cl = c < CH ? c : CH - 1;
                    const int w = cl / (NT / 8), cc = cl % (NT / 8);
                    const char* p = src + (size_t)w * wave_off + ((size_t)RD + (size_t)(st + pf_ahead) * NT) * 1024 + (size_t)cc * 8192 + lane * 128;
                    if (pf_on) pend[i] = *reinterpret_cast<const float*>(p);
                }
                lds_barrier(); lds_barrier();
            }
#pragma unroll
            for (int i = 0; i < CW; ++i) keep += pend[i];
            sink[blockIdx.x * 64 * (NW + 1) + tid] = keep;
            return;
        }
    }
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
    long long twait = 0;
    if constexpr (SYNC == 13) { if (wave >= 4) __builtin_amdgcn_s_setprio(3); }   // the second wave of every SIMD at the highest priority
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
        // WRAP_STAGES (round 5): the wave re-reads the same few stages - a working set that stays in the XCD's L2 (no first-touch misses)
        if (wrap_stages && (st + 1) % wrap_stages == 0) sb -= (size_t)wrap_stages * NT * tile_stride;
        if constexpr (SYNC == 2) { lds_barrier(); lds_barrier(); }
        if constexpr (SYNC >= 20) { lds_barrier(); lds_barrier(); }
        if constexpr (SYNC == 3) { lds_barrier(); }                                            // one barrier per stage
        if constexpr (SYNC == 4) { lds_barrier(); lds_barrier(); if (wave >= 4) __builtin_amdgcn_s_sleep(1); }   // partner waves restart 64 cycles later
        if constexpr (SYNC == 5) {                                                             // two barriers, time spent in them accounted
            const long long b0 = __builtin_amdgcn_s_memtime();
            lds_barrier(); lds_barrier();
            twait += __builtin_amdgcn_s_memtime() - b0;
        }
        if constexpr (SYNC == 7) {                                                             // software barrier (LDS counter + s_sleep) instead of s_barrier
            volatile int* bar = reinterpret_cast<volatile int*>(lds + 8 * WS);
            if (lane == 0) {
                __hip_atomic_fetch_add((int*)bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (*bar < NW * (st + 1)) __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (SYNC == 8 || SYNC == 9) {                                                // software barriers over SUBSETS of the waves:
            // 8: the two waves of a SIMD {w, w + 4}; 9: the first waves of the SIMDs {0..3} and the second waves {4..7} separately
            volatile int* bar = reinterpret_cast<volatile int*>(lds + 8 * WS) + 1 + (SYNC == 8 ? (wave & 3) : (wave >> 2));
            constexpr int NMEM = SYNC == 8 ? 2 : 4;
            if (lane == 0) {
                __hip_atomic_fetch_add((int*)bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (*bar < NMEM * (st + 1)) __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (SYNC == 12) {                                                            // two barriers, arrival / departure stamps of workgroup 0
            const long long b0 = __builtin_amdgcn_s_memtime();
            lds_barrier(); lds_barrier();
            const long long b1 = __builtin_amdgcn_s_memtime();
            if (blockIdx.x == 0 && lane == 0 && st < 16) { cycles[4096 + (st * 8 + wave) * 2] = b0 - t0; cycles[4096 + (st * 8 + wave) * 2 + 1] = b1 - t0; }
        }
        if constexpr (SYNC == 13 || SYNC == 14) {                                              // two barriers + s_setprio
            lds_barrier(); lds_barrier();
            if constexpr (SYNC == 14) {                                                        // the favoured wave of a SIMD alternates per stage
                if (((st & 1) ^ (wave >> 2)) != 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
            }
        }
        if constexpr (SYNC == 6) { lds_barrier(); lds_barrier(); if (wave >= 4) __builtin_amdgcn_s_sleep(8); }   // ... 512 cycles later
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    static_for<0, RD>([&](auto ic) { constexpr int i = decltype(ic)::value; wait_for<AREG, 0>(ring[i]); s += ring[i].x; });
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) s += acc[k][rb][0] + acc[k][rb][1] + acc[k][rb][2] + acc[k][rb][3];
    sink[blockIdx.x * 64 * NW + tid] = s;
    if (lane == 0) cycles[blockIdx.x * NW + wave] = SYNC == 5 && getwait ? twait : t1 - t0;
}

// LAYOUT 0: every wave its own contiguous region; 1: the waves' tiles interleaved (tile t of wave w at (t NW + w) KiB)
template <int NW, int RD, int B, int RB, int AREG, int VADDR, int ILV, int SYNC, int NT, int LAYOUT>
static void run(const char* src, size_t region, int n_wg, float* sink, long long* cyc) {
    const size_t wave_bytes = region / NW;
    const int n_stages = (int)(wave_bytes / 1024 / NT) - 1;
    auto kern = k_stream2<NW, RD, B, RB, AREG, VADDR, ILV, SYNC, NT, LAYOUT>;
    const size_t lds = (size_t)8 * (4 * NT + 4) * 4 + 64;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t wave_off = LAYOUT ? 1024 : wave_bytes;
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * (NW + (SYNC >= 20 ? 1 : 0))), lds, 0, src, wave_off, n_stages, sink, cyc, getenv("GETWAIT") ? 1 : 0, getenv("WRAP_STAGES") ? atoi(getenv("WRAP_STAGES")) : 0,
                           getenv("PF_AHEAD") ? atoi(getenv("PF_AHEAD")) : 1, getenv("PF_OFF") ? 0 : 1);
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
    if (SYNC == 12) {
        std::vector<long long> st((size_t)16 * 8 * 2);
        (void)hipMemcpy(st.data(), cyc + 4096, st.size() * 8, hipMemcpyDeviceToHost);
        printf("   workgroup 0, per stage: arrival at / departure from the barriers (cycles since start), waves 0 1 | 4 5\n");
        for (int s_ = 0; s_ < 8; ++s_) {
            printf("   stage %d:", s_);
            for (int w : {0, 1, 4, 5}) if (w < NW) printf("  w%d %lld/%lld", w, st[(s_ * 8 + w) * 2], st[(s_ * 8 + w) * 2 + 1]);
            printf("\n");
        }
    }
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
    printf("-- one barrier per stage / partner waves delayed after the barriers / time inside the barriers (GETWAIT=1: the per-wave figures are that time)\n");
    run<8, 24, 1, 1, 0, 0, 0, 3, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 4, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 6, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 5, 48, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 0, 0, 5, 96, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 2, 192, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 7, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 12, 48, 0>(src, region, W, sink, cyc);
    printf("-- a prefetch wave (sync 20 / 21 / 22: it touches 1/32, 1/8, all of next stage's lines)\n");
    run<4, 24, 1, 1, 1, 0, 0, 2, 96, 0>(src, region, W, sink, cyc);      // (ring of 24: five waves of <= 256 registers without spills)
    run<4, 24, 1, 1, 1, 0, 0, 20, 96, 0>(src, region, W, sink, cyc);
    run<4, 24, 1, 1, 1, 0, 0, 21, 96, 0>(src, region, W, sink, cyc);
    run<4, 16, 1, 1, 1, 0, 0, 2, 96, 0>(src, region, W, sink, cyc);
    run<4, 16, 1, 1, 1, 0, 0, 20, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 0, 0, 20, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 0, 0, 21, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 0, 0, 22, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 0, 0, 2, 96, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 21, 48, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 2, 1, 0, 0, 21, 96, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 2, 1, 0, 0, 2, 96, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 2, 48, 1>(src, region, W, sink, cyc);     // interleaved layout with barriers
    run<8, 24, 1, 1, 0, 0, 0, 0, 48, 1>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 0, 0, 2, 96, 1>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 13, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 14, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 8, 48, 0>(src, region, W, sink, cyc);
    run<8, 24, 1, 1, 0, 0, 0, 9, 48, 0>(src, region, W, sink, cyc);
    run<4, 32, 1, 1, 1, 0, 0, 7, 96, 0>(src, region, W, sink, cyc);
    printf("-- 16 chains per workgroup (RB 4), with two barriers per stage: is the 16-chain shape sensitive to L2 first touches? (WRAP_STAGES)\n");
    run<4, 24, 1, 4, 1, 0, 0, 2, 96, 0>(src, region, W, sink, cyc);
    run<4, 24, 1, 4, 1, 0, 0, 0, 96, 0>(src, region, W, sink, cyc);
    if (getenv("ONLY_R5")) return 0;
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
