// Micro-benchmark for the structure VERDICT r3 item 1(a) names: the B operand of the 4x4x1 stream GEMMs from an LDS RING that
// LOADER wave(s) fill with global_load_lds_dwordx4, so that the multiplying waves never block on their own weight request
// (nsplit.hip: a wave that requests its own 1-KiB tile loses ~34 cycles of issue per tile next to 64 cycles of MFMAs).
// NC = 4 consumer waves (one per SIMD), each with its own tile stream (as stream_r8.h); NL loader waves (wave index >= 4) that
// share SIMDs with consumers but issue no MFMA.  Ring: R rows of 4 tiles (one per consumer) = 4 R KiB of LDS.
//   loader:   for row r: wait until every consumer has consumed row r - R (LDS progress words), 4 DMA requests, then
//             s_waitcnt vmcnt(4 (KF - 1)) -> row r - KF + 1 has landed -> publish `filled` (LDS word)
//   consumer: for row r: wait until filled > r (cached; re-read only when behind), ds_read_b128 of its tile, 4 RB MFMAs,
//             publish its progress every second row
// Stage = NQS rows, then an epilogue (LDS writes) and a workgroup barrier that the loader joins R - 1 rows late (when it
// would have to wait for a consumer that is behind that barrier anyway).
// Prints cycles per tile and wave (slowest consumer), B/clk per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const float4* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// progress words: explicit DS instructions (a `volatile` LDS pointer makes hipcc emit FLAT accesses with sc0 sc1 and a
// vmcnt(0) behind each - that drains the loader's DMA queue at every publication)
__device__ __forceinline__ int flag_read(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int flag_read_min4(unsigned addr) {
    int4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(min(min(v.x, v.y), min(v.z, v.w)));
}
__device__ __forceinline__ void flag_write(unsigned addr, int val) {
    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(val) : "memory");
}

// MODE 0: stream + MFMA, 1: stream only (consumers read the tile, one add), 2: MFMA only (no waiting for the loader)
template <int NL, int R, int KF, int RB, int MODE, int NQS, int EPI, int LDBG = 0>
__global__ __launch_bounds__(64 * (4 + NL)) void k_lr(const float4* __restrict__ src, int n_stages, size_t wave_stride,
                                                      float* __restrict__ sink, long long* __restrict__ cycles) {
    static_assert(KF <= R && 4 * KF / NL <= 64, "in-flight rows");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NC = 4;
    constexpr int WS = 4 * NQS + 4;
    // LDS: ring [R][4][256 floats] | flags: filled, consumed[4] (16 ints) | A tile [RB*4][WS] | out [RB*4][260]
    float* ring = lds;
    int* flags = reinterpret_cast<int*>(lds + R * 4 * 256);
    const unsigned fl_b = (unsigned)(size_t)flags;     // filled[4] at +0, consumed[4] at +16
    float* act = lds + R * 4 * 256 + 16;
    float* out = act + RB * 4 * WS;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, arow = lane & 3;
    for (int e = tid; e < RB * 4 * WS; e += 64 * (NC + NL)) act[e] = 0.001f * (float)(e % 97);
    if (tid < 16) flags[tid] = 0;
    __syncthreads();
    const int total = n_stages * NQS;
    const unsigned ring_b = (unsigned)(size_t)ring;
    if (wave >= NC) {
        // ---------------- loader ----------------
        const int li = wave - NC;                       // loader li serves consumers li, li + NL, ...
        constexpr int PER = NC / NL;                    // tiles per row this loader requests
        int cmin = 0, bars = 0;
        const long long lt0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < total; ++r) {
            const int need = r - R + 1;                 // rows that must have been consumed
            if (need > 0) {
                while (bars < (need - 1) / NQS) { lds_barrier(); ++bars; }
                while (!(LDBG & 2) && cmin < need) {
                    cmin = flag_read_min4(fl_b + 16);
                    if (cmin < need) __builtin_amdgcn_s_sleep(1);
                }
            }
            const int slot = r % R;
#pragma unroll
            for (int i = 0; i < ((LDBG & 4) ? 0 : PER); ++i) {
                const int w = li + i * NL;
                glds16(src + (size_t)w * wave_stride + (size_t)r * 64 + lane, ring_b + (unsigned)((slot * 4 + w) * 1024));
            }
            if (r >= KF - 1) {
                vm_wait<PER * (KF - 1)>();
                if constexpr (LDBG & 1) continue;
                if (lane == 0) flag_write(fl_b + 4 * li, r - KF + 2);  // rows landed (this loader's share)
            }
        }
        vm_wait<0>();
        const long long lt1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && li == 0) cycles[gridDim.x * 4 + blockIdx.x] = lt1 - lt0;
        if (lane == 0) flag_write(fl_b + 4 * li, total);
        while (bars < n_stages) { lds_barrier(); ++bars; }
        return;
    }
    // ---------------- consumers ----------------
    f32x4 acc[4][RB];
    float tot = 0.f;
    int avail = 0;
    const float* tile0 = ring + wave * 256 + lane * 4;
    const long long t0 = __builtin_amdgcn_s_memtime();
    int row = 0, slot = 0;
    for (int st = 0; st < n_stages; ++st) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[k][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* ap = act + arow * WS;
#pragma unroll 4
        for (int q = 0; q < (MODE == 3 ? 0 : NQS); ++q, ++row) {
            if constexpr (MODE != 2) {
                while (avail <= row) {
                    if constexpr (NL == 1) avail = flag_read(fl_b);
                    else if constexpr (NL == 2) avail = min(flag_read(fl_b), flag_read(fl_b + 4));
                    else avail = flag_read_min4(fl_b);
                }
                asm volatile("" ::: "memory");
            }
            const f32x4 b = *reinterpret_cast<const f32x4*>(tile0 + slot * 1024);
            slot = slot + 1 == R ? 0 : slot + 1;
            if constexpr (MODE == 1) {
                acc[0][0] += b;
            } else {
                float4 a[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const float4*>(ap + rb * 4 * WS + 4 * q);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[0][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].x, b.x, acc[0][rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[1][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].y, b.y, acc[1][rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[2][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].z, b.z, acc[2][rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[3][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].w, b.w, acc[3][rb], 0, 0, 0);
            }
            if ((q & 1) == 1) {                          // the tile reads above have returned (their values were used)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) flag_write(fl_b + 16 + 4 * wave, row + 1);
            }
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

template <int NL, int R, int KF, int RB, int MODE, int NQS, int EPI, int LDBG = 0>
static void run(const char* name, const float4* src, size_t region_bytes, int n_wg, float* sink, long long* cyc) {
    const size_t wave_bytes = region_bytes / 4;
    const int n_stages = (int)(wave_bytes / 1024 / NQS) - 1;
    const size_t lds = (size_t)(R * 4 * 256 + 16 + RB * 4 * (4 * NQS + 4) + RB * 4 * 260 + 256) * 4;
    auto kern = k_lr<NL, R, KF, RB, MODE, NQS, EPI, LDBG>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * (4 + NL)), lds, 0, src, n_stages, wave_bytes / 16, sink, cyc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: failed: %s\n", name, hipGetErrorString(e)); return; }
    std::vector<long long> h((size_t)n_wg * 5);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double lmean = 0;
    for (int g = 0; g < n_wg; ++g) lmean += (double)h[(size_t)n_wg * 4 + g];
    lmean /= n_wg;
    double mean = 0;
    for (int g = 0; g < n_wg; ++g) {
        long long mx = 0;
        for (int w = 0; w < 4; ++w) mx = h[(size_t)g * 4 + w] > mx ? h[(size_t)g * 4 + w] : mx;
        mean += (double)mx;
    }
    mean /= n_wg;
    const double tiles = (double)n_stages * NQS;
    printf("%-28s NL=%d R=%2d KF=%d RB=%d NQS=%3d epi=%d dbg=%d %3d WGs: %6.1f cycles per tile and wave  %5.1f B/clk/CU  (loader: %6.1f per row)\n",
           name, NL, R, KF, RB, NQS, EPI, LDBG, n_wg, mean / tiles, tiles * 4 * 1024.0 / mean, lmean / tiles);
    fflush(stdout);
}

int main() {
    const size_t region = 10u << 20;
    float4* src; float* sink; long long* cyc;
    (void)hipMalloc((void**)&src, region + (2u << 20)); (void)hipMemset(src, 0, region + (2u << 20));
    (void)hipMalloc((void**)&sink, 1024 * 1024 * 4); (void)hipMalloc((void**)&cyc, 1024 * 16 * 8);
    const int n_wg = 256;
    run<1, 12, 6, 2, 3, 64, 0, 3>("idle consumers, bare loader", src, region, n_wg, sink, cyc);
    run<1, 12, 12, 2, 3, 64, 0, 3>("idle consumers, bare loader", src, region, n_wg, sink, cyc);
    run<1, 16, 15, 2, 3, 64, 0, 3>("idle consumers, bare loader", src, region, n_wg, sink, cyc);
    run<1, 12, 3, 2, 3, 64, 0, 3>("idle consumers, bare loader", src, region, n_wg, sink, cyc);
    run<2, 12, 12, 2, 3, 64, 0, 3>("idle consumers, bare loader", src, region, n_wg, sink, cyc);
    run<4, 12, 12, 2, 3, 64, 0, 3>("idle consumers, bare loader", src, region, n_wg, sink, cyc);
    run<1, 12, 12, 2, 3, 64, 0, 3>("1 workgroup", src, region, 1, sink, cyc);
    return 0;
}
