// <= 256-register waves, TWO per SIMD, for the W x W stage of the 8-chain stream kernels (flow_r8.h /
// spline_r8.h: RB = 2 row blocks of 4 chains, one 1-KiB tile = 8 v_mfma_f32_4x4x1 + 1 request, csrc/stream_r8.h).
// Today: 4 waves, one per SIMD, 98 cycles per tile - the requesting wave is the multiplying wave, and a request blocks it for
// ~34 cycles.  Here: the SAME stream code (stream_r8.h, included) with 8 waves per workgroup, the second wave of a SIMD
// (wave w + 4) taking the other half of the K range of wave w's 64 columns; per stage (NT tiles per wave) the pair's partial
// sums meet in LDS the way a product stage would have to do it (partner partials -> LDS, barrier, sum + ReLU + activation
// store by waves 0 .. 3, barrier).  Prints cycles per tile and SIMD (4 waves: per tile and wave) and B / clk / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../fab_torch_amd/csrc/stream_r8.h"

using namespace fab;

// SYNC 0: free-running (no stage boundary); 1: the stage boundary described above; 2: two bare barriers per stage
template <int NW, int RD, int NT, int SYNC, int RB>
__global__ __launch_bounds__(64 * NW) void k_wave2(const float4* __restrict__ src, size_t wave_f4, int n_stages,
                                                    float* __restrict__ sink, long long* __restrict__ cycles) {
    static_assert(NT % RD == 0, "a stage must leave the ring phase unchanged");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KW = 4 * NT * (NW / 4);                       // K of the stage = activation width
    constexpr int WS = KW + 4;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    float* act = lds;                                           // [8 chains][WS]
    float* part = lds + 8 * WS;                                 // [4 partner waves][RB][64 lanes] float4
    for (int e = tid; e < 8 * WS; e += 64 * NW) act[e] = 0.001f * (float)(e % 97);
    __syncthreads();
    S8StreamT<RD> s;
    s8_stream_init(s, lane);
    s8_prologue(s, src + (size_t)wave * wave_f4);
    const int khalf = wave >> 2;                                // which half of K this wave multiplies
    const float* ap = act + (lane & 3) * WS + khalf * 4 * NT;
    float keep = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < n_stages; ++st) {
        S8Acc<RB> acc;
        s8_zero(acc);
        s8_run_k<4, 0, NT, S8_INF>(s, ap, 4 * WS, acc);
        f32x4 o[RB];
        s8_fold(acc, o);
        if constexpr (SYNC == 1) {
            if constexpr (NW == 8) {
                if (wave >= 4) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) reinterpret_cast<f32x4*>(part)[((wave - 4) * RB + rb) * 64 + lane] = o[rb];
                }
                s8_barrier();
                if (wave < 4) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const f32x4 p = reinterpret_cast<const f32x4*>(part)[(wave * RB + rb) * 64 + lane];
                        o[rb] += p;
                    }
                }
            } else {
                s8_barrier();
            }
            if (wave < 4) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) act[(4 * rb + r) * WS + (64 * wave + lane) % KW] = fmaxf(o[rb][r], 0.f) * 1e-6f;
            }
            s8_barrier();
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) keep += o[rb][0] + o[rb][1] + o[rb][2] + o[rb][3];
            if constexpr (SYNC == 2) { s8_barrier(); s8_barrier(); }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    s8_drain(s);
    sink[blockIdx.x * 64 * NW + tid] = keep + act[tid] + s.r[0][0];
    if (lane == 0) cycles[blockIdx.x * NW + wave] = t1 - t0;
}

template <int NW, int RD, int NT, int SYNC, int RB = 2>
static void run(const char* name, const float4* src, size_t region, int n_wg, float* sink, long long* cyc) {
    const size_t wave_bytes = region / NW;
    const int n_stages = (int)(wave_bytes / 1024 / NT) - 1;
    auto kern = k_wave2<NW, RD, NT, SYNC, RB>;
    const size_t lds = (size_t)(8 * (4 * NT * (NW / 4) + 4) + 4 * RB * 64 * 4) * 4;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * NW), lds, 0, src, wave_bytes / 16, n_stages, sink, cyc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s failed: %s\n", name, hipGetErrorString(e)); return; }
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
    printf("%-30s NW=%d RD=%2d NT=%3d RB=%d sync=%d %3d WGs: %6.1f cycles per tile and SIMD  %5.1f B/clk/CU  (MFMA floor %d)\n", name, NW, RD,
           NT, RB, SYNC, n_wg, mean / tiles_simd, tiles_simd * 4 * 1024 / mean, 32 * RB);
    fflush(stdout);
}

int main() {
    const size_t region = 10u << 20;
    float4* src; float* sink; long long* cyc;
    (void)hipMalloc((void**)&src, region + (2u << 20)); (void)hipMemset(src, 0, region + (2u << 20));
    (void)hipMalloc((void**)&sink, 1024 * 512 * 4); (void)hipMalloc((void**)&cyc, 1024 * 8 * 8);
    for (int n_wg : {256}) {
        run<4, 32, 96, 0>("4 waves, free", src, region, n_wg, sink, cyc);
        run<4, 32, 96, 2>("4 waves, 2 barriers", src, region, n_wg, sink, cyc);
        run<4, 32, 96, 1>("4 waves, stage", src, region, n_wg, sink, cyc);
        run<4, 40, 80, 1>("4 waves, stage", src, region, n_wg, sink, cyc);
        run<8, 24, 48, 0>("8 waves, free", src, region, n_wg, sink, cyc);
        run<8, 24, 48, 2>("8 waves, 2 barriers", src, region, n_wg, sink, cyc);
        run<8, 24, 48, 1>("8 waves, stage", src, region, n_wg, sink, cyc);
        run<8, 16, 48, 0>("8 waves, free", src, region, n_wg, sink, cyc);
        run<8, 16, 48, 1>("8 waves, stage", src, region, n_wg, sink, cyc);
        run<8, 12, 48, 1>("8 waves, stage", src, region, n_wg, sink, cyc);
        run<8, 20, 40, 1>("8 waves, stage", src, region, n_wg, sink, cyc);
        // the 4-chain shape (RB = 1: 4 MFMAs per tile)
        run<4, 32, 96, 1, 1>("4 waves, stage, 4 chains", src, region, n_wg, sink, cyc);
        run<8, 24, 48, 0, 1>("8 waves, free, 4 chains", src, region, n_wg, sink, cyc);
        run<8, 24, 48, 1, 1>("8 waves, stage, 4 chains", src, region, n_wg, sink, cyc);
        // the 16-chain shape (RB = 4: 16 MFMAs per tile) - what one member of a 4-CU group would run on its quarter of the
        // columns (tools/ubench/xcu.hip): is a 16-chain stream stage MFMA-bound?
        run<4, 32, 96, 0, 4>("4 waves, free, 16 chains", src, region, n_wg, sink, cyc);
        run<4, 32, 96, 1, 4>("4 waves, stage, 16 chains", src, region, n_wg, sink, cyc);
        run<4, 16, 32, 1, 4>("4 waves, stage, 16 chains", src, region, n_wg, sink, cyc);
    }
    return 0;
}
