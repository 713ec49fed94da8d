// Micro-benchmark: how fast can ONE workgroup of 4 waves (one per SIMD) stream weights L2/MALL -> VGPR?
// Each wave issues 16-byte-per-lane loads of consecutive 1-KiB blocks (the B-operand stream of the flow kernels),
// NB loads in flight, with different cache-policy modifiers.  Prints bytes / shader clock / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void ld(f32x4& v, unsigned voff, const float4* sbase) {      // SGPR base + 32-bit lane offset
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase));
    else if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v) : "v"(voff), "s"(sbase));
    else if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc0" : "=v"(v) : "v"(voff), "s"(sbase));
    else if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(voff), "s"(sbase));
    else asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(voff), "s"(sbase));
}

// SHARE 0: every workgroup its own region; 1: all workgroups the same region, in lockstep; 2: the same region, each
// workgroup starting at a different offset (rotated by blockIdx)
template <int MODE, int NB, int SHARE>
__global__ __launch_bounds__(256) void k_stream_sh(const float4* __restrict__ src, long n_blocks_per_wave, long stride_blocks,
                                                   float* __restrict__ sink, long long* __restrict__ cycles) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float4* base = src + ((size_t)wave * stride_blocks) * 64;
    const long rot = SHARE == 2 ? ((long)blockIdx.x * 37 * NB) % n_blocks_per_wave / NB * NB : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (long b0 = 0; b0 < n_blocks_per_wave; b0 += NB) {
        long b = b0 + rot; if (b >= n_blocks_per_wave) b -= n_blocks_per_wave;
        f32x4 r[NB];
        const float4* sb = base + (size_t)b * 64;
#pragma unroll
        for (int i = 0; i < NB; ++i) ld<MODE>(r[i], (unsigned)((i * 64 + lane) * 16), sb);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[i]) : "n"(NB - 1 - i));
            acc += r[i];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) sink[blockIdx.x * 4 + wave] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NB, int SHARE>
static void run_shared(const char* name, const float4* src, size_t region_bytes, int n_wg, float* sink, long long* cyc) {
    const long per_wave = (long)(region_bytes / 1024 / 4) / NB * NB;      // 4 waves split the region
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL((k_stream_sh<0, NB, SHARE>), dim3(n_wg), dim3(256), 0, 0, src, per_wave, per_wave, sink, cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(n_wg);
    (void)hipMemcpy(h.data(), cyc, n_wg * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= n_wg;
    printf("%-28s NB=%2d  %3d workgroups  region %6.1f MB : %6.1f B/clk per CU\n", name, NB, n_wg, region_bytes / 1e6,
           per_wave * 4 * 1024.0 / mean);
}

template <int MODE, int NB>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ src, long n_blocks_per_wave, long stride_blocks,
                                                float* __restrict__ sink, long long* __restrict__ cycles) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float4* base = src + ((size_t)(blockIdx.x * 4 + wave) * stride_blocks) * 64;      // wave-uniform
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (long b = 0; b < n_blocks_per_wave; b += NB) {
        f32x4 r[NB];
        const float4* sb = base + (size_t)b * 64;
#pragma unroll
        for (int i = 0; i < NB; ++i) ld<MODE>(r[i], (unsigned)((i * 64 + lane) * 16), sb);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[i]) : "n"(NB - 1 - i));      // loads return in order
            acc += r[i];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) sink[blockIdx.x * 4 + wave] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE, int NB>
static void run(const char* name, const float4* src, size_t buf_bytes, int n_wg, float* sink, long long* cyc) {
    // every wave streams its own slice of the buffer (the flow kernels: every wave owns its columns), 3 passes
    const long blocks_total = (long)(buf_bytes / 1024);
    const long per_wave = (blocks_total / (n_wg * 4)) / NB * NB;
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL((k_stream<MODE, NB>), dim3(n_wg), dim3(256), 0, 0, src, per_wave, per_wave, sink, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(n_wg);
    hipMemcpy(h.data(), cyc, n_wg * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= n_wg;
    printf("%-28s NB=%2d  %3d workgroups  buffer %6.1f MB : %6.1f B/clk per CU\n", name, NB, n_wg, buf_bytes / 1e6,
           per_wave * 4 * 1024.0 / mean);
}

int main() {
    const size_t big = 64u << 20;
    float4* src; float* sink; long long* cyc;
    hipMalloc((void**)&src, big); hipMemset(src, 0, big);
    hipMalloc((void**)&sink, 4096 * 4); hipMalloc((void**)&cyc, 1024 * 8);
    for (size_t region : {(size_t)1 << 20, (size_t)10 << 20}) {     // all workgroups read the SAME weights (the flow kernels)
        for (int n_wg : {1, 8, 64, 256}) {
            run_shared<25, 1>("same region, lockstep", src, region, n_wg, sink, cyc);
            run_shared<25, 2>("same region, rotated", src, region, n_wg, sink, cyc);
        }
    }
    for (size_t bytes : {(size_t)12 << 20}) {
        for (int n_wg : {1, 64}) {
            run<0, 25>("own region, plain", src, bytes, n_wg, sink, cyc);
            run<1, 25>("own region, nt", src, bytes, n_wg, sink, cyc);
        }
    }
    return 0;
}
