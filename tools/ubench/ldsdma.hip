// How fast can wave(s) of one workgroup move an L2 / MALL-resident stream into LDS with global_load_lds_dwordx4 (1 KiB per
// wave-instruction)?  NW waves each issue N requests back to back (KF in flight), 256 workgroups read the same 10 MB.
// Variants: M0 saved / restored around every request (the guide's recipe), M0 written once per request, M0 fixed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int V>
__device__ __forceinline__ void glds16(const float4* gsrc, unsigned lds_dst) {
    if constexpr (V == 0) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    } else if constexpr (V == 1) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
    } else {
        asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(gsrc) : "memory");
    }
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int NW, int KF, int V, int REG>
__global__ __launch_bounds__(64 * NW) void k_dma(const float4* __restrict__ src, int n, size_t wave_stride, float* sink, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned base = (unsigned)(size_t)lds + wave * 16 * 1024;
    const float4* p = src + (size_t)wave * wave_stride + lane;
    if constexpr (V == 2) asm volatile("s_mov_b32 m0, %0" :: "s"(base) : "memory");
    __syncthreads();
    float4 accv = make_float4(0, 0, 0, 0);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 ring[16];                                    // destinations of the register variant: accumulation registers, kept live
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (REG) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(ring[j]) : "v"(p + (size_t)(i + j) * 64) : "memory");
                vm_wait<KF - 1>();
            } else if constexpr (V == 3) {                                   // scalar base + lane offset + immediate (moves the LDS address too)
                const float4* sb = src + (size_t)wave * wave_stride + (size_t)(i + (j & ~3)) * 64;
                const unsigned voff = lane * 16;
                const unsigned dst = base + (j & ~3) * 1024;
                if (j % 4 == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0" :: "v"(voff), "s"(sb), "s"(dst) : "memory");
                if (j % 4 == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(voff), "s"(sb), "s"(dst) : "memory");
                if (j % 4 == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(voff), "s"(sb), "s"(dst) : "memory");
                if (j % 4 == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff), "s"(sb), "s"(dst) : "memory");
                vm_wait<KF - 1>();
            } else {
                glds16<V>(p + (size_t)(i + j) * 64, base + j * 1024);
                vm_wait<KF - 1>();
            }
        }
    }
    vm_wait<0>();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { asm volatile("s_waitcnt vmcnt(0)" : "+a"(ring[j])); accv.x += ring[j].x; }
    }
    sink[blockIdx.x * 64 * NW + tid] = lds[tid] + accv.x;
    if (lane == 0) cycles[blockIdx.x * NW + wave] = t1 - t0;
}

template <int NW, int KF, int V, int REG>
static void run(const char* name, const float4* src, size_t region, int n_wg, float* sink, long long* cyc) {
    const size_t wave_bytes = region / NW;
    const int n = (int)(wave_bytes / 1024) / 16 * 16;
    auto kern = k_dma<NW, KF, V, REG>;
    const size_t lds = (size_t)NW * 16 * 1024;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(n_wg), dim3(64 * NW), lds, 0, src, n, wave_bytes / 16, sink, cyc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s failed: %s\n", name, hipGetErrorString(e)); return; }
    std::vector<long long> h((size_t)n_wg * NW);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int g = 0; g < n_wg; ++g) { long long mx = 0; for (int w = 0; w < NW; ++w) mx = h[(size_t)g * NW + w] > mx ? h[(size_t)g * NW + w] : mx; mean += (double)mx; }
    mean /= n_wg;
    printf("%-26s NW=%d KF=%2d V=%d: %6.1f cycles per request and wave  %5.1f B/clk/CU\n", name, NW, KF, V, mean / n, (double)n * NW * 1024 / mean);
    fflush(stdout);
}

int main() {
    const size_t region = 10u << 20;
    float4* src; float* sink; long long* cyc;
    (void)hipMalloc((void**)&src, region + (2u << 20)); (void)hipMemset(src, 0, region + (2u << 20));
    (void)hipMalloc((void**)&sink, 1024 * 1024 * 4); (void)hipMalloc((void**)&cyc, 1024 * 16 * 8);
    run<1, 16, 1, 0>("lds dma, m0 write", src, region, 256, sink, cyc);
    run<1, 16, 3, 0>("lds dma, saddr + imm", src, region, 256, sink, cyc);
    run<1, 8, 3, 0>("lds dma, saddr + imm", src, region, 256, sink, cyc);
    run<4, 8, 3, 0>("lds dma, saddr + imm", src, region, 256, sink, cyc);
    run<4, 8, 1, 0>("lds dma, m0 write", src, region, 256, sink, cyc);
    run<8, 8, 1, 0>("lds dma, m0 write", src, region, 256, sink, cyc);
    return 0;
}
