// Correctness of global_load_lds_dwordx4 on gfx950 (saddr + voffset + imm form; M0 = LDS byte address of the tile):
// (a) the lane -> LDS layout (lane * 16 bytes), (b) the immediate offset moving BOTH addresses, (c) LDS addresses above 64 KB,
// (d) s_waitcnt vmcnt(0) as the only synchronisation before the wave's own ds_read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, float4* out, unsigned lds_off_bytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned base = (unsigned)(size_t)lds + lds_off_bytes + wave * 5 * 1024;
    const float4* sb = src + (size_t)wave * 5 * 64;
    const unsigned voff0 = lane * 16, voff1 = lane * 16 + 4096;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:0\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff0), "s"(sb), "s"(base) : "memory");
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:0" :: "v"(voff1), "s"(sb), "s"(base + 4096) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float4* l4 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds) + lds_off_bytes) + wave * 5 * 64;
    for (int g = 0; g < 5; ++g) out[(size_t)blockIdx.x * 1280 + (wave * 5 + g) * 64 + lane] = l4[g * 64 + lane];
}
int main() {
    const int n = 4 * 5 * 64;
    std::vector<float4> h(n);
    for (int i = 0; i < n; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *src, *out;
    (void)hipMalloc((void**)&src, n * 16); (void)hipMalloc((void**)&out, 4 * n * 16);
    (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    for (unsigned off : {0u, 40960u, 98304u, 140000u & ~15u}) {
        const size_t bytes = off + 20 * 1024;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        (void)hipMemset(out, 0, 4 * n * 16);
        hipLaunchKernelGGL(k, dim3(4), dim3(256), bytes, 0, src, out, off);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float4> r(4 * n);
        (void)hipMemcpy(r.data(), out, 4 * n * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int b = 0; b < 4; ++b) for (int i = 0; i < n; ++i) { const float4 a = r[b * n + i], x = h[i]; if (a.x != x.x || a.y != x.y || a.z != x.z || a.w != x.w) ++bad; }
        printf("lds offset %6u: %s, %d of %d float4 wrong (first: %g %g)\n", off, hipGetErrorString(e), bad, 4 * n, r[0].x, r[64].x);
    }
    return 0;
}
