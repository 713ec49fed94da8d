// dev check: v_mfma_f32_4x4x4_16b_bf16 with vdst overlapping srcB, registers fixed by hand
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ unsigned short bf(float v) { return (unsigned short)(__float_as_uint(v) >> 16); }
__global__ void k(const unsigned* tiles, float* o, int mode) {
    const int l = threadIdx.x, i = l & 3;
    s16x4 a;
    for (int kk = 0; kk < 4; ++kk) a[kk] = (short)bf((float)(i + 1) * (kk + 1));
    float r0, r1, r2, r3;
    if (mode == 0)       // dst a[0:3] overlaps srcB a[2:3]
        asm volatile("global_load_dwordx4 a[0:3], %4, off\n\ts_waitcnt vmcnt(0)\n\ts_nop 7\n\t"
                     "v_mfma_f32_4x4x4_16b_bf16 a[0:3], %5, a[2:3], 0\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(tiles + l * 4), "v"(a) : "a0", "a1", "a2", "a3", "memory");
    else                 // disjoint: dst a[4:7]
        asm volatile("global_load_dwordx4 a[0:3], %4, off\n\ts_waitcnt vmcnt(0)\n\ts_nop 7\n\t"
                     "v_mfma_f32_4x4x4_16b_bf16 a[4:7], %5, a[2:3], 0\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\tv_accvgpr_read_b32 %2, a6\n\tv_accvgpr_read_b32 %3, a7"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(tiles + l * 4), "v"(a)
                     : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "memory");
    o[l * 4 + 0] = r0; o[l * 4 + 1] = r1; o[l * 4 + 2] = r2; o[l * 4 + 3] = r3;
}
int main() {
    unsigned h_t[256];
    for (int l = 0; l < 64; ++l) {
        int j = l & 3;
        for (int w = 0; w < 4; ++w) {
            float f0 = (float)((j + 1) + 4 * (2 * w)), f1 = (float)((j + 1) + 4 * (2 * w + 1));
            h_t[l * 4 + w] = ((*(unsigned*)&f0) >> 16) | (((*(unsigned*)&f1) >> 16) << 16);
        }
    }
    unsigned* dt; float* d; hipMalloc(&dt, sizeof(h_t)); hipMalloc(&d, 1024); hipMemcpy(dt, h_t, sizeof(h_t), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dt, d, mode);
        float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            float e = 0; int j = l & 3;
            for (int kk = 0; kk < 4; ++kk) e += (float)(r + 1) * (kk + 1) * (float)((j + 1) + 4 * (4 + kk));
            bad += h[l * 4 + r] != e;
            if (l < 2) printf("%g(%g) ", h[l * 4 + r], e);
        }
        printf("\nmode %d (%s) mismatches %d\n", mode, mode == 0 ? "vdst overlaps srcB" : "disjoint", bad);
    }
    return 0;
}
