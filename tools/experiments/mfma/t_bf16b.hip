// dev check: v_mfma_f32_4x4x4_16b_bf16 with vdst overlapping srcB (what hipcc allocates for a consumed ring tile)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ unsigned short bf(float v) { return (unsigned short)(__float_as_uint(v) >> 16); }
__global__ void k(const unsigned* tiles, float* o, int mode) {
    const int l = threadIdx.x, i = l & 3;
    s16x4 a;
    for (int kk = 0; kk < 4; ++kk) a[kk] = (short)bf((float)(i + 1) * (kk + 1));
    f32x4 tile;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=a"(tile) : "v"(tiles + l * 4));
    f32x4 acc;
    {
        acc = (f32x4){0, 0, 0, 0};
        f32x2 hi = __builtin_shufflevector(tile, tile, 2, 3);
        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, __builtin_bit_cast(s16x4, hi), acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) o[l * 4 + r] = acc[r];
}
int main() {
    unsigned h_t[256];
    for (int l = 0; l < 64; ++l) {
        int j = l & 3;
        float b[8];
        for (int kk = 0; kk < 8; ++kk) b[kk] = (float)((j + 1) + 4 * kk);   // B[k][j]
        for (int w = 0; w < 4; ++w) {
            unsigned lo, hi; float f0 = b[2 * w], f1 = b[2 * w + 1];
            lo = (*(unsigned*)&f0) >> 16; hi = (*(unsigned*)&f1) >> 16;
            h_t[l * 4 + w] = lo | (hi << 16);
        }
    }
    unsigned* dt; float* d; hipMalloc(&dt, sizeof(h_t)); hipMalloc(&d, 1024); hipMemcpy(dt, h_t, sizeof(h_t), hipMemcpyHostToDevice);
    for (int mode = 1; mode >= 1; --mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dt, d, mode);
        float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            float e = 0; int j = l & 3;
            for (int kk = 0; kk < 4; ++kk) e += (float)(r + 1) * (kk + 1) * (float)((j + 1) + 4 * (4 + kk));   // second quad (.zw)
            bad += h[l * 4 + r] != e;
            if (l < 2) printf("%g(%g) ", h[l * 4 + r], e);
        }
        printf("\nmode %d mismatches %d\n", mode, bad);
    }
    return 0;
}
