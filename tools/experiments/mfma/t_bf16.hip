// dev check: operand layout of v_mfma_f32_4x4x4_16b_bf16 (block b = lane / 4: D[i][j] = sum_k A[i][k] B[k][j])
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ unsigned short bf(float v) { return (unsigned short)(__float_as_uint(v) >> 16); }
__global__ void k(float* o) {
    const int l = threadIdx.x, i = l & 3;
    // A[i][k] = 10 i + k + 1 ; B[k][j] = (k == 0) + 100 (j + 1) (k == 1)   (exact in bf16)
    s16x4 a, b;
    for (int kk = 0; kk < 4; ++kk) a[kk] = (short)bf((float)(10 * i + kk + 1));
    const int j = l & 3;
    b[0] = (short)bf(1.f); b[1] = (short)bf(100.f * (j + 1)); b[2] = 0; b[3] = 0;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) o[l * 4 + r] = acc[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected D[i][j] = A[i][0] * 1 + A[i][1] * 100 (j + 1) = (10 i + 1) + (10 i + 2) * 100 (j + 1): lane 4b + j, register i
    int bad = 0;
    for (int l = 0; l < 8; ++l) { for (int r = 0; r < 4; ++r) { float e = (10 * r + 1) + (10 * r + 2) * 100.f * ((l & 3) + 1); printf("%g(%g) ", h[l * 4 + r], e); bad += h[l*4+r] != e; } printf("\n"); }
    printf("mismatches in first 8 lanes: %d\n", bad);
    return 0;
}
