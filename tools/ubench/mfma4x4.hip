// Lane layout check of v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks, D[b][i][j] += A[b][i] * B[b][j].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.f + (l % 4) + 10.f * (l / 4);        // A value identifies (block, i = l % 4)
    const float b = 100.f * (1 + l % 4) + 1000.f * (l / 4) * 0.f + 0.5f * (l / 4);   // B identifies (block, j = l % 4)
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc((void**)&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // hypothesis: lane l = 4 b + j holds column j of block b, VGPR r = row i = r:  D = A[b][r] * B[b][j]
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4;
            const float A = 1.f + r + 10.f * blk, B = 100.f * (1 + j) + 0.5f * blk;
            if (h[l * 4 + r] != A * B) { if (bad < 8) printf("lane %d vgpr %d: got %g expected %g\n", l, r, h[l * 4 + r], A * B); ++bad; }
        }
    printf("mismatches under hypothesis (lane = 4 b + col, vgpr = row): %d\n", bad);
    return 0;
}
