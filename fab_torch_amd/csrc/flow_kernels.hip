// RealNVP flow: parameter packing, log_prob (+ d/dx) and sampling kernels + their C ABI.
#include "flow_device.h"
#include "flow_r4.h"
#include "flow_r4f.h"
#include "flow_r8.h"
#include "launch.h"
#include <stdlib.h>

namespace fab {

// ------------------------------------------------------------------------------------------------
// InvertibleAffine assembly (normflows InvertibleAffine._assemble_W): one workgroup per layer.
//   W    = (P @ (tril(L,-1)+I)) @ (triu(U,1) + diag(sign_S * exp(log_S)))
//   Winv = (fl32(inv64(Um)) @ fl32(inv64(Lm))) @ P^T
// Triangular inverses by substitution in float64, one column per thread.
// ------------------------------------------------------------------------------------------------
constexpr int LBATCH = 16;      // layers per launch (pointer tables travel as kernel arguments)
struct AffineTab {
    const float *L[LBATCH], *U[LBATCH], *logS[LBATCH], *signS[LBATCH], *P[LBATCH];
    const float *an_s[LBATCH], *an_t[LBATCH];              // ActNorm after the affine map (nullptr: none)
};
struct MlpTab {
    const float *w1[LBATCH], *b1[LBATCH], *w2[LBATCH], *b2[LBATCH], *w3[LBATCH], *b3[LBATCH];
};

// workgroup y handles layer k0 + y: W / Winv to the scratch area of the packed image, sum(log_S) to the layer block
__global__ __launch_bounds__(1024) void k_affine_assemble(FlowDims f, AffineTab tab, int k0, float* __restrict__ packed,
                                                         int with_inverse) {
    const int D = f.D, layer = k0 + blockIdx.x;
    const float* __restrict__ Lraw = tab.L[blockIdx.x];
    const float* __restrict__ Uraw = tab.U[blockIdx.x];
    const float* __restrict__ logS = tab.logS[blockIdx.x];
    const float* __restrict__ signS = tab.signS[blockIdx.x];
    const float* __restrict__ P = tab.P[blockIdx.x];
    const float* __restrict__ an_s = tab.an_s[blockIdx.x];
    const float* __restrict__ an_t = tab.an_t[blockIdx.x];
    float* __restrict__ Lblock = packed + (size_t)layer * f.layer_stride;
    float* __restrict__ Wout = packed + f.o_scratch + (size_t)layer * 2 * D * D;
    float* __restrict__ Winvout = Wout + D * D;
    float* __restrict__ logS_sum = packed + (size_t)layer * f.layer_stride + f.o_logS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* Xd = reinterpret_cast<double*>(smem_raw);          // [D][D] fp64 scratch
    float* Li = reinterpret_cast<float*>(Xd + D * D);          // [D][D]
    float* Ui = Li + D * D;                                    // [D][D]
    float* T = reinterpret_cast<float*>(Xd);                   // aliases Xd once the inverses are cast
    float* Ls = Ui + D * D;                                    // Lm, Um, P staged once (the loops below re-read them)
    float* Us = Ls + D * D;
    float* Ps = Us + D * D;
    const int tid = threadIdx.x;
    for (int e = tid; e < D * D; e += blockDim.x) {
        const int i = e / D, j = e % D;
        Ls[e] = i == j ? 1.f : (i > j ? Lraw[e] : 0.f);
        Us[e] = i == j ? signS[i] * expf(logS[i]) : (i < j ? Uraw[e] : 0.f);
        Ps[e] = P[e];
    }
    __syncthreads();
    auto Lm = [&](int i, int j) -> float { return Ls[i * D + j]; };
    auto Um = [&](int i, int j) -> float { return Us[i * D + j]; };
    if (with_inverse) {                                        // W^-1: only the sampling direction needs it
    // inverse of unit-lower Lm, column j
    if (tid < D) {
        const int j = tid;
        for (int i = 0; i < D; ++i) Xd[i * D + j] = 0.0;
        Xd[j * D + j] = 1.0;
        for (int i = j + 1; i < D; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s += (double)Lm(i, k) * Xd[k * D + j];
            Xd[i * D + j] = -s;
        }
    }
    __syncthreads();
    for (int e = tid; e < D * D; e += blockDim.x) Li[e] = (float)Xd[e];
    __syncthreads();
    // inverse of upper Um, column j
    if (tid < D) {
        const int j = tid;
        for (int i = 0; i < D; ++i) Xd[i * D + j] = 0.0;
        Xd[j * D + j] = 1.0 / (double)Um(j, j);
        for (int i = j - 1; i >= 0; --i) {
            double s = 0.0;
            for (int k = i + 1; k <= j; ++k) s += (double)Um(i, k) * Xd[k * D + j];
            Xd[i * D + j] = -s / (double)Um(i, i);
        }
    }
    __syncthreads();
    for (int e = tid; e < D * D; e += blockDim.x) Ui[e] = (float)Xd[e];
    __syncthreads();
    for (int e = tid; e < D * D; e += blockDim.x) {           // T = Ui @ Li
        const int i = e / D, j = e % D;
        float s = 0.f;
        for (int k = 0; k < D; ++k) s = fmaf(Ui[i * D + k], Li[k * D + j], s);
        T[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < D * D; e += blockDim.x) {           // Winv = T @ P^T
        const int i = e / D, j = e % D;
        float s = 0.f;
        for (int k = 0; k < D; ++k) s = fmaf(T[i * D + k], Ps[j * D + k], s);
        Winvout[e] = an_s ? s * expf(an_s[j]) : s;            // ActNorm.forward after the map: W^-1 diag(e^s)
    }
    __syncthreads();
    }
    for (int e = tid; e < D * D; e += blockDim.x) {           // Li <- P @ Lm
        const int i = e / D, j = e % D;
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < D; ++k) s = fmaf(Ps[i * D + k], Lm(k, j), s);
        Li[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < D * D; e += blockDim.x) {           // W = (P Lm) @ Um
        const int i = e / D, j = e % D;
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < D; ++k) s = fmaf(Li[i * D + k], Um(k, j), s);
        Wout[e] = an_s ? expf(-an_s[i]) * s : s;              // ActNorm.inverse before the map: diag(e^-s) W
    }
    __syncthreads();
    if (tid < 64) {                                           // additive terms (zero without ActNorm / beyond D)
        float ac = 0.f, at = 0.f;
        if (an_s && tid < D) {
            for (int i = 0; i < D; ++i) ac = fmaf(-an_t[i], Wout[i * D + tid], ac);    // -(t e^-s) @ W = -t @ W'
            at = an_t[tid];
        }
        Lblock[f.o_ac + tid] = ac;
        Lblock[f.o_at + tid] = at;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int k = 0; k < D; ++k) s += logS[k] - (an_s ? an_s[k] : 0.f);
        *logS_sum = s;
    }
}

struct LayerSrc {
    const float *w1, *b1, *w2, *b2, *w3, *b3, *W, *Winv;
};

// decode an offset inside a tiled K x N matrix -> (k, n)
__device__ __forceinline__ void tile_kn(int off, int KB, int& k, int& n) {
    const int tile = off >> 8, within = off & 255;
    const int lane = within >> 2, tt = within & 3;
    const int c = tile / KB, S = tile % KB;
    k = 16 * S + 4 * (lane >> 4) + tt;
    n = 16 * c + (lane & 15);
}

// packed coupling-parameter column/row p in [0, 2*DOp) -> original MLP output index (interleaved
// shift/scale: param[:, 0::2] = shift, param[:, 1::2] = scale) or -1 for padding
__device__ __forceinline__ int prm_orig(int p, int DO, int DOp) {
    if (p < DOp) return p < DO ? 2 * p : -1;
    const int j = p - DOp;
    return j < DO ? 2 * j + 1 : -1;
}

// off_begin > 0 (the training image of a flow whose tape forward runs on the 8-chain stream tiles): only the bias blocks
// [o_b1, o_logS) - what r8_load_heads reads - and, from block (0, 0), the base distribution (one launch fewer)
__global__ __launch_bounds__(256) void k_pack_layer(FlowDims f, MlpTab tab, int k0, float* __restrict__ packed, int off_begin,
                                                    const float* __restrict__ loc, const float* __restrict__ log_scale) {
    const int D = f.D, d = f.d, DO = f.DO, W = f.W;
    const int y = blockIdx.y, layer = k0 + y;
    if (loc && blockIdx.x == 0 && y == 0 && (int)threadIdx.x < f.Dp) {
        const int j = threadIdx.x;
        packed[f.o_base + j] = j < f.D ? loc[j] : 0.f;
        packed[f.o_base + f.Dp + j] = j < f.D ? log_scale[j] : 0.f;
    }
    float* __restrict__ dst = packed + (size_t)layer * f.layer_stride;
    const float* Wm = packed + f.o_scratch + (size_t)layer * 2 * D * D;
    const LayerSrc s{tab.w1[y], tab.b1[y], tab.w2[y], tab.b2[y], tab.w3[y], tab.b3[y], Wm, Wm + D * D};
    for (int off = off_begin + blockIdx.x * blockDim.x + threadIdx.x; off < f.o_logS; off += gridDim.x * blockDim.x) {
        float v = 0.f;
        int k, n;
        if (off < f.o_AWT) {                       // AW: B[k][n] = W[k][n]
            tile_kn(off - f.o_AW, f.KBD, k, n);
            if (k < D && n < D) v = s.W[k * D + n];
        } else if (off < f.o_AWI) {                // AWT: B[k][n] = W[n][k]
            tile_kn(off - f.o_AWT, f.KBD, k, n);
            if (k < D && n < D) v = s.W[n * D + k];
        } else if (off < f.o_W1) {                 // AWI: B[k][n] = Winv[k][n]
            tile_kn(off - f.o_AWI, f.KBD, k, n);
            if (k < D && n < D) v = s.Winv[k * D + n];
        } else if (off < f.o_W2) {                 // W1: B[k][n] = w1[n][k]
            tile_kn(off - f.o_W1, f.KBd, k, n);
            if (k < d && n < W) v = s.w1[n * d + k];
        } else if (off < f.o_W3) {                 // W2: B[k][n] = w2[n][k]
            tile_kn(off - f.o_W2, f.KBW, k, n);
            if (k < W && n < W) v = s.w2[n * W + k];
        } else if (off < f.o_W3T) {                // W3: B[k][p] = w3[orig(p)][k]
            tile_kn(off - f.o_W3, f.KBW, k, n);
            const int o = prm_orig(n, DO, f.DOp);
            if (k < W && o >= 0) v = s.w3[o * W + k];
        } else if (off < f.o_W2T) {                // W3T: B[p][n] = w3[orig(p)][n]
            tile_kn(off - f.o_W3T, f.KBO, k, n);
            const int o = prm_orig(k, DO, f.DOp);
            if (o >= 0 && n < W) v = s.w3[o * W + n];
        } else if (off < f.o_W1T) {                // W2T: B[k][n] = w2[k][n]
            tile_kn(off - f.o_W2T, f.KBW, k, n);
            if (k < W && n < W) v = s.w2[k * W + n];
        } else if (off < f.o_b1) {                 // W1T: B[k][n] = w1[k][n]
            tile_kn(off - f.o_W1T, f.KBW, k, n);
            if (k < W && n < d) v = s.w1[k * d + n];
        } else if (off < f.o_b2) {
            const int j = off - f.o_b1;
            if (j < W) v = s.b1[j];
        } else if (off < f.o_b3) {
            const int j = off - f.o_b2;
            if (j < W) v = s.b2[j];
        } else {
            const int o = prm_orig(off - f.o_b3, DO, f.DOp);
            if (o >= 0) v = s.b3[o];
        }
        dst[off] = v;
    }
    if (off_begin > 0) return;
    // (W'^-1)^T, appended after the bf16 images: B[k][n] = Winv[n][k]
    for (int off = blockIdx.x * blockDim.x + threadIdx.x; off < f.KBD * f.NTD * 256; off += gridDim.x * blockDim.x) {
        int k, n;
        tile_kn(off, f.KBD, k, n);
        dst[f.o_AWIT + off] = (k < D && n < D) ? s.Winv[n * D + k] : 0.f;
    }
}

// fast mode: bf16 images of W2 (B[k][n] = w2[n][k]) and W2^T (B[k][n] = w2[k][n]) in v_mfma_f32_16x16x32_bf16 B-operand
// tiles (flow_device.h): float slot `off` of an image holds the bf16 pair (j, j + 1), j = 2 (off & 3), of lane
// (off >> 2) & 63 of tile off >> 8 = c * KB2 + S.  Round to nearest even, like v_cvt_pk_bf16_f32 on the activations.
__device__ __forceinline__ unsigned bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;        // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__global__ __launch_bounds__(256) void k_pack_bf16(FlowDims f, MlpTab tab, int k0, float* __restrict__ packed) {
    const int y = blockIdx.y, layer = k0 + y, W = f.W, half = f.Wp * f.Wp / 2, KB2 = f.Wp / 32;
    const float* __restrict__ w2 = tab.w2[y];
    unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(packed + (size_t)layer * f.layer_stride + f.o_W2h);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * half; e += gridDim.x * blockDim.x) {
        const bool tr = e >= half;
        const int off = tr ? e - half : e;
        const int tile = off >> 8, lane = (off >> 2) & 63, j = 2 * (off & 3);
        const int c = tile / KB2, S = tile % KB2;
        const int k = 32 * S + 8 * (lane >> 4) + j, n = 16 * c + (lane & 15);
        float v0 = 0.f, v1 = 0.f;
        if (n < W) {
            if (k < W) v0 = tr ? w2[k * W + n] : w2[n * W + k];
            if (k + 1 < W) v1 = tr ? w2[(k + 1) * W + n] : w2[n * W + k + 1];
        }
        dst[e] = bf16_rne(v0) | (bf16_rne(v1) << 16);
    }
}

__global__ void k_pack_base(FlowDims f, const float* __restrict__ loc, const float* __restrict__ log_scale,
                            float* __restrict__ dst) {
    const int j = threadIdx.x;
    if (j < f.Dp) {
        dst[f.o_base + j] = j < f.D ? loc[j] : 0.f;
        dst[f.o_base + f.Dp + j] = j < f.D ? log_scale[j] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// tile helpers shared with the transition kernels
// ------------------------------------------------------------------------------------------------
template <int NTWM, bool GRAD, bool FAST>
__device__ __forceinline__ void flow_log_prob_body(const FlowDims& f, const FlowLds& l, const float* __restrict__ packed,
                                                   const float* __restrict__ x, float* __restrict__ log_q,
                                                   float* __restrict__ grad, long B, float* lds) {
    Tid t;
    const long row0 = (long)blockIdx.x * ROWS;
    for (int e = t.tid; e < ROWS * l.PS; e += NTHREADS) lds[l.o_DP + e] = 0.f;
    for (int e = t.tid; e < ROWS * l.DS; e += NTHREADS) {
        const int r = e / l.DS, j = e % l.DS;
        const long g = row0 + r;
        lds[l.o_U0 + e] = (j < f.D && g < B) ? x[g * f.D + j] : 0.f;
    }
    __syncthreads();
    int goff = 0;
    const float lq = flow_log_prob_tile<NTWM, GRAD, false, FAST>(f, l, packed, lds, t, &goff);
    if (t.c == 0 && row0 + t.row < B) log_q[row0 + t.row] = lq;
    if (GRAD) {
        for (int e = t.tid; e < ROWS * f.D; e += NTHREADS) {
            const int r = e / f.D, j = e % f.D;
            const long g = row0 + r;
            if (g < B) grad[g * f.D + j] = lds[goff + r * l.DS + j];
        }
    }
}

template <int NTWM, bool GRAD>
__global__ __launch_bounds__(NTHREADS) void k_flow_log_prob(FlowDims f, FlowLds l, const float* __restrict__ packed,
                                                            const float* __restrict__ x, float* __restrict__ log_q,
                                                            float* __restrict__ grad, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    flow_log_prob_body<NTWM, GRAD, false>(f, l, packed, x, log_q, grad, B, lds);
}
// fast mode (bf16 W x W GEMMs): with the gradient only - a plain density evaluation stays fp32
template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_flow_log_prob_fast(FlowDims f, FlowLds l, const float* __restrict__ packed,
                                                                 const float* __restrict__ x, float* __restrict__ log_q,
                                                                 float* __restrict__ grad, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    flow_log_prob_body<NTWM, true, true>(f, l, packed, x, log_q, grad, B, lds);
}

template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_flow_sample(FlowDims f, FlowLds l, const float* __restrict__ packed,
                                                          const float* __restrict__ eps, float* __restrict__ x,
                                                          float* __restrict__ log_q, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid t;
    const long row0 = (long)blockIdx.x * ROWS;
    for (int e = t.tid; e < ROWS * l.DS; e += NTHREADS) {
        const int r = e / l.DS, j = e % l.DS;
        const long g = row0 + r;
        lds[l.o_U0 + e] = (j < f.D && g < B) ? eps[g * f.D + j] : 0.f;
    }
    __syncthreads();
    int xoff = 0;
    const float lq = flow_sample_tile<NTWM>(f, l, packed, lds, t, &xoff);
    if (t.c == 0 && row0 + t.row < B) log_q[row0 + t.row] = lq;
    for (int e = t.tid; e < ROWS * f.D; e += NTHREADS) {
        const int r = e / f.D, j = e % f.D;
        const long g = row0 + r;
        if (g < B) x[g * f.D + j] = lds[xoff + r * l.DS + j];
    }
}

// the same on 4-chain tiles (flow_r4.h: flow_sample_r4s; FUSED: flow_r4f.h: flow_sample_r4f) - 256 workgroups for 1024 chains
// instead of 64
template <int NTWM, bool FUSED>
__global__ __launch_bounds__(NTHREADS) void k_flow_sample_r4(FlowDims f, R4Dims rd, R4Lds l, const float* __restrict__ packed,
                                                             const float* __restrict__ eps, float* __restrict__ x,
                                                             float* __restrict__ log_q, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid4 t4;
    const long row0 = (long)blockIdx.x * R4;
    for (int e = t4.tid; e < R4 * R4_DS; e += NTHREADS) {
        const int r = e / R4_DS, j = e % R4_DS;
        const long g = row0 + r;
        lds[l.o_X0 + e] = (j < f.D && g < B) ? eps[g * f.D + j] : 0.f;
    }
    if constexpr (FUSED)
        r4f_load_bias(packed + f.o_r4fb + (size_t)f.K * r4f_bias_stride(f.Wp), lds + l.o_BIAS, (f.K + 1) * r4f_bias_stride(f.Wp),
                      t4.tid);
    __syncthreads();
    int xoff = 0;
    float lq;
    if constexpr (FUSED) lq = flow_sample_r4f<NTWM>(f, l, packed, lds, t4, &xoff);
    else lq = flow_sample_r4s<NTWM>(f, rd, l, packed, lds, t4, &xoff);
    if (t4.tid < 64 && (t4.tid & 15) == 0 && row0 + (t4.tid >> 4) < B) log_q[row0 + (t4.tid >> 4)] = lq;
    for (int e = t4.tid; e < R4 * f.D; e += NTHREADS) {
        const int r = e / f.D, j = e % f.D;
        const long g = row0 + r;
        if (g < B) x[g * f.D + j] = lds[xoff + r * R4_DS + j];
    }
}

template <int NTWM>
static int launch_log_prob(const FlowDims& f, const float* packed, const float* x, float* log_q, float* grad,
                           long B, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div((int)B, ROWS)), block(NTHREADS);
    if (grad) {
        const FlowLds l = make_flow_lds(f, true);
        const size_t bytes = (size_t)l.total * 4;
        if (f.fast) {
            FAB_TRY(set_max_lds((const void*)k_flow_log_prob_fast<NTWM>, bytes));
            hipLaunchKernelGGL((k_flow_log_prob_fast<NTWM>), grid, block, bytes, st, f, l, packed, x, log_q, grad, B);
            return check_launch();
        }
        FAB_TRY(set_max_lds((const void*)k_flow_log_prob<NTWM, true>, bytes));
        hipLaunchKernelGGL((k_flow_log_prob<NTWM, true>), grid, block, bytes, st, f, l, packed, x, log_q, grad, B);
    } else {
        const FlowLds l = make_flow_lds(f, false);
        const size_t bytes = (size_t)l.total * 4;
        FAB_TRY(set_max_lds((const void*)k_flow_log_prob<NTWM, false>, bytes));
        hipLaunchKernelGGL((k_flow_log_prob<NTWM, false>), grid, block, bytes, st, f, l, packed, x, log_q, grad, B);
    }
    return check_launch();
}

// 4-chain-tile image (flow_r4.h): float4 tile (q, g) of a matrix B[K][N], lane l = { B[4 q + kk][64 g + l] } kk < 4,
// tiles q-major.  One workgroup column per layer (blockIdx.y), grid-stride over the layer block.
__global__ __launch_bounds__(256) void k_pack_r4(FlowDims f, R4Dims rd, MlpTab tab, int k0, float* __restrict__ packed) {
    const int D = f.D, d = f.d, DO = f.DO, W = f.W;
    const int y = blockIdx.y, layer = k0 + y;
    float* __restrict__ dst = packed + f.o_r4 + (size_t)layer * rd.layer_stride;
    const float* Wm = packed + f.o_scratch + (size_t)layer * 2 * D * D;       // W' (assembled, ActNorm folded)
    const float *w1 = tab.w1[y], *w2 = tab.w2[y], *w3 = tab.w3[y];
    for (int off = blockIdx.x * blockDim.x + threadIdx.x; off < rd.layer_stride; off += gridDim.x * blockDim.x) {
        int base, G;
        if (off < rd.o_AWT) { base = rd.o_AW; G = 1; }
        else if (off < rd.o_W1) { base = rd.o_AWT; G = 1; }
        else if (off < rd.o_W2) { base = rd.o_W1; G = rd.G; }
        else if (off < rd.o_W3) { base = rd.o_W2; G = rd.G; }
        else if (off < rd.o_W3T) { base = rd.o_W3; G = 1; }
        else if (off < rd.o_W2T) { base = rd.o_W3T; G = rd.G; }
        else if (off < rd.o_W1T) { base = rd.o_W2T; G = rd.G; }
        else { base = rd.o_W1T; G = 1; }
        const int e = off - base, kk = e & 3, l = (e >> 2) & 63, tile = e >> 8;
        int k, n;
        if (base == rd.o_W3 || base == rd.o_W1T) {           // dense narrow tiles (r4_dense_n16): NSUB k-quads side by side
            const int CW = base == rd.o_W3 ? 2 * f.DOp : pad16(d), NSUB = 64 / CW;
            k = 4 * (NSUB * tile + l / CW) + kk; n = l % CW;   // (tiles are wave-major, 4 NTW / NSUB per wave: quads stay in order)
        } else {
            const int g = tile % G, q = tile / G;
            k = 4 * q + kk; n = 64 * g + l;
        }
        float v = 0.f;
        if (base == rd.o_AW) { if (k < D && n < D) v = Wm[k * D + n]; }
        else if (base == rd.o_AWT) { if (k < D && n < D) v = Wm[n * D + k]; }
        else if (base == rd.o_W1) { if (k < d && n < W) v = w1[n * d + k]; }
        else if (base == rd.o_W2) { if (k < W && n < W) v = w2[n * W + k]; }
        else if (base == rd.o_W3) { const int o = prm_orig(n, DO, f.DOp); if (k < W && n < 2 * f.DOp && o >= 0) v = w3[o * W + k]; }
        else if (base == rd.o_W3T) { const int o = prm_orig(k, DO, f.DOp); if (k < 2 * f.DOp && o >= 0 && n < W) v = w3[o * W + n]; }
        else if (base == rd.o_W2T) { if (k < W && n < W) v = w2[k * W + n]; }
        else { if (k < W && n < d) v = w1[k * d + n]; }
        dst[off] = v;
    }
}

// 8-chain-tile image (flow_r8.h): ONE contiguous stream per wave - [layers K-1 .. 0 forward | layers 0 .. K-1 reverse | tail] -
// of 1-KiB tiles (lane = column, float4 = 4 consecutive k) in the order the wave consumes them, every layer and direction
// padded to r8_tiles_p(G) tiles (a multiple of the ring depth).  Forward [AW 4 | W1 4 (+1) | W2 16 G (+4 G) | W3 2 G],
// reverse [W3T 8 (+2) | W2T 16 G (+4 G) | W1T G | AWT 4]; N-split matrices: columns 64 wave + lane, all of K; (+..): the
// fifth column group of a 320-wide layer, N-split as well (r6): columns 256 + 16 wave + (lane & 15), all of K as DENSE tiles
// (4 k-quads side by side in the lane quarters); K-split (W3, W1T): this
// wave's quarter of K as DENSE tiles (2 / 4 k-quads side by side); the D x D maps (dense) are the same tiles for every wave.
__global__ __launch_bounds__(256) void k_pack_r8(FlowDims f, MlpTab tab, int k0, float* __restrict__ packed) {
    const int D = f.D, d = f.d, DO = f.DO, W = f.W, G = f.Wp / 64, EX = G - 4, K = f.K;
    const int TF = r8_tiles_fwd(G), TR = r8_tiles_rev(G), TP = r8_tiles_p(G);
    const int NQW = 16 * G, NQK = 4 * G;
    const int y = blockIdx.y, layer = k0 + y;
    float* __restrict__ img = packed + f.o_r8;
    const long WT = r8_wave_tiles(G, K);
    const float* Wm = packed + f.o_scratch + (size_t)layer * 2 * D * D;       // W' (assembled, ActNorm folded)
    const float *w1 = tab.w1[y], *w2 = tab.w2[y], *w3 = tab.w3[y];
    const int LT = NWAVE * 2 * TP * 256;                                        // floats of this layer, over waves and directions
    for (int off = blockIdx.x * blockDim.x + threadIdx.x; off < LT; off += gridDim.x * blockDim.x) {
        const int kk = off & 3, lane = (off >> 2) & 63;
        int tl = off >> 8;
        const int wave = tl / (2 * TP);
        tl -= wave * 2 * TP;
        const bool fwd = tl < TP;
        int ti = fwd ? tl : tl - TP;
        // position of the tile in the wave's stream: forward layers run K-1 .. 0, then reverse layers 0 .. K-1
        const long pos = fwd ? (long)(K - 1 - layer) * TP + ti : (long)(K + layer) * TP + ti;
        float* dstp = img + ((size_t)wave * WT + pos) * 256 + ((size_t)lane << 2) + kk;
        if (ti >= (fwd ? TF : TR)) { *dstp = 0.f; continue; }                  // the padding tile
        float v = 0.f;
        // (matrix, k-quad, column) of the tile: mat 0 AW / AWT, 1 W1 / W1T, 2 W2 / W2T, 3 W3 / W3T
        int mat, q, n;
        if (fwd) {
            if (ti < R8_TD) { mat = 0; q = 2 * ti + (lane >> 5); n = lane & 31; }               // dense: 2 k-quads x 32 columns
            else if ((ti -= R8_TD) < R8_Kd4) { mat = 1; q = ti; n = 64 * wave + lane; }
            else if ((ti -= R8_Kd4) < EX * (R8_Kd4 / 4)) { mat = 1; q = 4 * ti + (lane >> 4); n = 256 + 16 * wave + (lane & 15); }
            else if ((ti -= EX * (R8_Kd4 / 4)) < NQW) { mat = 2; q = ti; n = 64 * wave + lane; }
            else if ((ti -= NQW) < EX * NQK) { mat = 2; q = 4 * ti + (lane >> 4); n = 256 + 16 * wave + (lane & 15); }
            else { ti -= EX * NQK; mat = 3; q = NQK * wave + 2 * ti + (lane >> 5); n = lane & 31; }   // dense: 2 k-quads x (shift | scale)
            const int k = 4 * q + kk;
            if (mat == 0) { if (k < D && n < D) v = Wm[k * D + n]; }
            else if (mat == 1) { if (k < d && n < W) v = w1[n * d + k]; }
            else if (mat == 2) { if (k < W && n < W) v = w2[n * W + k]; }
            else { const int o = prm_orig(n, DO, f.DOp); if (k < W && n < 2 * f.DOp && o >= 0) v = w3[o * W + k]; }
        } else {
            if (ti < R8_Ko4) { mat = 3; q = ti; n = 64 * wave + lane; }
            else if ((ti -= R8_Ko4) < EX * (R8_Ko4 / 4)) { mat = 3; q = 4 * ti + (lane >> 4); n = 256 + 16 * wave + (lane & 15); }
            else if ((ti -= EX * (R8_Ko4 / 4)) < NQW) { mat = 2; q = ti; n = 64 * wave + lane; }
            else if ((ti -= NQW) < EX * NQK) { mat = 2; q = 4 * ti + (lane >> 4); n = 256 + 16 * wave + (lane & 15); }
            else if ((ti -= EX * NQK) < NQK / 4) { mat = 1; q = NQK * wave + 4 * ti + (lane >> 4); n = lane & 15; }   // dense: 4 k-quads x 16
            else { ti -= NQK / 4; mat = 0; q = 2 * ti + (lane >> 5); n = lane & 31; }
            const int k = 4 * q + kk;
            if (mat == 3) { const int o = prm_orig(k, DO, f.DOp); if (k < 2 * f.DOp && o >= 0 && n < W) v = w3[o * W + n]; }
            else if (mat == 2) { if (k < W && n < W) v = w2[k * W + n]; }
            else if (mat == 1) { if (k < W && n < d) v = w1[k * d + n]; }
            else { if (k < D && n < D) v = Wm[n * D + k]; }
        }
        *dstp = v;
    }
}

// 8-chain-tile image with fused stages (flow_r8.h, FlowDims::o_r8f): the same per-wave stream, per layer forward [AW 4 | W1' 8 (+2) |
// W2 16 G (+4 G) | W3 2 G], reverse [W3T 8 (+2) | W2T 16 G (+4 G) | W1'T 2 G | AWT 1], padded to r8f_tiles_p(G).  W1' = W'[:, :d] W1^T
// is a float64 product rounded once (the bias b1' comes from k_pack_r4f's density blocks); W1'T: this wave's quarter of the hidden
// rows as dense tiles (2 k-quads x 32 state columns); AWT: ONE dense tile per wave - rows 8 w .. 8 w + 7 of W'^T.
__global__ __launch_bounds__(256) void k_pack_r8f(FlowDims f, MlpTab tab, int k0, float* __restrict__ packed) {
    const int D = f.D, d = f.d, DO = f.DO, W = f.W, G = f.Wp / 64, EX = G - 4, K = f.K;
    const int TF = r8f_tiles_fwd(G), TR = r8f_tiles_rev(G), TP = r8f_tiles_p(G);
    const int NQW = 16 * G, NQK = 4 * G, NQ1 = R8_KD4;
    const int y = blockIdx.y, layer = k0 + y;
    float* __restrict__ img = packed + f.o_r8f;
    const long WT = r8f_wave_tiles(G, K);
    const float* Wm = packed + f.o_scratch + (size_t)layer * 2 * D * D;       // W' (assembled, ActNorm folded)
    const float *w1 = tab.w1[y], *w2 = tab.w2[y], *w3 = tab.w3[y];
    auto fusedW = [&](int r, int n) -> float {                                 // W1'[r][n] = sum_j W'[r][j] W1[n][j]
        if (r >= D || n >= W) return 0.f;
        double acc = 0.0;
        for (int j = 0; j < d; ++j) acc += (double)Wm[r * D + j] * (double)w1[n * d + j];
        return (float)acc;
    };
    const int LT = NWAVE * 2 * TP * 256;                                        // floats of this layer, over waves and directions
    for (int off = blockIdx.x * blockDim.x + threadIdx.x; off < LT; off += gridDim.x * blockDim.x) {
        const int kk = off & 3, lane = (off >> 2) & 63;
        int tl = off >> 8;
        const int wave = tl / (2 * TP);
        tl -= wave * 2 * TP;
        const bool fwd = tl < TP;
        int ti = fwd ? tl : tl - TP;
        const long pos = fwd ? (long)(K - 1 - layer) * TP + ti : (long)(K + layer) * TP + ti;
        float* dstp = img + ((size_t)wave * WT + pos) * 256 + ((size_t)lane << 2) + kk;
        if (ti >= (fwd ? TF : TR)) { *dstp = 0.f; continue; }                  // the padding tiles
        float v = 0.f;
        int mat, q, n;                                                          // mat 0 AW / AWT, 1 W1' / W1'T, 2 W2 / W2T, 3 W3 / W3T
        if (fwd) {
            if (ti < R8_TD) { mat = 0; q = 2 * ti + (lane >> 5); n = lane & 31; }
            else if ((ti -= R8_TD) < NQ1) { mat = 1; q = ti; n = 64 * wave + lane; }
            else if ((ti -= NQ1) < EX * (NQ1 / 4)) { mat = 1; q = 4 * ti + (lane >> 4); n = 256 + 16 * wave + (lane & 15); }
            else if ((ti -= EX * (NQ1 / 4)) < NQW) { mat = 2; q = ti; n = 64 * wave + lane; }
            else if ((ti -= NQW) < EX * NQK) { mat = 2; q = 4 * ti + (lane >> 4); n = 256 + 16 * wave + (lane & 15); }
            else { ti -= EX * NQK; mat = 3; q = NQK * wave + 2 * ti + (lane >> 5); n = lane & 31; }
            const int k = 4 * q + kk;
            if (mat == 0) { if (k < D && n < D) v = Wm[k * D + n]; }
            else if (mat == 1) v = fusedW(k, n);
            else if (mat == 2) { if (k < W && n < W) v = w2[n * W + k]; }
            else { const int o = prm_orig(n, DO, f.DOp); if (k < W && n < 2 * f.DOp && o >= 0) v = w3[o * W + k]; }
        } else {
            if (ti < R8_Ko4) { mat = 3; q = ti; n = 64 * wave + lane; }
            else if ((ti -= R8_Ko4) < EX * (R8_Ko4 / 4)) { mat = 3; q = 4 * ti + (lane >> 4); n = 256 + 16 * wave + (lane & 15); }
            else if ((ti -= EX * (R8_Ko4 / 4)) < NQW) { mat = 2; q = ti; n = 64 * wave + lane; }
            else if ((ti -= NQW) < EX * NQK) { mat = 2; q = 4 * ti + (lane >> 4); n = 256 + 16 * wave + (lane & 15); }
            else if ((ti -= EX * NQK) < NQK / 2) { mat = 1; q = NQK * wave + 2 * ti + (lane >> 5); n = lane & 31; }   // dense: 2 k-quads x 32
            else { mat = 0; q = 2 * wave + (lane >> 5); n = lane & 31; }
            const int k = 4 * q + kk;
            if (mat == 3) { const int o = prm_orig(k, DO, f.DOp); if (k < 2 * f.DOp && o >= 0 && n < W) v = w3[o * W + n]; }
            else if (mat == 2) { if (k < W && n < W) v = w2[k * W + n]; }
            else if (mat == 1) v = fusedW(n, k);                                // (W1')^T[k][n]: hidden row k, state column n
            else { if (k < D && n < D) v = Wm[n * D + k]; }
        }
        *dstp = v;
    }
}

// Stream image (flow_r4.h: R4Stream): the r4 tiles copied into the order a wave consumes them.  float4 index of a
// stream element = ((item * 4 + wave) * G + g) * 64 + lane, item = global item number (forward layers K-1 .. 0, then
// reverse layers 0 .. K-1, C = 4 G + 4 items each).  Source tiles come from the r4 image k_pack_r4 has just written.
__global__ __launch_bounds__(256) void k_pack_r4s(FlowDims f, R4Dims rd, float* __restrict__ packed) {
    const int G = rd.G, C = 4 * G + 4, K = f.K;
    const int nqD = rd.KD / 16;
    const long total4 = (long)(3 * K * C + 16) * 4 * G * 64;                 // float4 elements incl. the padding items
    const int D = f.D;
    const float4* r4 = reinterpret_cast<const float4*>(packed + f.o_r4);
    float4* dst = reinterpret_cast<float4*>(packed + f.o_r4s);
    const long LS4 = rd.layer_stride / 4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        long r = e >> 6;
        const int g = (int)(r % G); r /= G;
        const int w = (int)(r & 3);
        const long item = r >> 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (item < 2L * K * C) {
            const bool fwd = item < (long)K * C;
            const long li = fwd ? item : item - (long)K * C;
            const int layer = fwd ? K - 1 - (int)(li / C) : (int)(li / C);
            const int i = (int)(li % C);
            const float4* L = r4 + (long)layer * LS4;
            long src = -1;                                                    // float4 index inside the layer's r4 block
            if (fwd) {
                if (i == 0) { if (g < nqD) src = rd.o_AW / 4 + (long)(nqD * w + g) * 64 + lane; }
                else if (i == 1) src = rd.o_W1 / 4 + ((long)(1 * w) * G + g) * 64 + lane;            // Kd = 16: one quad per wave
                else if (i < 2 + 4 * G) src = rd.o_W2 / 4 + ((long)(4 * G * w + (i - 2)) * G + g) * 64 + lane;
                else { const int fi = (i - 2 - 4 * G) * G + g, Q = fi / 2, ct = fi % 2;              // W3: G k-tiles x 2 column tiles
                       src = rd.o_W3 / 4 + ((long)(G * w + Q) * 2 + ct) * 64 + lane; }
            } else {
                if (i < 2) src = rd.o_W3T / 4 + ((long)(2 * w + i) * G + g) * 64 + lane;             // Ko = 32: two quads per wave
                else if (i < 2 + 4 * G) src = rd.o_W2T / 4 + ((long)(4 * G * w + (i - 2)) * G + g) * 64 + lane;
                else if (i == 2 + 4 * G) src = rd.o_W1T / 4 + ((long)(G * w + g) * 1) * 64 + lane;  // W1T: k-tile Q = g, one column tile
                else { if (g < nqD) src = rd.o_AWT / 4 + (long)(nqD * w + g) * 64 + lane; }
            }
            if (src >= 0) v = L[src];
        } else if (item >= 2L * K * C + 8 && item < 3L * K * C + 8) {       // sampling section (flow_sample_r4s): layers 0 .. K-1
            const long li = item - (2L * K * C + 8);
            const int layer = (int)(li / C), i = (int)(li % C);
            const float4* L = r4 + (long)layer * LS4;
            if (i == 0) v = L[rd.o_W1 / 4 + ((long)(1 * w) * G + g) * 64 + lane];
            else if (i < 1 + 4 * G) v = L[rd.o_W2 / 4 + ((long)(4 * G * w + (i - 1)) * G + g) * 64 + lane];
            else if (i < 3 + 4 * G) { const int fi = (i - 1 - 4 * G) * G + g, Q = fi / 2, ct = fi % 2;
                                      v = L[rd.o_W3 / 4 + ((long)(G * w + Q) * 2 + ct) * 64 + lane]; }
            else if (g < nqD) {                                               // W'^-1: tile q = nqD w + g, lane = column
                const float* Winv = packed + f.o_scratch + (size_t)layer * 2 * D * D + (size_t)D * D;
                const int k0 = 4 * (nqD * w + g);
                float t4[4];
                for (int kk = 0; kk < 4; ++kk) t4[kk] = (k0 + kk < D && lane < D) ? Winv[(k0 + kk) * D + lane] : 0.f;
                v = make_float4(t4[0], t4[1], t4[2], t4[3]);
            }
        }
        dst[e] = v;
    }
}

// Fused-stage stream image + bias blocks (flow_r4f.h), built from what the launches before this one have written: the r4 tiles
// (k_pack_r4: W1, W2, W2^T, W3, W3^T in fp32, copied), the assembled D x D maps W' / W'^-1 (k_affine_assemble) and the layer
// blocks' biases.  The fused matrices W1' = W'[:, :d] W1^T (density; sampling: W'^-1 of the layer before) and the fused biases
// are float64 products rounded once.  Layout: see flow_r4f.h.
__global__ __launch_bounds__(256) void k_pack_r4f(FlowDims f, R4Dims rd, float* __restrict__ packed, int fast) {
    const int G = rd.G, K = f.K, D = f.D, d = f.d, Wp = f.Wp;
    // fast = 1: the bf16 image (density sections only; W x W items = two k-quads per tile)
    const int NQW = fast ? 2 * G : 4 * G;
    const int TL = fast ? r4f_tl_fast(G) : r4f_tl(G), CR = NQW + 5, I_N = NQW + 3;
    const long total4 = (long)(fast ? 2 * K + 1 : 3 * K + 3) * TL * NWAVE * 64;
    const float4* r4 = reinterpret_cast<const float4*>(packed + f.o_r4);
    const long LS4 = rd.layer_stride / 4;
    float4* dst = reinterpret_cast<float4*>(packed + (fast ? f.o_r4fh : f.o_r4f));
    auto Wm = [&](int layer) { return packed + f.o_scratch + (size_t)layer * 2 * D * D; };           // W' (ActNorm folded)
    auto Winv = [&](int layer) { return packed + f.o_scratch + (size_t)layer * 2 * D * D + (size_t)D * D; };
    // W1[k][n] (k < 16: conditioner input, n < Wp: hidden column) of a layer, from its r4 tiles (zero beyond d / W)
    auto W1 = [&](int layer, int k, int n) -> float {
        return packed[f.o_r4 + (size_t)layer * rd.layer_stride + rd.o_W1 + ((size_t)(k >> 2) * G + (n >> 6)) * 256 + (n & 63) * 4 + (k & 3)];
    };
    // (M[r][:d] . W1[:, n]) in float64: the fused first Linear of `layer` behind the D x D map M
    auto fusedW = [&](const float* M, int r, int layer, int n) -> float {
        double acc = 0.0;
        for (int j = 0; j < d; ++j) acc += (double)M[r * D + j] * (double)W1(layer, j, n);
        return (float)acc;
    };
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        const long tt = e >> 6;
        const int slot = (int)(tt / (NWAVE * TL));
        int r = (int)(tt % (NWAVE * TL));
        int sec = -1, layer = 0;                           // 0 density forward, 1 density reverse, 2 sampling (virtual layer)
        if (slot < K) { sec = 0; layer = K - 1 - slot; }
        else if (slot < 2 * K) { sec = 1; layer = slot - K; }
        else if (!fast && slot > 2 * K && slot <= 3 * K + 1) { sec = 2; layer = slot - (2 * K + 1); }
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (sec >= 0) {
            int I, w, g;
            if (r < 2 * NWAVE * G) { I = r / (NWAVE * G); r -= I * NWAVE * G; w = r / G; g = r % G; }
            else if (r < 2 * NWAVE * G + NWAVE) { I = 2; w = r - 2 * NWAVE * G; g = 0; }
            else { r -= 2 * NWAVE * G + NWAVE; I = 3 + r / (NWAVE * G); r %= NWAVE * G; w = r / G; g = r % G; }
            const int sblk = lane >> 5, c = lane & 31;
            long src = -1;                                 // float4 index inside the layer's r4 block
            const int lsrc = layer < K ? layer : K - 1;    // (virtual sampling layer K has no coupling block)
            const bool coupling = sec != 2 || layer < K;
            if (I < 2) {                                   // S1: k-quad 2 w + I of the fused first Linear | S4: of W3^T
                const int q = 2 * w + I, n = 64 * g + lane;
                if (sec == 1) src = rd.o_W3T / 4 + ((long)q * G + g) * 64 + lane;
                else if (coupling) {
                    for (int j = 0; j < 4; ++j) {
                        const int k = 4 * q + j;
                        if (k >= D) continue;
                        if (sec == 0) v[j] = fusedW(Wm(layer), k, layer, n);
                        else if (layer == 0) v[j] = k < d ? W1(0, k, n) : 0.f;
                        else v[j] = fusedW(Winv(layer - 1), k, layer, n);
                    }
                }
            } else if (I == 2) {                           // dense D x D tile: k-quad 2 w + sblk, column c
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * (2 * w + sblk) + j;
                    if (k >= D || c >= D) continue;
                    if (sec == 0) v[j] = Wm(layer)[k * D + c];
                    else if (sec == 1) v[j] = Wm(layer)[c * D + k];
                    else if (layer == 0) v[j] = k == c ? 1.f : 0.f;
                    else v[j] = Winv(layer - 1)[k * D + c];
                }
            } else if (I < I_N) {                          // W x W: k-quad 4 G w + (I - 3)   (fast: k-quads 4 G w + 2 (I - 3), + 1 as bf16)
                if (!fast) {
                    if (coupling) src = (sec == 1 ? rd.o_W2T : rd.o_W2) / 4 + ((long)(4 * G * w + (I - 3)) * G + g) * 64 + lane;
                } else {
                    const long b4 = (long)lsrc * LS4 + (sec == 1 ? rd.o_W2T : rd.o_W2) / 4;
                    const int q0 = 4 * G * w + 2 * (I - 3);
                    const float4 lo = r4[b4 + ((long)q0 * G + g) * 64 + lane], hi = r4[b4 + ((long)(q0 + 1) * G + g) * 64 + lane];
                    v[0] = __uint_as_float(bf16_rne(lo.x) | (bf16_rne(lo.y) << 16));
                    v[1] = __uint_as_float(bf16_rne(lo.z) | (bf16_rne(lo.w) << 16));
                    v[2] = __uint_as_float(bf16_rne(hi.x) | (bf16_rne(hi.y) << 16));
                    v[3] = __uint_as_float(bf16_rne(hi.z) | (bf16_rne(hi.w) << 16));
                }
            } else if (I < CR) {                           // dense narrow tiles: T = (I - I_N) G + g, k-quads 4 G w + 2 T + sblk
                const int T = (I - I_N) * G + g;
                if (sec == 1) {                            // (W1')^T: hidden row k, state column c
                    for (int j = 0; j < 4; ++j) {
                        const int k = 4 * (4 * G * w + 2 * T + sblk) + j;
                        if (c < D && k < Wp) v[j] = fusedW(Wm(layer), c, layer, k);
                    }
                } else if (coupling) src = rd.o_W3 / 4 + ((long)(2 * G * w + T)) * 64 + lane;
            }
            if (src >= 0) { const float4 t4 = r4[(long)lsrc * LS4 + src]; v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w; }
        }
        dst[e] = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (fast) return;
    // bias blocks: K density (layer k), K + 1 sampling (virtual layer v: coupling of layer v behind the affine map of layer v - 1)
    const int BS = r4f_bias_stride(Wp);
    float* bdst = packed + f.o_r4fb;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)(2 * K + 1) * BS; e += (long)gridDim.x * blockDim.x) {
        const int blk = (int)(e / BS), o = (int)(e % BS);
        const bool samp = blk >= K;
        const int layer = samp ? blk - K : blk;            // density: layer; sampling: virtual layer v
        const bool coupling = !samp || layer < K;
        const float* Lp = packed + (size_t)(coupling ? layer : K - 1) * f.layer_stride;
        // additive term of the D x D map in front of this block's first Linear, and its log-det
        const float* add = nullptr;
        float logS = 0.f;
        if (!samp) { add = Lp + f.o_ac; logS = Lp[f.o_logS]; }
        else if (layer > 0) { const float* Lq = packed + (size_t)(layer - 1) * f.layer_stride; add = Lq + f.o_at; logS = Lq[f.o_logS]; }
        float val = 0.f;
        if (o < Wp) {                                      // b1' = b1 + add[:d] . W1
            if (coupling) {
                double acc = (double)Lp[f.o_b1 + o];
                if (add) for (int j = 0; j < d; ++j) acc += (double)add[j] * (double)W1(layer, j, o);
                val = (float)acc;
            }
        } else if (o < 2 * Wp) { if (coupling) val = Lp[f.o_b2 + (o - Wp)]; }
        else if (o < 2 * Wp + 32) { const int cc = o - 2 * Wp; if (add && cc < D) val = add[cc]; }
        else if (o < 2 * Wp + 48) { if (coupling) val = Lp[f.o_b3 + (o - 2 * Wp - 32)]; }
        else if (o < 2 * Wp + 64) { if (coupling) val = Lp[f.o_b3 + f.DOp + (o - 2 * Wp - 48)]; }
        else if (o == 2 * Wp + 64) val = logS;
        bdst[e] = val;
    }
}

template <int NTWM>
static int launch_sample(const FlowDims& f, const float* packed, const float* eps, float* x, float* log_q, long B,
                         hipStream_t st) {
    if constexpr (NTWM >= 2 && NTWM <= 5) {
        // batches the transitions run on 4-chain tiles (<= 1152 chains): the sample on 4-chain tiles as well
        if (use_r4_tiles(f, B) && f.o_r4s >= 0 && option(FABHIP_OPT_R4_STREAM) != 0) {
            const R4Dims rd = make_r4_dims(f);
            const bool fused = use_r4_fused(f);
            const R4Lds l4 = make_r4_lds(f, fused);
            const size_t bytes4 = (size_t)l4.total * 4;
            const dim3 grid4((unsigned)((B + R4 - 1) / R4));
            if (fused) {
                FAB_TRY(set_max_lds((const void*)k_flow_sample_r4<NTWM, true>, bytes4));
                hipLaunchKernelGGL((k_flow_sample_r4<NTWM, true>), grid4, dim3(NTHREADS), bytes4, st, f, rd, l4, packed, eps, x, log_q, B);
            } else {
                FAB_TRY(set_max_lds((const void*)k_flow_sample_r4<NTWM, false>, bytes4));
                hipLaunchKernelGGL((k_flow_sample_r4<NTWM, false>), grid4, dim3(NTHREADS), bytes4, st, f, rd, l4, packed, eps, x, log_q, B);
            }
            return check_launch();
        }
    }
    const dim3 grid((unsigned)ceil_div((int)B, ROWS)), block(NTHREADS);
    const FlowLds l = make_flow_lds(f, false);
    const size_t bytes = (size_t)l.total * 4;
    FAB_TRY(set_max_lds((const void*)k_flow_sample<NTWM>, bytes));
    hipLaunchKernelGGL((k_flow_sample<NTWM>), grid, block, bytes, st, f, l, packed, eps, x, log_q, B);
    return check_launch();
}

}  // namespace fab

using namespace fab;

// dev-only stage timeline (FABHIP_TIMELINE=1): 64 s_memtime stamps written by workgroup 0 of fabhip_flow_log_prob
static long long* g_timeline = nullptr;
namespace fab {
// dev-only (FABHIP_TIMELINE=1): the 64-stamp buffer, zeroed on `st`; nullptr when the switch is off
long long* debug_timeline(hipStream_t st) {
    if (!option(FABHIP_OPT_TIMELINE)) return nullptr;
    if (!g_timeline && hipMalloc((void**)&g_timeline, 64 * 8) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(g_timeline, 0, 64 * 8, st);
    return g_timeline;
}
}  // namespace fab

namespace fab {
bool r4f_lds_fits(const FlowDims& f) {
    const R4Lds l = make_r4_lds(f, true);
    return (size_t)(l.total + 4 * R4 * f.D + 4) * 4 <= 160 * 1024;
}
static int g_fast_mode = 0;
int fast_mode() { return g_fast_mode; }

// developer switches: defaults, overridden once from the environment when the library is loaded
struct Options {
    int v[FABHIP_OPT_COUNT];
    Options() {
        static const struct { int key; const char* env; int dflt; } tab[FABHIP_OPT_COUNT] = {
            {FABHIP_OPT_TILE_SHAPE, "FABHIP_TILE", 0},           {FABHIP_OPT_R4_STREAM, "FABHIP_R4_STREAM", 2},
            {FABHIP_OPT_SCAN_VARIANT, "FABHIP_SCAN_VARIANT", 3}, {FABHIP_OPT_SYSTEMATIC_VARIANT, "FABHIP_SYSTEMATIC_VARIANT", 1},
            {FABHIP_OPT_SPLINE_STAGED, "FABHIP_SPLINE_STAGED", 0}, {FABHIP_OPT_TIMELINE, "FABHIP_TIMELINE", 0},
            {FABHIP_OPT_SPLINE_MFMA, "FABHIP_SPLINE_MFMA", 0},   {FABHIP_OPT_SPLINE_LEAP, "FABHIP_SPLINE_LEAP", 1},
            {FABHIP_OPT_FUSED_TAIL, "FABHIP_FUSED_TAIL", 1},     {FABHIP_OPT_ADAPT_FOLD, "FABHIP_ADAPT_FOLD", 1},
            {FABHIP_OPT_PGRAD, "FABHIP_PGRAD", 1},               {FABHIP_OPT_TAPE_TILES, "FABHIP_TAPE_TILES", 0}};
        for (const auto& t : tab) {
            const char* e = getenv(t.env);
            v[t.key] = (e && e[0]) ? atoi(e) : t.dflt;
        }
    }
};
static Options g_options;
int option(int key) { return g_options.v[key]; }
}  // namespace fab

extern "C" {

int fabhip_set_fast_mode(int on) {
    const int prev = fab::g_fast_mode;
    fab::g_fast_mode = on ? 1 : 0;
    return prev;
}

int fabhip_get_fast_mode(void) { return fab::g_fast_mode; }

int fabhip_set_option(int key, int value) {
    if (key < 0 || key >= FABHIP_OPT_COUNT) return FABHIP_EINVAL;
    const int prev = fab::g_options.v[key];
    fab::g_options.v[key] = value;
    return prev;
}

int fabhip_get_option(int key) { return (key < 0 || key >= FABHIP_OPT_COUNT) ? FABHIP_EINVAL : fab::g_options.v[key]; }

int fabhip_debug_timeline(int64_t* host_out, int32_t n) {
    if (!g_timeline || !host_out || n < 1 || n > 64) return FABHIP_EINVAL;
    return hipMemcpy(host_out, g_timeline, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess ? FABHIP_OK : FABHIP_ELAUNCH;
}

int64_t fabhip_flow_packed_floats(int32_t dim, int32_t n_layers, int32_t width) {
    if (check_flow_shape(dim, n_layers, width) != FABHIP_OK) return -1;
    return (int64_t)make_flow_dims(dim, n_layers, width).total;
}

static int flow_pack_impl(const fabhip_flow_params* p, float* packed, int with_inverse, fabhip_stream_t stream) {
    if (!p || !packed) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(p->dim, p->n_layers, p->width));
    if (!p->loc || !p->log_scale) return FABHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const FlowDims f = make_flow_dims(p->dim, p->n_layers, p->width);
    const int D = f.D;
    const size_t smem = (size_t)D * D * (8 + 4 * 5);
    FAB_TRY(set_max_lds((const void*)k_affine_assemble, smem));
    for (int k = 0; k < f.K; ++k)
        if (!p->w1[k] || !p->b1[k] || !p->w2[k] || !p->b2[k] || !p->w3[k] || !p->b3[k] || !p->lu_L[k] ||
            !p->lu_U[k] || !p->log_S[k] || !p->sign_S[k] || !p->perm_P[k] || (!p->an_s[k] != !p->an_t[k]))
            return FABHIP_EINVAL;
    bool base_done = false;
    for (int k0 = 0; k0 < f.K; k0 += LBATCH) {             // all layers of a batch in one launch each
        const int nl = f.K - k0 < LBATCH ? f.K - k0 : LBATCH;
        AffineTab at;
        MlpTab mt;
        for (int y = 0; y < LBATCH; ++y) {
            const int k = k0 + (y < nl ? y : 0);
            at.L[y] = p->lu_L[k]; at.U[y] = p->lu_U[k]; at.logS[y] = p->log_S[k]; at.signS[y] = p->sign_S[k];
            at.P[y] = p->perm_P[k];
            at.an_s[y] = p->an_s[k]; at.an_t[y] = p->an_t[k];
            mt.w1[y] = p->w1[k]; mt.b1[y] = p->b1[k]; mt.w2[y] = p->w2[k]; mt.b2[y] = p->b2[k];
            mt.w3[y] = p->w3[k]; mt.b3[y] = p->b3[k];
        }
        hipLaunchKernelGGL(k_affine_assemble, dim3(nl), dim3(f.D <= 32 ? 1024 : 256), smem, st, f, at, k0, packed, with_inverse == 1 ? 1 : 0);
        // training image + 8-chain tape tiles: the bias blocks only, with the base distribution in the same launch
        const bool heads_only = with_inverse == 2 && f.o_r8 >= 0 && option(FABHIP_OPT_TAPE_TILES) != 16;
        const int off_begin = heads_only ? f.o_b1 : 0;
        const int nblk = ceil_div(f.o_logS - off_begin, 256 * 4);
        hipLaunchKernelGGL(k_pack_layer, dim3(nblk, nl), dim3(256), 0, st, f, mt, k0, packed, off_begin,
                           heads_only && k0 == 0 ? p->loc : (const float*)nullptr, p->log_scale);
        base_done = base_done || (heads_only && k0 == 0);
        if (f.o_r8 >= 0)
            hipLaunchKernelGGL(k_pack_r8, dim3(ceil_div(NWAVE * 2 * r8_tiles_p(f.Wp / 64) * 256, 256 * 8), nl), dim3(256), 0, st, f, mt, k0, packed);
        if (with_inverse == 2) continue;                  // training image: what fabhip_flow_log_prob_tape reads, nothing else
        hipLaunchKernelGGL(k_pack_bf16, dim3(ceil_div(f.Wp * f.Wp, 256 * 4), nl), dim3(256), 0, st, f, mt, k0, packed);
        const R4Dims rd = make_r4_dims(f);
        hipLaunchKernelGGL(k_pack_r4, dim3(ceil_div(rd.layer_stride, 256 * 8), nl), dim3(256), 0, st, f, rd, mt, k0, packed);
        if (f.o_r8f >= 0)
            hipLaunchKernelGGL(k_pack_r8f, dim3(ceil_div(NWAVE * 2 * r8f_tiles_p(f.Wp / 64) * 256, 256 * 8), nl), dim3(256), 0, st, f, mt, k0, packed);
    }
    if (f.o_r4s >= 0 && with_inverse != 2)
        hipLaunchKernelGGL(k_pack_r4s, dim3(1024), dim3(256), 0, st, f, make_r4_dims(f), packed);
    if (f.o_r4f >= 0 && with_inverse != 2) {
        hipLaunchKernelGGL(k_pack_r4f, dim3(2048), dim3(256), 0, st, f, make_r4_dims(f), packed, 0);
        hipLaunchKernelGGL(k_pack_r4f, dim3(1024), dim3(256), 0, st, f, make_r4_dims(f), packed, 1);
    }
    if (!base_done) hipLaunchKernelGGL(k_pack_base, dim3(1), dim3(64), 0, st, f, p->loc, p->log_scale, packed);
    return check_launch();
}

int fabhip_flow_pack(const fabhip_flow_params* p, float* packed, fabhip_stream_t stream) {
    return flow_pack_impl(p, packed, 1, stream);
}

int fabhip_flow_pack_density(const fabhip_flow_params* p, float* packed, fabhip_stream_t stream) {
    return flow_pack_impl(p, packed, 0, stream);
}

int fabhip_flow_pack_train(const fabhip_flow_params* p, float* packed, fabhip_stream_t stream) {
    return flow_pack_impl(p, packed, 2, stream);
}

int fabhip_flow_log_prob(const fabhip_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                         fabhip_stream_t stream) {
    if (!flow || !flow->packed || !x || !log_q || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(flow->dim, flow->n_layers, flow->width));
    if (B == 0) return FABHIP_OK;
    FlowDims f = flow_dims_of(*flow);
    if (option(FABHIP_OPT_TIMELINE)) {        // diagnostics only: never set in production (allocates once)
        if (!g_timeline && hipMalloc((void**)&g_timeline, 64 * 8) != hipSuccess) return FABHIP_ELAUNCH;
        hipMemsetAsync(g_timeline, 0, 64 * 8, (hipStream_t)stream);
        f.timeline = g_timeline;
    }
    FAB_DISPATCH_NTW(f, launch_log_prob, f, flow->packed, x, log_q, grad_x, (long)B, (hipStream_t)stream);
}

int fabhip_flow_sample(const fabhip_flow* flow, const float* eps, float* x, float* log_q, int64_t B,
                       fabhip_stream_t stream) {
    if (!flow || !flow->packed || !eps || !x || !log_q || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(flow->dim, flow->n_layers, flow->width));
    if (B == 0) return FABHIP_OK;
    const FlowDims f = make_flow_dims(flow->dim, flow->n_layers, flow->width);
    FAB_DISPATCH_NTW(f, launch_sample, f, flow->packed, eps, x, log_q, (long)B, (hipStream_t)stream);
}

}  // extern "C"
