// Flow training path: log q(x) with a tape, and the parameter gradients sum_b coef[b] * d log q(x_b) / d theta
// (the backward of `flow.log_prob(x)` in fab/train_with_prioritised_buffer.py:162-177 and
// fab/core.py:112-118, where coef = d loss / d log_q).
//
//   k_flow_log_prob_tape   forward + reverse sweep of the 16-chain tile kernel (flow_device.h, TAPE = true):
//                          layer inputs / hidden activations / back-propagated deltas go to HBM once
//   k_param_grad           every weight / bias gradient as a batch-reduction GEMM  C[p][q] = sum_b c_b Y[b][p] X[b][q]
//                          on the matrix cores (v_mfma_f32_16x16x4_f32, 64 x 64 block per workgroup, LDS double
//                          buffered, fixed summation order => deterministic), written straight into the flat
//                          gradient image in the parameters' own (PyTorch [out][in]) layouts
//                          (the DiagGaussian's dloc / dlog_scale and sum(coef) are one more such GEMM against a ones column)
//   k_affine_grads         InvertibleAffine: dW -> (dL, dU, dlog_S) through W = P (tril(L,-1)+I)(triu(U,1)+diag(s e^logS))
#include "flow_device.h"
#include "launch.h"
#include "train_common.h"

namespace fab {

template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_flow_log_prob_tape(FlowDims f, FlowLds l, TapeDims td,
                                                                 const float* __restrict__ packed,
                                                                 const float* __restrict__ x,
                                                                 float* __restrict__ log_q, float* __restrict__ grad,
                                                                 float* __restrict__ tape, long B,
                                                                 const int64_t* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid t;
    const long row0 = (long)blockIdx.x * ROWS;
    for (int e = t.tid; e < ROWS * l.PS; e += NTHREADS) lds[l.o_DP + e] = 0.f;
    for (int e = t.tid; e < ROWS * l.DS; e += NTHREADS) {
        const int r = e / l.DS, j = e % l.DS;
        const long g = row0 + r;
        lds[l.o_U0 + e] = (j < f.D && g < B) ? x[(rows ? (long)rows[g] : g) * f.D + j] : 0.f;
        lds[l.o_U1 + e] = 0.f;
    }
    __syncthreads();
    int goff = 0;
    const float lq = flow_log_prob_tile<NTWM, true, true>(f, l, packed, lds, t, &goff, &td, tape, row0);
    if (t.c == 0 && row0 + t.row < B) log_q[row0 + t.row] = lq;
    if (grad) {
        for (int e = t.tid; e < ROWS * f.D; e += NTHREADS) {
            const int r = e / f.D, j = e % f.D;
            const long g = row0 + r;
            if (g < B) grad[g * f.D + j] = lds[goff + r * l.DS + j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the SAMPLING direction (x, log q = flow.sample(eps); the reparameterised losses flow_reverse_kl /
// flow_alpha_2_div_nis of fab/core.py:130-152): given x = T(eps), gx = d loss / dx and gl = d loss / d log q, one sweep
// x -> eps re-evaluates each layer from its output (the conditioner sees the same z1 in both directions) and at the same
// time propagates the cotangent through the layer's transposed maps, writing the SAME tape the density path writes:
//   Z1 | H1 | H2   the conditioner's input / hidden activations           DP | E2 | E1   the true cotangents of
//   (shift, s) and of the two pre-activations      ZA = the affine map's x-side value, GZ = -(cotangent of its z-side
//   value u): x = (u - ac) W'^-1  =>  d loss / dW' = -x^T (g_x W'^-T) = ZA^T GZ (implicit-function form, so the LU /
//   ActNorm chain rule of k_affine_grads applies unchanged)         TB = [g_z0 | g_z0 (z0 - loc) - gl | gl]
// => fabhip_flow_param_grad(tape, coef = 1) returns d loss / d theta (its sum(coef) column is sum(gl): the log-det terms).
// Stages issue their own prologues (not a hot path: one call per optimiser step of a baseline loss).
// ------------------------------------------------------------------------------------------------
template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_flow_sample_bwd(FlowDims f, FlowLds l, TapeDims td,
                                                              const float* __restrict__ packed,
                                                              const float* __restrict__ x, const float* __restrict__ gx,
                                                              const float* __restrict__ gl, float* __restrict__ g_eps,
                                                              float* __restrict__ tape, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DW = depth_w<NTWM>();
    Tid t;
    const long row0 = (long)blockIdx.x * ROWS;
    const int o_U2 = l.total, o_MK = o_U2 + ROWS * l.DS;
    for (int e = t.tid; e < ROWS * l.PS; e += NTHREADS) lds[l.o_DP + e] = 0.f;
    for (int e = t.tid; e < ROWS * l.DS; e += NTHREADS) {
        const int r = e / l.DS, j = e % l.DS;
        const long g = row0 + r;
        const bool in = j < f.D && g < B;
        lds[l.o_U0 + e] = in ? x[g * f.D + j] : 0.f;
        lds[l.o_U1 + e] = in ? gx[g * f.D + j] : 0.f;
        lds[o_U2 + e] = 0.f;
    }
    const float glr = row0 + t.row < B ? gl[row0 + t.row] : 0.f;
    __syncthreads();
    int bx = l.o_U0, bg = l.o_U1, bf = o_U2;              // state (x side), its cotangent, free buffer
    float* PART = lds + l.o_PART;
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
    float* DP = lds + l.o_DP;
    unsigned* mk = reinterpret_cast<unsigned*>(lds + o_MK);
    for (int layer = f.K - 1; layer >= 0; --layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        float* tl_layer = tape + (size_t)layer * td.layer_stride;
        tape_copy(tl_layer + td.o_ZA + row0 * td.wz, td.wz, lds + bx, l.DS, t);
        // u = x @ W' + ac (the density direction's affine map = the inverse of the sampling direction's)
        dense_small(lds + bx, l.DS, f.D, f.KBD, reinterpret_cast<const float4*>(Lp + f.o_AW), f.NTD, lds + bf, l.DS, t,
                    Lp + f.o_ac);
        __syncthreads();
        // g_u = g_x @ (W'^-1)^T, into the buffer x just left
        dense_small(lds + bg, l.DS, f.D, f.KBD, reinterpret_cast<const float4*>(Lp + f.o_AWIT), f.NTD, lds + bx, l.DS, t);
        __syncthreads();
        { const int tmp = bg; bg = bx; bx = bf; bf = tmp; }
        float* U = lds + bx;
        float* G = lds + bg;
        {
            float* GZ = tl_layer + td.o_GZ + row0 * td.wz;
            const int wz = td.wz;
            for (int e = t.tid; e < ROWS * wz; e += NTHREADS) {
                const int r = e / wz, j = e - r * wz;
                GZ[(long)r * wz + j] = -G[r * l.DS + j];
            }
            float* Z1 = tl_layer + td.o_Z1 + row0 * td.w1;
            const int w1 = td.w1, c1 = w1 - 16;
            for (int e = t.tid; e < ROWS * w1; e += NTHREADS) {
                const int r = e / w1, j = e - r * w1;
                Z1[(long)r * w1 + j] = j < f.d ? U[r * l.DS + j] : (j == c1 ? 1.f : 0.f);
            }
            const int r = t.tid >> 4, j = t.tid & 15;
            const float one = j == 0 ? 1.f : 0.f;
            tl_layer[td.o_H1 + (row0 + r) * td.wh + f.Wp + j] = one;
            tl_layer[td.o_H2 + (row0 + r) * td.wh + f.Wp + j] = one;
        }
        // conditioner, forward: the ReLU sign words stay in LDS for the transposed pass below
        dense_relu<NTWM, 2, true, true, true>(U, l.DS, f.d, f.KBd, reinterpret_cast<const float4*>(Lp + f.o_W1),
                                              Lp + f.o_b1, HA, l.WS, mk, t, tl_layer + td.o_H1 + row0 * td.wh, td.wh);
        __syncthreads();
        dense_relu<NTWM, DW, false, true, true>(HA, l.WS, f.Wp, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_W2),
                                                Lp + f.o_b2, HB, l.WS, mk + NTHREADS, t,
                                                tl_layer + td.o_H2 + row0 * td.wh, td.wh);
        __syncthreads();
        gemm_ksplit<NTWM>(HB, l.WS, reinterpret_cast<const float4*>(Lp + f.o_W3), f.NTO, PART, l.PN, t);
        __syncthreads();
        // coupling: sampling direction b2 = a2 e^s + t (log q -= sum s), here u2 = b2 is known and a2 is recovered
        for (int j = t.c; j < f.DO; j += 16) {
            const float shift = part_sum(PART, l.PN, t.row, j) + Lp[f.o_b3 + j];
            const float s = part_sum(PART, l.PN, t.row, f.DOp + j) + Lp[f.o_b3 + f.DOp + j];
            const float u2 = U[t.row * l.DS + f.d + j], g2 = G[t.row * l.DS + f.d + j];
            U[t.row * l.DS + f.d + j] = (u2 - shift) * expf(-s);           // a2
            DP[t.row * l.PS + j] = g2;                                     // d/d shift
            DP[t.row * l.PS + f.DOp + j] = g2 * (u2 - shift) - glr;        // d/d s  (a2 e^s = u2 - shift; -gl: log-det)
            G[t.row * l.DS + f.d + j] = g2 * expf(s);                      // d/d a2
        }
        __syncthreads();
        tape_copy(tl_layer + td.o_DP + row0 * td.wp, td.wp, DP, l.PS, t);
        dense_masked<NTWM, 2, true>(DP, l.PS, f.KBO, reinterpret_cast<const float4*>(Lp + f.o_W3T), HA, l.WS,
                                    mk + NTHREADS, t, tl_layer + td.o_E2 + row0 * td.we, td.we);
        __syncthreads();
        dense_masked<NTWM, DW, true>(HA, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_W2T), HB, l.WS, mk, t,
                                     tl_layer + td.o_E1 + row0 * td.we, td.we);
        __syncthreads();
        gemm_ksplit<NTWM>(HB, l.WS, reinterpret_cast<const float4*>(Lp + f.o_W1T), f.NTd, PART, l.PN, t);
        __syncthreads();
        for (int j = t.c; j < f.d; j += 16) G[t.row * l.DS + j] += part_sum(PART, l.PN, t.row, j);
        __syncthreads();
    }
    // base distribution: z0 = loc + e^{log_scale} eps, log q has -sum(log_scale)
    const float* base = packed + f.o_base;
    const float* Z = lds + bx;
    const float* G = lds + bg;
    float* TB = tape + td.o_TB + (row0 + t.row) * td.wb;
    for (int j = t.c; j < f.D; j += 16) {
        const float g = G[t.row * l.DS + j], dz = Z[t.row * l.DS + j] - base[j];
        TB[j] = g;
        TB[td.wz + j] = g * dz - glr;
        if (g_eps && row0 + t.row < B) {                 // eps = dz / sc; log q has -eps^2 / 2
            const float sc = expf(base[f.Dp + j]);
            g_eps[(row0 + t.row) * f.D + j] = g * sc - glr * (dz / sc);
        }
    }
    TB[2 * td.wz + t.c] = t.c == 0 ? glr : 0.f;
}

// ------------------------------------------------------------------------------------------------
// batch-reduction GEMMs.  One workgroup = one 64 x 64 block of one (layer, matrix); wave w owns rows
// [16 w, 16 w + 16) of the block and all four 16-column tiles.
// ------------------------------------------------------------------------------------------------
constexpr int GK = 32;          // batch rows per LDS chunk
constexpr int GLD = 80;         // LDS leading dim: 80 mod 64 = 16 -> the 4 k-groups of a wave hit disjoint banks

struct GemmBlocks {
    int n1, n2, n3, nA, per_layer;     // 64 x 64 blocks of G1 (dW1|db1), G2, G3, GA
    int q1, q2, q3, qA;                // blocks along Q
};

FAB_HD GemmBlocks make_gemm_blocks(const FlowDims& f, const TapeDims& td) {
    GemmBlocks g;
    g.q1 = ceil_div(td.w1, 64); g.q2 = ceil_div(td.wh, 64); g.q3 = g.q2; g.qA = ceil_div(td.wz, 64);
    g.n1 = ceil_div(td.we, 64) * g.q1;
    g.n2 = ceil_div(td.we, 64) * g.q2;
    g.n3 = ceil_div(td.wp, 64) * g.q3;
    g.nA = ceil_div(td.wz, 64) * g.qA;
    g.per_layer = g.n1 + g.n2 + g.n3 + g.nA;
    return g;
}

__device__ __forceinline__ int prm_row(int p, int DO, int DOp) {      // packed [shift | scale] row -> interleaved row
    if (p < DOp) return p < DO ? 2 * p : -1;
    const int j = p - DOp;
    return j < DO ? 2 * j + 1 : -1;
}

// The same 64 x 64-block batch-reduction GEMM for ANY tape (the spline flow's, fabhip_spline_log_prob_tape): L problems
// of one shape, C[l][p][q] = sum_b coef[b] Y[l][b][p] X[l][b][q] and, from the blocks with q0 = 0, the bias gradients
// S[l][p] = sum_b coef[b] Y[l][b][p].  Rows are added in chunk order, chunks by the matrix cores' k order: deterministic.
struct TapeGemm {
    const float *Y, *X, *coef;
    long y_stride, x_stride, B;
    int ldy, ldx, P, Q, pblocks, qblocks;
    float *C, *S;                      // [L][P][Q], [L][P] (S may be nullptr)
};

__global__ __launch_bounds__(256) void k_tape_gemm(TapeGemm g) {
    __shared__ __attribute__((aligned(16))) float Ys[2][GK * GLD];
    __shared__ __attribute__((aligned(16))) float Xs[2][GK * GLD];
    __shared__ float Ss[16][64];
    const int per = g.pblocks * g.qblocks;
    const int layer = blockIdx.x / per, b = blockIdx.x % per;
    const int p0 = 64 * (b / g.qblocks), q0 = 64 * (b % g.qblocks);
    const float* Y = g.Y + (size_t)layer * g.y_stride;
    const float* X = g.X + (size_t)layer * g.x_stride;
    const int P = g.P, Q = g.Q, ldy = g.ldy, ldx = g.ldx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kg = lane >> 4;
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    const bool pvalid = p0 + 16 * wave < P;
    const bool want_s = g.S != nullptr && q0 == 0;
    float4 ry[2], rx[2];
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld4 = [&](const float* base, long k, int ld, int c0, int lim) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < g.B) {
            const float* r = base + k * ld;
            if (c0 + 3 < lim && ((ld | c0) & 3) == 0) v = *reinterpret_cast<const float4*>(r + c0);
            else {
                if (c0 + 0 < lim) v.x = r[c0 + 0];
                if (c0 + 1 < lim) v.y = r[c0 + 1];
                if (c0 + 2 < lim) v.z = r[c0 + 2];
                if (c0 + 3 < lim) v.w = r[c0 + 3];
            }
        }
        return v;
    };
    auto gload = [&](long k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long k = k0 + lr + 16 * h;
            const float c = k < g.B ? g.coef[k] : 0.f;
            const float4 y = ld4(Y, k, ldy, p0 + lc, P);
            ry[h] = make_float4(y.x * c, y.y * c, y.z * c, y.w * c);
            rx[h] = ld4(X, k, ldx, q0 + lc, Q);
            if (want_s) { ssum.x += ry[h].x; ssum.y += ry[h].y; ssum.z += ry[h].z; ssum.w += ry[h].w; }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&Ys[buf][(lr + 16 * h) * GLD + lc]) = ry[h];
            *reinterpret_cast<float4*>(&Xs[buf][(lr + 16 * h) * GLD + lc]) = rx[h];
        }
    };
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nch = (int)((g.B + GK - 1) / GK);
    gload(0);
    sstore(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nch) gload((long)(ch + 1) * GK);
        if (pvalid) {
#pragma unroll
            for (int s = 0; s < GK / 4; ++s) {
                const float a = Ys[buf][(4 * s + kg) * GLD + 16 * wave + n];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = mfma4(a, Xs[buf][(4 * s + kg) * GLD + 16 * j + n], acc[j]);
            }
        }
        if (ch + 1 < nch) sstore(buf ^ 1);
        __syncthreads();
    }
    if (want_s) {                                           // column sums: 16 row groups x 64 columns, added in row-group order
        Ss[lr][lc] = ssum.x; Ss[lr][lc + 1] = ssum.y; Ss[lr][lc + 2] = ssum.z; Ss[lr][lc + 3] = ssum.w;
        __syncthreads();
        if (tid < 64 && p0 + tid < P) {
            float s = 0.f;
            for (int r = 0; r < 16; ++r) s += Ss[r][tid];
            g.S[(size_t)layer * P + p0 + tid] = s;
        }
    }
    if (!pvalid) return;
    float* C = g.C + (size_t)layer * P * Q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = q0 + 16 * j + n;
        if (q >= Q) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = p0 + 16 * wave + 4 * kg + r;
            if (p < P) C[(size_t)p * Q + q] = acc[j][r];
        }
    }
}

__global__ __launch_bounds__(256) void k_param_grad(FlowDims f, TapeDims td, GemmBlocks gb, GradLayout gl,
                                                    const float* __restrict__ tape, const float* __restrict__ coef,
                                                    long B, float* __restrict__ grads, float* __restrict__ ga_ws) {
    __shared__ __attribute__((aligned(16))) float Ys[2][GK * GLD];
    __shared__ __attribute__((aligned(16))) float Xs[2][GK * GLD];
    // XCD-aware order: consecutive blockIdx go to different XCDs (blockIdx % 8), so give XCD x the x-th eighth of
    // the logical block list -> the blocks that re-read the same Y / X panels share one L2
    const int nlog = f.K * gb.per_layer;
    int logical = blockIdx.x;
    if ((int)blockIdx.x < (nlog & ~7)) logical = (blockIdx.x & 7) * (nlog >> 3) + (blockIdx.x >> 3);
    int layer = logical / gb.per_layer;
    int b = logical % gb.per_layer;
    // largest problem first; the blocks after the K layers are the base distribution (kind 5)
    int kind, qblocks;
    if (layer >= f.K) { b = logical - f.K * gb.per_layer; layer = 0; kind = 5; qblocks = gb.q1; }
    else if (b < gb.n2) { kind = 2; qblocks = gb.q2; }
    else if ((b -= gb.n2) < gb.n1) { kind = 1; qblocks = gb.q1; }
    else if ((b -= gb.n1) < gb.n3) { kind = 3; qblocks = gb.q3; }
    else { b -= gb.n3; kind = 4; qblocks = gb.qA; }
    const int p0 = 64 * (b / qblocks), q0 = 64 * (b % qblocks);
    const float* Lt = tape + (size_t)layer * td.layer_stride;
    const float *Y, *X;
    int ldy, ldx, P, Q;
    if (kind == 1) { Y = Lt + td.o_E1; ldy = P = td.we; X = Lt + td.o_Z1; ldx = Q = td.w1; }
    else if (kind == 2) { Y = Lt + td.o_E2; ldy = P = td.we; X = Lt + td.o_H1; ldx = Q = td.wh; }
    else if (kind == 3) { Y = Lt + td.o_DP; ldy = P = td.wp; X = Lt + td.o_H2; ldx = Q = td.wh; }
    else if (kind == 4) { Y = Lt + td.o_ZA; ldy = P = td.wz; X = Lt + td.o_GZ; ldx = Q = td.wz; }
    else { Y = tape + td.o_TB; ldy = P = td.wb; X = Lt + td.o_Z1; ldx = Q = td.w1; }   // x the ones column of Z1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kg = lane >> 4;
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    const bool yok = p0 + lc < P, xok = q0 + lc < Q;
    const bool pvalid = p0 + 16 * wave < P;
    float4 ry[2], rx[2];
    auto gload = [&](long k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long k = k0 + lr + 16 * h;
            const bool in = k < td.Bp;
            const float c = k < B ? coef[k] : 0.f;
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f), xv = y;
            if (in && yok) y = *reinterpret_cast<const float4*>(Y + k * ldy + p0 + lc);
            if (in && xok) xv = *reinterpret_cast<const float4*>(X + k * ldx + q0 + lc);
            ry[h] = make_float4(y.x * c, y.y * c, y.z * c, y.w * c);
            rx[h] = xv;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&Ys[buf][(lr + 16 * h) * GLD + lc]) = ry[h];
            *reinterpret_cast<float4*>(&Xs[buf][(lr + 16 * h) * GLD + lc]) = rx[h];
        }
    };
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nch = (int)((B + GK - 1) / GK);
    gload(0);
    sstore(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nch) gload((long)(ch + 1) * GK);
        if (pvalid) {
#pragma unroll
            for (int s = 0; s < GK / 4; ++s) {
                const float a = Ys[buf][(4 * s + kg) * GLD + 16 * wave + n];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = mfma4(a, Xs[buf][(4 * s + kg) * GLD + 16 * j + n], acc[j]);
            }
        }
        if (ch + 1 < nch) sstore(buf ^ 1);
        __syncthreads();
    }
    if (!pvalid) return;
    float* G = grads + (size_t)layer * gl.layer_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = q0 + 16 * j + n;
        if (q >= Q) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = p0 + 16 * wave + 4 * kg + r;
            const float v = acc[j][r];
            if (kind == 1) {
                if (p < f.W) {
                    if (q < f.d) G[gl.w1 + (long)p * f.d + q] = v;
                    else if (q == td.w1 - 16) G[gl.b1 + p] = v;
                }
            } else if (kind == 2) {
                if (p < f.W) {
                    if (q < f.W) G[gl.w2 + (long)p * f.W + q] = v;
                    else if (q == f.Wp) G[gl.b2 + p] = v;
                }
            } else if (kind == 3) {
                const int row = prm_row(p, f.DO, f.DOp);
                if (row >= 0) {
                    if (q < f.W) G[gl.w3 + (long)row * f.W + q] = v;
                    else if (q == f.Wp) G[gl.b3 + row] = v;
                }
            } else if (kind == 4) {
                ga_ws[((size_t)layer * td.wz + p) * td.wz + q] = v;
            } else if (q == td.w1 - 16) {                       // base: dloc | dlog_scale | sum(coef)
                if (p < f.D) grads[gl.loc + p] = v;
                else if (p >= td.wz && p < td.wz + f.D) grads[gl.log_scale + p - td.wz] = v;
                else if (p == 2 * td.wz) ga_ws[(size_t)f.K * td.wz * td.wz] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// blocks 0 .. K-1: InvertibleAffine layer `blockIdx.x`; block K: DiagGaussian base.
// ------------------------------------------------------------------------------------------------
constexpr int AFF_THREADS = 1024;      // one output entry per thread and D x D product for D <= 32 (the products are latency chains)
// one workgroup per InvertibleAffine layer: (dL, dU, dlog_S) from dW = ga_ws[layer]; everything staged in LDS.
// With an ActNorm after the map (density direction z = a @ W, a = (x - t) e^-s, log_det -= sum(s)) ga_ws holds
// dW' = sum_b c_b x_b^T g_b for the FOLDED map W' = diag(e^-s) W.  With dc = sum_b c_b g_b (reduced here from the
// tape's GZ rows):  dW = diag(e^-s) (dW' - t (x) dc),  ds_i = -sum_j W_ij dW_ij - sum_b c_b,  dt_i = -e^-s_i (W dc)_i.
__global__ __launch_bounds__(AFF_THREADS) void k_affine_grads(FlowDims f, TapeDims td, GradLayout gl, AffineSrcTab tab,
                                                      int k0, const float* __restrict__ ga_ws,
                                                      float* __restrict__ grads, const float* __restrict__ tape,
                                                      const float* __restrict__ coef, long B) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, D = f.D, DD = D * D;
    const int y = blockIdx.x, layer = k0 + y;
    const AffineSrc src{tab.L[y], tab.U[y], tab.logS[y], tab.signS[y], tab.P[y], tab.an_s[y], tab.an_t[y]};
    float* dW = sm;                 // [D][D] each
    float* Lm = dW + DD;
    float* Um = Lm + DD;            // rows D + 1 floats apart: T = dW Um^T reads a COLUMN of lanes (j) per k - with rows D apart
    const int US = D + 1;           // (D = 32) all 32 lanes of a row hit one bank, 32 times per output entry
    float* Ps = Um + DD + D;
    float* PL = Ps + DD;            // P @ Lm
    float* T = PL + DD;             // dW @ Um^T
    float* dc = T + DD;             // [4][64] partial column sums of c_b g_b, then [64] (ActNorm only)
    const float csum = ga_ws[(size_t)f.K * td.wz * td.wz];          // sum_b coef_b (k_param_grad, base block)
    const bool an = src.an_s != nullptr;
    if (an) {
        const float* GZ = tape + (size_t)layer * td.layer_stride + td.o_GZ;
        const int j = tid & 63, r = tid >> 6;
        if (tid < 256) {
            float s = 0.f;
            if (j < D)
                for (long b = r; b < B; b += 4) s = fmaf(coef[b], GZ[b * td.wz + j], s);
            dc[r * 64 + j] = s;
        }
        __syncthreads();
        if (tid < 64) dc[tid] = (dc[tid] + dc[64 + tid]) + (dc[128 + tid] + dc[192 + tid]);
        __syncthreads();
    }
    for (int e = tid; e < DD; e += AFF_THREADS) {
        const int i = e / D, j = e - i * D;
        const float raw = ga_ws[((size_t)layer * td.wz + i) * td.wz + j];
        dW[e] = an ? expf(-src.an_s[i]) * (raw - src.an_t[i] * dc[j]) : raw;
        Lm[e] = i == j ? 1.f : (i > j ? src.L[e] : 0.f);
        Um[i * US + j] = i == j ? src.signS[i] * expf(src.logS[i]) : (i < j ? src.U[e] : 0.f);
        Ps[e] = src.P[e];
    }
    __syncthreads();
    for (int e = tid; e < DD; e += AFF_THREADS) {
        const int i = e / D, j = e - i * D;
        float s = 0.f, st = 0.f;
#pragma unroll 8
        for (int k = 0; k < D; ++k) {
            s = fmaf(Ps[i * D + k], Lm[k * D + j], s);
            st = fmaf(dW[i * D + k], Um[j * US + k], st);            // T = dW Um^T
        }
        PL[e] = s;
        T[e] = st;
    }
    __syncthreads();
    float* G = grads + (size_t)layer * gl.layer_stride;
    if (an && tid < D) {                                             // ActNorm gradients
        float gs = 0.f, gt = 0.f;
        const int i = tid;
        for (int j = 0; j < D; ++j) {
            float w = 0.f;
            for (int k = 0; k < D; ++k) w = fmaf(PL[i * D + k], Um[k * US + j], w);          // W_ij = (P Lm Um)_ij
            gs = fmaf(-w, dW[i * D + j], gs);
            gt = fmaf(w, dc[j], gt);
        }
        grads[gl.an_base + (long)layer * 2 * D + i] = gs - csum;
        grads[gl.an_base + (long)layer * 2 * D + D + i] = -expf(-src.an_s[i]) * gt;
    }
    for (int e = tid; e < DD; e += AFF_THREADS) {
        const int i = e / D, j = e - i * D;
        float su = 0.f, sl = 0.f;
#pragma unroll 8
        for (int k = 0; k < D; ++k) {
            su = fmaf(PL[k * D + i], dW[k * D + j], su);            // dUm = (P Lm)^T dW
            sl = fmaf(Ps[k * D + i], T[k * D + j], sl);             // dLm = P^T (dW Um^T)
        }
        G[gl.U + e] = i < j ? su : 0.f;
        G[gl.L + e] = i > j ? sl : 0.f;
        if (i == j) G[gl.logS + i] = su * Um[i * US + j] + csum;             // d/dlog_S of s e^{log_S} (+ the +sum(log_S) log-det)
    }
}

template <int NTWM>
static int launch_log_prob_tape(const FlowDims& f, const TapeDims& td, const float* packed, const float* x,
                                float* log_q, float* grad, float* tape, long B, const int64_t* rows, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div((int)B, ROWS)), block(NTHREADS);
    const FlowLds l = make_flow_lds(f, true);
    const size_t bytes = (size_t)l.total * 4;
    FAB_TRY(set_max_lds((const void*)k_flow_log_prob_tape<NTWM>, bytes));
    hipLaunchKernelGGL((k_flow_log_prob_tape<NTWM>), grid, block, bytes, st, f, l, td, packed, x, log_q, grad, tape, B, rows);
    return check_launch();
}

template <int NTWM>
static int launch_sample_bwd(const FlowDims& f, const TapeDims& td, const float* packed, const float* x,
                             const float* gx, const float* gl, float* g_eps, float* tape, long B, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div((int)B, ROWS)), block(NTHREADS);
    const FlowLds l = make_flow_lds(f, false);
    const size_t bytes = ((size_t)l.total + ROWS * l.DS + 2 * NTHREADS) * 4;
    FAB_TRY(set_max_lds((const void*)k_flow_sample_bwd<NTWM>, bytes));
    hipLaunchKernelGGL((k_flow_sample_bwd<NTWM>), grid, block, bytes, st, f, l, td, packed, x, gx, gl, g_eps, tape, B);
    return check_launch();
}

int launch_affine_grads(const FlowDims& f, const TapeDims& td, const GradLayout& gl, const fabhip_flow_params* params,
                        const float* ga, float* grads, const float* tp, const float* coef, long B, hipStream_t st) {
    const size_t smem = ((size_t)6 * f.D * f.D + f.D + 256) * 4;
    FAB_TRY(set_max_lds((const void*)k_affine_grads, smem));
    for (int k = 0; k < f.K; ++k)
        if (!params->lu_L[k] || !params->lu_U[k] || !params->log_S[k] || !params->sign_S[k] || !params->perm_P[k] ||
            (!params->an_s[k] != !params->an_t[k]))
            return FABHIP_EINVAL;
    for (int k0 = 0; k0 < f.K; k0 += LBATCH_T) {
        const int nl = f.K - k0 < LBATCH_T ? f.K - k0 : LBATCH_T;
        AffineSrcTab tab;
        for (int y = 0; y < LBATCH_T; ++y) {
            const int k = k0 + (y < nl ? y : 0);
            tab.L[y] = params->lu_L[k]; tab.U[y] = params->lu_U[k]; tab.logS[y] = params->log_S[k];
            tab.signS[y] = params->sign_S[k]; tab.P[y] = params->perm_P[k];
            tab.an_s[y] = params->an_s[k]; tab.an_t[y] = params->an_t[k];
        }
        hipLaunchKernelGGL(k_affine_grads, dim3(nl), dim3(AFF_THREADS), smem, st, f, td, gl, tab, k0, ga, grads, tp, coef, B);
    }
    return check_launch();
}

}  // namespace fab

using namespace fab;

extern "C" {

int64_t fabhip_flow_grad_floats(int32_t dim, int32_t n_layers, int32_t width) {
    if (check_flow_shape(dim, n_layers, width) != FABHIP_OK) return -1;
    return (int64_t)make_grad_layout(make_flow_dims(dim, n_layers, width)).total;
}

int fabhip_flow_grad_layout(int32_t dim, int32_t n_layers, int32_t width, int64_t* out15) {
    if (!out15) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(dim, n_layers, width));
    const GradLayout g = make_grad_layout(make_flow_dims(dim, n_layers, width));
    const long v[15] = {g.layer_stride, g.w1, g.b1, g.w2, g.b2, g.w3, g.b3, g.L, g.U, g.logS, g.loc, g.log_scale,
                        g.total, g.an_base, g.total_an};
    for (int i = 0; i < 15; ++i) out15[i] = v[i];
    return FABHIP_OK;
}

size_t fabhip_flow_tape_bytes(int32_t dim, int32_t n_layers, int32_t width, int64_t B) {
    if (check_flow_shape(dim, n_layers, width) != FABHIP_OK || B < 0) return 0;
    const FlowDims f = make_flow_dims(dim, n_layers, width);
    return tape_floats(f, make_tape_dims(f, (long)B)) * sizeof(float);
}

int fabhip_flow_tape_layout(int32_t dim, int32_t n_layers, int32_t width, int64_t B, int64_t* out18) {
    if (!out18 || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(dim, n_layers, width));
    const TapeDims t = make_tape_dims(make_flow_dims(dim, n_layers, width), (long)B);
    const long v[18] = {t.Bp, t.wz, t.w1, t.wh, t.wp, t.we, t.wb, t.o_ZA, t.o_GZ, t.o_Z1, t.o_H1, t.o_H2, t.o_DP,
                        t.o_E2, t.o_E1, t.layer_stride, t.o_TB, t.total};
    for (int i = 0; i < 18; ++i) out18[i] = v[i];
    return FABHIP_OK;
}

// rows != nullptr: batch row g is row rows[g] of x (fabhip_flow_log_prob_tape_rows)
static int log_prob_tape_impl(const fabhip_flow* flow, const float* x, const int64_t* rows, float* log_q, float* grad_x, int64_t B,
                              void* tape, size_t tape_bytes, fabhip_stream_t stream) {
    if (!flow || !flow->packed || !x || !log_q || !tape || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(flow->dim, flow->n_layers, flow->width));
    if (B == 0) return FABHIP_OK;
    const FlowDims f = make_flow_dims(flow->dim, flow->n_layers, flow->width);
    const TapeDims td = make_tape_dims(f, (long)B);
    if (tape_bytes < tape_floats(f, td) * sizeof(float)) return FABHIP_ENOSPC;
    // 8-chain stream tiles where the flow has that image (D <= 32, hidden width padded to 256 / 320): twice the workgroups of the
    // 16-chain kernel on the same batch - a 2048-row minibatch fills the chip
    if (option(FABHIP_OPT_TAPE_TILES) != 16 && f.o_r8 >= 0)
        return launch_log_prob_tape_r8(f, td, flow->packed, x, rows, log_q, grad_x, (float*)tape, (long)B, (hipStream_t)stream);
    FAB_DISPATCH_NTW(f, launch_log_prob_tape, f, td, flow->packed, x, log_q, grad_x, (float*)tape, (long)B, rows,
                     (hipStream_t)stream);
}

int fabhip_flow_log_prob_tape(const fabhip_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                              void* tape, size_t tape_bytes, fabhip_stream_t stream) {
    return log_prob_tape_impl(flow, x, nullptr, log_q, grad_x, B, tape, tape_bytes, stream);
}

int fabhip_flow_log_prob_tape_rows(const fabhip_flow* flow, const float* x, const int64_t* rows, float* log_q, float* grad_x,
                                   int64_t B, void* tape, size_t tape_bytes, fabhip_stream_t stream) {
    if (!rows) return FABHIP_EINVAL;
    return log_prob_tape_impl(flow, x, rows, log_q, grad_x, B, tape, tape_bytes, stream);
}

int fabhip_flow_sample_grad_tape(const fabhip_flow* flow, const float* x, const float* grad_x, const float* grad_log_q,
                                 float* grad_eps, int64_t B, void* tape, size_t tape_bytes, fabhip_stream_t stream) {
    if (!flow || !flow->packed || !x || !grad_x || !grad_log_q || !tape || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(flow->dim, flow->n_layers, flow->width));
    if (B == 0) return FABHIP_OK;
    const FlowDims f = make_flow_dims(flow->dim, flow->n_layers, flow->width);
    const TapeDims td = make_tape_dims(f, (long)B);
    if (tape_bytes < tape_floats(f, td) * sizeof(float)) return FABHIP_ENOSPC;
    FAB_DISPATCH_NTW(f, launch_sample_bwd, f, td, flow->packed, x, grad_x, grad_log_q, grad_eps, (float*)tape, (long)B,
                     (hipStream_t)stream);
}

int fabhip_flow_param_grad(const fabhip_flow_params* params, const fabhip_flow* flow, const void* tape,
                           size_t tape_bytes, const float* coef, int64_t B, float* grads, fabhip_stream_t stream) {
    if (!params || !flow || !flow->packed || !tape || !coef || !grads || B < 1) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(flow->dim, flow->n_layers, flow->width));
    if (params->dim != flow->dim || params->n_layers != flow->n_layers || params->width != flow->width)
        return FABHIP_EINVAL;
    const FlowDims f = make_flow_dims(flow->dim, flow->n_layers, flow->width);
    const TapeDims td = make_tape_dims(f, (long)B);
    if (tape_bytes < tape_floats(f, td) * sizeof(float)) return FABHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    const GradLayout gl = make_grad_layout(f);
    const float* tp = (const float*)tape;
    float* ga = const_cast<float*>(tp) + tape_ga_offset(td);
    if (option(FABHIP_OPT_PGRAD) != 0) {
        // one workgroup per output tile, operands straight from the tape into the matrix cores (train_step.hip)
        FAB_TRY(launch_param_grad_tiles(f, td, gl, tp, coef, (long)B, grads, ga, st));
    } else {
        const GemmBlocks gb = make_gemm_blocks(f, td);
        const int nbase = ceil_div(td.wb, 64) * gb.q1;               // base-distribution blocks after the K layers
        hipLaunchKernelGGL(k_param_grad, dim3((unsigned)(f.K * gb.per_layer + nbase)), dim3(256), 0, st, f, td, gb, gl,
                           tp, coef, (long)B, grads, ga);
    }
    return launch_affine_grads(f, td, gl, params, ga, grads, tp, coef, (long)B, st);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Optimiser step on the flat parameter image: global-norm gradient clipping
// (torch.nn.utils.clip_grad_norm_, fab/train_with_prioritised_buffer.py:174) + Adam
// (torch.optim.Adam single-tensor formulas) in two launches, no host synchronisation: a non-finite
// gradient norm skips the update on the device (the reference skips it on the host, :175-179).
// ------------------------------------------------------------------------------------------------
namespace fab {

constexpr int ADAM_BLOCKS = 512;

// the minibatch's loss and logging statistics from the partial sums the tape kernel's tail left (MbTail): one extra workgroup of the
// norm launch, thread t adds partials t, t + 256, ... in order, then a fixed tree over the threads; the flag `loss_ok` makes
// k_adam_clip skip the update of a non-finite loss (fab/train_with_prioritised_buffer.py:172-181)
__device__ void mb_finish(const float* __restrict__ part, int n_part, long B, float* __restrict__ stats, int* __restrict__ loss_ok,
                          float (*red)[256]) {
    const int tid = threadIdx.x;
    float t[6] = {0.f, 0.f, 0.f, INFINITY, -INFINITY, 0.f};
    for (int i = tid; i < n_part; i += 256) {
        const float* p = part + (size_t)i * MB_PART;
        t[0] += p[0]; t[1] += p[1]; t[2] += p[2]; t[3] = fminf(t[3], p[3]); t[4] = fmaxf(t[4], p[4]); t[5] += p[5];
    }
    for (int q = 0; q < 6; ++q) red[q][tid] = t[q];
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) {
            red[0][tid] += red[0][tid + k]; red[1][tid] += red[1][tid + k]; red[2][tid] += red[2][tid + k];
            red[3][tid] = fminf(red[3][tid], red[3][tid + k]); red[4][tid] = fmaxf(red[4][tid], red[4][tid + k]);
            red[5][tid] += red[5][tid + k];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float Bf = (float)B, loss = -(red[0][0] / Bf);
        const bool sawnan = red[5][0] > 0.f;                  // torch.min / max propagate NaN
        stats[0] = loss; stats[1] = red[1][0] / Bf; stats[2] = sawnan ? NAN : red[3][0]; stats[3] = sawnan ? NAN : red[4][0];
        stats[4] = red[2][0] / Bf; stats[6] = 0.f; stats[7] = 0.f;
        *loss_ok = isfinite(loss) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void k_sqnorm_partial(const float* __restrict__ g, long n, double* __restrict__ part, int nblk,
                                                        int* __restrict__ steps_copy, const int* __restrict__ step_count,
                                                        const float* __restrict__ mb_part, int n_mb, long B,
                                                        float* __restrict__ stats) {
    __shared__ double red[256];
    __shared__ float redf[6][256];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x == nblk) {                            // the extra workgroup (launched only with a minibatch tail)
        mb_finish(mb_part, n_mb, B, stats, steps_copy + 1, redf);
        return;
    }
    // the step counter as k_adam_clip (the launch behind this one) sees it: its blocks read this copy, its block 0 writes the
    // counter itself - no third launch and no block reads what another one writes
    if (blockIdx.x == 0 && tid == 0) { steps_copy[0] = *step_count; if (!mb_part) steps_copy[1] = 1; }
    double s = 0.0;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * 256 + tid; i < n4; i += (long)nblk * 256) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && tid < (int)(n & 3)) { const float v = g[(n4 << 2) + tid]; s += (double)v * v; }
    red[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) red[tid] += red[tid + k];
        __syncthreads();
    }
    if (tid == 0) part[blockIdx.x] = red[0];
}

struct AdamK {
    float lr, beta1, beta2, eps, max_norm;
};

__device__ __forceinline__ void adam_one(float& th, float gi, float& mi, float& vi, float coef, const AdamK& a, float step_size,
                                         float bc2_sqrt) {
    gi *= coef;
    mi = mi + (gi - mi) * (1.f - a.beta1);                    // exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * a.beta2 + (1.f - a.beta2) * (gi * gi);          // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
    th = th - step_size * (mi / denom);                       // addcdiv_(exp_avg, denom, -step_size)
}

// VEC: all four images 16-byte aligned (float4 accesses; the tail of n mod 4 by block 0).  Block 0 advances the step counter (the
// blocks read k_sqnorm_partial's copy of it).
template <bool VEC>
__global__ __launch_bounds__(256) void k_adam_clip(float* __restrict__ theta, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, AdamK a,
                                                   const double* __restrict__ part, int nparts, int* __restrict__ step_count,
                                                   float* __restrict__ norm_out, const int* __restrict__ steps_copy) {
    // the norm from the block partials (one wave, fixed order), the clip coefficient and Adam's bias corrections: once per block
    __shared__ float bc[4];
    const int tid = threadIdx.x;
    if (tid < 64) {
        double s = 0.0;
        for (int i = tid; i < nparts; i += 64) s += part[i];
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_down(s, o);
        if (tid == 0) {
            const float total = (float)sqrt(s);
            float coef = 1.f;
            if (a.max_norm > 0.f) { coef = a.max_norm / (total + 1e-6f); coef = coef > 1.f ? 1.f : coef; }
            // t = applied steps so far + 1 (a skipped step does not advance the bias correction, like the reference, which simply
            // does not call optimizer.step()); bias corrections in double as torch computes them on the host
            const double t = (double)(*steps_copy + 1);
            bc[0] = total; bc[1] = coef;
            bc[2] = (float)((double)a.lr / (1.0 - pow((double)a.beta1, t)));
            bc[3] = (float)sqrt(1.0 - pow((double)a.beta2, t));
        }
    }
    __syncthreads();
    const bool loss_ok = steps_copy[1] != 0;                    // (a non-finite minibatch loss: "nan loss in replay step")
    const float total = loss_ok ? bc[0] : NAN;
    if (blockIdx.x == 0 && tid == 0) *norm_out = total;
    const bool apply = isfinite(total);                         // "nan grad norm": no step
    if (apply) {
        const float coef = bc[1], step_size = bc[2], bc2_sqrt = bc[3];
        if constexpr (VEC) {
            const long n4 = n >> 2;
            float4* th4 = reinterpret_cast<float4*>(theta);
            const float4* g4 = reinterpret_cast<const float4*>(g);
            float4* m4 = reinterpret_cast<float4*>(m);
            float4* v4 = reinterpret_cast<float4*>(v);
            for (long i = (long)blockIdx.x * 256 + tid; i < n4; i += (long)gridDim.x * 256) {
                float4 th = th4[i], mi = m4[i], vi = v4[i];
                const float4 gi = g4[i];
                adam_one(th.x, gi.x, mi.x, vi.x, coef, a, step_size, bc2_sqrt);
                adam_one(th.y, gi.y, mi.y, vi.y, coef, a, step_size, bc2_sqrt);
                adam_one(th.z, gi.z, mi.z, vi.z, coef, a, step_size, bc2_sqrt);
                adam_one(th.w, gi.w, mi.w, vi.w, coef, a, step_size, bc2_sqrt);
                th4[i] = th; m4[i] = mi; v4[i] = vi;
            }
            if (blockIdx.x == 0 && tid < (int)(n & 3)) {
                const long i = (n4 << 2) + tid;
                adam_one(theta[i], g[i], m[i], v[i], coef, a, step_size, bc2_sqrt);
            }
        } else {
            for (long i = (long)blockIdx.x * 256 + tid; i < n; i += (long)gridDim.x * 256)
                adam_one(theta[i], g[i], m[i], v[i], coef, a, step_size, bc2_sqrt);
        }
    }
    if (blockIdx.x == 0 && tid == 0 && apply) *step_count = *steps_copy + 1;
}

int adam_clip_step_impl(float* theta, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        int32_t* step_count, float max_norm, float* grad_norm_out, void* workspace, size_t workspace_bytes,
                        const float* mb_partials, int n_partials, float* stats, long B, hipStream_t st) {
    if (!theta || !grad || !m || !v || n < 1 || !step_count || !workspace) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_adam_workspace_bytes(n)) return FABHIP_ENOSPC;
    if (((uintptr_t)grad & 15) != 0 || ((uintptr_t)workspace & 7) != 0) return FABHIP_EINVAL;
    if (mb_partials && (!stats || n_partials < 1 || B < 1)) return FABHIP_EINVAL;
    double* part = (double*)workspace;
    int* steps_copy = (int*)(part + ADAM_BLOCKS);             // [0] the step counter as the update launch sees it, [1] loss finite
    const long per = 256 * 4;
    int nb = (int)((n + per - 1) / per);
    if (nb > ADAM_BLOCKS) nb = ADAM_BLOCKS;
    hipLaunchKernelGGL(k_sqnorm_partial, dim3(nb + (mb_partials ? 1 : 0)), dim3(256), 0, st, grad, (long)n, part, nb, steps_copy,
                       (const int*)step_count, mb_partials, n_partials, B, stats);
    AdamK a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm;
    float* norm = grad_norm_out ? grad_norm_out : (float*)(steps_copy + 2);
    const bool vec = (((uintptr_t)theta | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    int nb2 = (int)((n + per - 1) / per);          // one float4 per thread (a second trip of a few threads doubles the launch)
    if (nb2 > 8192) nb2 = 8192;
    if (vec)
        hipLaunchKernelGGL(k_adam_clip<true>, dim3(nb2), dim3(256), 0, st, theta, grad, m, v, (long)n, a, part, nb,
                           (int*)step_count, norm, (const int*)steps_copy);
    else
        hipLaunchKernelGGL(k_adam_clip<false>, dim3(nb2), dim3(256), 0, st, theta, grad, m, v, (long)n, a, part, nb,
                           (int*)step_count, norm, (const int*)steps_copy);
    return check_launch();
}

}  // namespace fab

extern "C" {

size_t fabhip_adam_workspace_bytes(int64_t n) { (void)n; return (size_t)fab::ADAM_BLOCKS * sizeof(double) + 32; }

int fabhip_adam_clip_step(float* theta, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                          float beta2, float eps, int32_t* step_count, float max_norm, float* grad_norm_out,
                          void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    return fab::adam_clip_step_impl(theta, grad, m, v, n, lr, beta1, beta2, eps, step_count, max_norm, grad_norm_out, workspace,
                                    workspace_bytes, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}

int fabhip_tape_gemm(const float* Y, int64_t y_layer_stride, int32_t ldy, int32_t P, const float* X, int64_t x_layer_stride,
                     int32_t ldx, int32_t Q, const float* coef, int64_t B, int32_t L, float* C, float* colsum,
                     fabhip_stream_t stream) {
    if (!Y || !X || !coef || !C || B < 1 || L < 1 || P < 1 || Q < 1 || ldy < P || ldx < Q) return FABHIP_EINVAL;
    fab::TapeGemm g;
    g.Y = Y; g.X = X; g.coef = coef; g.y_stride = y_layer_stride; g.x_stride = x_layer_stride; g.B = B;
    g.ldy = ldy; g.ldx = ldx; g.P = P; g.Q = Q; g.pblocks = ceil_div(P, 64); g.qblocks = ceil_div(Q, 64);
    g.C = C; g.S = colsum;
    hipLaunchKernelGGL(fab::k_tape_gemm, dim3((unsigned)(L * g.pblocks * g.qblocks)), dim3(256), 0, (hipStream_t)stream, g);
    return check_launch();
}

}  // extern "C"
