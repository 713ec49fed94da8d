// Device-side RealNVP evaluation on a 16-chain tile held by one 256-thread workgroup.
//
// Every linear map is a [16 x K] @ [K x N] product on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, bit-identical to an fmaf chain): activations live in LDS
// (row-major, leading dimension padded by 4 floats), weights are streamed from the L2-resident
// packed image straight into VGPRs (each weight is used by exactly one wave of the workgroup, so an
// LDS round trip would be pure overhead) with a software prefetch ring.
//
// Packed B-operand tile (c, S) of a K x N matrix: 64 lanes x float4, lane l = (q = l>>4, n = l&15)
// holds  B[16S + 4q + t][16c + n], t = 0..3.  MFMA step t of k-block S therefore multiplies
// A[row][16S + 4q + t] (one ds_read_b128 per lane per k-block) with that register.
#pragma once
#include "fabhip_common.h"

namespace fab {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float row16_sum(float v) {
    // sum over the 16 lanes that share a chain row (tid = row*16 + c); fixed xor tree => deterministic
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

struct Tid {
    int tid, wave, lane, q, n, row, c;
    __device__ __forceinline__ Tid() {
        tid = threadIdx.x;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        lane = tid & 63;
        q = lane >> 4;
        n = lane & 15;
        row = tid >> 4;   // elementwise mapping: 16 rows x 16 column lanes
        c = tid & 15;
    }
};

// ---- weight-tile loads hidden from hipcc's s_waitcnt bookkeeping --------------------------------
// hipcc drains vmcnt(0) at the head of every loop iteration that carries loads in flight, which
// serialises a register prefetch ring.  The ring loads are therefore issued through inline asm and
// waited for with hand-counted s_waitcnt vmcnt(N) (loads return in order, so N = number of ring loads
// issued after the one needed; any other load in flight only makes the wait more conservative).
__device__ __forceinline__ void gload16(f32x4& dst, const float4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}

__device__ __forceinline__ void gload4(float& dst, const float* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p));
}

template <int N, int NT>
__device__ __forceinline__ void wait_vals(float (&r)[NT]) {       // vmcnt(N) with the registers as operands
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[0]) : "n"(N));
    else if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N));
    else if constexpr (NT == 4)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
    else if constexpr (NT == 5)
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : "n"(N));
    else if constexpr (NT == 8)
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                     : "n"(N));
    else static_assert(NT == 1, "unsupported tile count");
}

template <int N, int NT>
__device__ __forceinline__ void wait_tiles(f32x4 (&r)[NT]) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[0]) : "n"(N));
    else if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N));
    else if constexpr (NT == 4)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
    else if constexpr (NT == 5)
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : "n"(N));
    else if constexpr (NT == 8)
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                     : "n"(N));
    else static_assert(NT == 1, "unsupported tile count");
}

// ---- N-split GEMM: wave w owns column tiles c_i = w + 4 i, i < NTWM (the tile count is padded to
// 4*NTWM with zero tiles at pack time, so the hot loop has no predication at all).
// acc[i] += A[16 x 16*KB] @ B[:, tile c_i];  KB % DEPTH == 0 (K padded at pack time).
// Weight tiles are fetched DEPTH k-blocks ahead into a register ring (DEPTH*NTWM KiB in flight per
// wave); the A fragment (one ds_read_b128 per k-block) is fetched one k-block ahead.
// With `bias` the accumulators start from bias[16 c_i + n] (loaded through the same hidden path, issued
// BEFORE the ring so that the ring stays in flight while they are waited for) instead of the caller's values.
template <int NTWM, int DEPTH, bool MASKK, bool BIAS = false>
__device__ __forceinline__ void gemm_nsplit(const float* __restrict__ A, int lda, int kmax, int KB,
                                            const float4* __restrict__ Bp, const Tid& t, f32x4 (&acc)[NTWM],
                                            const float* __restrict__ bias = nullptr) {
    const float* arow = A + t.n * lda + 4 * t.q;
    const float4* bt = Bp + (size_t)t.wave * KB * 64 + t.lane;     // tile c_i at bt + i*4*KB*64
    const size_t tstride = (size_t)4 * KB * 64;
    float bv[NTWM];
    if (BIAS) {
#pragma unroll
        for (int i = 0; i < NTWM; ++i) gload4(bv[i], bias + 16 * (t.wave + 4 * i) + t.n);
    }
    f32x4 ring[DEPTH][NTWM];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < NTWM; ++i) gload16(ring[d][i], bt + i * tstride + (size_t)d * 64);
    float4 a_nxt = *reinterpret_cast<const float4*>(arow);
    if (BIAS) {
        wait_vals<DEPTH * NTWM, NTWM>(bv);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){bv[i], bv[i], bv[i], bv[i]};
    }
    for (int S0 = 0; S0 < KB; S0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int S = S0 + d;
            float4 a = a_nxt;
            const int Sn = (S + 1 < KB) ? S + 1 : S;
            a_nxt = *reinterpret_cast<const float4*>(arow + 16 * Sn);
            if (MASKK) {
                const int k0 = 16 * S + 4 * t.q;
                a.x = (k0 + 0 < kmax) ? a.x : 0.f;
                a.y = (k0 + 1 < kmax) ? a.y : 0.f;
                a.z = (k0 + 2 < kmax) ? a.z : 0.f;
                a.w = (k0 + 3 < kmax) ? a.w : 0.f;
            }
            // ring slot d holds k-block S; (DEPTH-1)*NTWM younger ring loads may stay in flight
            wait_tiles<(DEPTH - 1) * NTWM, NTWM>(ring[d]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.x, ring[d][i].x, acc[i]);
#pragma unroll
            for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.y, ring[d][i].y, acc[i]);
#pragma unroll
            for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.z, ring[d][i].z, acc[i]);
#pragma unroll
            for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.w, ring[d][i].w, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
            // refill this slot with k-block S + DEPTH (clamped: the tail re-reads the last block, which
            // keeps the in-flight count constant so that the hand-counted vmcnt stays exact)
            const int Sp = (S + DEPTH < KB) ? S + DEPTH : KB - 1;
#pragma unroll
            for (int i = 0; i < NTWM; ++i) gload16(ring[d][i], bt + i * tstride + (size_t)Sp * 64);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // The clamped tail loads are still in flight.  Drain them with the ring registers as operands of the
    // wait: otherwise hipcc, which believes they are dead, may re-allocate them above the wait and the
    // landing loads would clobber live values.
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) wait_tiles<0, NTWM>(ring[d]);
    __builtin_amdgcn_sched_barrier(0);
}

// ---- K-split GEMM for narrow outputs (N = 16*NT <= 64): wave w sums k-blocks S = w, w+4, ... of
// every column tile; the four partial [16 x 16*NT] products go to LDS part[w][row][PN] and are added by
// the caller.  Tiles are the outer (runtime) loop so that no register array is indexed by NT.
__device__ __forceinline__ void gemm_ksplit(const float* __restrict__ A, int lda, int KB,
                                            const float4* __restrict__ Bp, int NT, float* __restrict__ part,
                                            int PN, const Tid& t) {
    const float* arow = A + t.n * lda + 4 * t.q;
    float* p = part + (size_t)t.wave * ROWS * PN;
    for (int i = 0; i < NT; ++i) {
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        const float4* bt = Bp + (size_t)i * KB * 64 + t.lane;
        int S = t.wave;
        for (; S + NWAVE < KB; S += 2 * NWAVE) {            // two independent accumulation chains
            const float4 b0 = bt[(size_t)S * 64], b1 = bt[(size_t)(S + NWAVE) * 64];
            const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * S);
            const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * (S + NWAVE));
            acc0 = mfma4(a0.x, b0.x, acc0); acc1 = mfma4(a1.x, b1.x, acc1);
            acc0 = mfma4(a0.y, b0.y, acc0); acc1 = mfma4(a1.y, b1.y, acc1);
            acc0 = mfma4(a0.z, b0.z, acc0); acc1 = mfma4(a1.z, b1.z, acc1);
            acc0 = mfma4(a0.w, b0.w, acc0); acc1 = mfma4(a1.w, b1.w, acc1);
        }
        if (S < KB) {
            const float4 b0 = bt[(size_t)S * 64];
            const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * S);
            acc0 = mfma4(a0.x, b0.x, acc0);
            acc0 = mfma4(a0.y, b0.y, acc0);
            acc0 = mfma4(a0.z, b0.z, acc0);
            acc0 = mfma4(a0.w, b0.w, acc0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) p[(4 * t.q + r) * PN + 16 * i + t.n] = acc0[r] + acc1[r];
    }
}

__device__ __forceinline__ float part_sum(const float* part, int PN, int row, int col) {
    float s = part[row * PN + col];
    s += part[(ROWS + row) * PN + col];
    s += part[(2 * ROWS + row) * PN + col];
    s += part[(3 * ROWS + row) * PN + col];
    return s;
}

// depth of the weight ring of the W x W GEMMs (must divide KBW = 4 * NTWM): how many k-blocks of weights
// are in flight per wave.  L2 misses go to the Infinity Cache (~1 us): 5 blocks x 640 MFMA-cycles cover it.
#ifndef FAB_DEPTH_W5
#define FAB_DEPTH_W5 4
#endif
template <int NTWM>
__host__ __device__ constexpr int depth_w() { return NTWM == 5 ? FAB_DEPTH_W5 : (NTWM == 8 ? 2 : 4); }

// hidden layer: OUT = relu(A @ B + bias).  With MASK the ReLU sign pattern of this lane's 4*NTWM outputs is
// kept as one 32-bit word per thread (bit 4 i + r) for the reverse sweep: the same lane of the same wave
// owns the same (tile, register) there, so no cross-lane exchange and a single LDS store per GEMM.
template <int NTWM, int DEPTH, bool MASKK, bool MASK>
__device__ __forceinline__ void dense_relu(const float* A, int lda, int kmax, int KB, const float4* Bp,
                                           const float* __restrict__ bias, float* OUT, int ldo,
                                           unsigned* mask, const Tid& t) {
    f32x4 acc[NTWM];
    gemm_nsplit<NTWM, DEPTH, MASKK, true>(A, lda, kmax, KB, Bp, t, acc, bias);
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const int c = t.wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[i][r];
            const bool pos = v > 0.f;
            OUT[(4 * t.q + r) * ldo + 16 * c + t.n] = pos ? v : 0.f;
            if (MASK) m |= (pos ? 1u : 0u) << (4 * i + r);
        }
    }
    if (MASK) mask[t.tid] = m;
}

// backward of a hidden layer: OUT = (A @ B) * mask
template <int NTWM, int DEPTH>
__device__ __forceinline__ void dense_masked(const float* A, int lda, int KB, const float4* Bp, float* OUT,
                                             int ldo, const unsigned* mask, const Tid& t) {
    f32x4 acc[NTWM];
#pragma unroll
    for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned m = mask[t.tid];
    gemm_nsplit<NTWM, DEPTH, false>(A, lda, 0, KB, Bp, t, acc);
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const int c = t.wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool pos = (m >> (4 * i + r)) & 1u;
            OUT[(4 * t.q + r) * ldo + 16 * c + t.n] = pos ? acc[i][r] : 0.f;
        }
    }
}

// OUT[16 x 16*NT] = A[16 x K] @ B   (NT <= 4: one column tile per wave), used for the D x D affine maps
__device__ __forceinline__ void dense_small(const float* A, int lda, int kmax, int KB, const float4* Bp,
                                            int NT, float* OUT, int ldo, const Tid& t) {
    if (t.wave < NT) {
        f32x4 acc[1];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gemm_nsplit<1, 2, true>(A, lda, kmax, KB, Bp, t, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) OUT[(4 * t.q + r) * ldo + 16 * t.wave + t.n] = acc[0][r];
    }
}

// conditioner MLP of one coupling layer: PART <- partial sums of relu(relu(z1 W1 + b1) W2 + b2) W3
template <int NTWM, bool MASK>
__device__ __forceinline__ void coupling_mlp(const FlowDims& f, const FlowLds& l, const float* Lp, float* lds,
                                             const float* Z, int layer, const Tid& t) {
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
    unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
    dense_relu<NTWM, 2, true, MASK>(Z, l.DS, f.d, f.KBd, reinterpret_cast<const float4*>(Lp + f.o_W1), Lp + f.o_b1,
                                    HA, l.WS, mk, t);
    __syncthreads();
    dense_relu<NTWM, depth_w<NTWM>(), false, MASK>(HA, l.WS, f.Wp, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_W2), Lp + f.o_b2,
                                     HB, l.WS, mk + NTHREADS, t);
    __syncthreads();
    gemm_ksplit(HB, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_W3), f.NTO, lds + l.o_PART, l.PN, t);
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// log q(x) (and d log q / dx) for the 16 rows in U0 (columns >= D must be zero).
// Returns log q of this thread's row (replicated over the row's 16 lanes).  With GRAD the gradient
// is left in the state buffer whose LDS offset is returned through *grad_off.
// normflows NormalizingFlow.log_prob: inverses in reversed layer order, log-dets added, base last.
// ------------------------------------------------------------------------------------------------
template <int NTWM, bool GRAD>
__device__ float flow_log_prob_tile(const FlowDims& f, const FlowLds& l, const float* __restrict__ packed,
                                    float* lds, const Tid& t, int* grad_off) {
    int cur = l.o_U0, nxt = l.o_U1;
    float logq = 0.f;
    float* PART = lds + l.o_PART;
    for (int layer = f.K - 1; layer >= 0; --layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        // InvertibleAffine.inverse: z <- z @ (P L U), log_det = +sum(log_S)
        dense_small(lds + cur, l.DS, f.D, f.KBD, reinterpret_cast<const float4*>(Lp + f.o_AW), f.NTD, lds + nxt,
                    l.DS, t);
        logq += Lp[f.o_logS];
        __syncthreads();
        float* Z = lds + nxt;
        coupling_mlp<NTWM, GRAD>(f, l, Lp, lds, Z, layer, t);
        // AffineCoupling.inverse: z2 <- (z2 - shift) * exp(-s), log_det = -sum(s)
        float ssum = 0.f;
        for (int j = t.c; j < f.DO; j += 16) {
            const float shift = part_sum(PART, l.PN, t.row, j) + Lp[f.o_b3 + j];
            const float s = part_sum(PART, l.PN, t.row, f.DOp + j) + Lp[f.o_b3 + f.DOp + j];
            const float es = expf(-s);
            const float v2 = (Z[t.row * l.DS + f.d + j] - shift) * es;
            Z[t.row * l.DS + f.d + j] = v2;
            if (GRAD) {
                lds[l.o_ES + ((size_t)layer * ROWS + t.row) * f.DOp + j] = es;
                lds[l.o_V2 + ((size_t)layer * ROWS + t.row) * f.DOp + j] = v2;
            }
            ssum += s;
        }
        logq += -row16_sum(ssum);
        __syncthreads();
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    // DiagGaussian.log_prob
    const float* base = packed + f.o_base;
    float* Zc = lds + cur;
    float bsum = 0.f;
    for (int j = t.c; j < f.D; j += 16) {
        const float ls = base[f.Dp + j];
        const float sc = expf(ls);
        const float zn = (Zc[t.row * l.DS + j] - base[j]) / sc;
        bsum += ls + 0.5f * (zn * zn);
        if (GRAD) Zc[t.row * l.DS + j] = -(zn / sc);      // d/dz of -0.5 ((z - loc)/sc)^2
    }
    logq += -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
    if (!GRAD) return logq;
    __syncthreads();

    // ---- reverse sweep: g = d log q / d(state), layers 0 .. K-1 -----------------------------------
    float* DP = lds + l.o_DP;
    for (int layer = 0; layer < f.K; ++layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        float* G = lds + cur;
        for (int j = t.c; j < f.DO; j += 16) {
            const float g2 = G[t.row * l.DS + f.d + j];
            const float es = lds[l.o_ES + ((size_t)layer * ROWS + t.row) * f.DOp + j];
            const float v2 = lds[l.o_V2 + ((size_t)layer * ROWS + t.row) * f.DOp + j];
            DP[t.row * l.PS + j] = -(g2 * es);                    // d/d shift
            DP[t.row * l.PS + f.DOp + j] = -(g2 * v2) - 1.f;       // d/d s  (incl. the -sum(s) log-det)
            G[t.row * l.DS + f.d + j] = g2 * es;                  // d/d z2
        }
        __syncthreads();
        const unsigned* mk = reinterpret_cast<const unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        dense_masked<NTWM, 2>(DP, l.PS, f.KBO, reinterpret_cast<const float4*>(Lp + f.o_W3T), lds + l.o_HA, l.WS,
                              mk + NTHREADS, t);
        __syncthreads();
        dense_masked<NTWM, depth_w<NTWM>()>(lds + l.o_HA, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_W2T), lds + l.o_HB,
                              l.WS, mk, t);
        __syncthreads();
        gemm_ksplit(lds + l.o_HB, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_W1T), f.NTd, PART, l.PN, t);
        __syncthreads();
        for (int j = t.c; j < f.d; j += 16) G[t.row * l.DS + j] += part_sum(PART, l.PN, t.row, j);
        __syncthreads();
        // through InvertibleAffine.inverse: g <- g @ W^T
        dense_small(G, l.DS, f.D, f.KBD, reinterpret_cast<const float4*>(Lp + f.o_AWT), f.NTD, lds + nxt, l.DS, t);
        __syncthreads();
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    *grad_off = cur;
    return logq;
}

// ------------------------------------------------------------------------------------------------
// x, log q = flow.sample given base noise in U0 (NormalizingFlow.sample): forward maps in layer order.
// Leaves x in the buffer at *x_off and returns log q of this thread's row.
// ------------------------------------------------------------------------------------------------
template <int NTWM>
__device__ float flow_sample_tile(const FlowDims& f, const FlowLds& l, const float* __restrict__ packed,
                                  float* lds, const Tid& t, int* x_off) {
    int cur = l.o_U0, nxt = l.o_U1;
    const float* base = packed + f.o_base;
    float* PART = lds + l.o_PART;
    float bsum = 0.f;
    {
        float* Z = lds + cur;
        for (int j = t.c; j < f.D; j += 16) {
            const float e = Z[t.row * l.DS + j];
            const float ls = base[f.Dp + j];
            Z[t.row * l.DS + j] = base[j] + expf(ls) * e;
            bsum += ls + 0.5f * (e * e);
        }
    }
    float logq = -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
    __syncthreads();
    for (int layer = 0; layer < f.K; ++layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        float* Z = lds + cur;
        coupling_mlp<NTWM, false>(f, l, Lp, lds, Z, layer, t);
        float ssum = 0.f;
        for (int j = t.c; j < f.DO; j += 16) {
            const float shift = part_sum(PART, l.PN, t.row, j) + Lp[f.o_b3 + j];
            const float s = part_sum(PART, l.PN, t.row, f.DOp + j) + Lp[f.o_b3 + f.DOp + j];
            Z[t.row * l.DS + f.d + j] = Z[t.row * l.DS + f.d + j] * expf(s) + shift;
            ssum += s;
        }
        logq -= row16_sum(ssum);
        __syncthreads();
        // InvertibleAffine.forward: z <- z @ W^-1, log_det = -sum(log_S)
        dense_small(Z, l.DS, f.D, f.KBD, reinterpret_cast<const float4*>(Lp + f.o_AWI), f.NTD, lds + nxt, l.DS, t);
        logq -= -Lp[f.o_logS];
        __syncthreads();
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    *x_off = cur;
    return logq;
}

}  // namespace fab
