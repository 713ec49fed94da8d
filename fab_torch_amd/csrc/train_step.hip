// Training half of a FAB iteration (fab/train_with_prioritised_buffer.py:158-185), round 6:
//
//   k_pgrad_tiles     every weight-gradient product of every layer, C[p][q] = sum_b c_b Y[b][p] X[b][q] over a RealNVP tape, on
//                     v_mfma_f32_16x16x4_f32 with the operands straight from HBM / L2 in registers: a lane loads 4 consecutive
//                     columns of a tape row (Y and X) - float4 component t of the A side x component u of the B side is one MFMA
//                     whose 16 x 16 output block is rows p0 + 4 m + t, columns q0 + 4 n + u - so a 64 x 64 tile advances 4 batch
//                     rows with two 16-byte loads per lane and 16 MFMAs, no LDS and no barrier in the main loop.  One workgroup of
//                     16 waves per output tile: the waves split the batch rows and are added as a fixed tree (deterministic); the
//                     bias gradients are the column sums of the same A operands (no ones column is read).
//   k_flow_log_prob_tape_r8   the tape forward on the 8-chain stream tiles (flow_r8.h, one stage per matrix): a 2048-row minibatch is
//                     256 workgroups.
#include "flow_r8.h"
#include "train_common.h"

namespace fab {

// ---------------------------------------------------------------------------------------------------------------------------------
// tile plan: one workgroup per output tile, its waves split the batch rows
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int PG_WAVES = 16;         // waves per workgroup (1024 threads = one CU at 4 waves per SIMD); k range split 16 ways
constexpr int PG_SLOT = 68 * 64;     // floats of one wave's register image: 16 accumulator quads + 4 column-sum registers x 64 lanes

struct PgKind {
    long yo, xo;               // Y / X: offset inside a layer block (the base block: inside TB)
    int ldy, P, ldx, Q;        // row lengths, valid columns
    int tp, tq;                // floats per lane along P / Q (tile extent / 16); tq = 0: column sums only
    int np, nq;                // tiles along P, Q
    int id;                    // destination rule: 1 dW1|db1, 2 dW2|db2, 3 dW3|db3, 4 affine dW scratch, 5 base distribution
};
struct PgPlan {
    PgKind k[4];               // per layer: k[0 .. nk-1]; k[3]: the base block (once; np = 0: absent)
    int nk, K, S;              // kinds per layer, layers, k-steps of 4 batch rows
    long layer_stride;         // tape floats per layer
    int tiles_kind[4], tiles_layer, tiles_total;
};

// `narrow` = false: the W x W products (dW2: 64 x 64 tiles - at the reference architecture 250 tiles for 256 CUs and 86 % of the
// flops); true: every other product - dW3 as 32 x 32 tiles (D > 32: 64 x 32), dW1 as 32 x 16 (64 x 32), the affine maps' dW scratch -
// and the base block's column sums: 212 tiles at the reference architecture.  Two launches: one kernel with all tile shapes needs
// more registers than four waves per SIMD leave.
static PgPlan make_pg_plan(const FlowDims& f, const TapeDims& td, long B, bool narrow) {
    PgPlan p;
    const int nw = f.Wp / 64;
    const bool big = f.D > 32;
    auto kind = [](long yo, int ldy, int P, long xo, int ldx, int Q, int tp, int tq, int id) {
        PgKind k;
        k.yo = yo; k.ldy = ldy; k.P = P; k.xo = xo; k.ldx = ldx; k.Q = Q; k.tp = tp; k.tq = tq;
        k.np = (P + 16 * tp - 1) / (16 * tp); k.nq = tq ? (Q + 16 * tq - 1) / (16 * tq) : 1; k.id = id;
        return k;
    };
    for (int i = 0; i < 4; ++i) { p.k[i] = kind(0, 4, 0, 0, 4, 0, 2, 2, 0); p.k[i].np = 0; }
    if (!narrow) {
        p.nk = 1;
        p.k[0] = kind(td.o_E2, td.we, td.we, td.o_H1, td.wh, f.Wp, 4, 4, 2);
    } else {
        p.nk = 3;
        p.k[0] = kind(td.o_DP, td.wp, td.wp, td.o_H2, td.wh, f.Wp, big ? 4 : 2, 2, 3);
        p.k[1] = kind(td.o_E1, td.we, td.we, td.o_Z1, td.w1, 16 * f.NTd, big ? 4 : 2, big ? 2 : 1, 1);
        p.k[2] = kind(td.o_ZA, td.wz, td.wz, td.o_GZ, td.wz, td.wz, big ? 4 : 2, big ? 4 : 2, 4);
        p.k[3] = kind(0, td.wb, td.wb, 0, td.wb, 0, 4, 0, 5);
    }
    p.K = f.K; p.S = (int)((B + 3) / 4);
    p.layer_stride = td.layer_stride;
    p.tiles_layer = 0;
    for (int i = 0; i < 4; ++i) {
        p.tiles_kind[i] = p.k[i].np * p.k[i].nq;
        if (i < 3) p.tiles_layer += p.tiles_kind[i];
    }
    p.tiles_total = p.K * p.tiles_layer + p.tiles_kind[3];
    return p;
}

struct PgTile {
    int layer, ki, ti;         // layer (K: the base block), kind index, tile inside the kind
};
// tile `index` (0 .. tiles_total - 1): kind-major inside a layer, so that the tiles of one product (which re-read each other's
// panels) run at the same time
__device__ __forceinline__ PgTile pg_tile_index(const PgPlan& p, int index) {
    PgTile t;
    int layer = index / p.tiles_layer;
    if (layer > p.K) layer = p.K;
    int rem = index - layer * p.tiles_layer;
    int ki = 3;
    if (layer < p.K) {
        ki = 0;
        while (ki < p.nk - 1 && rem >= p.tiles_kind[ki]) { rem -= p.tiles_kind[ki]; ++ki; }
    }
    t.layer = layer; t.ki = ki; t.ti = rem;
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// per-wave accumulation of one fragment
// ---------------------------------------------------------------------------------------------------------------------------------
// Latency is hidden by occupancy, not by a deep register ring: a wave keeps TWO chunks of U k-steps (the one it multiplies and the
// one in flight) and four workgroups share a CU, so a chunk has three other waves' MFMA blocks of cover on its SIMD.  (An 8-deep
// ring in one wave was tried first: hipcc's scheduler gathers the re-loads of all slots at one end of the unrolled body and its
// wait-count pass then drains them there; with inline-asm loads the loop-carried slots are copied at the back edge while in flight.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N> struct PgVec;
template <> struct PgVec<4> { f32x4 v; __device__ __forceinline__ float get(int i) const { return v[i]; } };
template <> struct PgVec<2> { f32x2 v; __device__ __forceinline__ float get(int i) const { return v[i]; } };
template <> struct PgVec<1> { float v; __device__ __forceinline__ float get(int) const { return v; } };
template <> struct PgVec<0> { __device__ __forceinline__ float get(int) const { return 0.f; } };
template <int N> __device__ __forceinline__ void pg_load(PgVec<N>& r, const float* p) {
    if constexpr (N == 4) r.v = *reinterpret_cast<const f32x4*>(p);
    else if constexpr (N == 2) r.v = *reinterpret_cast<const f32x2*>(p);
    else if constexpr (N == 1) r.v = *p;
}

template <int TP, int TQ>
struct PgAcc {
    f32x4 c[TP][TQ ? TQ : 1];
    float s[TP];               // this lane's column sums of the A operands (rows k = lane >> 4 mod 4)
};

template <int TP, int TQ, int U>
struct PgChunk {
    PgVec<TP> y[U];
    PgVec<TQ> x[U];
    float c[U];
};

// Y / X: this lane's column of row 0 (already offset by tile and lane); rows 4 s + kg for s in [s0, s1)
template <int TP, int TQ, int U>
__device__ __forceinline__ void pg_run(const float* __restrict__ Y, int ldy, const float* __restrict__ X, int ldx,
                                       const float* __restrict__ coef, int B, bool ymask, int s0, int s1, int kg,
                                       PgAcc<TP, TQ>& acc) {
    const int last = B - 1;
    auto load = [&](PgChunk<TP, TQ, U>& ch, int s) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            int sj = s + j;
            sj = sj < s1 ? sj : s1 - 1;                                 // (behind the range: a valid row, multiplied by zero)
            int row = 4 * sj + kg;
            row = row < B ? row : last;
            pg_load<TP>(ch.y[j], Y + (long)row * ldy);
            if constexpr (TQ > 0) pg_load<TQ>(ch.x[j], X + (long)row * ldx);
            ch.c[j] = coef[row];
        }
    };
    auto mult = [&](const PgChunk<TP, TQ, U>& ch, int s) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            // (no branch around the products of the steps behind s1: a ragged last chunk multiplies zeros)
            const float cm = (ymask && s + j < s1 && 4 * (s + j) + kg < B) ? ch.c[j] : 0.f;
            float a[TP];
#pragma unroll
            for (int t = 0; t < TP; ++t) { a[t] = ch.y[j].get(t) * cm; acc.s[t] += a[t]; }
            if constexpr (TQ > 0) {
#pragma unroll
                for (int t = 0; t < TP; ++t)
#pragma unroll
                    for (int u = 0; u < TQ; ++u) acc.c[t][u] = mfma4(a[t], ch.x[j].get(u), acc.c[t][u]);
            }
        }
    };
    PgChunk<TP, TQ, U> ca, cb;
    load(ca, s0);
    for (int s = s0; s < s1; s += 2 * U) {
        load(cb, s + U);
        mult(ca, s);
        load(ca, s + 2 * U);
        mult(cb, s + U);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// destinations
// ---------------------------------------------------------------------------------------------------------------------------------
struct PgOut {
    int W, d, D, DO, DOp, K, wz;
    long gl_stride, gl_w1, gl_b1, gl_w2, gl_b2, gl_w3, gl_b3, gl_loc, gl_log_scale;
    float* grads;
    float* ga_ws;
};
__device__ __forceinline__ int pg_prm_row(int p, int DO, int DOp) {      // packed [shift | scale] row -> interleaved row
    if (p < DOp) return p < DO ? 2 * p : -1;
    const int j = p - DOp;
    return j < DO ? 2 * j + 1 : -1;
}
__device__ __forceinline__ void pg_store(const PgOut& o, int id, int layer, int p, int q, float v) {
    float* G = o.grads + (size_t)layer * o.gl_stride;
    if (id == 1) { if (p < o.W && q < o.d) G[o.gl_w1 + (long)p * o.d + q] = v; }
    else if (id == 2) { if (p < o.W && q < o.W) G[o.gl_w2 + (long)p * o.W + q] = v; }
    else if (id == 3) { const int row = pg_prm_row(p, o.DO, o.DOp); if (row >= 0 && q < o.W) G[o.gl_w3 + (long)row * o.W + q] = v; }
    else if (id == 4) { if (p < o.wz && q < o.wz) o.ga_ws[((size_t)layer * o.wz + p) * o.wz + q] = v; }
}
__device__ __forceinline__ void pg_store_sum(const PgOut& o, int id, int layer, int p, float v) {
    float* G = o.grads + (size_t)layer * o.gl_stride;
    if (id == 1) { if (p < o.W) G[o.gl_b1 + p] = v; }
    else if (id == 2) { if (p < o.W) G[o.gl_b2 + p] = v; }
    else if (id == 3) { const int row = pg_prm_row(p, o.DO, o.DOp); if (row >= 0) G[o.gl_b3 + row] = v; }
    else if (id == 5) {                                               // base: dloc | dlog_scale | sum(coef)
        if (p < o.D) o.grads[o.gl_loc + p] = v;
        else if (p >= o.wz && p < o.wz + o.D) o.grads[o.gl_log_scale + p - o.wz] = v;
        else if (p == 2 * o.wz) o.ga_ws[(size_t)o.K * o.wz * o.wz] = v;
    }
}

// a finished tile (registers of one wave) to its destination
template <int TP, int TQ>
__device__ __forceinline__ void pg_emit(const PgOut& o, const PgKind& k, int layer, int ti, int lane, PgAcc<TP, TQ>& acc) {
    const int pi = ti / k.nq, qi = ti - pi * k.nq;
    const int p0 = pi * 16 * TP, q0 = qi * 16 * TQ;
    const int n = lane & 15, kg = lane >> 4;
    if constexpr (TQ > 0) {
#pragma unroll
        for (int t = 0; t < TP; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + TP * (4 * kg + r) + t;
#pragma unroll
                for (int u = 0; u < TQ; ++u) pg_store(o, k.id, layer, p, q0 + TQ * n + u, acc.c[t][u][r]);
            }
    }
    if (qi == 0) {                                                    // column sums: the four row groups (kg) in a fixed order
#pragma unroll
        for (int t = 0; t < TP; ++t) {
            const float v = acc.s[t];
            const float v1 = __shfl(v, n + 16), v2 = __shfl(v, n + 32), v3 = __shfl(v, n + 48), v0 = __shfl(v, n);
            const float s = (v0 + v1) + (v2 + v3);
            if (kg == 0) pg_store_sum(o, k.id, layer, p0 + TP * n + t, s);
        }
    }
}

// register image <-> memory: 16 accumulator quads [quad][lane] (16 bytes per lane: coalesced, conflict-free), then the column sums
template <int TP, int TQ>
__device__ __forceinline__ void pg_put(float* img, int lane, const PgAcc<TP, TQ>& acc) {
    if constexpr (TQ > 0) {
        f32x4* q = reinterpret_cast<f32x4*>(img);
#pragma unroll
        for (int t = 0; t < TP; ++t)
#pragma unroll
            for (int u = 0; u < TQ; ++u) q[(t * TQ + u) * 64 + lane] = acc.c[t][u];
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) img[(64 + t) * 64 + lane] = acc.s[t];
}
template <int TP, int TQ>
__device__ __forceinline__ void pg_add(const float* img, int lane, PgAcc<TP, TQ>& acc) {
    if constexpr (TQ > 0) {
        const f32x4* q = reinterpret_cast<const f32x4*>(img);
#pragma unroll
        for (int t = 0; t < TP; ++t)
#pragma unroll
            for (int u = 0; u < TQ; ++u) acc.c[t][u] += q[(t * TQ + u) * 64 + lane];
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) acc.s[t] += img[(64 + t) * 64 + lane];
}
template <int TP, int TQ>
__device__ __forceinline__ void pg_zero(PgAcc<TP, TQ>& acc) {
#pragma unroll
    for (int t = 0; t < TP; ++t) {
        acc.s[t] = 0.f;
#pragma unroll
        for (int u = 0; u < (TQ ? TQ : 1); ++u) acc.c[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// one tile: the workgroup's waves split the k-steps, wave 0 ends with the sum and writes it out
template <int TP, int TQ, int U>
__device__ __forceinline__ void pg_tile(const PgPlan& p, const PgOut& o, const PgTile& t, const float* __restrict__ tape, long o_TB,
                                        const float* __restrict__ coef, int B, float* red) {
    const PgKind& k = p.k[t.ki];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kg = lane >> 4;
    const int pi = t.ti / k.nq, qi = t.ti - pi * k.nq;
    const float* base = t.layer < p.K ? tape + (size_t)t.layer * p.layer_stride : tape + o_TB;
    int pc = pi * 16 * TP + TP * n, qc = qi * 16 * TQ + TQ * n;
    const bool ymask = pc < k.P;
    if (!ymask) pc = 0;                                               // (a valid column of the same row; its products are discarded)
    if (qc >= k.Q) qc = 0;
    const float* Y = base + k.yo + pc;
    const float* X = base + k.xo + qc;
    PgAcc<TP, TQ> acc;
    pg_zero(acc);
    const int w0 = __builtin_amdgcn_readfirstlane((int)((long)p.S * wave / PG_WAVES));
    const int w1 = __builtin_amdgcn_readfirstlane((int)((long)p.S * (wave + 1) / PG_WAVES));
    if (w0 < w1) pg_run<TP, TQ, U>(Y, k.ldy, X, k.ldx, coef, B, ymask, w0, w1, kg, acc);
    // the waves' sums as a fixed tree (wave w + wave w + half, half = 8, 4, 2, 1)
#pragma unroll
    for (int half = PG_WAVES / 2; half >= 1; half /= 2) {
        if (wave >= half && wave < 2 * half) pg_put(red + (size_t)(wave - half) * PG_SLOT, lane, acc);
        __syncthreads();
        if (wave < half) pg_add(red + (size_t)wave * PG_SLOT, lane, acc);
        __syncthreads();
    }
    if (wave == 0) pg_emit(o, k, t.layer, t.ti, lane, acc);
}

// tile shapes of the narrow launch by flow class: D <= 32: dW3 (2,2), dW1 (2,1), affine (2,2); D > 32: dW3 (4,2), dW1 (4,2),
// affine (4,4); the base block's sums (4,0) in both
template <int CLS>      // 0: the W x W products; 1 / 2: the narrow products of a flow with D <= 32 / D > 32
__global__ __launch_bounds__(64 * PG_WAVES, 4) void k_pgrad_tiles(PgPlan p, PgOut o, const float* __restrict__ tape, long o_TB,
                                                                  const float* __restrict__ coef, int B) {
    extern __shared__ __attribute__((aligned(16))) float red[];        // PG_WAVES / 2 register images
    // consecutive workgroups go to different XCDs (blockIdx % 8), each with its own L2: give XCD x the x-th eighth of the tile list,
    // so that the tiles of one product - which re-read each other's tape panels - share an L2
    const int per = (p.tiles_total + 7) >> 3;
    const int index = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (index >= p.tiles_total) return;
    const PgTile t = pg_tile_index(p, index);
    const int c = p.k[t.ki].tp * 8 + p.k[t.ki].tq;
    if constexpr (CLS == 0) {
        pg_tile<4, 4, 2>(p, o, t, tape, o_TB, coef, B, red);
    } else if constexpr (CLS == 1) {
        if (c == 2 * 8 + 2) pg_tile<2, 2, 4>(p, o, t, tape, o_TB, coef, B, red);
        else if (c == 2 * 8 + 1) pg_tile<2, 1, 4>(p, o, t, tape, o_TB, coef, B, red);
        else pg_tile<4, 0, 4>(p, o, t, tape, o_TB, coef, B, red);
    } else {
        if (c == 4 * 8 + 2) pg_tile<4, 2, 2>(p, o, t, tape, o_TB, coef, B, red);
        else if (c == 4 * 8 + 4) pg_tile<4, 4, 1>(p, o, t, tape, o_TB, coef, B, red);
        else pg_tile<4, 0, 4>(p, o, t, tape, o_TB, coef, B, red);
    }
}

int launch_param_grad_tiles(const FlowDims& f, const TapeDims& td, const GradLayout& gl, const float* tape, const float* coef,
                            long B, float* grads, float* ga_ws, hipStream_t st) {
    if (B < 1 || B > (1L << 28)) return FABHIP_EINVAL;
    PgOut o;
    o.W = f.W; o.d = f.d; o.D = f.D; o.DO = f.DO; o.DOp = f.DOp; o.K = f.K; o.wz = td.wz;
    o.gl_stride = gl.layer_stride; o.gl_w1 = gl.w1; o.gl_b1 = gl.b1; o.gl_w2 = gl.w2; o.gl_b2 = gl.b2; o.gl_w3 = gl.w3;
    o.gl_b3 = gl.b3; o.gl_loc = gl.loc; o.gl_log_scale = gl.log_scale;
    o.grads = grads; o.ga_ws = ga_ws;
    const size_t lds = (size_t)(PG_WAVES / 2) * PG_SLOT * sizeof(float);
    const bool big = f.D > 32;
    const PgPlan pw = make_pg_plan(f, td, B, false), pn = make_pg_plan(f, td, B, true);
    const dim3 block(64 * PG_WAVES);
    FAB_TRY(set_max_lds((const void*)k_pgrad_tiles<0>, lds));
    hipLaunchKernelGGL(k_pgrad_tiles<0>, dim3(8 * ((pw.tiles_total + 7) / 8)), block, lds, st, pw, o, tape, td.o_TB, coef, (int)B);
    if (big) {
        FAB_TRY(set_max_lds((const void*)k_pgrad_tiles<2>, lds));
        hipLaunchKernelGGL(k_pgrad_tiles<2>, dim3(8 * ((pn.tiles_total + 7) / 8)), block, lds, st, pn, o, tape, td.o_TB, coef, (int)B);
    } else {
        FAB_TRY(set_max_lds((const void*)k_pgrad_tiles<1>, lds));
        hipLaunchKernelGGL(k_pgrad_tiles<1>, dim3(8 * ((pn.tiles_total + 7) / 8)), block, lds, st, pn, o, tape, td.o_TB, coef, (int)B);
    }
    return check_launch();
}

}  // namespace fab
