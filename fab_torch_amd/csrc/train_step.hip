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
    // the sixteen waves' sums in a fixed order, three barriers: waves 8 .. 15 leave their register images in the eight LDS slots and
    // waves 0 .. 7 add them (w + (w + 8)); those eight sums go back to the slots, and wave w adds accumulator quad w (one quad =
    // 4 rows x 64 lanes of one (t, u) pair) over the slots as ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)) and writes it out - the
    // finished tile leaves from ALL waves (written by wave 0 alone, the 64 scattered stores per lane cost a fifth of the main loop)
    static_assert(PG_WAVES == 16, "the reduction below is written for 16 waves");
    if (wave >= 8) pg_put(red + (size_t)(wave - 8) * PG_SLOT, lane, acc);
    __syncthreads();
    if (wave < 8) pg_add(red + (size_t)wave * PG_SLOT, lane, acc);
    __syncthreads();
    if (wave < 8) pg_put(red + (size_t)wave * PG_SLOT, lane, acc);
    __syncthreads();
    auto slot_sum = [&](auto get) {
        return ((get(0) + get(1)) + (get(2) + get(3))) + ((get(4) + get(5)) + (get(6) + get(7)));
    };
    if constexpr (TQ > 0) {
        if (wave < TP * TQ) {
            const int tt = wave / TQ, u = wave - tt * TQ;
            const f32x4 q = slot_sum([&](int sl) { return reinterpret_cast<const f32x4*>(red + (size_t)sl * PG_SLOT)[wave * 64 + lane]; });
            const int p0 = pi * 16 * TP, q0 = qi * 16 * TQ;
#pragma unroll
            for (int r = 0; r < 4; ++r) pg_store(o, k.id, t.layer, p0 + TP * (4 * kg + r) + tt, q0 + TQ * n + u, q[r]);
        }
    }
    if (wave == PG_WAVES - 1 && qi == 0) {                            // column sums: the eight slots, then the four row groups (kg)
#pragma unroll
        for (int tt = 0; tt < TP; ++tt) {
            const float v = slot_sum([&](int sl) { return red[(size_t)sl * PG_SLOT + (64 + tt) * 64 + lane]; });
            const float v1 = __shfl(v, n + 16), v2 = __shfl(v, n + 32), v3 = __shfl(v, n + 48), v0 = __shfl(v, n);
            const float sum = (v0 + v1) + (v2 + v3);
            if (kg == 0) pg_store_sum(o, k.id, t.layer, pi * 16 * TP + TP * n + tt, sum);
        }
    }
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

// ---------------------------------------------------------------------------------------------------------------------------------
// tape forward on the 8-chain stream tiles (flow_r8.h, one stage per matrix: the tape needs z and the full cotangent of z, which the
// fused stages never form).  `rows` != nullptr: row g of the batch is row rows[g] of `x` (a minibatch of the replay buffer read in
// place, fab/utils/prioritised_replay_buffer.py:88-99 `self.buffer.x[indices]`).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(NTHREADS) void k_flow_log_prob_tape_r8(FlowDims f, R8Lds l, TapeDims td, const float* __restrict__ packed,
                                                                    const float* __restrict__ x, const int64_t* __restrict__ rows,
                                                                    float* __restrict__ log_q, float* __restrict__ grad,
                                                                    float* __restrict__ tape, long B, MbTail mb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = NTHREADS;
    Tid8f t8;
    const int D = f.D;
    const long row0 = (long)blockIdx.x * R8;
    r8_load_heads(f, l, packed, lds, t8.tid, NT);
    for (int e = t8.tid; e < R8 * R4_DS; e += NT) { lds[l.o_DP + e] = 0.f; lds[l.o_PRM + e] = 0.f; }
    for (int e = t8.tid; e < R8 * R4_DS; e += NT) {
        const int r = e / R4_DS, j = e % R4_DS;
        const long g = row0 + r;
        float v = 0.f;
        if (j < D && g < B) v = x[(rows ? (long)rows[g] : g) * D + j];
        lds[l.o_X0 + e] = v;
    }
    R8Stream s;
    s8_stream_init(s, t8.lane);
    __syncthreads();
    const R8Tape tp{&td, tape, row0};
    int goff = 0;
    const float lq = flow_log_prob_r8<G, false, true>(f, l, packed, lds, t8, s, &goff, &tp);
    if (t8.tid >= 16 * R8) return;
    const long g = row0 + t8.row;
    if (g < B) {
        if (t8.c == 0) log_q[g] = lq;
        if (grad)
            for (int j = t8.c; j < D; j += 16) grad[g * D + j] = lds[goff + t8.row * R4_DS + j];
    }
    if (!mb.coef) return;
    // the minibatch arithmetic (train_with_prioritised_buffer.py:162-170, prioritised_replay_buffer.py:117-131) on this row
    float s_wl = 0.f, s_w = 0.f, s_l = 0.f, mn = INFINITY, mx = -INFINITY, nanf_ = 0.f;
    if (g < B && t8.c == 0) {
        const long row = rows ? (long)rows[g] : g;
        const float lqo = mb.log_q_old[mb.rows_old ? row : g];
        const float adj = mb.one_minus_alpha * (lq - lqo);
        const float wp = expf(adj);
        const float w = (mb.w_clip > 0.f && wp > mb.w_clip) ? mb.w_clip : wp;     // torch.clip(max=): NaN stays NaN
        mb.log_w_adjust[g] = adj;
        mb.coef[g] = w * mb.neg_inv_B;
        s_wl = w * lq; s_w = wp; s_l = lq; mn = wp; mx = wp; nanf_ = wp != wp ? 1.f : 0.f;
        if (mb.buf_log_w) {
            const bool valid = isfinite(adj) && isfinite(lq);
            mb.buf_log_w[row] = valid ? mb.buf_log_w[row] + adj : -INFINITY;
            if (valid) mb.buf_log_q_old[row] = lq;
        }
    }
    // the wave's four rows (lanes 0, 16, 32, 48) in row order
    const int lane = t8.lane;
    auto rowsum = [&](float v) { return (__shfl(v, 0) + __shfl(v, 16)) + (__shfl(v, 32) + __shfl(v, 48)); };
    const float a0 = rowsum(s_wl), a1 = rowsum(s_w), a2 = rowsum(s_l), a5 = rowsum(nanf_);
    const float a3 = fminf(fminf(__shfl(mn, 0), __shfl(mn, 16)), fminf(__shfl(mn, 32), __shfl(mn, 48)));
    const float a4 = fmaxf(fmaxf(__shfl(mx, 0), __shfl(mx, 16)), fmaxf(__shfl(mx, 32), __shfl(mx, 48)));
    if (lane == 0) {
        float* p = mb.partials + ((size_t)2 * blockIdx.x + t8.wave) * MB_PART;
        p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3; p[4] = a4; p[5] = a5;
    }
}

int launch_log_prob_tape_r8(const FlowDims& f, const TapeDims& td, const float* packed, const float* x, const int64_t* rows,
                            float* log_q, float* grad, float* tape, long B, hipStream_t st, const MbTail* mb) {
    if (!r8_shape_ok(f)) return FABHIP_ENOTSUP;
    const R8Lds l = make_r8_lds(f);
    const size_t bytes = (size_t)l.total * 4;
    const dim3 grid((unsigned)(td.Bp / R8));                  // every row of the tape (Bp: a multiple of 16) is written
    MbTail m;
    if (mb) m = *mb; else { m = MbTail(); m.coef = nullptr; }
#define FAB_R8_TAPE(G)                                                                                              \
    do {                                                                                                            \
        FAB_TRY(set_max_lds((const void*)k_flow_log_prob_tape_r8<G>, bytes));                                       \
        hipLaunchKernelGGL((k_flow_log_prob_tape_r8<G>), grid, dim3(NTHREADS), bytes, st, f, l, td, packed, x, rows, log_q, grad, \
                           tape, B, m);                                                                                \
    } while (0)
    if (f.Wp == 320) FAB_R8_TAPE(5);
    else if (f.Wp == 256) FAB_R8_TAPE(4);
    else return FABHIP_ENOTSUP;
#undef FAB_R8_TAPE
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The arithmetic between `log_q_x = flow.log_prob(x)` and `loss.backward()` of one replay-buffer minibatch
// (fab/train_with_prioritised_buffer.py:162-172) and the buffer's `adjust` (fab/utils/prioritised_replay_buffer.py:117-131), one
// workgroup:   log_w_adjust = (1 - alpha) (log_q - log_q_old);  w = clip(exp(log_w_adjust), max = w_clip);  loss = -mean(w log_q);
//   coef_b = d loss / d log_q_b = w_b * (-1 / B)   (w is detached in the reference's loss), times NaN when the loss is not finite
//   (the update is then skipped by the optimiser's finite-norm test: the reference's two host-side checks, :172-181);
//   buffer: log_w[idx] += log_w_adjust, log_q_old[idx] = log_q where both are finite, log_w[idx] = -inf elsewhere.
// stats: [0] loss, [1] mean(w before the clip), [2] min, [3] max, [4] mean(log_q)   (the reference's logging keys, :188-196)
// ---------------------------------------------------------------------------------------------------------------------------------
struct MinibatchK {
    const float* log_q;
    const float* log_q_old;            // [B], or the buffer's log_q_old when `rows` is set
    const int64_t* rows;               // buffer rows of the minibatch (nullptr: no gather, no adjust)
    float one_minus_alpha, w_clip, neg_inv_B;
    float* coef;                       // [B] out
    float* log_w_adjust;               // [B] out
    float* buf_log_w;                  // adjusted in place at `rows` (nullptr: no adjust)
    float* buf_log_q_old;
    float* stats;                      // [8] out
    long B;
};

__global__ __launch_bounds__(1024) void k_buffer_minibatch(MinibatchK a) {
    __shared__ float red[5][16];
    __shared__ float poison_s;
    constexpr int KEEP = 4;                                 // weights kept in registers between the two passes (B <= 4096)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s_wl = 0.f, s_w = 0.f, s_l = 0.f, mn = INFINITY, mx = -INFINITY;
    bool nan_w = false;
    float wk[KEEP];
    // the first KEEP trips as three rounds of independent loads (row -> its stored values -> arithmetic): one memory latency per
    // round instead of one per dependent access
    long rowk[KEEP];
    float lqk[KEEP], lqok[KEEP], blwk[KEEP];
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
        const long b = tid + 1024l * it;
        rowk[it] = b < a.B ? (a.rows ? (long)a.rows[b] : b) : 0;
        lqk[it] = b < a.B ? a.log_q[b] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
        const long b = tid + 1024l * it;
        lqok[it] = b < a.B ? a.log_q_old[rowk[it]] : 0.f;
        blwk[it] = (b < a.B && a.buf_log_w) ? a.buf_log_w[rowk[it]] : 0.f;
    }
    auto one = [&](long b, long row, float lq, float lqo, float blw, float& w_out) {
        const float adj = a.one_minus_alpha * (lq - lqo);
        const float wp = expf(adj);
        const float w = (a.w_clip > 0.f && wp > a.w_clip) ? a.w_clip : wp;       // torch.clip(max=): NaN stays NaN
        a.log_w_adjust[b] = adj;
        w_out = w;
        s_wl += w * lq; s_w += wp; s_l += lq;
        nan_w = nan_w || (wp != wp);
        mn = fminf(mn, wp); mx = fmaxf(mx, wp);
        if (a.buf_log_w) {
            const bool valid = isfinite(adj) && isfinite(lq);
            a.buf_log_w[row] = valid ? blw + adj : -INFINITY;
            if (valid) a.buf_log_q_old[row] = lq;
        }
    };
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
        const long b = tid + 1024l * it;
        wk[it] = 0.f;
        if (b < a.B) one(b, rowk[it], lqk[it], lqok[it], blwk[it], wk[it]);
    }
    for (long b = tid + 1024l * KEEP; b < a.B; b += 1024) {
        const long row = a.rows ? (long)a.rows[b] : b;
        float w;
        one(b, row, a.log_q[b], a.log_q_old[row], a.buf_log_w ? a.buf_log_w[row] : 0.f, w);
        a.coef[b] = w;
    }
    if (nan_w) { mn = NAN; mx = NAN; }                                           // torch.min / max propagate NaN
    // fixed-order sums: the lanes of a wave by shuffles, then the 16 waves in order
    for (int o = 32; o >= 1; o >>= 1) {
        s_wl += __shfl_down(s_wl, o); s_w += __shfl_down(s_w, o); s_l += __shfl_down(s_l, o);
        const float m1 = __shfl_down(mn, o), m2 = __shfl_down(mx, o);
        mn = (mn != mn || m1 != m1) ? NAN : fminf(mn, m1);
        mx = (mx != mx || m2 != m2) ? NAN : fmaxf(mx, m2);
    }
    if (lane == 0) { red[0][wave] = s_wl; red[1][wave] = s_w; red[2][wave] = s_l; red[3][wave] = mn; red[4][wave] = mx; }
    __syncthreads();
    if (tid == 0) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = INFINITY, t4 = -INFINITY;
        for (int w = 0; w < 16; ++w) {
            t0 += red[0][w]; t1 += red[1][w]; t2 += red[2][w];
            t3 = (t3 != t3 || red[3][w] != red[3][w]) ? NAN : fminf(t3, red[3][w]);
            t4 = (t4 != t4 || red[4][w] != red[4][w]) ? NAN : fmaxf(t4, red[4][w]);
        }
        const float Bf = (float)a.B;
        const float loss = -(t0 / Bf);
        a.stats[0] = loss; a.stats[1] = t1 / Bf; a.stats[2] = t3; a.stats[3] = t4; a.stats[4] = t2 / Bf;
        a.stats[6] = 0.f; a.stats[7] = 0.f;                 // ([5]: the gradient norm, written by the optimiser step)
        poison_s = isfinite(loss) ? 1.f : NAN;
    }
    __syncthreads();
    const float poison = poison_s;
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
        const long b = tid + 1024l * it;
        if (b < a.B) a.coef[b] = wk[it] * a.neg_inv_B * poison;
    }
    for (long b = tid + 1024l * KEEP; b < a.B; b += 1024) a.coef[b] = a.coef[b] * a.neg_inv_B * poison;
}

}  // namespace fab

using namespace fab;

extern "C" {

size_t fabhip_train_step_workspace_bytes(int32_t dim, int32_t n_layers, int32_t width, int64_t B, int64_t n_params) {
    const size_t tape = fabhip_flow_tape_bytes(dim, n_layers, width, B);
    if (tape == 0 || B < 1 || n_params < 1) return 0;
    // tape | optimiser scratch | loss partials of the tape kernel's tail (two per 8-row workgroup)
    return ((tape + 255) & ~(size_t)255) + ((fabhip_adam_workspace_bytes(n_params) + 255) & ~(size_t)255) +
           (((size_t)(B + 15) / 16 * 4 * MB_PART * sizeof(float) + 255) & ~(size_t)255);
}

int fabhip_buffer_train_step(const fabhip_train_step_args* a, fabhip_stream_t stream) {
    if (!a || a->struct_bytes != sizeof(fabhip_train_step_args)) return FABHIP_EINVAL;
    if (!a->params || !a->packed || !a->x || !a->log_q_old || !a->log_q || !a->log_w_adjust || !a->coef || !a->grads ||
        !a->stats || !a->theta || !a->m || !a->v || !a->step_count || !a->workspace || a->B < 1 || a->n_params < 1)
        return FABHIP_EINVAL;
    if ((a->buf_log_w != nullptr) != (a->buf_log_q_old != nullptr)) return FABHIP_EINVAL;
    if ((a->buf_log_w || a->log_q_old_rows) && !a->rows) return FABHIP_EINVAL;
    const fabhip_flow_params* p = a->params;
    FAB_TRY(check_flow_shape(p->dim, p->n_layers, p->width));
    const size_t tape_bytes = fabhip_flow_tape_bytes(p->dim, p->n_layers, p->width, a->B);
    const size_t tape_al = (tape_bytes + 255) & ~(size_t)255;
    if (a->workspace_bytes < fabhip_train_step_workspace_bytes(p->dim, p->n_layers, p->width, a->B, a->n_params)) return FABHIP_ENOSPC;
    if (((uintptr_t)a->workspace & 255) != 0) return FABHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // 1. the image of the current parameters (the optimiser step of the previous minibatch changed them)
    if (a->repack) FAB_TRY(fabhip_flow_pack_train(p, a->packed, stream));
    // 2. log q(x) with the tape, the minibatch read in place from the buffer; 3. weights of the loss, its value, the buffer's
    // adjustment: in the tail of the 8-chain tape kernel where the flow has those tiles, else a launch of their own
    fabhip_flow fl;
    fl.dim = p->dim; fl.n_layers = p->n_layers; fl.width = p->width; fl.precision = FABHIP_PRECISION_FP32; fl.packed = a->packed;
    if (!a->log_q_old_rows && a->buf_log_w) return FABHIP_EINVAL;       // (adjusting in place needs the buffer's own log_q_old)
    const FlowDims f = make_flow_dims(p->dim, p->n_layers, p->width);
    const TapeDims td = make_tape_dims(f, (long)a->B);
    char* adam_ws = (char*)a->workspace + tape_al;
    const size_t adam_bytes = a->workspace_bytes - tape_al;
    const bool tail = f.o_r8 >= 0 && option(FABHIP_OPT_TAPE_TILES) == 0;       // (8: the 8-chain tiles without the fused tail, A/B)
    const int n_part = 2 * (int)(td.Bp / R8);
    float* partials = (float*)(adam_ws + ((fabhip_adam_workspace_bytes(a->n_params) + 255) & ~(size_t)255));
    if (tail) {
        MbTail mb;
        mb.log_q_old = a->log_q_old; mb.rows_old = a->log_q_old_rows ? 1 : 0;
        mb.one_minus_alpha = 1.f - a->alpha; mb.w_clip = a->w_adjust_max_clip; mb.neg_inv_B = -1.f / (float)a->B;
        mb.coef = a->coef; mb.log_w_adjust = a->log_w_adjust; mb.buf_log_w = a->buf_log_w; mb.buf_log_q_old = a->buf_log_q_old;
        mb.partials = partials;
        FAB_TRY(launch_log_prob_tape_r8(f, td, a->packed, a->x, a->rows, a->log_q, nullptr, (float*)a->workspace, (long)a->B, st, &mb));
    } else {
        if (a->rows) FAB_TRY(fabhip_flow_log_prob_tape_rows(&fl, a->x, a->rows, a->log_q, nullptr, a->B, a->workspace, tape_bytes, stream));
        else FAB_TRY(fabhip_flow_log_prob_tape(&fl, a->x, a->log_q, nullptr, a->B, a->workspace, tape_bytes, stream));
        MinibatchK k;
        k.log_q = a->log_q; k.log_q_old = a->log_q_old; k.rows = (a->log_q_old_rows || a->buf_log_w) ? a->rows : nullptr;
        k.one_minus_alpha = 1.f - a->alpha; k.w_clip = a->w_adjust_max_clip; k.neg_inv_B = -1.f / (float)a->B;
        k.coef = a->coef; k.log_w_adjust = a->log_w_adjust; k.buf_log_w = a->buf_log_w; k.buf_log_q_old = a->buf_log_q_old;
        k.stats = a->stats; k.B = (long)a->B;
        hipLaunchKernelGGL(k_buffer_minibatch, dim3(1), dim3(1024), 0, st, k);
    }
    // 4. d loss / d theta, 5. clipped Adam step (skipped on the device when the norm - or, with the tail, the loss - is not finite)
    FAB_TRY(fabhip_flow_param_grad(p, &fl, a->workspace, tape_bytes, a->coef, a->B, a->grads, stream));
    return adam_clip_step_impl(a->theta, a->grads, a->m, a->v, a->n_params, a->lr, a->beta1, a->beta2, a->eps, a->step_count,
                               a->max_grad_norm, a->stats + 5, adam_ws, adam_bytes, tail ? partials : nullptr, tail ? n_part : 0,
                               a->stats, (long)a->B, st);
}

}  // extern "C"
