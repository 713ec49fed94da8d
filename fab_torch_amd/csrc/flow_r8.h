// 8-chain tiles (R8) for the RealNVP density + d/dx: EIGHT chains per workgroup (4 waves) on v_mfma_f32_4x4x1_16b_f32, hidden
// width padded to 256 or 320 (G = Wp / 64 = 4 or 5 column groups), D <= 32.
//
// Why a third tile shape: the 4-chain kernel (flow_r4.h) is bound by the L2 -> CU weight stream - every CU streams the whole
// flow for 4 chains (0.96 MB per layer pair at ~45 of the ~52 B/clk the path delivers) - and the 16-chain kernel leaves CUs
// idle below 4096 chains.  Eight chains per workgroup halve the bytes per chain; with the structure of spline_r8.h the
// extra MFMAs hide behind the stream:
//   * the products into the hidden width (W1, W2, W3T, W2T) are N-split: wave w owns columns 64 w .. 64 w + 63 for all of
//     K, k mod 4 on four accumulators per row block - no partial sums through LDS, one LDS-only barrier per stage.  The
//     fifth column group of a 320-wide layer is K-split over the four waves (a quarter of its k-quads each, partial sums
//     through LDS, wave w finishes rows 2 w, 2 w + 1): a fifth wave would share a SIMD with wave 0, and the 4x4x1 MFMA
//     pipe (8 cycles per instruction) of that SIMD then bounds every stage (measured: 156 instead of 95 cycles per k-quad);
//   * the narrow products out of the hidden width (W3 -> shift | scale, W1T -> the d input gradients) are K-split (wave w:
//     k = 64 w .. 64 w + 63) with their partials summed by the element-wise stage that consumes them;
//   * the D x D maps are evaluated by every wave, wave w stores chains 2 w, 2 w + 1;
//   * r4: the narrow products are streamed as DENSE tiles - 2 k-quads of a 32-column matrix (D x D maps, W3), 4 k-quads of a
//     16-column one (W1T) side by side in one 1-KiB tile, each lane half / quarter multiplying its own k-quad: 33 tiles (12 %)
//     fewer per wave and layer pair; the 2 / 4 partial products per wave are summed with the other waves' by the consumer;
//   * every wave reads its tiles of a layer and direction as ONE stream through a 32-tile AGPR ring (stream_r8.h): forward
//     [AW 4 | W1 4 (+1) | W2 16 G (+4 G) | W3 2 G], reverse [W3T 8 (+2) | W2T 16 G (+4 G) | W1T G | AWT 4] tiles of 1 KiB
//     ((+..): its share of the fifth group); the next layer's ring is requested before the layer's last element-wise stage;
//   * biases / log-det constants of ALL layers sit in LDS (copied once per kernel), ReLU decisions as one word per thread and layer in LDS.
// Same arithmetic as flow_log_prob_r4 / flow_log_prob_tile up to the summation order inside the GEMMs.
#pragma once
#include "flow_r4.h"
#include "stream_r8.h"

namespace fab {

constexpr int R8 = 8;                  // chains per workgroup
constexpr int R8_RD = 40;              // ring depth: 1-KiB tiles in flight per wave.  r4: ONE stream per wave for a whole evaluation (all
                                       // layers forward, then reverse), never drained - a layer's padded tile count (80 / 120) is a
                                       // multiple of the depth, so every slot index is a compile-time constant
constexpr int R8_TAIL = 64;            // tiles of slack behind a wave's stream (the last requests read past its end)
using R8Stream = S8StreamT<R8_RD>;
constexpr int R8_KD4 = 8, R8_Kd4 = 4, R8_Ko4 = 8;    // k-quads of the short K extents, padded to D = 32 / d = 16 / 2 DOp = 32
constexpr int R8_TD = R8_KD4 / 2;                    // dense tiles of a D x D map (2 k-quads x 32 columns each)

FAB_HD bool r8_shape_ok(const FlowDims& f) { return f.o_r8 >= 0; }
// tiles per wave, layer and direction (G = 4: 80 / 80, G = 5: 119 / 119); make_flow_dims sizes the image with the same sums
FAB_HD int r8_tiles_fwd(int G) { const int EX = G - 4; return R8_TD + R8_Kd4 + EX + 16 * G + 4 * G * EX + 2 * G; }
FAB_HD int r8_tiles_rev(int G) { const int EX = G - 4; return R8_Ko4 + 2 * EX + 16 * G + 4 * G * EX + G + R8_TD; }
FAB_HD int r8_tiles_pad(int G) { return G == 5 ? 1 : 0; }               // zero tiles at the end of a layer and direction: 119 -> 120
FAB_HD int r8_tiles_p(int G) { return r8_tiles_fwd(G) + r8_tiles_pad(G); }   // per layer and direction, padded (fwd == rev: 80 / 120)
FAB_HD long r8_wave_tiles(int G, int K) { return 2L * K * r8_tiles_p(G) + R8_TAIL; }   // one wave's stream
FAB_HD long r8_image_floats(int G, int K) { return (long)NWAVE * r8_wave_tiles(G, K) * 256; }

// ---- fused stages (r5; the 8-chain form of flow_r4f.h) ----------------------------------------------------------------
// The InvertibleAffine of a layer is multiplied INTO the conditioner's first Linear at pack time (k_pack_r8f: W1' = W'[:, :d] W1^T,
// b1' = b1 + ac[:d] W1^T, float64 products rounded once), so that
//   forward: ONE stage reads y and makes z = y W' + ac (every wave, 4 dense tiles; needed by the coupling three stages later)
//            AND h1 = relu(y W1' + b1') (N-split, K = 32: 8 k-quads) - no stage and no barrier for the D x D map alone;
//   reverse: ONE K-split stage makes g_y = dh1 W1'^T + g_z W'^T (2 G dense tiles of this wave's quarter of K = Wp plus ONE dense
//            tile of the D x D map: this wave's 8 rows of it) and its consumer - the sum of the 8 partial products - forms the next
//            layer's coupling cotangents: three stages instead of five, the "+=" stage and the every-wave D x D product gone.
// Stream per wave and layer: forward [AW 4 | W1' 8 (+2) | W2 16 G (+4 G) | W3 2 G], reverse [W3T 8 (+2) | W2T 16 G (+4 G) |
// W1'T 2 G | AWT 1], padded to r8f_tiles_p (84 / 126 = 2 / 3 ring depths of 42).  Biases: the density blocks of flow_r4f.h's
// image (FlowDims::o_r4fb: b1' | b2 | ac | shift | scale | logS), copied into the head blocks' layout.
constexpr int R8F_RD = 42;
using R8FStream = S8StreamT<R8F_RD>;
FAB_HD int r8f_tiles_fwd(int G) { const int EX = G - 4; return R8_TD + R8_KD4 + EX * (R8_KD4 / 4) + 16 * G + 4 * G * EX + 2 * G; }
FAB_HD int r8f_tiles_rev(int G) { const int EX = G - 4; return R8_Ko4 + 2 * EX + 16 * G + 4 * G * EX + 2 * G + 1; }
FAB_HD int r8f_tiles_p(int G) { return G == 5 ? 3 * R8F_RD : 2 * R8F_RD; }                  // >= 124 / 84 forward, 121 / 81 reverse
FAB_HD long r8f_wave_tiles(int G, int K) { return 2L * K * r8f_tiles_p(G) + R8_TAIL; }
FAB_HD long r8f_image_floats(int G, int K) { return (long)NWAVE * r8f_wave_tiles(G, K) * 256; }
FAB_HD bool r8f_shape_ok(const FlowDims& f) { return f.o_r8f >= 0; }

// LDS plan of an r8 workgroup (floats)
struct R8Lds {
    int WS, HF;                        // leading dim of the hidden tiles; floats per layer of the head block
    int o_X0, o_X1, o_HA, o_HB, o_PRM, o_DP, o_PART, o_ES, o_V2, o_MASK, o_HEAD, total;
};
// head block of a layer: ac[64] | b1[Wp] | b2[Wp] | b3[64] (shift | scale at 0 / DOp) | logS[16]
FAB_HD R8Lds make_r8_lds(const FlowDims& f) {
    R8Lds l;
    const int G = f.Wp / 64;
    l.WS = f.Wp + 4;
    l.HF = 64 + 2 * f.Wp + 64 + 16;
    int o = 0;
    l.o_X0 = o; o += R8 * R4_DS;
    l.o_X1 = o; o += R8 * R4_DS;
    l.o_HA = o; o += R8 * l.WS;
    l.o_HB = o; o += R8 * l.WS;
    l.o_PRM = o; o += R8 * R4_DS;      // (unused columns stay zero)
    l.o_DP = o; o += R8 * R4_DS;
    l.o_PART = o; o += NWAVE * R8 * R4_DS;           // K-split partials: narrow outputs / the fifth column group
    l.o_ES = o; o += f.K * R8 * f.DOp;
    l.o_V2 = o; o += f.K * R8 * f.DOp;
    l.o_MASK = o; o += f.K * NWAVE * 64;             // ReLU decisions: one word per layer and thread (see r8_dense_wide)
    l.o_HEAD = o; o += f.K * l.HF;
    l.total = (o + 3) & ~3;
    return l;
}

struct Tid8f {
    int tid, wave, lane, arow, row, c;
    __device__ __forceinline__ Tid8f() {
        tid = threadIdx.x;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        lane = tid & 63;
        arow = lane & 3;
        row = tid >> 4;                // element-wise stages: threads < 128 = 8 chains x 16 lanes (the 16-chain code's mapping)
        c = tid & 15;
    }
};

// copy the head blocks of all layers into LDS (once per kernel; plain loads, nothing else in flight)
__device__ __forceinline__ void r8_load_heads(const FlowDims& f, const R8Lds& l, const float* __restrict__ packed, float* lds,
                                              int tid, int nthreads) {
    for (int e = tid; e < f.K * l.HF; e += nthreads) {
        const int layer = e / l.HF, i = e - layer * l.HF;
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        float v;
        if (i < 64) v = Lp[f.o_ac + i];
        else if (i < 64 + f.Wp) v = Lp[f.o_b1 + i - 64];
        else if (i < 64 + 2 * f.Wp) v = Lp[f.o_b2 + i - 64 - f.Wp];
        else if (i < 128 + 2 * f.Wp) { const int j = i - 64 - 2 * f.Wp; v = j < 2 * f.DOp ? Lp[f.o_b3 + j] : 0.f; }
        else v = Lp[f.o_logS + i - 128 - 2 * f.Wp];
        lds[l.o_HEAD + e] = v;
    }
}

// fused stages: the same head layout from the density bias blocks of the fused 4-chain image (b1' instead of b1)
__device__ __forceinline__ void r8f_load_heads(const FlowDims& f, const R8Lds& l, const float* __restrict__ packed, float* lds,
                                               int tid, int nthreads) {
    const int BS = r4f_bias_stride(f.Wp);
    for (int e = tid; e < f.K * l.HF; e += nthreads) {
        const int layer = e / l.HF, i = e - layer * l.HF;
        const float* Bp = packed + f.o_r4fb + (size_t)layer * BS;            // b1'[Wp] | b2[Wp] | ac[32] | shift[16] | scale[16] | logS
        float v = 0.f;
        if (i < 64) { if (i < 32) v = Bp[2 * f.Wp + i]; }
        else if (i < 64 + 2 * f.Wp) v = Bp[i - 64];
        else if (i < 128 + 2 * f.Wp) { const int j = i - 64 - 2 * f.Wp; if (j < 2 * f.DOp) v = Bp[2 * f.Wp + 32 + j]; }   // (DOp == 16)
        else if (i == 128 + 2 * f.Wp) v = Bp[2 * f.Wp + 64];
        lds[l.o_HEAD + e] = v;
    }
}

// An opaque zero, re-made in every layer iteration and added to the LDS base: hipcc otherwise hoists every LDS address of the
// (large) layer body out of the loop - several hundred registers, spilled to scratch (G = 5: 500 VGPRs).
__device__ __forceinline__ int r8_opaque_zero() {
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}

__device__ __forceinline__ float r8_part_sum(const float* p) {          // the 4 waves' partials of one output, fixed order
    return (p[0] + p[R8 * R4_DS]) + (p[2 * R8 * R4_DS] + p[3 * R8 * R4_DS]);
}
// dense narrow products: NP = 8 (W3: 4 waves x 2 k-quad halves, [NP][8][32]) or 16 (W1T: 4 waves x 4 quarters, [NP][8][16])
// partial products of one output, `st` floats apart, added as a fixed balanced tree
template <int NP>
__device__ __forceinline__ float r8_part_sum_n(const float* p, int st) {
    float v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = p[i * st];
#pragma unroll
    for (int w = 1; w < NP; w *= 2)
#pragma unroll
        for (int i = 0; i < NP; i += 2 * w) v[i] = v[i] + v[i + w];
    return v[0];
}
// sum of the two k-quad halves of a dense D x D product (lanes l and l ^ 32 hold the partial products of column l & 31), on
// every lane: v_permlane32_swap (gfx950) makes [lower | lower] and [upper | upper] of the register - no LDS round trip
__device__ __forceinline__ float r8_sum_halves(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// sum of the four k-quad quarters of a dense 16-column product (lanes l, l ^ 16, l ^ 32, l ^ 48 hold the partial products of
// column l & 15), on every lane: (q0 + q1) + (q2 + q3) with v_permlane16_swap / v_permlane32_swap
__device__ __forceinline__ float r8_sum_quarters(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return r8_sum_halves(__uint_as_float(r[0]) + __uint_as_float(r[1]));
}

// One product into the hidden width: OUT[8][Wp] = epilogue(ACT[8][4 NQ] @ B).  Stream tiles T0 .. : NQ of this wave's own
// column group, then (G == 5) NQ / 4 of the fifth group (this wave's quarter of K).  `epi(v, col)` -> stored value of an
// output BEFORE masking; EP 1: ReLU, decisions kept in mb; EP 2: multiplied by the decisions in mb.
// mb: THIS thread's two decision bytes of this layer and stage - byte 0: bit i = output row i of its column in the 64-column groups,
// byte 1: its two rows of the fifth group (r6: a lane keeps the decisions of its own outputs - the reverse sweep's products have the
// same thread-to-output mapping - instead of 8 + 2 wave ballots per stage written by lane 0 and read back as broadcasts: ~55 VALU /
// LDS instructions fewer per stage in the epilogues, which nothing overlaps).  Ends with a workgroup barrier.
// (leading dimensions are template parameters: with run-time strides hipcc hoists one address register per LDS access out of
// the layer loop - several hundred of them - and spills)
template <int G, int T0, int NQ, int TOTAL, int EP, int lda, int ldo, class ST, class Bias>
__device__ __forceinline__ void r8_dense_wide(ST& s, const float* act, float* out, float* PART,
                                              unsigned char* mb, const Tid8f& t, Bias bias) {
    constexpr int EX = G - 4;
    f32x4 o[2];
    {
        S8Acc<2> acc;
        s8_zero(acc);
        s8_run<T0, NQ, TOTAL>(s, act + t.arow * lda, 4 * lda, acc);
        s8_fold(acc, o);
    }
    {
        const int col = 64 * t.wave + t.lane;
        const float bv = bias(col);
        unsigned bits = 0;
        if constexpr (EP == 2) bits = mb[0];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = o[rb][r] + bv;
                if constexpr (EP == 1) {
                    const bool p = v > 0.f;
                    bits |= p ? (1u << (4 * rb + r)) : 0u;
                    v = p ? v : 0.f;
                } else if constexpr (EP == 2) {
                    v = (bits & (1u << (4 * rb + r))) ? v : 0.f;
                }
                out[(4 * rb + r) * ldo + col] = v;
            }
        if constexpr (EP == 1) mb[0] = (unsigned char)bits;
    }
    // (the main epilogue comes FIRST: with the two GEMMs back to back hipcc's allocator runs out of registers - 500 spilled; its
    //  rows issued among the fifth group's MFMAs - tried in r6 - cost more than they hide: an LDS access there sits in front of every
    //  tile's operand wait, and the register-only form of the rows spills)
    if constexpr (EX) {
        // the fifth column group, N-split like the other four: wave w owns columns 256 + 16 w .. + 15 for ALL of K, streamed as
        // dense tiles (4 k-quads side by side in the lane quarters, each quarter multiplying its own k-quad); the four quarters'
        // partial products are added by lane swaps - no partial sums through LDS, no barrier, no second epilogue pass
        // (r6; before: K split over the waves, PART round trip + barrier + finish: ~0.5 k cycles per stage, 4 stages per layer pair)
        static_assert(NQ % 4 == 0, "a dense 16-column tile holds 4 k-quads");
        f32x4 ox[2];
        {
            S8Acc<2> acc;
            s8_zero(acc);
            s8_run_k<16, T0 + NQ, NQ / 4, TOTAL>(s, act + t.arow * lda + 4 * (t.lane >> 4), 4 * lda, acc);
            s8_fold(acc, ox);
        }
        // transpose-reduce: 8 rows x 4 quarters of partial products per lane -> lane (q, n) keeps the totals of rows 2 q and 2 q + 1
        // of column n.  permlane32_swap(a, b) -> [a.lo | b.lo], [a.hi | b.hi]: their sum holds a's half-sums in the lower half and
        // b's in the upper; permlane16_swap the same one level down.  Every total = (q0 + q2) + (q1 + q3).
        auto pair32 = [](float a, float b) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            return __uint_as_float(r[0]) + __uint_as_float(r[1]);
        };
        auto pair16 = [](float a, float b) {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            return __uint_as_float(r[0]) + __uint_as_float(r[1]);
        };
        float tt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tt[r] = pair32(ox[0][r], ox[1][r]);          // lower half: row r, upper half: row r + 4
        float u2[2] = {pair16(tt[0], tt[2]), pair16(tt[1], tt[3])};             // quarter q: row 2 q | row 2 q + 1
        const int q = t.lane >> 4;
        const int col = 256 + 16 * t.wave + (t.lane & 15);
        const float bv = bias(col);
        unsigned bits = 0;
        if constexpr (EP == 2) bits = mb[1];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v = u2[i] + bv;
            if constexpr (EP == 1) {
                const bool p = v > 0.f;
                bits |= p ? (1u << i) : 0u;
                v = p ? v : 0.f;
            } else if constexpr (EP == 2) {
                v = (bits & (1u << i)) ? v : 0.f;
            }
            out[(2 * q + i) * ldo + col] = v;
        }
        if constexpr (EP == 1) mb[1] = (unsigned char)bits;
    }
    s8_barrier();
}

// The training tape of an 8-chain tile (TapeDims layout, rows row0 .. row0 + 7; fabhip_flow_log_prob_tape): every matrix the
// parameter-gradient GEMMs read is copied from LDS to HBM by all 256 threads right behind the barrier that completes it.  The
// stores are compiler-tracked vector-memory operations issued while the weight ring is in flight: they only make the ring's
// hand-counted waits conservative (gfx9 retires loads and stores of a wave in issue order; stream_r8.h).
struct R8Tape {
    const TapeDims* td;
    float* tape;
    long row0;
};
// 8 rows x 64 G floats (a hidden tile, leading dimension 64 G + 4) to tape rows `dld` floats apart: 16 G float4 per row, thread
// (row = tid >> 5, c = tid & 31) takes columns c, c + 32, c + 64 - no division by a run-time width
template <int G>
__device__ __forceinline__ void r8_tape_wide(float* __restrict__ dst, int dld, const float* src, int tid) {
    constexpr int WS = 64 * G + 4, W4 = 16 * G;
    const int r = tid >> 5, c0 = tid & 31;
    const float* sp = src + r * WS;
    float* dp = dst + (long)r * dld;
#pragma unroll
    for (int c = 0; c < W4; c += 32)
        if (c + c0 < W4) *reinterpret_cast<float4*>(dp + 4 * (c + c0)) = *reinterpret_cast<const float4*>(sp + 4 * (c + c0));
}
// 8 rows x w floats (w = 16 or 32) of a state-shaped buffer (leading dimension R4_DS) to tape rows w floats apart
__device__ __forceinline__ void r8_tape_narrow(float* __restrict__ dst, int w, const float* src, int tid) {
    const int sh = w == 32 ? 3 : 2;                                  // float4 per row: 8 / 4
    if (tid < (R8 << sh)) {
        const int r = tid >> sh, c = tid & ((1 << sh) - 1);
        *reinterpret_cast<float4*>(dst + (long)r * w + 4 * c) = *reinterpret_cast<const float4*>(src + r * R4_DS + 4 * c);
    }
}

// log q(x) and d log q / dx for the 8 rows in X0 (columns >= D zero; DP and PRM zeroed by the caller); the gradient is left
// in the state buffer whose offset is returned through *grad_off.  Returns log q of row `tid >> 4` on threads < 128.
// All 256 threads of the workgroup must call it; it ends with a workgroup barrier.
template <int G, bool FUSED = false, bool TAPE = false, class ST>
__device__ __forceinline__ float flow_log_prob_r8(const FlowDims& f, const R8Lds& l, const float* __restrict__ packed, float* lds,
                                  const Tid8f& t, ST& s, int* grad_off, const R8Tape* tp = nullptr) {
    static_assert(!(TAPE && FUSED), "the tape holds z and the full cotangent of z: one stage per matrix");
    constexpr int EX = G - 4, NQW = 16 * G, NQK = 4 * G;                  // k-quads of K = Wp; of a wave's quarter of it
    constexpr int NQ1 = FUSED ? R8_KD4 : R8_Kd4;                            // k-quads of the first Linear's K (fused: the whole state)
    constexpr int F_AW = 0, F_W1 = R8_TD, F_W2 = F_W1 + NQ1 + EX * (NQ1 / 4), F_W3 = F_W2 + NQW + EX * NQK, TF = F_W3 + NQK / 2;
    constexpr int B_W3T = 0, B_W2T = R8_Ko4 + EX * (R8_Ko4 / 4), B_W1T = B_W2T + NQW + EX * NQK,
                  B_AWT = B_W1T + (FUSED ? NQK / 2 : NQK / 4), TR = B_AWT + (FUSED ? 1 : R8_TD);
    constexpr int TP = FUSED ? (G == 5 ? 3 : 2) * R8F_RD : TF + EX;         // padded tiles per layer and direction (zero tiles behind)
    static_assert(ST::RD == (FUSED ? R8F_RD : R8_RD) && TP % ST::RD == 0 && TP >= TF && TP >= TR && (FUSED || TF == TR),
                  "a layer's padded tile count must be a multiple of the ring depth");
    constexpr int CONT = S8_INF;                                            // "tiles left in the stream": never drained
    const int h2 = t.lane >> 5, h4 = t.lane >> 4;                           // this lane's k-quad inside a dense tile of 2 / 4
    constexpr int WS = 64 * G + 4;                                          // = l.WS, as a constant (see r8_dense_wide)
    const float* img = packed + (FUSED ? f.o_r8f : f.o_r8);
    int cur = l.o_X0, nxt = l.o_X1;
    float* const lds0 = lds;
    const bool ew = t.tid < 128;
    const int row = t.row, c = t.c;
    const int DOp = f.DOp;
    float logq = 0.f;
    // this wave's stream: [layers K-1 .. 0 forward | layers 0 .. K-1 reverse | tail], TP tiles per layer and direction
    s8_prologue(s, reinterpret_cast<const float4*>(img) + (size_t)t.wave * (FUSED ? r8f_wave_tiles(G, f.K) : r8_wave_tiles(G, f.K)) * 64);
    auto pad_tiles = [&](float* lds, auto tc) {                             // the zero tiles that round a layer up to TP
        constexpr int T = decltype(tc)::value;
        if constexpr (T < TP) {
            S8Acc<2> acc;
            s8_iter_k<4, TP - T, 0, T % ST::RD, CONT>(s, lds + l.o_X0 + t.arow * R4_DS, 4 * R4_DS, acc);
        }
    };
    for (int layer = f.K - 1; layer >= 0; --layer) {
        float* lds = lds0 + r8_opaque_zero();
        float* HA = lds + l.o_HA;
        float* HB = lds + l.o_HB;
        float* PART = lds + l.o_PART;
        const float* HD = lds + l.o_HEAD + (size_t)layer * l.HF;
        unsigned char* mk = reinterpret_cast<unsigned char*>(lds + l.o_MASK + (size_t)layer * (NWAVE * 64) + t.tid);   // bytes 0, 1: h1; 2, 3: h2
        const bool tl = layer == f.K - 2;
        if (tl) FAB_TL(f, 0);
        float* tl_layer = nullptr;
        if constexpr (TAPE) {
            tl_layer = tp->tape + (size_t)layer * tp->td->layer_stride;
            r8_tape_narrow(tl_layer + tp->td->o_ZA + tp->row0 * tp->td->wz, tp->td->wz, lds + cur, t.tid);
        }
        {   // InvertibleAffine.inverse (+ folded ActNorm): z <- z @ W' + ac   (every wave; wave 0 stores)
            f32x4 o[2];
            S8Acc<2> acc;
            s8_zero(acc);
            s8_run_k<8, F_AW, R8_TD, CONT + F_AW>(s, lds + cur + t.arow * R4_DS + 4 * h2, 4 * R4_DS, acc);
            s8_fold(acc, o);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[rb][r] = r8_sum_halves(o[rb][r]);
            {   // every wave holds the whole product: wave w stores chains 2 w, 2 w + 1 (a quarter of the epilogue's latency each)
                const float bv = HD[t.lane];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (((4 * rb + r) >> 1) == t.wave) lds[nxt + (4 * rb + r) * R4_DS + t.lane] = t.lane < 32 ? o[rb][r] + bv : 0.f;
            }
        }
        logq += HD[128 + 2 * f.Wp];
        float* Z = lds + nxt;
        if constexpr (!FUSED) {
            s8_barrier();
            if (tl) FAB_TL(f, 1);
        }
        // conditioner: HA = relu(z[:, :d] W1 + b1), HB = relu(HA W2 + b2)   (fused: HA = relu(y W1' + b1') from the layer's INPUT: z is
        // first read by the coupling, three barriers from here)
        r8_dense_wide<G, F_W1, NQ1, CONT + TP, 1, R4_DS, WS>(s, FUSED ? lds + cur : Z, HA, PART, mk, t, [&](int col) { return HD[64 + col]; });
        if (tl) FAB_TL(f, 2);
        r8_dense_wide<G, F_W2, NQW, CONT + TP, 1, WS, WS>(s, HA, HB, PART, mk + 2, t, [&](int col) { return HD[64 + f.Wp + col]; });
        if (tl) FAB_TL(f, 3);
        {   // (shift | scale) = HB W3: K split over the waves, partial [8][64] products to PART
            f32x4 o[2];
            S8Acc<2> acc;
            s8_zero(acc);
            s8_run_k<8, F_W3, NQK / 2, CONT + F_W3>(s, HB + t.arow * WS + 4 * NQK * t.wave + 4 * h2, 4 * WS, acc);
            s8_fold(acc, o);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) PART[((2 * t.wave + h2) * R8 + 4 * rb + r) * 32 + (t.lane & 31)] = o[rb][r];
        }
        pad_tiles(lds, std::integral_constant<int, TF>{});   // (the stream runs on into the next layer: nothing to request here)
        s8_barrier();
        if (tl) FAB_TL(f, 4);
        // AffineCoupling.inverse: z2 <- (z2 - shift) exp(-s), log_det = -sum(s)
        if (ew) {
            float ssum = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int j = c + 16 * it;
                if (j < f.DO) {
                    const float shift = r8_part_sum_n<8>(PART + row * 32 + j, R8 * 32) + HD[64 + 2 * f.Wp + j];
                    const float sv = r8_part_sum_n<8>(PART + row * 32 + DOp + j, R8 * 32) + HD[64 + 2 * f.Wp + DOp + j];
                    const float es = expf(-sv);
                    const float v2 = (Z[row * R4_DS + f.d + j] - shift) * es;
                    Z[row * R4_DS + f.d + j] = v2;
                    lds[l.o_ES + ((size_t)layer * R8 + row) * DOp + j] = es;
                    lds[l.o_V2 + ((size_t)layer * R8 + row) * DOp + j] = v2;
                    ssum += sv;
                }
            }
            logq += -row16_sum(ssum);
        }
        s8_barrier();
        if (tl) FAB_TL(f, 5);
        if constexpr (TAPE) {
            // ONE burst of stores per layer (a store sits in the wave's vmcnt queue until it is acknowledged, and the ring's waits
            // count it): z1 | ones column, h1 = HA, h2 = HB (both intact until the next layer's conditioner) and their ones columns.
            // (the layer's input went out at the layer's top: the next affine stage overwrites its buffer without a barrier)
            const TapeDims& td = *tp->td;
            {
                const int r = t.tid >> 5, j = t.tid & 31;            // w1 = 32 (d <= 16)
                tl_layer[td.o_Z1 + (tp->row0 + r) * td.w1 + j] = j < f.d ? Z[r * R4_DS + j] : (j == td.w1 - 16 ? 1.f : 0.f);
            }
            if (t.tid < R8 * 16) {
                const int r = t.tid >> 4, j = t.tid & 15;
                const float one = j == 0 ? 1.f : 0.f;
                tl_layer[td.o_H1 + (tp->row0 + r) * td.wh + f.Wp + j] = one;
                tl_layer[td.o_H2 + (tp->row0 + r) * td.wh + f.Wp + j] = one;
            }
            r8_tape_wide<G>(tl_layer + td.o_H1 + tp->row0 * td.wh, td.wh, HA, t.tid);
            r8_tape_wide<G>(tl_layer + td.o_H2 + tp->row0 * td.wh, td.wh, HB, t.tid);
        }
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    // DiagGaussian.log_prob, the seed of the reverse sweep, and the first layer's coupling cotangents
    if (ew) {
        float* DP = lds + l.o_DP;
        const float* base = packed + f.o_base;
        float* Zc = lds + cur;
        float bsum = 0.f;
        for (int j = c; j < f.D; j += 16) {
            const float ls = base[f.Dp + j];
            const float sc = expf(ls);
            const float zn = (Zc[row * R4_DS + j] - base[j]) / sc;
            bsum += ls + 0.5f * (zn * zn);
            if constexpr (TAPE) {                         // what d/dloc and d/dlog_scale reduce over the batch
                float* TB = tp->tape + tp->td->o_TB + (tp->row0 + row) * tp->td->wb;
                TB[j] = zn / sc;
                TB[tp->td->wz + j] = zn * zn - 1.f;
            }
            Zc[row * R4_DS + j] = -(zn / sc);
        }
        if constexpr (TAPE) tp->tape[tp->td->o_TB + (tp->row0 + row) * tp->td->wb + 2 * tp->td->wz + c] = c == 0 ? 1.f : 0.f;
        logq += -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
        // (the 16 lanes of a row wrote the whole row: same wave, program order)
        for (int j = c; j < f.DO; j += 16) {
            const float g2 = Zc[row * R4_DS + f.d + j];
            const float es = lds[l.o_ES + (size_t)row * DOp + j];
            const float v2 = lds[l.o_V2 + (size_t)row * DOp + j];
            DP[row * R4_DS + j] = -(g2 * es);
            DP[row * R4_DS + DOp + j] = -(g2 * v2) - 1.f;
            Zc[row * R4_DS + f.d + j] = g2 * es;
        }
    }
    s8_barrier();
    // reverse sweep: g = d log q / d(state), layers 0 .. K-1
    for (int layer = 0; layer < f.K; ++layer) {
        float* lds = lds0 + r8_opaque_zero();
        float* HA = lds + l.o_HA;
        float* HB = lds + l.o_HB;
        float* DP = lds + l.o_DP;
        float* PART = lds + l.o_PART;
        unsigned char* mk = reinterpret_cast<unsigned char*>(lds + l.o_MASK + (size_t)layer * (NWAVE * 64) + t.tid);   // bytes 0, 1: h1; 2, 3: h2
        float* Gs = lds + cur;
        const bool tl = layer == 1;
        if (tl) FAB_TL(f, 16);
        float* tl_layer = nullptr;
        if constexpr (TAPE) {
            tl_layer = tp->tape + (size_t)layer * tp->td->layer_stride;
            r8_tape_narrow(tl_layer + tp->td->o_DP + tp->row0 * tp->td->wp, tp->td->wp, DP, t.tid);
        }
        // d relu(h2) = DP W3T masked by h2 > 0 -> HA;  d relu(h1) = HA W2T masked by h1 > 0 -> HB
        r8_dense_wide<G, B_W3T, R8_Ko4, CONT + TP, 2, R4_DS, WS>(s, DP, HA, PART, mk + 2, t, [](int) { return 0.f; });
        if (tl) FAB_TL(f, 17);
        r8_dense_wide<G, B_W2T, NQW, CONT + TP, 2, WS, WS>(s, HA, HB, PART, mk, t, [](int) { return 0.f; });
        if (tl) FAB_TL(f, 18);
        if constexpr (FUSED) {
            {   // g_y = dh1 W1'^T + g_z W'^T: K split over the waves (this wave's quarter of the hidden width + 8 columns of g_z)
                f32x4 o[2];
                S8Acc<2> acc;
                s8_zero(acc);
                s8_run_k<8, B_W1T, NQK / 2, CONT + B_W1T>(s, HB + t.arow * WS + 4 * NQK * t.wave + 4 * h2, 4 * WS, acc);
                s8_run_k<8, B_AWT, 1, CONT + B_AWT>(s, Gs + t.arow * R4_DS + 8 * t.wave + 4 * h2, 4 * R4_DS, acc);
                s8_fold(acc, o);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) PART[((2 * t.wave + h2) * R8 + 4 * rb + r) * 32 + (t.lane & 31)] = o[rb][r];
            }
            pad_tiles(lds, std::integral_constant<int, TR>{});
            s8_barrier();
            if (tl) FAB_TL(f, 19);
            if (ew) {   // the sum of the 8 partial products; where the NEXT layer's g2 appears, its coupling cotangents
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int col = c + 16 * it, j = col - f.d;
                    float v = col < f.D ? r8_part_sum_n<8>(PART + row * 32 + col, R8 * 32) : 0.f;
                    if (layer + 1 < f.K && j >= 0 && j < f.DO) {
                        const float es = lds[l.o_ES + ((size_t)(layer + 1) * R8 + row) * DOp + j];
                        const float v2 = lds[l.o_V2 + ((size_t)(layer + 1) * R8 + row) * DOp + j];
                        DP[row * R4_DS + j] = -(v * es);
                        DP[row * R4_DS + DOp + j] = -(v * v2) - 1.f;
                        v = v * es;
                    }
                    lds[nxt + row * R4_DS + col] = v;
                }
            }
        } else {
        {   // conditioner input gradient = HB W1T: K split over the waves
            f32x4 o[2];
            S8Acc<2> acc;
            s8_zero(acc);
            s8_run_k<16, B_W1T, NQK / 4, CONT + B_W1T>(s, HB + t.arow * WS + 4 * NQK * t.wave + 4 * h4, 4 * WS, acc);
            s8_fold(acc, o);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) PART[((4 * t.wave + h4) * R8 + 4 * rb + r) * 16 + (t.lane & 15)] = o[rb][r];
        }
        s8_barrier();
        if (tl) FAB_TL(f, 19);
        if (ew && c < f.d) Gs[row * R4_DS + c] += r8_part_sum_n<16>(PART + row * 16 + c, R8 * 16);   // g[:, :d] += ...   (d <= 16)
        s8_barrier();
        if (tl) FAB_TL(f, 20);
        if constexpr (TAPE) {                             // the reverse sweep's burst: d relu(h2) = HA, d relu(h1) = HB, the cotangent of z
            r8_tape_wide<G>(tl_layer + tp->td->o_E2 + tp->row0 * tp->td->we, tp->td->we, HA, t.tid);
            r8_tape_wide<G>(tl_layer + tp->td->o_E1 + tp->row0 * tp->td->we, tp->td->we, HB, t.tid);
            r8_tape_narrow(tl_layer + tp->td->o_GZ + tp->row0 * tp->td->wz, tp->td->wz, Gs, t.tid);
        }
        {   // g <- g W'^T (every wave; wave 0 stores, and forms the NEXT layer's coupling cotangents where its g2 appears)
            f32x4 o[2];
            S8Acc<2> acc;
            s8_zero(acc);
            s8_run_k<8, B_AWT, R8_TD, CONT + B_AWT>(s, Gs + t.arow * R4_DS + 4 * h2, 4 * R4_DS, acc);
            s8_fold(acc, o);
            pad_tiles(lds, std::integral_constant<int, TR>{});
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[rb][r] = r8_sum_halves(o[rb][r]);
            {   // wave w finishes chains 2 w, 2 w + 1 (every wave holds the whole product)
                const int j = t.lane - f.d;
                const bool cpl = layer + 1 < f.K && j >= 0 && j < f.DO;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = 4 * rb + r;
                        if ((rr >> 1) != t.wave) continue;
                        float v = t.lane < 32 ? o[rb][r] : 0.f;
                        if (cpl) {
                            const float es = lds[l.o_ES + ((size_t)(layer + 1) * R8 + rr) * DOp + j];
                            const float v2 = lds[l.o_V2 + ((size_t)(layer + 1) * R8 + rr) * DOp + j];
                            DP[rr * R4_DS + j] = -(v * es);
                            DP[rr * R4_DS + DOp + j] = -(v * v2) - 1.f;
                            v = v * es;
                        }
                        lds[nxt + rr * R4_DS + t.lane] = v;
                    }
            }
        }
        }
        s8_barrier();
        if (tl) FAB_TL(f, 21);
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    s8_drain(s);                       // the requests past the end of the stream (R8_TAIL)
    *grad_off = cur;
    return logq;
}

// host side: fused stages where the image exists and FABHIP_OPT_R4_STREAM >= 2 (the switch of the 4-chain kernel: one A/B for both)
int option(int key);                                   // (launch.h)
static inline bool use_r8_fused(const FlowDims& f) { return r8f_shape_ok(f) && option(FABHIP_OPT_R4_STREAM) >= 2; }

}  // namespace fab
