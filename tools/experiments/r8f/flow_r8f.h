// 8-chain tiles, FUSED STAGES (r4): the arithmetic and the weight stream of flow_r8.h, with 3 workgroup barriers (+ 2 cheap
// ones at width 320) per layer and direction instead of 7.
//
// Why: the stage timeline of flow_r8.h (profiles/r3/hmc_r8_stage_timeline.txt) shows 15.6 k of a layer pair's 35.2 k cycles
// in eight SHORT stages - 5 .. 20 weight tiles each, 1.0 - 3.1 k cycles where the stream needs 0.4 - 1.6 k - because every
// stage is a dependent chain (barrier release -> LDS read -> MFMA chain -> fold -> epilogue -> LDS write -> barrier) of
// ~0.8 k cycles whatever its size.  A layer has only three products whose inputs come from OTHER waves (W1 -> W2 -> W3,
// W3T -> W2T -> W1T); everything between W3 / W1T and the next layer's W1 / W3T is small (8 x 32 values) and is now done
// REDUNDANTLY BY EVERY WAVE behind a wave-local `s_waitcnt lgkmcnt(0)`:
//   forward  F1: coupling of the previous layer (4 partials + bias, exp, log-det) -> state | D x D map | W1 (+ fifth-group partials)
//            F2: [fifth group of h1] W2 (+ partials)           F3: [fifth group of h2] W3 K-split partials, next ring
//   reverse  R1: sum of the previous layer's W1T partials | D x D map^T, next ring, coupling cotangents | W3T (+ partials)
//            R2: [fifth group] W2T (+ partials)                R3: [fifth group] W1T K-split partials
// All waves write IDENTICAL values to the shared state buffers (a wave reads only after its own write has completed, so it
// sees the value whoever wrote last); buffers rotate over three slots so that no wave overwrites what a slower wave of the
// same stage still reads.  The fifth column group of a 320-wide layer (K-split over the waves) is finished at the START of
// the stage that consumes it, and only the k-quads that need it wait at a second barrier all waves reach at once.
// Same sums in the same order as flow_r8.h: bit-identical results (tests/test_gpu_hmc_shapes.py).
#pragma once
#include "flow_r8.h"

namespace fab {

__device__ __forceinline__ void r8f_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// the fifth column group (columns 256 + lane) of a hidden activation from its K-split partials: wave w finishes chains 2 w, 2 w + 1
template <int EP, int ldo, class Bias>
__device__ __forceinline__ void r8f_finish5(const float* PX, float* act, unsigned long long* mk, const Tid8f& t, Bias bias) {
    const int col = 256 + t.lane;
    const float bv = bias(col);
    unsigned long long* mw = mk + 4 * R8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = 2 * t.wave + i;
        float v = r8_part_sum(PX + rr * R4_DS + t.lane) + bv;
        if constexpr (EP == 1) {
            const unsigned long long m = __ballot(v > 0.f);
            if (t.lane == 0) mw[rr] = m;
            v = v > 0.f ? v : 0.f;
        } else if constexpr (EP == 2) {
            v = ((mw[rr] >> t.lane) & 1ull) ? v : 0.f;
        }
        act[rr * ldo + col] = v;
    }
}

// One product into the hidden width (see r8_dense_wide): OUT[8][Wp] = epilogue(ACT[8][4 NQ] @ B); the partial products of
// the fifth column group go to PX and are finished by the NEXT stage (r8f_finish5).  SPLIT: ACT's own fifth group was
// finished at the start of this stage - the k-quads from 64 on wait for it at a barrier.  Ends with a workgroup barrier.
template <int G, int T0, int NQ, int TOTAL, int EP, int lda, int ldo, bool SPLIT, class Bias>
__device__ __forceinline__ void r8f_wide(R8Stream& s, const float* act, float* out, float* PX, unsigned long long* mk,
                                         const Tid8f& t, Bias bias) {
    constexpr int EX = G - 4;
    f32x4 o[2];
    {
        S8Acc<2> acc;
        s8_zero(acc);
        if constexpr (SPLIT && EX) {
            static_assert(NQ > 64, "the fifth group starts at k-quad 64");
            s8_run<T0, 64, TOTAL>(s, act + t.arow * lda, 4 * lda, acc);
            s8_barrier();
            s8_run<T0 + 64, NQ - 64, TOTAL>(s, act + t.arow * lda + 256, 4 * lda, acc);
        } else {
            s8_run<T0, NQ, TOTAL>(s, act + t.arow * lda, 4 * lda, acc);
        }
        s8_fold(acc, o);
    }
    {
        const int col = 64 * t.wave + t.lane;
        const float bv = bias(col);
        unsigned long long* mw = mk + t.wave * R8;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = o[rb][r] + bv;
                if constexpr (EP == 1) {
                    const unsigned long long m = __ballot(v > 0.f);
                    if (t.lane == 0) mw[4 * rb + r] = m;
                    v = v > 0.f ? v : 0.f;
                } else if constexpr (EP == 2) {
                    v = ((mw[4 * rb + r] >> t.lane) & 1ull) ? v : 0.f;
                }
                out[(4 * rb + r) * ldo + col] = v;
            }
    }
    if constexpr (EX) {                // partial products of the fifth group: k-quads [w NQ / 4, (w + 1) NQ / 4)
        static_assert(NQ % 4 == 0, "the fifth column group is K-split over 4 waves");
        f32x4 ox[2];
        S8Acc<2> acc;
        s8_zero(acc);
        s8_run<T0 + NQ, NQ / 4, TOTAL>(s, act + t.arow * lda + NQ * t.wave, 4 * lda, acc);
        s8_fold(acc, ox);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) PX[(t.wave * R8 + 4 * rb + r) * R4_DS + t.lane] = ox[rb][r];
    }
    s8_barrier();
}

// A narrow product out of the hidden width, K split over the waves (wave w: k = NQK w .. NQK (w + 1) quads): partial [8][64]
// products to PART.  No barrier.
template <int T0, int NQK, int TOTAL, int lda>
__device__ __forceinline__ void r8f_ksplit(R8Stream& s, const float* act, float* PART, const Tid8f& t) {
    f32x4 o[2];
    S8Acc<2> acc;
    s8_zero(acc);
    s8_run<T0, NQK, TOTAL>(s, act + t.arow * lda + 4 * NQK * t.wave, 4 * lda, acc);
    s8_fold(acc, o);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) PART[(t.wave * R8 + 4 * rb + r) * R4_DS + t.lane] = o[rb][r];
}

// log q(x) and d log q / dx for the 8 rows in X0 (columns >= D zero; DP and PRM zeroed by the caller): the contract of
// flow_log_prob_r8.
template <int G>
__device__ __forceinline__ float flow_log_prob_r8f(const FlowDims& f, const R8Lds& l, const float* __restrict__ packed, float* lds,
                                                   const Tid8f& t, R8Stream& s, int* grad_off) {
    constexpr int EX = G - 4, NQW = 16 * G, NQK = 4 * G;
    constexpr int F_AW = 0, F_W1 = R8_KD4, F_W2 = F_W1 + R8_Kd4 + EX * (R8_Kd4 / 4), F_W3 = F_W2 + NQW + EX * NQK, TF = F_W3 + NQK;
    constexpr int B_W3T = 0, B_W2T = R8_Ko4 + EX * (R8_Ko4 / 4), B_W1T = B_W2T + NQW + EX * NQK, B_AWT = B_W1T + NQK,
                  TR = B_AWT + R8_KD4;
    constexpr int LF = NWAVE * (TF + TR) * 256;
    constexpr int WS = 64 * G + 4;
    const float* img = packed + f.o_r8;
    float* const lds0 = lds;
    const int DOp = f.DOp;
    const int er = t.lane >> 4, ec = t.lane & 15;                           // redundant element-wise steps: rows er, er + 4; column ec
    float lq0 = 0.f, lq1 = 0.f;                                             // log q of rows er / er + 4 (every wave)
    auto fwd_base = [&](int layer) { return reinterpret_cast<const float4*>(img + (size_t)layer * LF) + (size_t)t.wave * TF * 64; };
    auto rev_base = [&](int layer) {
        return reinterpret_cast<const float4*>(img + (size_t)layer * LF) + (size_t)(NWAVE * TF + t.wave * TR) * 64;
    };
    // forward: state S (coupling output, D x D map input); D x D map output alternating ZN / ZP
    const int S = l.o_X0;
    int ZN = l.o_X1, ZP = l.o_X2;
    // AffineCoupling.inverse of layer `cl` from its W3 partials: [z1 | (z2 - shift) exp(-s)] -> S, log_det = -sum(s)
    auto coupling = [&](float* lds, int cl, int zp) {
        const float* HDc = lds + l.o_HEAD + (size_t)cl * l.HF;
        const float* PART = lds + l.o_PART;
        const float* Zp = lds + zp;
        float* Sd = lds + S;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = er + 4 * it;
            float ssum = 0.f;
            if (ec < f.DO) {
                const int j = ec;
                const float shift = r8_part_sum(PART + row * R4_DS + j) + HDc[64 + 2 * f.Wp + j];
                const float sv = r8_part_sum(PART + row * R4_DS + DOp + j) + HDc[64 + 2 * f.Wp + DOp + j];
                const float es = expf(-sv);
                const float v2 = (Zp[row * R4_DS + f.d + j] - shift) * es;
                Sd[row * R4_DS + f.d + j] = v2;
                if (t.wave == 0) {
                    lds[l.o_ES + ((size_t)cl * R8 + row) * DOp + j] = es;
                    lds[l.o_V2 + ((size_t)cl * R8 + row) * DOp + j] = v2;
                }
                ssum += sv;
            }
            if (ec < f.d) Sd[row * R4_DS + ec] = Zp[row * R4_DS + ec];
            const float sd = -row16_sum(ssum);
            if (it == 0) lq0 += sd; else lq1 += sd;
        }
    };
    s8_prologue(s, fwd_base(f.K - 1));
    for (int layer = f.K - 1; layer >= 0; --layer) {
        float* lds = lds0 + r8_opaque_zero();
        float* HA = lds + l.o_HA;
        float* HB = lds + l.o_HB;
        float* PART = lds + l.o_PART;
        float* PX = lds + l.o_PARTX;
        const float* HD = lds + l.o_HEAD + (size_t)layer * l.HF;
        unsigned long long* mk = reinterpret_cast<unsigned long long*>(lds + l.o_MASK) + (size_t)layer * 2 * G * R8;
        const bool tl = layer == f.K - 2;
        if (tl) FAB_TL(f, 0);
        // ---- F1: (previous coupling, at the end of the previous iteration) | D x D map | W1 -------------------------------
        {   // InvertibleAffine.inverse (+ folded ActNorm): z <- z @ W' + ac   (every wave computes and stores all of it)
            f32x4 o[2];
            S8Acc<2> acc;
            s8_zero(acc);
            s8_run<F_AW, R8_KD4, TF>(s, lds + S + t.arow * R4_DS, 4 * R4_DS, acc);
            s8_fold(acc, o);
            const float bv = HD[t.lane];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[ZN + (4 * rb + r) * R4_DS + t.lane] = o[rb][r] + bv;
        }
        {
            const float ls = HD[128 + 2 * f.Wp];
            lq0 += ls; lq1 += ls;
        }
        r8f_lds_wait();
        float* Z = lds + ZN;
        r8f_wide<G, F_W1, R8_Kd4, TF, 1, R4_DS, WS, false>(s, Z, HA, PX, mk, t, [&](int col) { return HD[64 + col]; });
        if (tl) FAB_TL(f, 1);
        // ---- F2: W2 ----------------------------------------------------------------------------------------------------------
        if constexpr (EX) r8f_finish5<1, WS>(PX, HA, mk, t, [&](int col) { return HD[64 + col]; });
        r8f_wide<G, F_W2, NQW, TF, 1, WS, WS, true>(s, HA, HB, PX, mk + G * R8, t, [&](int col) { return HD[64 + f.Wp + col]; });
        if (tl) FAB_TL(f, 2);
        // ---- F3: (shift | scale) = h2 W3, K split --------------------------------------------------------------------------
        if constexpr (EX) {
            r8f_finish5<1, WS>(PX, HB, mk + G * R8, t, [&](int col) { return HD[64 + f.Wp + col]; });
            s8_barrier();
        }
        r8f_ksplit<F_W3, NQK, TF, WS>(s, HB, PART, t);
        // the ring is empty here: request the next layer's (or the reverse sweep's first) tiles
        s8_prologue(s, layer > 0 ? fwd_base(layer - 1) : rev_base(0));
        s8_barrier();
        if (tl) FAB_TL(f, 3);
        coupling(lds, layer, ZN);           // (belongs to the next layer's F1; here so that the loop carries nothing but the fresh ring)
        r8f_lds_wait();
        if (tl) FAB_TL(f, 4);
        const int tmp = ZN; ZN = ZP; ZP = tmp;
    }
    s8_barrier();
    // DiagGaussian.log_prob, the seed of the reverse sweep, and the first layer's coupling cotangents (threads < 128, as flow_r8.h)
    const bool ew = t.tid < 128;
    const int row = t.row, c = t.c;
    float logq = t.wave == 0 ? lq0 : lq1;                                   // rows tid >> 4: wave 0 holds 0 .. 3, wave 1 holds 4 .. 7
    if (ew) {
        float* DP = lds + l.o_DP;
        const float* base = packed + f.o_base;
        float* Zc = lds + S;
        float bsum = 0.f;
        for (int j = c; j < f.D; j += 16) {
            const float ls = base[f.Dp + j];
            const float sc = expf(ls);
            const float zn = (Zc[row * R4_DS + j] - base[j]) / sc;
            bsum += ls + 0.5f * (zn * zn);
            Zc[row * R4_DS + j] = -(zn / sc);
        }
        logq += -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
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
    // reverse sweep: g = d log q / d(state), layers 0 .. K-1.  g alternates between GC / GN; C holds g + the W1T sum.
    int GC = S, GN = l.o_X2;
    const int C = l.o_X1;
    // g[:, :d] += previous layer's W1T partials -> C;  g <- C W'^T (tiles: the end of layer `pl`'s stream) -> GN, and layer
    // pl + 1's coupling cotangents where its g2 appears
    auto affine_t = [&](float* lds, int pl) {
        const float* PART = lds + l.o_PART;
        const float* Gs = lds + GC;
        float* Cd = lds + C;
        float* DP = lds + l.o_DP;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int rw = er + 4 * it;
            if (ec < f.d) Cd[rw * R4_DS + ec] = Gs[rw * R4_DS + ec] + r8_part_sum(PART + rw * R4_DS + ec);
            if (ec < f.DO) Cd[rw * R4_DS + f.d + ec] = Gs[rw * R4_DS + f.d + ec];
        }
        r8f_lds_wait();
        f32x4 o[2];
        S8Acc<2> acc;
        s8_zero(acc);
        s8_run<B_AWT, R8_KD4, TR>(s, Cd + t.arow * R4_DS, 4 * R4_DS, acc);
        s8_fold(acc, o);
        if (pl + 1 < f.K) s8_prologue(s, rev_base(pl + 1));
        const int j = t.lane - f.d;
        const bool cpl = pl + 1 < f.K && j >= 0 && j < f.DO;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 4 * rb + r;
                float v = o[rb][r];
                if (cpl) {
                    const float es = lds[l.o_ES + ((size_t)(pl + 1) * R8 + rr) * DOp + j];
                    const float v2 = lds[l.o_V2 + ((size_t)(pl + 1) * R8 + rr) * DOp + j];
                    DP[rr * R4_DS + j] = -(v * es);
                    DP[rr * R4_DS + DOp + j] = -(v * v2) - 1.f;
                    v = v * es;
                }
                lds[GN + rr * R4_DS + t.lane] = v;
            }
        const int tmp = GC; GC = GN; GN = tmp;
    };
    for (int layer = 0; layer < f.K; ++layer) {
        float* lds = lds0 + r8_opaque_zero();
        float* HA = lds + l.o_HA;
        float* HB = lds + l.o_HB;
        float* DP = lds + l.o_DP;
        float* PART = lds + l.o_PART;
        float* PX = lds + l.o_PARTX;
        unsigned long long* mk = reinterpret_cast<unsigned long long*>(lds + l.o_MASK) + (size_t)layer * 2 * G * R8;
        const bool tl = layer == 1;
        if (tl) FAB_TL(f, 16);
        // ---- R1: (previous layer's input gradient + D x D map^T, at the end of the previous iteration) | W3T --------------
        // d relu(h2) = DP W3T masked by h2 > 0 -> HA;  d relu(h1) = HA W2T masked by h1 > 0 -> HB
        r8f_wide<G, B_W3T, R8_Ko4, TR, 2, R4_DS, WS, false>(s, DP, HA, PX, mk + G * R8, t, [](int) { return 0.f; });
        if (tl) FAB_TL(f, 17);
        // ---- R2: W2T ---------------------------------------------------------------------------------------------------------
        if constexpr (EX) r8f_finish5<2, WS>(PX, HA, mk + G * R8, t, [](int) { return 0.f; });
        r8f_wide<G, B_W2T, NQW, TR, 2, WS, WS, true>(s, HA, HB, PX, mk, t, [](int) { return 0.f; });
        if (tl) FAB_TL(f, 18);
        // ---- R3: conditioner input gradient = HB W1T, K split -----------------------------------------------------------
        if constexpr (EX) {
            r8f_finish5<2, WS>(PX, HB, mk, t, [](int) { return 0.f; });
            s8_barrier();
        }
        r8f_ksplit<B_W1T, NQK, TR, WS>(s, HB, PART, t);
        s8_barrier();
        if (tl) FAB_TL(f, 19);
        affine_t(lds, layer);               // (belongs to the next layer's R1; its latch re-requests the ring, as flow_r8.h's)
        r8f_lds_wait();
        if (tl) FAB_TL(f, 20);
    }
    s8_barrier();
    *grad_off = GC;
    return logq;
}

}  // namespace fab
