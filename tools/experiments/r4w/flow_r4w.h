// 4- and 8-chain tiles, second generation ("r4w"): the flow density + d/dx for 4 RB chains per workgroup (RB = 1, 2 row
// blocks of 4 chains) on v_mfma_f32_4x4x1_16b_f32, with the short GEMM stages WAVE-PRIVATE.
//
// flow_r4.h splits every product over the 4 waves along K and adds the four partial products through LDS: 2 workgroup
// barriers + one LDS round trip per stage, 8 stages per layer and direction, ~1 k cycles each - 10 k of the 28 k cycles a
// layer pair takes while the two W x W stages (the only ones that carry real work) take 18 k.  Here only the W x W
// products (and the narrow products behind them) are K-split; what feeds them is computed by the wave that consumes it:
//   * the D x D affine maps are evaluated redundantly by every wave (32 MFMAs) - no exchange at all;
//   * the products INTO the hidden width (W1: d -> W, W3T: [shift | scale] -> W) are column-split so that wave w
//     produces exactly the 16 G hidden columns that form its K range of the following W x W product, applies bias /
//     ReLU / sign bits in registers and parks them in a wave-private LDS slice (the MFMA A operand is read from LDS);
//   * a W x W product writes its partial [4 RB][Wp] to LDS, ONE barrier, and each wave reduces only its own 16 G
//     columns (again the K range of the product that follows: W3 / W1T, K-split, dense narrow tiles);
//   * their partials cross one more barrier and are reduced by every wave redundantly (coupling transform, input
//     gradient), so the state of the next layer is wave-private again.
// 4 barriers per layer pair instead of 17.  The ARITHMETIC is that of flow_r4.h, addition by addition: a wave that now
// covers a K range four waves used to share keeps one accumulator per former wave and adds them in the former order
// ((P0 + P1) + (P2 + P3)) + bias, so both kernels give identical bits (tests/test_gpu_hmc_shapes.py).
//
// Two row blocks (RB = 2, 8 chains per workgroup) share every weight tile: the bytes a CU streams stay the same and the
// MFMA work doubles - for 1152 < chains <= 2304 per GPU (BASELINE cfg 3 / cfg 4: 2048), where 16-chain tiles fill half
// the chip and 4-chain tiles need two rounds of workgroups.
//
// Weight stream: ONE stream of 1-KiB tiles per wave in consumption order (k_pack_r4w), a ring of RT tiles in registers;
// the tile RT places down the stream is requested into a slot as soon as the slot's tile has been multiplied - across
// stage, layer and direction boundaries.  Per layer body the stream has TF (forward) / TR (reverse) tiles, padded to a
// multiple of RT so that every slot index is a compile-time constant inside the layer body.
#pragma once
#include "flow_r4.h"

namespace fab {

FAB_HD bool r4w_supported(const FlowDims& f) { return f.D <= 32 && f.Wp >= 128; }

// LDS plan (floats).  Wave-private regions are [wave][...]; RB4 = 4 RB rows.
struct R4WLds {
    int RB4, ZL, HL, PN;
    int o_X0, o_Z, o_DP, o_HS, o_PART, o_P3, o_ES, o_V2, o_MASK, o_TL, total;
};
FAB_HD R4WLds make_r4w_lds(const FlowDims& f, int RB) {
    R4WLds l;
    const int G = f.Wp / 64, SG = (16 * G + 63) / 64;
    l.RB4 = 4 * RB; l.ZL = R4_DS; l.HL = 64 * SG + 4; l.PN = f.Wp;
    int o = 0;
    l.o_X0 = o; o += l.RB4 * l.ZL;                        // the caller's input rows (shared)
    l.o_Z = o; o += NWAVE * 2 * l.RB4 * l.ZL;             // state ping-pong, per wave
    l.o_DP = o; o += NWAVE * l.RB4 * l.ZL;                // coupling-parameter cotangents, per wave
    l.o_HS = o; o += NWAVE * l.RB4 * l.HL;                // the wave's slice of the hidden activations / cotangents
    l.o_PART = o; o += NWAVE * l.RB4 * l.PN;              // W x W partial products (shared)
    l.o_P3 = o; o += 16 * l.RB4 * 16;                     // narrow partial products: 8 x [RB4][32] or 16 x [RB4][16] (shared)
    l.o_ES = o; o += f.K * l.RB4 * f.DOp;
    l.o_V2 = o; o += f.K * l.RB4 * f.DOp;
    l.o_MASK = o; o += f.K * NTHREADS;                    // one ReLU sign word per thread and layer
    l.o_TL = o; o += 128;                                 // dev-only stage stamps (64 x 8 bytes; written to LDS: a global store
                                                          // inside the loops would cost the production schedule waits)
    l.total = (o + 3) & ~3;
    return l;
}

template <int NTWM, int QD, int RB>
struct R4WCtx {
    static constexpr R4WGeom GM = make_r4w_geom(NTWM, QD);
    static constexpr int G = NTWM, RT = GM.RT, SG = GM.SG, SW = GM.SW;
    float4 ring[RT];
    const float4* sp;                                     // this wave's lane pointer at tile 0 of the current layer body

    // the tile at position P of the current body has been consumed: request the tile RT places down the stream into its
    // slot.  Positions past the body continue into the next one (contiguous in the image); padding tiles are skipped
    // where that is known at compile time (the body after the last forward body is a reverse body: both counts apply).
    // The tail of a body (everything behind its W x W product) does NOT request: the compiler drains vmcnt(0) at the
    // back edge of the layer loop (loop-carried registers of loads in flight), and a request issued just before it
    // costs a full memory latency there (~2.2 k cycles).  Those slots are refilled at the head of the next body
    // instead (`deferred`), so the youngest request at the back edge is ~2 k cycles old.
    // (the same number of slots after a forward and after a reverse body: ONE code path, no branch around loads - hipcc
    // equalises the counter at a join by waiting)
    static constexpr int DEF = (GM.TFP - GM.PW3) > (GM.TRP - GM.PW1T) ? (GM.TFP - GM.PW3) : (GM.TRP - GM.PW1T);
    static_assert(DEF <= RT, "deferred tail larger than the ring");
    static constexpr int defer_from(bool fwd) { return (fwd ? GM.TFP : GM.TRP) - DEF; }
    template <int P, bool FWD>
    __device__ __forceinline__ void refill() {
        constexpr int TP = FWD ? GM.TFP : GM.TRP, T = FWD ? GM.TF : GM.TR;
        constexpr int nx = P + RT;
        constexpr int TMAX = GM.TF > GM.TR ? GM.TF : GM.TR;
        constexpr bool real = nx < TP ? (nx < T) : ((nx - TP) < TMAX);
        if constexpr (real && P < defer_from(FWD)) ring[P % RT] = sp[(size_t)nx * 256];
    }
    // head of a body: the requests the previous body's tail left out = tiles [RT - DEF, RT) of THIS body (TP is a multiple
    // of RT, so slot = position); in the very first body they complete the initial fill
    __device__ __forceinline__ void deferred() {
        constexpr int TMAX = GM.TF > GM.TR ? GM.TF : GM.TR;
        static_for<RT - DEF, RT>([&](auto pc) {
            constexpr int nx = decltype(pc)::value;
            if constexpr (nx < TMAX) ring[nx] = sp[(size_t)nx * 256];
        });
    }
    template <int P>
    __device__ __forceinline__ const float4& tile() const { return ring[P % RT]; }
    // end of a layer body (the padding positions belong to the deferred tail)
    template <bool FWD>
    __device__ __forceinline__ void end_body() { sp += (size_t)(FWD ? GM.TFP : GM.TRP) * 256; }
};

__device__ __forceinline__ float4 r4w_mask_k(float4 a, int k0, int kmax) {
    if (k0 + 0 >= kmax) a.x = 0.f;
    if (k0 + 1 >= kmax) a.y = 0.f;
    if (k0 + 2 >= kmax) a.z = 0.f;
    if (k0 + 3 >= kmax) a.w = 0.f;
    return a;
}

// ---- D x D affine map, evaluated by every wave: out[r][lane] = ((P0 + P1) + (P2 + P3)) + bias, Pp = the product over
// k-quads [p QD / 4, (p + 1) QD / 4) (what wave p contributed in flow_r4.h).  `post(r, col, v)` maps the finished value.
template <int NTWM, int QD, int RB, int P0, bool FWD, class Post>
__device__ __forceinline__ void r4w_affine(R4WCtx<NTWM, QD, RB>& cx, const float* zin, float* zout, int ZL, int kmax,
                                           float bias, const Tid4& t, Post post, long long* tlb = nullptr) {
    constexpr int NQ = QD / 4;
    auto stamp = [&](int i) { if (tlb && t.tid == 0 && blockIdx.x == 0) tlb[i] = (long long)__builtin_amdgcn_s_memtime(); };
    f32x4 acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[rb][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<0, NQ>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        float4 a[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int k0 = 4 * (p * NQ + s);
                a[rb][p] = r4w_mask_k(*reinterpret_cast<const float4*>(zin + (4 * rb + t.arow) * ZL + k0), k0, kmax);
            }
        static_for<0, 4>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            static_for<0, 4>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                const float4& w = cx.template tile<P0 + p * NQ + s>();
                const float wv = kk == 0 ? w.x : (kk == 1 ? w.y : (kk == 2 ? w.z : w.w));
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float4& av = a[rb][p];
                    const float aa = kk == 0 ? av.x : (kk == 1 ? av.y : (kk == 2 ? av.z : av.w));
                    acc[rb][p] = mfma44(aa, wv, acc[rb][p]);
                }
            });
        });
        static_for<0, 4>([&](auto pc) { cx.template refill<P0 + decltype(pc)::value * NQ + s, FWD>(); });
        stamp(8 + s);
    });
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = ((acc[rb][0][i] + acc[rb][1][i]) + (acc[rb][2][i] + acc[rb][3][i])) + bias;
            zout[(4 * rb + i) * ZL + t.lane] = post(4 * rb + i, t.lane, v);
        }
    stamp(10);
}

// ---- product into the hidden width, this wave's SW columns only: K = 4 NQT k-quads (NQT / 4 per former wave),
// SG column groups.  EP 1: + bias, ReLU, sign bits appended to `m`; EP 2: (+ 0) multiplied by the sign bits in `m`.
// Bit b of group g2, row block rb, row i: (BIT0 + (g2 RB + rb) 4 + i).
template <int NTWM, int QD, int RB, int P0, bool FWD, int NQT, int EP, int BIT0>
__device__ __forceinline__ void r4w_into_hidden(R4WCtx<NTWM, QD, RB>& cx, const float* ain, int lda, int kmax,
                                                const float (&bias)[R4WCtx<NTWM, QD, RB>::SG], float* hs, int HL,
                                                unsigned& m, const Tid4& t) {
    using C = R4WCtx<NTWM, QD, RB>;
    constexpr int SG = C::SG, NQ = NQT / 4;
    f32x4 acc[RB][SG][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < SG; ++g)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[rb][g][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<0, NQ>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        float4 a[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int k0 = 4 * (p * NQ + s);
                a[rb][p] = r4w_mask_k(*reinterpret_cast<const float4*>(ain + (4 * rb + t.arow) * lda + k0), k0, kmax);
            }
        static_for<0, 4>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            static_for<0, 4>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                static_for<0, SG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const float4& w = cx.template tile<P0 + (p * NQ + s) * SG + g>();
                    const float wv = kk == 0 ? w.x : (kk == 1 ? w.y : (kk == 2 ? w.z : w.w));
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const float4& av = a[rb][p];
                        const float aa = kk == 0 ? av.x : (kk == 1 ? av.y : (kk == 2 ? av.z : av.w));
                        acc[rb][g][p] = mfma44(aa, wv, acc[rb][g][p]);
                    }
                });
            });
        });
        static_for<0, 4>([&](auto pc) {
            static_for<0, SG>([&](auto gc) {
                cx.template refill<P0 + (decltype(pc)::value * NQ + s) * SG + decltype(gc)::value, FWD>();
            });
        });
    });
#pragma unroll
    for (int g = 0; g < SG; ++g) {
        const bool valid = 64 * g + t.lane < C::SW;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = ((acc[rb][g][0][i] + acc[rb][g][1][i]) + (acc[rb][g][2][i] + acc[rb][g][3][i])) + bias[g];
                const int bit = BIT0 + (g * RB + rb) * 4 + i;
                if (EP == 1) { const bool pos = v > 0.f; m |= (pos ? 1u : 0u) << bit; v = pos ? v : 0.f; }
                if (EP == 2) v = ((m >> bit) & 1u) ? v : 0.f;
                if (valid) hs[(4 * rb + i) * HL + 64 * g + t.lane] = v;
            }
    }
}

// ---- W x W product, K-split: this wave's 4 G k-quads (its slice `hs`) times all G column groups; the partial product
// goes to PART[wave]; after the barrier the wave reduces its own SW columns (bias, ReLU / sign bits as above) into `hs`.
template <int NTWM, int QD, int RB, int P0, bool FWD, int EP, int BIT0>
__device__ __forceinline__ void r4w_wide(R4WCtx<NTWM, QD, RB>& cx, float* hs, int HL,
                                         const float (&bias)[R4WCtx<NTWM, QD, RB>::SG], float* part, int PN, unsigned& m,
                                         const Tid4& t) {
    using C = R4WCtx<NTWM, QD, RB>;
    constexpr int G = NTWM, SG = C::SG, RB4 = 4 * RB;
    f32x4 acc[RB][G];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[rb][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<0, 4 * G>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        float4 a[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const float4*>(hs + (4 * rb + t.arow) * HL + 4 * q);
        static_for<0, 4>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float aa = kk == 0 ? a[rb].x : (kk == 1 ? a[rb].y : (kk == 2 ? a[rb].z : a[rb].w));
                static_for<0, G>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const float4& w = cx.template tile<P0 + q * G + g>();
                    const float wv = kk == 0 ? w.x : (kk == 1 ? w.y : (kk == 2 ? w.z : w.w));
                    acc[rb][g] = mfma44(aa, wv, acc[rb][g]);
                });
            }
        });
        static_for<0, G>([&](auto gc) { cx.template refill<P0 + q * G + decltype(gc)::value, FWD>(); });
        __builtin_amdgcn_sched_barrier(0);
    });
    float* pw = part + (size_t)t.wave * RB4 * PN + t.lane;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[(4 * rb + i) * PN + 64 * g] = acc[rb][g][i];
    r4_barrier();
#pragma unroll
    for (int g = 0; g < SG; ++g) {
        const bool valid = 64 * g + t.lane < C::SW;
        const int col = valid ? C::SW * t.wave + 64 * g + t.lane : 0;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* p = part + (4 * rb + i) * PN + col;
                float v = ((p[0] + p[RB4 * PN]) + (p[2 * RB4 * PN] + p[3 * RB4 * PN])) + bias[g];
                const int bit = BIT0 + (g * RB + rb) * 4 + i;
                if (EP == 1) { const bool pos = v > 0.f; m |= (pos ? 1u : 0u) << bit; v = pos ? v : 0.f; }
                if (EP == 2) v = ((m >> bit) & 1u) ? v : 0.f;
                if (valid) hs[(4 * rb + i) * HL + 64 * g + t.lane] = v;
            }
    }
}

// ---- narrow product behind a W x W product (r4_dense_n16): K = this wave's slice, CW = 16 NT output columns, NSUB =
// 4 / NT k-quads side by side in a tile; the 4 NSUB partial products go to P3[p][row][CW], one barrier, and EVERY wave
// then reads what it needs (r4w_narrow_sum: the fixed pairwise tree of flow_r4.h).
template <int NTWM, int QD, int RB, int P0, bool FWD, int NT>
__device__ __forceinline__ void r4w_narrow(R4WCtx<NTWM, QD, RB>& cx, const float* hs, int HL, float* p3, const Tid4& t) {
    constexpr int NSUB = 4 / NT, CW = 16 * NT, NTILE = NTWM * NT, RB4 = 4 * RB;
    const int sblk = t.lane / CW, col = t.lane % CW;
    f32x4 acc0[RB], acc1[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { acc0[rb] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[rb] = acc0[rb]; }
    static_for<0, NTILE>([&](auto Tc) {
        constexpr int T = decltype(Tc)::value;
        const float4& w = cx.template tile<P0 + T>();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float4 a = *reinterpret_cast<const float4*>(hs + (4 * rb + t.arow) * HL + 4 * sblk + 4 * NSUB * T);
            acc0[rb] = mfma44(a.x, w.x, acc0[rb]);
            acc1[rb] = mfma44(a.y, w.y, acc1[rb]);
            acc0[rb] = mfma44(a.z, w.z, acc0[rb]);
            acc1[rb] = mfma44(a.w, w.w, acc1[rb]);
        }
        cx.template refill<P0 + T, FWD>();
    });
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        acc0[rb] += acc1[rb];
        float* pw = p3 + ((size_t)(t.wave * NSUB + sblk) * RB4 + 4 * rb) * CW + col;
#pragma unroll
        for (int i = 0; i < 4; ++i) pw[i * CW] = acc0[rb][i];
    }
    r4_barrier();
}

template <int RB, int NT>
__device__ __forceinline__ float r4w_narrow_sum(const float* p3, int row, int c) {
    constexpr int NSUB = 4 / NT, CW = 16 * NT, NP = 4 * NSUB, RB4 = 4 * RB;
    float v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = p3[((size_t)i * RB4 + row) * CW + c];
#pragma unroll
    for (int n = NP; n > 1; n >>= 1)
#pragma unroll
        for (int i = 0; i < n / 2; ++i) v[i] = v[2 * i] + v[2 * i + 1];
    return v[0];
}

struct R4WIdentity {
    __device__ __forceinline__ float operator()(int, int, float v) const { return v; }
};

// ------------------------------------------------------------------------------------------------
// log q(x) and d log q / dx for the 4 RB rows in X0 (columns >= D zero).  Every wave ends with the full gradient in
// its own state buffer (offset returned through *grad_off, leading dimension ZL); logq[rb] is the density of row
// 4 rb + lane / 16 (identical in every wave).
// ------------------------------------------------------------------------------------------------
#define R4W_TL(idx) do { if (TIMED && t.tid == 0 && blockIdx.x == 0) tlb[idx] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

template <int NTWM, int QD, int RB, bool TIMED = false>
__device__ void flow_log_prob_r4w(const FlowDims& f, const R4WLds& l, const float* __restrict__ packed, float* lds,
                                  const Tid4& t, float (&logq)[RB], int* grad_off) {
    using C = R4WCtx<NTWM, QD, RB>;
    constexpr R4WGeom GM = C::GM;
    constexpr int SG = C::SG, RT = C::RT, RB4 = 4 * RB;
    const int ZL = l.ZL, HL = l.HL, PN = l.PN;
    int cur = l.o_Z + (t.wave * 2 + 0) * RB4 * ZL, nxt = l.o_Z + (t.wave * 2 + 1) * RB4 * ZL;
    float* DP = lds + l.o_DP + t.wave * RB4 * ZL;
    float* HS = lds + l.o_HS + t.wave * RB4 * HL;
    float* PART = lds + l.o_PART;
    float* P3 = lds + l.o_P3;
    long long* tlb = reinterpret_cast<long long*>(lds + l.o_TL);
    if (TIMED && t.tid < 64) tlb[t.tid] = 0;
    const int er = t.lane >> 4, c = t.lane & 15;          // element-wise mapping inside a wave: row er of each row block, column c
    C cx;
    cx.sp = reinterpret_cast<const float4*>(packed + f.o_r4w) + (size_t)t.wave * 64 + t.lane;
#pragma unroll
    for (int i = 0; i < RT - C::DEF; ++i) cx.ring[i] = cx.sp[(size_t)i * 256];      // (the rest: `deferred` of the first body)
    // private copy of the input rows; the cotangent buffer starts at zero (its unused columns meet zero weights)
    for (int e = t.lane; e < RB4 * ZL; e += 64) { lds[cur + e] = lds[l.o_X0 + e]; DP[e] = 0.f; }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) logq[rb] = 0.f;
    float zero_b[SG];
#pragma unroll
    for (int g = 0; g < SG; ++g) zero_b[g] = 0.f;

    for (int layer = f.K - 1; layer >= 0; --layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * NTHREADS + t.tid;
        const bool tl = TIMED && layer == f.K - 2;
        if (tl) R4W_TL(0);
        if (tl) R4W_TL(6);
        cx.deferred();
        if (tl) R4W_TL(7);
        // the layer's biases arrive as the first two tiles of the body (k_pack_r4w): no load outside the stream, whose
        // result would return behind every tile already in flight
        float b1v[SG], b2v[SG];
        {
            const float4& t0 = cx.template tile<GM.PB>();
            const float4& t1 = cx.template tile<GM.PB + 1>();
            b1v[0] = t0.x; b2v[0] = t0.z;
            if constexpr (SG > 1) { b1v[1] = t0.y; b2v[1] = t0.w; }
            static_assert(SG <= 2, "bias tile holds two column groups");
        }
        const float bA = cx.template tile<GM.PB + 1>().x, b3s = cx.template tile<GM.PB + 1>().y,
                    b3c = cx.template tile<GM.PB + 1>().z;
        cx.template refill<GM.PB, true>();
        cx.template refill<GM.PB + 1, true>();
        // InvertibleAffine.inverse (+ folded ActNorm): z <- z @ W' + ac
        r4w_affine<NTWM, QD, RB, GM.PA, true>(cx, lds + cur, lds + nxt, ZL, f.D, bA, t, R4WIdentity(), tl ? tlb : nullptr);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) logq[rb] += Lp[f.o_logS];
        float* Z = lds + nxt;
        if (tl) R4W_TL(1);
        unsigned m = 0u;
        // conditioner: first layer (this wave's hidden columns), W x W, coupling parameters
        r4w_into_hidden<NTWM, QD, RB, GM.PW1, true, 4, 1, 0>(cx, Z, ZL, f.d, b1v, HS, HL, m, t);
        if (tl) R4W_TL(2);
        r4w_wide<NTWM, QD, RB, GM.PW2, true, 1, SG * RB * 4>(cx, HS, HL, b2v, PART, PN, m, t);
        *mk = m;
        if (tl) R4W_TL(3);
        r4w_narrow<NTWM, QD, RB, GM.PW3, true, 2>(cx, HS, HL, P3, t);
        if (tl) R4W_TL(4);
        // AffineCoupling.inverse: z2 <- (z2 - shift) exp(-s), log_det = -sum(s)      (every wave, its own copy)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int row = 4 * rb + er;
            float ssum = 0.f;
            if (c < f.DO) {
                const float shift = r4w_narrow_sum<RB, 2>(P3, row, c) + b3s;
                const float s = r4w_narrow_sum<RB, 2>(P3, row, f.DOp + c) + b3c;
                const float es = expf(-s);
                const float v2 = (Z[row * ZL + f.d + c] - shift) * es;
                Z[row * ZL + f.d + c] = v2;
                lds[l.o_ES + ((size_t)layer * RB4 + row) * f.DOp + c] = es;
                lds[l.o_V2 + ((size_t)layer * RB4 + row) * f.DOp + c] = v2;
                ssum += s;
            }
            logq[rb] += -row16_sum(ssum);
        }
        if (tl) R4W_TL(5);
        cx.template end_body<true>();
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    // DiagGaussian.log_prob, and the seed of the reverse sweep
    {
        const float* base = packed + f.o_base;
        float* Zc = lds + cur;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int row = 4 * rb + er;
            float bsum = 0.f;
            for (int j = c; j < f.D; j += 16) {
                const float ls = base[f.Dp + j];
                const float sc = expf(ls);
                const float zn = (Zc[row * ZL + j] - base[j]) / sc;
                bsum += ls + 0.5f * (zn * zn);
                Zc[row * ZL + j] = -(zn / sc);
            }
            logq[rb] += -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
        }
    }
    // reverse sweep: g = d log q / d(state), layers 0 .. K-1
    for (int layer = 0; layer < f.K; ++layer) {
        cx.deferred();
        unsigned m = reinterpret_cast<const unsigned*>(lds + l.o_MASK)[(size_t)layer * NTHREADS + t.tid];
        float* Gs = lds + cur;
        const bool tl = TIMED && layer == 1;
        if (tl) R4W_TL(16);
        if (layer == 0) {                                 // (layers > 0: formed by the previous layer's D x D epilogue, `couple`)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = 4 * rb + er;
                if (c < f.DO) {
                    const float g2 = Gs[row * ZL + f.d + c];
                    const float es = lds[l.o_ES + (size_t)row * f.DOp + c];
                    const float v2 = lds[l.o_V2 + (size_t)row * f.DOp + c];
                    DP[row * ZL + c] = -(g2 * es);
                    DP[row * ZL + f.DOp + c] = -(g2 * v2) - 1.f;
                    Gs[row * ZL + f.d + c] = g2 * es;
                }
            }
        }
        if (tl) R4W_TL(17);
        // (shift | scale) -> hidden (this wave's columns, masked by the second ReLU), W x W transposed, hidden -> d
        r4w_into_hidden<NTWM, QD, RB, GM.PW3T, false, 8, 2, SG * RB * 4>(cx, DP, ZL, 2 * f.DOp, zero_b, HS, HL, m, t);
        if (tl) R4W_TL(18);
        r4w_wide<NTWM, QD, RB, GM.PW2T, false, 2, 0>(cx, HS, HL, zero_b, PART, PN, m, t);
        if (tl) R4W_TL(19);
        r4w_narrow<NTWM, QD, RB, GM.PW1T, false, 1>(cx, HS, HL, P3, t);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {                 // g[:, :d] += (conditioner input gradient)
            const int row = 4 * rb + er;
            Gs[row * ZL + c] += r4w_narrow_sum<RB, 1>(P3, row, c);
        }
        if (tl) R4W_TL(20);
        // cotangents of the NEXT layer's coupling parameters, formed where its input gradient is produced
        auto couple = [&](int r, int col, float v) -> float {
            if (layer + 1 < f.K && col >= f.d && col < f.d + f.DO) {
                const int j = col - f.d;
                const float es = lds[l.o_ES + ((size_t)(layer + 1) * RB4 + r) * f.DOp + j];
                const float v2 = lds[l.o_V2 + ((size_t)(layer + 1) * RB4 + r) * f.DOp + j];
                DP[r * ZL + j] = -(v * es);
                DP[r * ZL + f.DOp + j] = -(v * v2) - 1.f;
                return v * es;
            }
            return v;
        };
        r4w_affine<NTWM, QD, RB, GM.PAT, false>(cx, Gs, lds + nxt, ZL, f.D, 0.f, t, couple);
        if (tl) R4W_TL(22);
        cx.template end_body<false>();
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    if (TIMED && f.timeline && blockIdx.x == 0 && t.tid < 64) f.timeline[t.tid] = tlb[t.tid];
    *grad_off = cur;
}

}  // namespace fab
