// 4-chain tiles, FUSED stages (round 5): the same flow density + d/dx (and the sampling direction) as flow_r4.h's stream
// variant in SIX stages per layer pair instead of ten.
//
// What is fused (the stage stamps of profiles/r2/hmc_r4_stage_timeline.txt price a stage of the 4-chain
// kernel at >= 1 k cycles whatever it multiplies - two LDS-only barriers, the partial sums' round trip, the epilogue):
//  * InvertibleAffine + first conditioner Linear.  Density direction, layer k: z = y A_k + ac, h1 = relu(z[:, :d] W1^T + b1)
//    = relu(y (A_k[:, :d] W1^T) + (ac[:d] W1^T + b1)).  W1' = A_k[:, :d] W1^T (D x W) and b1' are formed at pack time in float64
//    (k_pack_r4f), and ONE stage multiplies y by [W1' | A_k]: the affine map is off the critical path.  Reverse sweep:
//    g_y = dh1 W1'^T + g_z A_k^T is ONE K = W + D product instead of W1^T, an add and A^T.
//  * the coupling transform runs in the epilogue of the W3 product (the thread that sums the partial shift / scale of a
//    coordinate transforms it), the coupling cotangents of the next layer in the epilogue of the K = W + D product (as before).
//  Per layer and direction: S1 [y -> h1, z] (K = 32), S2 [W x W], S3 [W -> shift | scale, coupling]; reverse: S4 [(shift | scale)
//  cotangents -> W] (K = 32), S5 [W x W], S6 [W + D -> D].  Sampling direction: the affine map of layer v - 1 is fused with
//  the first Linear of layer v the same way (virtual layers 0 .. K: layer 0 starts from the identity, layer K is the last
//  affine map alone).
//
// Arithmetic: the stage sums are the 4-chain tiles' (K split over the 4 waves, partial sums added in a fixed order); h1 differs
// from flow_r4.h's by the rounding of W1' (one fp32 rounding of a float64 product instead of two fp32 GEMM stages) - results
// agree to ~1e-6 relative, not bit for bit (tests/test_gpu_hmc_shapes.py pins the two against each other and the oracle).
//
// Weight stream (k_pack_r4f; FlowDims::o_r4f): per layer and direction TL = (4 G + 4) G + 1 tiles of 1 KiB per wave in
// consumption order, the same ITEM pattern in every section so that the ring's slots are compile-time constants across layer,
// direction and evaluation boundaries:
//    item 0, 1   G tiles each   S1: the two k-quads of W1' this wave multiplies      | S4: the two k-quads of W3^T
//    item 2      1 tile         S1: dense tile of A (2 k-quads x 32 columns)         | S6: dense tile of A^T (parked in registers over S5)
//    item 3 ..   G tiles each   S2: 4 G k-quads of W2                                | S5: 4 G k-quads of W2^T
//    last two    G tiles each   S3: 2 G dense tiles of W3 (2 k-quads x 32 columns)   | S6: 2 G dense tiles of W1'^T
// float4 index of (layer slot s, item I, wave w, tile g, lane): ((s TL + toff(I)) 4 + w ntiles(I) + g) 64 + lane.
// Biases / log-dets of all layers live in LDS for the whole kernel (r4f_load_bias: one copy per launch, no bias loads in the
// layer bodies).
#pragma once
#include "flow_r4.h"

namespace fab {

template <int NTWM>
struct R4F {
    static constexpr int G = NTWM;
    static constexpr int CR = 4 * G + 5;                   // items with tiles per layer and direction
#ifdef FAB_R4F_RD
    static constexpr int RD = FAB_R4F_RD;                  // (timing experiments: tools/experiments/price)
#else
    static constexpr int RD = CR % 5 == 0 ? 5 : (CR % 7 == 0 ? 7 : (CR % 6 == 0 ? 6 : 7));
#endif
    static constexpr int C = (CR + RD - 1) / RD * RD;      // padded with empty items: the ring size divides the items per layer
    static constexpr int TL = (4 * G + 4) * G + 1;         // tiles per wave, layer and direction
    static constexpr int I_A = 2, I_W = 3, I_N = 4 * G + 3;
    static constexpr int ntiles(int I) { return I == I_A ? 1 : (I < CR ? G : 0); }
    static constexpr int toff(int I) { return I <= I_A ? I * G : (I - 1) * G + 1; }
};

// bias block of one layer in LDS / in the image (r4f_bias_stride floats, fabhip_common.h):
// b1' [Wp] | b2 [Wp] | ac [32] | b3 shift [16] | b3 scale [16] | log-det [16]

// eight independent accumulation chains for the dense narrow products (v_mfma_f32_4x4x1 has a dependent latency of ~54 cycles
// against 8 cycles of issue: two chains, as r4_dense_n16 keeps, run at a quarter of the issue rate)
struct R4FAcc {
    f32x4 a[8];
    __device__ __forceinline__ R4FAcc() {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    template <int T>
    __device__ __forceinline__ void tile(const float4& x, const float4& w) {
        constexpr int o = 4 * (T & 1);
        a[o + 0] = mfma44(x.x, w.x, a[o + 0]);
        a[o + 1] = mfma44(x.y, w.y, a[o + 1]);
        a[o + 2] = mfma44(x.z, w.z, a[o + 2]);
        a[o + 3] = mfma44(x.w, w.w, a[o + 3]);
    }
    // this lane's partial (wave, k sub-block) -> PZ[(wave 2 + sblk)][row][32]
    __device__ __forceinline__ void store(float* __restrict__ pz, const Tid4& t) const {
        const f32x4 s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        float* p = pz + ((size_t)(t.wave * 2 + (t.lane >> 5)) * R4) * 32 + (t.lane & 31);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r * 32] = s[r];
    }
};

// sum of the 8 partials of output (row, c) of a dense narrow product, fixed tree
__device__ __forceinline__ float r4f_tree8(const float* __restrict__ pz, int row, int c) {
    const float* p = pz + row * 32 + c;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[(size_t)i * R4 * 32];
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

template <int G>
__device__ __forceinline__ void r4f_bias(float (&bv)[G], const float* __restrict__ b, const Tid4& t) {
    constexpr int N = 64 * G;
#pragma unroll
    for (int i = 0; i < G; ++i) bv[i] = b[(256 * i + t.tid) % N];
}

// the bias blocks of `nblk` layers: image -> LDS (all threads; the caller synchronises)
__device__ __forceinline__ void r4f_load_bias(const float* __restrict__ src, float* __restrict__ dst, int nfloats, int tid) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int e = tid; e < nfloats / 4; e += NTHREADS) d4[e] = s4[e];
}

// ring of RD items over the fused stream; `sp` points at (layer slot, lane) of the section being consumed
template <int NTWM>
struct R4FRing {
    using S = R4F<NTWM>;
    static constexpr int G = S::G, RD = S::RD, C = S::C;
    float4 r[RD][G];
    const float4* sp;
    int wG, w1;
    __device__ __forceinline__ R4FRing(const float4* base, const Tid4& t) : sp(base + t.lane), wG(t.wave * G * 64), w1(t.wave * 64) {
        static_for<0, RD>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            load<I, 0>(IC<I % RD>{});
        });
    }
    // item J of the layer `LOFF` slots ahead -> ring slot SL
    template <int J, int LOFF, int SL>
    __device__ __forceinline__ void load(IC<SL>) {
        constexpr int n = S::ntiles(J);
        constexpr long off = ((long)LOFF * S::TL + S::toff(J)) * 4 * 64;
#pragma unroll
        for (int g = 0; g < n; ++g) r[SL][g] = sp[off + (n == G ? wG : w1) + g * 64];
    }
    // item I of the current layer has been consumed: request the item RD places further down the stream into its slot
    template <int I>
    __device__ __forceinline__ void refill(IC<I>) {
        constexpr int J = I + RD;
        if constexpr (J < C) load<J, 0>(IC<I % RD>{});
        else load<J - C, 1>(IC<I % RD>{});
    }
    // the empty items that pad a layer to a multiple of the ring size
    __device__ __forceinline__ void skip_pad() {
        static_for<S::CR, C>([&](auto ic) { refill(ic); });
    }
    __device__ __forceinline__ void next_layer() { sp += (size_t)S::TL * 4 * 64; }
};

// S1 / S4: OUT[4][64 G] = epilogue(ACT[4][32] @ B) with the two k-quads of this wave in items I0, I0 + 1; separate accumulators
// per k-quad (10 chains of 4 dependent MFMAs instead of 5 chains of 8).  Leaves the partials in PART; the caller adds its own
// products, the barrier and the epilogue.
template <int NTWM, int I0, class Ring>
__device__ __forceinline__ void r4f_short_mma(const float4& a0, const float4& a1, Ring& ring, float* part, int PN, const Tid4& t) {
    constexpr int G = NTWM, RD = Ring::RD;
    f32x4 acc0[G], acc1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { acc0[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[g] = acc0[g]; }
    r4_quad<G>(a0, ring.r[I0 % RD], acc0);
    ring.refill(IC<I0>{});
    __builtin_amdgcn_sched_barrier(0);
    r4_quad<G>(a1, ring.r[(I0 + 1) % RD], acc1);
    ring.refill(IC<I0 + 1>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc0[g] += acc1[g];
    r4_store_part<G>(acc0, part, PN, t);
}

// S2 / S5: the W x W stage on the ring, quad q = item I0 + q
template <int NTWM, int EP, class Ring>
__device__ __forceinline__ void r4f_dense_wide(const float* act, int lda, Ring& ring, const float* __restrict__ bias, float* out,
                                               int ldo, unsigned* mask, float* part, int PN, const Tid4& t) {
    using S = R4F<NTWM>;
    constexpr int G = NTWM, NQ = 4 * NTWM, RD = S::RD, I0 = S::I_W;
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow = act + t.arow * lda + 16 * NTWM * t.wave;
    static_for<0, NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 a = *reinterpret_cast<const float4*>(arow + 4 * q);
        r4_quad<G>(a, ring.r[(I0 + q) % RD], acc);
        ring.refill(IC<I0 + q>{});
        __builtin_amdgcn_sched_barrier(0);
    });
    r4_store_part<G>(acc, part, PN, t);
    float bv[G];
    if (bias) r4f_bias<G>(bv, bias, t);
    else {
#pragma unroll
        for (int g = 0; g < G; ++g) bv[g] = 0.f;
    }
    r4_barrier();
    r4_epilogue<G, EP>(part, PN, bv, out, ldo, mask, t);
    r4_barrier();
}

// the 2 G dense tiles of items I_N, I_N + 1 against ACT[4][Wp] (this wave's K range: quads 4 G w + 2 T + sblk)
template <int NTWM, class Ring>
__device__ __forceinline__ void r4f_narrow_mma(const float* act, int lda, Ring& ring, R4FAcc& p, const Tid4& t) {
    using S = R4F<NTWM>;
    constexpr int G = NTWM, RD = S::RD, I0 = S::I_N;
    const float* arow = act + t.arow * lda + 16 * NTWM * t.wave + 4 * (t.lane >> 5);
    static_for<0, 2 * G>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        const float4 a = *reinterpret_cast<const float4*>(arow + 8 * T);
        p.template tile<T>(a, ring.r[(I0 + T / G) % RD][T % G]);
        if constexpr (T % G == G - 1) {
            ring.refill(IC<I0 + T / G>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}

// ------------------------------------------------------------------------------------------------
// log q(x) and d log q / dx for the 4 rows in X0 (columns >= D zero).  The gradient is left in X0 (*grad_off = l.o_X0, leading
// dimension R4_DS); returns log q of row `tid >> 4` on wave 0.  The density bias table must be in LDS at l.o_BIAS.
// ------------------------------------------------------------------------------------------------
template <int NTWM>
__device__ float flow_log_prob_r4f(const FlowDims& f, const R4Lds& l, const float* __restrict__ packed, float* lds,
                                   const Tid4& t, int* grad_off) {
    using S = R4F<NTWM>;
    constexpr int G = NTWM, RD = S::RD;
    float* X = lds + l.o_X0;
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
    float* DP = lds + l.o_DP;
    float* PART = lds + l.o_PART;
    float* PZ = lds + l.o_PZ;
    const float* BT = lds + l.o_BIAS;
    const int BS = r4f_bias_stride(f.Wp);
    const bool ew = t.tid < 64;
    const int row = t.tid >> 4, c = t.tid & 15;            // element-wise mapping of wave 0 (coupling, base distribution)
    const int zrow = t.tid >> 5, zc = t.tid & 31;          // (row, column) of the 4 x 32 outputs of a dense narrow product
    const int sblk = t.lane >> 5;
    R4FRing<NTWM> ring(reinterpret_cast<const float4*>(packed + f.o_r4f), t);
    float logq = 0.f;
    for (int layer = f.K - 1; layer >= 0; --layer) {
        const float* bt = BT + (size_t)layer * BS;
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        const bool tl = layer == f.K - 2;
        if (tl) FAB_TL(f, 0);
        {   // S1: h1 = relu(y W1' + b1'), z = y A + ac   (K = 32: k in [8 w, 8 w + 8) on this wave)
            const float* xr = X + t.arow * R4_DS + 8 * t.wave;
            const float4 a0 = *reinterpret_cast<const float4*>(xr), a1 = *reinterpret_cast<const float4*>(xr + 4);
            r4f_short_mma<NTWM, 0>(a0, a1, ring, PART, l.PN, t);
            R4FAcc z;
            z.template tile<0>(sblk ? a1 : a0, ring.r[S::I_A % RD][0]);
            ring.refill(IC<S::I_A>{});
            z.store(PZ, t);
            float bv[G];
            r4f_bias<G>(bv, bt, t);
            r4_barrier();
            float zv = 0.f;                                // (read before the wide epilogue's writes: one LDS round trip less)
            if (t.tid < 128) zv = r4f_tree8(PZ, zrow, zc) + bt[2 * f.Wp + zc];
            r4_epilogue<G, 1>(PART, l.PN, bv, HA, l.WS, mk, t);
            if (t.tid < 128) X[zrow * R4_DS + zc] = zv;
            r4_barrier();
        }
        logq += bt[2 * f.Wp + 64];
        if (tl) FAB_TL(f, 1);
        r4f_dense_wide<NTWM, 1>(HA, l.WS, ring, bt + f.Wp, HB, l.WS, mk + NTHREADS, PART, l.PN, t);
        if (tl) FAB_TL(f, 3);
        {   // S3: (shift | s) = h2 W3 + b3, then AffineCoupling.inverse: z2 <- (z2 - shift) exp(-s), log_det = -sum(s)
            R4FAcc p;
            r4f_narrow_mma<NTWM>(HB, l.WS, ring, p, t);
            p.store(PZ, t);
            ring.skip_pad();
            r4_barrier();
            if (ew) {
                float ssum = 0.f;
                if (c < f.DO) {
                    const float shift = r4f_tree8(PZ, row, c) + bt[2 * f.Wp + 32 + c];
                    const float s = r4f_tree8(PZ, row, 16 + c) + bt[2 * f.Wp + 48 + c];
                    const float es = expf(-s);
                    const float v2 = (X[row * R4_DS + f.d + c] - shift) * es;
                    X[row * R4_DS + f.d + c] = v2;
                    lds[l.o_ES + ((size_t)layer * R4 + row) * f.DOp + c] = es;
                    lds[l.o_V2 + ((size_t)layer * R4 + row) * f.DOp + c] = v2;
                    ssum = s;
                }
                logq += -row16_sum(ssum);
            }
            r4_barrier();
        }
        if (tl) FAB_TL(f, 5);
        ring.next_layer();
    }
    // DiagGaussian.log_prob, and the seed of the reverse sweep (with the coupling cotangents of layer 0)
    if (ew) {
        const float* base = packed + f.o_base;
        float bsum = 0.f;
        for (int j = c; j < f.D; j += 16) {
            const float ls = base[f.Dp + j];
            const float sc = expf(ls);
            const float zn = (X[row * R4_DS + j] - base[j]) / sc;
            bsum += ls + 0.5f * (zn * zn);
            float g = -(zn / sc);
            if (j >= f.d) {                               // (D <= 32: j - d < DO <= 16)
                const int jj = j - f.d;
                const float es = lds[l.o_ES + (size_t)row * f.DOp + jj];
                const float v2 = lds[l.o_V2 + (size_t)row * f.DOp + jj];
                DP[row * R4_DS + jj] = -(g * es);
                DP[row * R4_DS + f.DOp + jj] = -(g * v2) - 1.f;
                g = g * es;
            }
            X[row * R4_DS + j] = g;
        }
        logq += -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
    }
    r4_barrier();
    // reverse sweep: g = d log q / d(state), layers 0 .. K-1
    for (int layer = 0; layer < f.K; ++layer) {
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        const bool tl = layer == 1;
        if (tl) FAB_TL(f, 16);
        float4 atile;
        {   // S4: (shift | scale) cotangents -> hidden (K = 32)
            const float* dr = DP + t.arow * R4_DS + 8 * t.wave;
            const float4 a0 = *reinterpret_cast<const float4*>(dr), a1 = *reinterpret_cast<const float4*>(dr + 4);
            r4f_short_mma<NTWM, 0>(a0, a1, ring, PART, l.PN, t);
            atile = ring.r[S::I_A % RD][0];               // A^T: used by S6 (stays in registers over S5)
            ring.refill(IC<S::I_A>{});
            float bv[G];
#pragma unroll
            for (int g = 0; g < G; ++g) bv[g] = 0.f;
            r4_barrier();
            r4_epilogue<G, 2>(PART, l.PN, bv, HA, l.WS, mk + NTHREADS, t);
            r4_barrier();
        }
        if (tl) FAB_TL(f, 18);
        r4f_dense_wide<NTWM, 2>(HA, l.WS, ring, nullptr, HB, l.WS, mk, PART, l.PN, t);
        if (tl) FAB_TL(f, 19);
        {   // S6: g_y = dh1 W1'^T + g_z A^T, then the coupling cotangents of layer + 1
            R4FAcc p;
            r4f_narrow_mma<NTWM>(HB, l.WS, ring, p, t);
            const float4 ag = *reinterpret_cast<const float4*>(X + t.arow * R4_DS + 8 * t.wave + 4 * sblk);
            p.template tile<0>(ag, atile);
            p.store(PZ, t);
            ring.skip_pad();
            r4_barrier();
            if (t.tid < 128) {
                float v = r4f_tree8(PZ, zrow, zc);
                if (layer + 1 < f.K && zc >= f.d && zc < f.d + f.DO) {
                    const int j = zc - f.d;
                    const float es = lds[l.o_ES + ((size_t)(layer + 1) * R4 + zrow) * f.DOp + j];
                    const float v2 = lds[l.o_V2 + ((size_t)(layer + 1) * R4 + zrow) * f.DOp + j];
                    DP[zrow * R4_DS + j] = -(v * es);
                    DP[zrow * R4_DS + f.DOp + j] = -(v * v2) - 1.f;
                    v = v * es;
                }
                X[zrow * R4_DS + zc] = v;
            }
            r4_barrier();
        }
        if (tl) FAB_TL(f, 22);
        ring.next_layer();
    }
    *grad_off = l.o_X0;
    return logq;
}

// ------------------------------------------------------------------------------------------------
// x, log q = flow.sample(eps) for the 4 rows whose base noise is in X0 (columns >= D zero); x is left in X0 (*x_off = l.o_X0).
// The SAMPLING bias table (K + 1 blocks) must be in LDS at l.o_BIAS.  Section 2 of the stream: virtual layers 0 .. K.
// ------------------------------------------------------------------------------------------------
template <int NTWM>
__device__ float flow_sample_r4f(const FlowDims& f, const R4Lds& l, const float* __restrict__ packed, float* lds, const Tid4& t,
                                 int* x_off) {
    using S = R4F<NTWM>;
    constexpr int G = NTWM, RD = S::RD;
    float* X = lds + l.o_X0;
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
    float* PART = lds + l.o_PART;
    float* PZ = lds + l.o_PZ;
    const float* BT = lds + l.o_BIAS;
    const int BS = r4f_bias_stride(f.Wp);
    const bool ew = t.tid < 64;
    const int row = t.tid >> 4, c = t.tid & 15;
    const int zrow = t.tid >> 5, zc = t.tid & 31;
    const int sblk = t.lane >> 5;
    unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK);                 // (ReLU signs: scratch here)
    R4FRing<NTWM> ring(reinterpret_cast<const float4*>(packed + f.o_r4f) + (size_t)(2 * f.K + 1) * S::TL * 4 * 64, t);
    float logq = 0.f;
    if (ew) {                                          // z = loc + exp(log_scale) eps ; log N(eps)
        const float* base = packed + f.o_base;
        float bsum = 0.f;
        for (int j = c; j < f.D; j += 16) {
            const float e = X[row * R4_DS + j];
            const float ls = base[f.Dp + j];
            X[row * R4_DS + j] = base[j] + expf(ls) * e;
            bsum += ls + 0.5f * (e * e);
        }
        logq = -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
    }
    r4_barrier();
    for (int v = 0; v <= f.K; ++v) {
        const float* bt = BT + (size_t)v * BS;
        {   // S1: h1 = relu(z W1'' + b1''), y = z W'^-1(layer v - 1) + at   (v = 0: y = z; v = K: the last affine map alone)
            const float* xr = X + t.arow * R4_DS + 8 * t.wave;
            const float4 a0 = *reinterpret_cast<const float4*>(xr), a1 = *reinterpret_cast<const float4*>(xr + 4);
            r4f_short_mma<NTWM, 0>(a0, a1, ring, PART, l.PN, t);
            R4FAcc z;
            z.template tile<0>(sblk ? a1 : a0, ring.r[S::I_A % RD][0]);
            ring.refill(IC<S::I_A>{});
            z.store(PZ, t);
            float bv[G];
            r4f_bias<G>(bv, bt, t);
            r4_barrier();
            float zv = 0.f;                                // (read before the wide epilogue's writes: one LDS round trip less)
            if (t.tid < 128) zv = r4f_tree8(PZ, zrow, zc) + bt[2 * f.Wp + zc];
            r4_epilogue<G, 1>(PART, l.PN, bv, HA, l.WS, mk, t);
            if (t.tid < 128) X[zrow * R4_DS + zc] = zv;
            r4_barrier();
        }
        logq -= -bt[2 * f.Wp + 64];
        if (v == f.K) break;
        r4f_dense_wide<NTWM, 1>(HA, l.WS, ring, bt + f.Wp, HB, l.WS, mk + NTHREADS, PART, l.PN, t);
        {   // S3 + AffineCoupling.forward: z2 <- z2 exp(s) + shift, log_det = sum(s)
            R4FAcc p;
            r4f_narrow_mma<NTWM>(HB, l.WS, ring, p, t);
            p.store(PZ, t);
            ring.skip_pad();
            r4_barrier();
            if (ew) {
                float ssum = 0.f;
                if (c < f.DO) {
                    const float shift = r4f_tree8(PZ, row, c) + bt[2 * f.Wp + 32 + c];
                    const float s = r4f_tree8(PZ, row, 16 + c) + bt[2 * f.Wp + 48 + c];
                    X[row * R4_DS + f.d + c] = X[row * R4_DS + f.d + c] * expf(s) + shift;
                    ssum = s;
                }
                logq -= row16_sum(ssum);
            }
            r4_barrier();
        }
        ring.next_layer();
    }
    *x_off = l.o_X0;
    return logq;
}

// host side: the fused-stage variant is chosen where its image exists, FABHIP_OPT_R4_STREAM >= 2 (default) and the bias blocks of
// all layers fit the CU's LDS next to the tile state (the transition kernel adds 4 x [4][D] floats of HMC state)
int option(int key);                                   // (launch.h)
static inline bool use_r4_fused(const FlowDims& f) {
    if (f.o_r4f < 0 || f.NTW / 4 < 2 || f.NTW / 4 > 5 || option(FABHIP_OPT_R4_STREAM) < 2) return false;
    const R4Lds l = make_r4_lds(f, true);
    return (size_t)(l.total + 4 * R4 * f.D + 4) * 4 <= 160 * 1024;
}

}  // namespace fab
