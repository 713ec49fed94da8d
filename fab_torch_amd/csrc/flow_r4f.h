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
#include "stream_r8.h"

namespace fab {

// FAST (fast mode, never the parity path): the W x W items hold bf16 tiles - 2 k-quads per 1-KiB tile (lane = column, 16 bytes =
// 8 consecutive k), so a wave's W x W stage is 2 G items instead of 4 G and streams half the bytes; everything else as fp32.
// NS > 0 (round 5, "stash"): the LAST NS items of a W x W stage are not ring items - they are copied into LDS by
// global_load_lds_dwordx4 (no register, one instruction per 1-KiB tile and wave) while the SHORT stages in front of that W x W
// stage run, and read back with ds_read_b128 when their turn comes.  Why: the W x W stages run at the rate the L2 -> CU path
// delivers; in the short stages the same path idles (five items consumed in ~3.9 k cycles: 42 % of what it could carry), because
// the ring can only hold RD - 1 items ahead.  The CU's free LDS is the one place where more of the stream can wait: NS G KiB per
// wave.  Same tiles, same MFMA order: bit-identical results.  MEASURED (profiles/r5/hmc_r4f_stash_timeline.txt, NS = 3): the W x W
// stages of a layer pair lose 1.9 k cycles, the short stages gain 1.7 k - ~55 cycles per copied tile and wave wherever the copy
// is issued: a 1-KiB vector-memory instruction occupies the CU's address path for 16 cycles, the four waves reach their request
// sites together (a barrier apart at most) and an in-order wave cannot do anything else while it waits its turn.  Level with the
// ring alone, so it is NOT the default (FABHIP_OPT_R4_STREAM = 3 selects it).  One stash PIECE (= one item, G tiles) is requested behind the
// ring's own requests while items I_N, I_N + 1 (S3 / S6: for the NEXT layer slot), 0, 1, 2 (S1 / S4: this slot) are consumed -
// the first NS of those; loads return in order, so the hand-counted waits simply count the pieces' loads as well.
template <int NTWM, bool FAST = false, int NS_ = 0>
struct R4F {
    static constexpr int G = NTWM;
    static constexpr int NS = NS_;
    static constexpr int NQW = FAST ? 2 * G : 4 * G;       // items of a W x W stage
    static constexpr int CR = NQW + 5;                     // items with tiles per layer and direction
    static constexpr int TL = (NQW + 4) * G + 1;           // tiles per wave, layer and direction
    static constexpr int I_A = 2, I_W = 3, I_N = NQW + 3;
    // stash item k = k-quad SQ0 + k SQD of the W x W stage: spread over its front - a stash item is consumed ~3 x faster than a
    // ring item, and the RD - 1 items behind one lose that much lead time; at the stage's END that starved S3 / S6 and S1 / S4 of
    // their tiles (measured: short stages + 2.3 k cycles per layer pair, more than the W x W stages gained)
    static constexpr int SQ0 = 1, SQD = NS > 0 ? (NQW - 5) / NS : 1;
    static constexpr int stash_k(int I) { return (I - I_W - SQ0) / SQD; }
    static constexpr bool is_stash(int I) {
        return NS > 0 && I >= I_W + SQ0 && I < I_N && (I - I_W - SQ0) % SQD == 0 && stash_k(I) < NS;
    }
    static constexpr int stash_item(int k) { return I_W + SQ0 + k * SQD; }
    static constexpr int ntiles(int I) { return I == I_A ? 1 : (I < CR && !is_stash(I) ? G : 0); }   // tiles the RING holds of item I
    // the stash piece requested while item I is consumed (-1: none), and whether it belongs to the next layer slot
    static constexpr int dma_piece(int I) {
        const int k = I >= I_N ? I - I_N : (I <= I_A ? I + 2 : -1);
        return k >= 0 && k < NS ? k : -1;
    }
    static constexpr bool dma_next_slot(int I) { return I >= I_N; }
    static constexpr int toff(int I) { return I <= I_A ? I * G : (I - 1) * G + 1; }
    // The ring (R4FRing) holds RD items in ACCUMULATION registers; its slots are compile-time constants when RD divides the
    // items per layer, so a layer is padded to C = a multiple of RD with E EMPTY items (no tiles, no request).  They sit inside
    // the W x W stage, one behind every STEP-th k-quad, where the extra request of the item behind them hides behind MFMAs
    // (round-5 measurement: all of them at the layer's end = one burst of requests = +2 k cycles in S3 / S6).
#ifdef FAB_R4F_RD
    static constexpr int RD = FAB_R4F_RD;                  // (timing experiments: tools/experiments/price)
#else
    // G = 5: RD = 5 (C = 25), G = 4: RD = 7 (C = 21), G = 2: RD = 13 (C = 13): no empty items in the shipped shapes.  Measured at
    // G = 5 (tools/experiments/price): RD = 9 (C = 27, two empty items) is 2 % SLOWER than RD = 5 - the W x W stages run at the
    // rate the L2 -> CU path delivers (~46 B/clk per CU with 256 CUs streaming) whatever is in flight, and a deeper queue only
    // delays the items of the short stages behind it
    static constexpr int RD = G >= 5 ? 5 : (G == 4 ? 7 : (FAST ? 9 : 13));   // (FAST: CR = 15 / 13 / 9 -> C = 15 / 14 / 9)
#endif
    static constexpr int C = (CR + RD - 1) / RD * RD;
    static constexpr int E = C - CR;
    static constexpr int STEP = NQW / (E + 1) > 0 ? NQW / (E + 1) : 1;
    // virtual position of item I in the padded sequence / the item at virtual position V (-1: empty)
    static constexpr int vidx(int I) {
        int q = I - I_W;
        q = q < 0 ? 0 : (q > NQW ? NQW : q);
        const int e = q / STEP;
        return I + (e < E ? e : E);
    }
    static constexpr int item_at(int V) {
        for (int I = 0; I < CR; ++I)
            if (vidx(I) == V) return I;
        return -1;
    }
    static constexpr int vtiles(int V) { return item_at(V % C) < 0 ? 0 : ntiles(item_at(V % C)); }
    // Request schedule: while item I (position V) is consumed, every position up to V - 1 + RD that has not been requested yet
    // is requested - i.e. the slot of the item consumed BEFORE I is re-filled (its last reader sits behind a scheduling
    // barrier: the old and the new contents of a slot are never live together, so hipcc needs no copy of a ring register;
    // stream_r8.h tops its ring up the same way).  RD - 1 items are in flight at most.
    static constexpr int prev_pos(int I) { return I > 0 ? vidx(I - 1) : vidx(CR - 1) - C; }
    static constexpr int req_hi(int I) { return vidx(I) - 1 + RD; }          // last position requested while item I is consumed
    static constexpr int req_lo(int I) { return prev_pos(I) - 1 + RD + 1; }  // first one
    // loads issued after the last tile of item I when its tiles are waited for (requests up to prev_pos(I) - 1 + RD are out)
    static constexpr int inflight_behind(int I) {
        if (NS > 0) return younger(I);
        int n = 0;
        for (int J = vidx(I) + 1; J <= prev_pos(I) - 1 + RD; ++J) n += vtiles(J);
        return n;
    }
    // NS > 0 (E == 0: positions are items): loads issued behind the tiles of item I - ring item: its request, stash item: its
    // piece - when they are waited for, i.e. after the refill of the item consumed before I.  The issue order is replayed
    // backwards: the item consumed `step` items before I requested ring position (that item) - 1 + RD and then its piece.
    static constexpr int younger(int I) {
        int n = 0;
        for (int step = 1; step <= C; ++step) {
            const int J = ((I - step) % C + C) % C;
            const int pk = dma_piece(J);
            if (pk >= 0) {
                if (is_stash(I) && pk == stash_k(I)) return n;
                n += G;
            }
            const int V = (I - step) - 1 + RD;                 // position relative to I's layer (may be < 0 or >= C)
            if (!is_stash(I) && V == I) return n;
            n += vtiles(((V % C) + C) % C);
        }
        return 63;
    }
    static_assert(E <= NQW / STEP, "every empty item needs an item of the W x W stage to sit behind");
    static_assert((RD - 1) * G < 64, "vmcnt is a 6-bit counter");
    static_assert(NS == 0 || (E == 0 && !FAST && NS <= 5 && SQD >= 2 && stash_item(NS - 1) + RD - 1 < I_N),
                  "the stash needs a layer without empty items; no stash item within the ring's reach of the short stages");
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
    template <int T, class WT>
    __device__ __forceinline__ void tile(const float4& x, const WT& w) {
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

// the bias blocks of `nblk` layers: image -> LDS (all threads; the caller synchronises)
__device__ __forceinline__ void r4f_load_bias(const float* __restrict__ src, float* __restrict__ dst, int nfloats, int tid) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int e = tid; e < nfloats / 4; e += NTHREADS) d4[e] = s4[e];
}

// Ring of RD items over the fused stream in ACCUMULATION registers (the structure of stream_r8.h): the tiles are requested by
// inline-asm loads hipcc does not see ("=a" destinations: the MFMAs read their B operand from there directly) and waited for
// with hand-counted s_waitcnt vmcnt(N) that name the registers - so the ring costs no architectural VGPR (round 5: with
// compiler-tracked loads 100 of 255 VGPRs, 136 v_accvgpr moves in S1 alone) and can be RD = 10 items deep: the short stages'
// idle memory pipe prefetches half of the next W x W stage.  Loads return in order, so an item is complete once at most
// `inflight_behind` younger loads are outstanding; any other load in flight (stage stamps) only makes a wait conservative.
// The build's ISA check (_isa_check.py) verifies on the generated code that no instruction touches a ring register whose
// load may still be in flight.
template <int NTWM, bool FAST = false, int NS = 0>
struct R4FRing {
    using S = R4F<NTWM, FAST, NS>;
    static constexpr int G = S::G, RD = S::RD, C = S::C;
    f32x4 r[RD][G];
    const float4 *spG, *sp1;           // (layer slot, wave) of the section being consumed: items of G tiles / of one tile
    unsigned voff0, voff1;             // lane * 16 (+ 4096: the fifth tile of an item is past the 12-bit immediate offset)
    unsigned stash_m0;                 // LDS byte address of this wave's stash (NS G tiles of 1 KiB)
    const float4* stash_rd;            // the same place for this lane's ds_read_b128: + (k G + g) 64 float4
    __device__ __forceinline__ R4FRing(const float4* base, const Tid4& t, float* stash = nullptr)
        : spG(base + (size_t)t.wave * G * 64), sp1(base + (size_t)t.wave * 64), voff0((unsigned)t.lane * 16u),
          voff1((unsigned)t.lane * 16u + 4096u),
          stash_m0((unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)stash + t.wave * (NS * G * 1024))),
          stash_rd(reinterpret_cast<const float4*>(stash) + (size_t)t.wave * (NS * G * 64) + t.lane) {
        if constexpr (NS == 0) static_for<0, RD - 1>([&](auto vc) { request<decltype(vc)::value>(); });
        else {
            // what the last RD - 1 items of a layer in front of this one would have requested, in their order (the waits count on it)
            static_for<0, RD - 1>([&](auto pc) {
                constexpr int P = decltype(pc)::value - (RD - 1);        // consumed position, < 0: item P + C of the slot before
                request<P - 1 + RD>();
                constexpr int I = P + C;
                if constexpr (S::dma_piece(I) >= 0) request_dma<S::dma_piece(I), false>();
            });
        }
    }
    // stash piece k of this layer slot (NEXT: of the one behind it): G tiles -> LDS
    template <int K, bool NEXT>
    __device__ __forceinline__ void request_dma() {
        constexpr int I = S::stash_item(K);
        const float4* b0 = spG + ((size_t)(NEXT ? 1 : 0) * S::TL + S::toff(I)) * 4 * 64;
        const unsigned long long bu = (unsigned long long)b0;
        const float4* b = reinterpret_cast<const float4*>(
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bu >> 32)) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bu & 0xffffffffull)));
        const unsigned dst = stash_m0 + K * G * 1024;
        // M0 = LDS byte address of the first tile; the immediate offset moves the global AND the LDS address (lane l -> + 16 l)
        // (M0 is declared clobbered: hipcc may hold an indexing base in it across the statement otherwise - ADVICE r5)
        static_assert(G >= 2 && G <= 5, "");
        if constexpr (G == 2)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:0\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(voff0), "s"(b), "s"(dst) : "memory", "m0");
        else if constexpr (G == 3)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:0\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048"
                         :: "v"(voff0), "s"(b), "s"(dst) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:0\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff0), "s"(b), "s"(dst) : "memory", "m0");
        if constexpr (G == 5)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0"
                         :: "v"(voff1), "s"(b), "s"(dst + 4096u) : "memory", "m0");
    }
    // the tiles of stash item I have landed in LDS (loads return in order: at most `inflight_behind` younger ones are outstanding)
    template <int I>
    __device__ __forceinline__ void wait_stash(IC<I>) {
        constexpr int N = S::inflight_behind(I);
        static_assert(S::is_stash(I) && N >= 0 && N < 64, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    }
    template <int I>
    __device__ __forceinline__ void read_stash(IC<I>, float4 (&w)[G]) const {
        constexpr int K = S::stash_k(I);
#pragma unroll
        for (int g = 0; g < G; ++g) w[g] = stash_rd[(K * G + g) * 64];
    }
    // request the item at virtual position V (V >= C: of the next layer) into its slot
    template <int V>
    __device__ __forceinline__ void request() {
        constexpr int I = S::item_at(V % C);
        if constexpr (I >= 0) {
            constexpr int n = S::ntiles(I), SL = V % RD;
            const float4* b0 = (n == G ? spG : sp1) + ((size_t)(V / C) * S::TL + S::toff(I)) * 4 * 64;
            // (wave-uniform by construction; made so explicitly: the "s" operand of the load must not end up in VGPRs)
            const unsigned long long bu = (unsigned long long)b0;
            const float4* b = reinterpret_cast<const float4*>(
                ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bu >> 32)) << 32) |
                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bu & 0xffffffffull)));
            static_for<0, n>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g == 0) s8_load_first<0>(r[SL][0], voff0, b);
                else if constexpr (g < 4) s8_load<g * 1024>(r[SL][g], voff0, b);
                else s8_load<(g - 4) * 1024>(r[SL][g], voff1, b);
            });
        }
    }
    template <int I>
    static constexpr int slot(IC<I>) { return S::vidx(I) % RD; }
    // the tiles of item I have landed (nothing below this line is scheduled above it)
    template <int I>
    __device__ __forceinline__ void wait(IC<I>) {
        constexpr int SL = S::vidx(I) % RD, N = S::inflight_behind(I);
        static_assert(!S::is_stash(I) && N >= 0 && N < 64, "vmcnt is a 6-bit counter");
        if constexpr (S::ntiles(I) == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+a"(r[SL][0]) : "n"(N));
        else if constexpr (G == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+a"(r[SL][0]), "+a"(r[SL][1]) : "n"(N));
        else if constexpr (G == 4)
            asm volatile("s_waitcnt vmcnt(%4)" : "+a"(r[SL][0]), "+a"(r[SL][1]), "+a"(r[SL][2]), "+a"(r[SL][3]) : "n"(N));
        else
            asm volatile("s_waitcnt vmcnt(%5)"
                         : "+a"(r[SL][0]), "+a"(r[SL][1]), "+a"(r[SL][2]), "+a"(r[SL][3]), "+a"(r[SL][4])
                         : "n"(N));
        __builtin_amdgcn_sched_barrier(0);
    }
    // called once while item I is consumed, between two scheduling barriers: top the ring up to position vidx(I) - 1 + RD
    template <int I>
    __device__ __forceinline__ void refill(IC<I>) {
        __builtin_amdgcn_sched_barrier(0);
        static_for<S::req_lo(I), S::req_hi(I) + 1>([&](auto pc) { request<decltype(pc)::value>(); });
        if constexpr (S::dma_piece(I) >= 0) request_dma<S::dma_piece(I), S::dma_next_slot(I)>();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void next_layer() { spG += (size_t)S::TL * 4 * 64; sp1 += (size_t)S::TL * 4 * 64; }
    // end of an evaluation: wait for the requests that ran past the end of the stream (into its padding slot)
    __device__ __forceinline__ void drain() {
        static_for<0, RD>([&](auto sc) {
            constexpr int SL = decltype(sc)::value;
            static_for<0, G>([&](auto gc) { asm volatile("s_waitcnt vmcnt(0)" : "+a"(r[SL][decltype(gc)::value])); });
        });
    }
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
    ring.wait(IC<I0>{});
    r4_quad<G>(a0, ring.r[ring.slot(IC<I0>{})], acc0);
    ring.refill(IC<I0>{});
    __builtin_amdgcn_sched_barrier(0);
    ring.wait(IC<I0 + 1>{});
    r4_quad<G>(a1, ring.r[ring.slot(IC<I0 + 1>{})], acc1);
    ring.refill(IC<I0 + 1>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc0[g] += acc1[g];
    (void)PN;
    r4_store_part_rf<G>(acc0, part, t);
}

// S2 / S5: the W x W stage on the ring, quad q = item I0 + q
template <int NTWM, int EP, class Ring>
__device__ __forceinline__ void r4f_dense_wide(const float* act, int lda, Ring& ring, const float* __restrict__ bias, float* out,
                                               int ldo, unsigned* mask, float* part, int PN, const Tid4& t) {
    using S = typename Ring::S;
    constexpr int G = NTWM, NQ = 4 * NTWM, I0 = S::I_W;
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow = act + t.arow * lda + 16 * NTWM * t.wave;
    float4 ws[2][G];                   // stash tiles out of LDS, read one item ahead of their MFMAs
    static_for<0, NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 a = *reinterpret_cast<const float4*>(arow + 4 * q);
        if constexpr (S::is_stash(I0 + q)) {
            r4_quad<G>(a, ws[S::stash_k(I0 + q) & 1], acc);
        } else {
            ring.wait(IC<I0 + q>{});
            r4_quad<G>(a, ring.r[ring.slot(IC<I0 + q>{})], acc);
        }
        ring.refill(IC<I0 + q>{});
        if constexpr (q + 1 < NQ && S::is_stash(I0 + q + 1)) {
            ring.wait_stash(IC<I0 + q + 1>{});
            ring.read_stash(IC<I0 + q + 1>{}, ws[S::stash_k(I0 + q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    (void)PN;
    r4_store_part_rf<G>(acc, part, t);
    float bv[G];
    if (bias) r4_bias_rf<G>(bv, bias, t);
    else {
#pragma unroll
        for (int g = 0; g < G; ++g) bv[g] = 0.f;
    }
    r4_barrier();
    r4_epilogue_rf<G, EP>(part, bv, out, ldo, mask, t);
    r4_barrier();
}

// the same stage in FAST mode: item i = k-quads 2 i, 2 i + 1 of this wave as bf16 tiles; the activations are rounded to bf16 as
// they are fetched (v_cvt_pk_bf16_f32, round to nearest even: what fast mode's 16-chain kernels do), one
// v_mfma_f32_4x4x4_16b_bf16 per k-quad and column group (fp32 accumulation; even / odd quads on separate accumulators, added at
// the end: 2 G chains of G dependent instructions)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x4 r4f_bf16x4(const float4& a) {
    union { unsigned u[2]; s16x4 v; } c;
    c.u[0] = cvt_pk_bf16(a.x, a.y);
    c.u[1] = cvt_pk_bf16(a.z, a.w);
    return c.v;
}
template <int NTWM, int EP, class Ring>
__device__ __forceinline__ void r4f_dense_wide_bf16(const float* act, int lda, Ring& ring, const float* __restrict__ bias, float* out,
                                                    int ldo, unsigned* mask, float* part, const Tid4& t) {
    using S = typename Ring::S;
    constexpr int G = NTWM, NI = 2 * NTWM, I0 = S::I_W;
    f32x4 acc0[G], acc1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { acc0[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[g] = acc0[g]; }
    const float* arow = act + t.arow * lda + 16 * NTWM * t.wave;
    static_for<0, NI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const s16x4 a0 = r4f_bf16x4(*reinterpret_cast<const float4*>(arow + 8 * i));
        const s16x4 a1 = r4f_bf16x4(*reinterpret_cast<const float4*>(arow + 8 * i + 4));
        ring.wait(IC<I0 + i>{});
        constexpr int SL = Ring::slot(IC<I0 + i>{});
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 w = ring.r[SL][g];
            acc0[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a0, __builtin_bit_cast(s16x4, (f32x2v)__builtin_shufflevector(w, w, 0, 1)),
                                                             acc0[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 w = ring.r[SL][g];
            acc1[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a1, __builtin_bit_cast(s16x4, (f32x2v)__builtin_shufflevector(w, w, 2, 3)),
                                                             acc1[g], 0, 0, 0);
        }
        // the item's tiles stay allocated until its last MFMA has issued: hipcc otherwise makes a consumed tile the destination of
        // the MFMA that reads its upper half (v_mfma a[68:71], v[..], a[70:71]), and on gfx950 the pipelined 4x4x4 bf16 MFMA then
        // returns wrong sums (round 5: odd k-quads wrong exactly for the column groups allocated that way; the f32 4x4x1 form with
        // the same overlap is fine, a single bf16 MFMA with it as well)
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : : "a"(ring.r[SL][g]));
        // (r6, accumulators in architectural registers - _build.py's -amdgpu-mfma-vgpr-form: the same holds for the A operands, which
        //  hipcc would otherwise hand to the last MFMA that reads them as its destination)
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : "+v"(acc0[g]), "+v"(acc1[g]) : "v"(a0), "v"(a1));
        ring.refill(IC<I0 + i>{});
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int g = 0; g < G; ++g) acc0[g] += acc1[g];
    r4_store_part_rf<G>(acc0, part, t);
    float bv[G];
    if (bias) r4_bias_rf<G>(bv, bias, t);
    else {
#pragma unroll
        for (int g = 0; g < G; ++g) bv[g] = 0.f;
    }
    r4_barrier();
    r4_epilogue_rf<G, EP>(part, bv, out, ldo, mask, t);
    r4_barrier();
}

// the 2 G dense tiles of items I_N, I_N + 1 against ACT[4][Wp] (this wave's K range: quads 4 G w + 2 T + sblk)
template <int NTWM, class Ring>
__device__ __forceinline__ void r4f_narrow_mma(const float* act, int lda, Ring& ring, R4FAcc& p, const Tid4& t) {
    using S = typename Ring::S;
    constexpr int G = NTWM, RD = S::RD, I0 = S::I_N;
    const float* arow = act + t.arow * lda + 16 * NTWM * t.wave + 4 * (t.lane >> 5);
    static_for<0, 2 * G>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        const float4 a = *reinterpret_cast<const float4*>(arow + 8 * T);
        if constexpr (T % G == 0) ring.wait(IC<I0 + T / G>{});
        p.template tile<T>(a, ring.r[ring.slot(IC<I0 + T / G>{})][T % G]);
        if constexpr (T % G == G - 1) {
            ring.refill(IC<I0 + T / G>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}

// L2 prefetch of the weight stream (round 5; tools/ubench/stream2.hip, profiles/r5/ubench_stream2_prefetch.txt).  The 47.9 MB a
// launch streams pass through every XCD's 4 MB L2 once per evaluation: the 32 workgroups of an XCD run in step, the first one
// to ask for a line misses (MALL / HBM, ~1.3 us) and the other 31 queue on that pending fill - a stage-synchronised 4-wave
// stream of this shape runs at 88 cycles per 1-KiB tile and SIMD on such data and at 70.5 when the data already sits in the L2.
// So each workgroup TOUCHES its share of the lines of a layer slot AHEAD slots before the stream gets there: one load per
// wave and slot, lane -> one 128-byte line (line j + n (64 wave + lane) of the slot, j = the workgroup's index on its XCD, n = the
// launch's workgroups per XCD, 32 at most: they cover every line once; waves whose lines all lie past the slot's end skip it) - in the
// micro-benchmark 88.2 -> 75.7 cycles per tile with a fifth wave doing it; here the waves do it themselves, right behind the
// last request of the slot's first short stage: loads return in order, so the ring items requested behind a prefetch wait
// for it, and at that point the next ring item is not needed for ~3 k cycles (the stage's epilogue + four W x W items).
// The load is an LDS-DMA (global_load_lds_dword: the dwords land in 256 bytes of junk LDS of this wave): it has NO destination
// register.  (First version: a global_load_dword into an accumulation register nobody reads - measured the same, but hipcc is
// free to move such a register while the load is in flight, and did so in two other instantiations: the build's ISA check
// refused them.)  The hand-counted waits stay valid: another load in the queue only makes them conservative.
#ifndef FAB_R4F_PF_AHEAD
#define FAB_R4F_PF_AHEAD 2             // slots ahead (1 and 2 measure alike: 0.511 ms per transition against 0.570 without; 2 leaves
                                       // more room for workgroups of an XCD that have drifted apart); 0 = no prefetch
#endif
// this lane's line of a slot (byte offset; ~0u: this wave has none) - once per evaluation: it takes an integer division
template <int TL>
__device__ __forceinline__ unsigned r4f_prefetch_line(const Tid4& t) {
    constexpr unsigned NL = (unsigned)TL * 32u;                    // 128-byte lines of a slot (TL tiles x 4 waves x 1 KiB)
    // shares = the workgroups this launch has on an XCD (workgroup b runs on XCD b mod 8), 32 at most: with fewer than 16 a
    // workgroup's 256 lanes no longer reach every line of its share - what they reach is still prefetched
    unsigned nsh = (gridDim.x + 7u) >> 3;
    nsh = nsh > 32u ? 32u : nsh;
    if (nsh * 64u * (unsigned)t.wave >= NL) return ~0u;            // (wave-uniform: all of this wave's lines lie past the slot's end)
    unsigned line = (blockIdx.x >> 3) % nsh + nsh * (unsigned)(64 * t.wave + t.lane);
    line = line < NL ? line : NL - 1u;
    return line * 128u;
}
template <int TL>
__device__ __forceinline__ void r4f_prefetch_l2(const float* junk, unsigned voff, const char* img, int slot, int nslots) {
    if constexpr (FAB_R4F_PF_AHEAD > 0) {
        if (voff == ~0u) return;                                   // (wave-uniform)
        int ts = slot + FAB_R4F_PF_AHEAD;
        ts = ts >= nslots ? ts - nslots : ts;                      // (the next evaluation starts at slot 0 again)
        const unsigned long long bu = (unsigned long long)(img + (size_t)ts * ((size_t)TL * 4096));
        const char* b = reinterpret_cast<const char*>(
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bu >> 32)) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bu & 0xffffffffull)));
        const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)junk);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(b), "s"(m0v) : "memory", "m0");
    }
}

// ------------------------------------------------------------------------------------------------
// log q(x) and d log q / dx for the 4 rows in X0 (columns >= D zero).  The gradient is left in X0 (*grad_off = l.o_X0, leading
// dimension R4_DS); returns log q of row `tid >> 4` on wave 0.  The density bias table must be in LDS at l.o_BIAS.
// ------------------------------------------------------------------------------------------------
template <int NTWM, bool FAST = false, int NS = 0>
__device__ float flow_log_prob_r4f(const FlowDims& f, const R4Lds& l, const float* __restrict__ packed, float* lds,
                                   const Tid4& t, int* grad_off, float* stash = nullptr) {
    using S = R4F<NTWM, FAST, NS>;
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
    R4FRing<NTWM, FAST, NS> ring(reinterpret_cast<const float4*>(packed + (FAST ? f.o_r4fh : f.o_r4f)), t, stash);
    const char* pf_img = reinterpret_cast<const char*>(packed + (FAST ? f.o_r4fh : f.o_r4f));
    const float* pf_junk = lds + l.o_PF + 64 * t.wave;             // (r4f_prefetch_l2)
    const unsigned pf_voff = r4f_prefetch_line<S::TL>(t);
    int pf_slot = 0;
    float logq = 0.f;
#pragma unroll 1                       // (an unrolled copy gets other ring registers, joined by copies of in-flight slots: ISA check)
    for (int layer = f.K - 1; layer >= 0; --layer) {
        const float* bt = BT + (size_t)layer * BS;
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        const bool tl = layer == f.K - 2;
        if (tl) FAB_TL(f, 0);
        {   // S1: h1 = relu(y W1' + b1'), z = y A + ac   (K = 32: k in [8 w, 8 w + 8) on this wave)
            const float* xr = X + t.arow * R4_DS + 8 * t.wave;
            const float4 a0 = *reinterpret_cast<const float4*>(xr), a1 = *reinterpret_cast<const float4*>(xr + 4);
#ifdef FAB_R4F_WAVETL                  // dev-only (tools/experiments/dephase): per-wave stamps around the refills of S1
            long long wt[8];
#define FAB_WT(k) do { __builtin_amdgcn_sched_barrier(0); wt[k] = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
            FAB_WT(0);
            {
                f32x4 acc0[G], acc1[G];
#pragma unroll
                for (int g = 0; g < G; ++g) { acc0[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[g] = acc0[g]; }
                ring.wait(IC<0>{});
                FAB_WT(1);
                r4_quad<G>(a0, ring.r[ring.slot(IC<0>{})], acc0);
                FAB_WT(2);
                ring.refill(IC<0>{});
                FAB_WT(3);
                ring.wait(IC<1>{});
                r4_quad<G>(a1, ring.r[ring.slot(IC<1>{})], acc1);
                FAB_WT(4);
                ring.refill(IC<1>{});
                FAB_WT(5);
#pragma unroll
                for (int g = 0; g < G; ++g) acc0[g] += acc1[g];
                r4_store_part_rf<G>(acc0, PART, t);
            }
#else
            r4f_short_mma<NTWM, 0>(a0, a1, ring, PART, l.PN, t);
#endif
            R4FAcc z;
            ring.wait(IC<S::I_A>{});
            z.template tile<0>(sblk ? a1 : a0, ring.r[ring.slot(IC<S::I_A>{})][0]);
            ring.refill(IC<S::I_A>{});
            r4f_prefetch_l2<S::TL>(pf_junk, pf_voff, pf_img, pf_slot++, 2 * f.K);
#ifdef FAB_R4F_WAVETL
            FAB_WT(6);
#endif
            z.store(PZ, t);
            float bv[G];
            r4_bias_rf<G>(bv, bt, t);
            r4_barrier();
#ifdef FAB_R4F_WAVETL
            FAB_WT(7);
            if (tl && f.timeline && blockIdx.x == 0 && t.lane == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) f.timeline[32 + 8 * t.wave + k] = wt[k];
            }
#endif
            float zv = 0.f;                                // (read before the wide epilogue's writes: one LDS round trip less)
            if (t.tid < 128) zv = r4f_tree8(PZ, zrow, zc) + bt[2 * f.Wp + zc];
            r4_epilogue_rf<G, 1>(PART, bv, HA, l.WS, mk, t);
            if (t.tid < 128) X[zrow * R4_DS + zc] = zv;
            r4_barrier();
        }
        logq += bt[2 * f.Wp + 64];
        if (tl) FAB_TL(f, 1);
        if constexpr (FAST) r4f_dense_wide_bf16<NTWM, 1>(HA, l.WS, ring, bt + f.Wp, HB, l.WS, mk + NTHREADS, PART, t);
        else r4f_dense_wide<NTWM, 1>(HA, l.WS, ring, bt + f.Wp, HB, l.WS, mk + NTHREADS, PART, l.PN, t);
        if (tl) FAB_TL(f, 3);
        {   // S3: (shift | s) = h2 W3 + b3, then AffineCoupling.inverse: z2 <- (z2 - shift) exp(-s), log_det = -sum(s)
            R4FAcc p;
            r4f_narrow_mma<NTWM>(HB, l.WS, ring, p, t);
            p.store(PZ, t);
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
#pragma unroll 1                       // (an unrolled copy gets other ring registers, joined by copies of in-flight slots: ISA check)
    for (int layer = 0; layer < f.K; ++layer) {
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        const bool tl = layer == 1;
        if (tl) FAB_TL(f, 16);
        f32x4 atile;
        {   // S4: (shift | scale) cotangents -> hidden (K = 32)
            const float* dr = DP + t.arow * R4_DS + 8 * t.wave;
            const float4 a0 = *reinterpret_cast<const float4*>(dr), a1 = *reinterpret_cast<const float4*>(dr + 4);
            r4f_short_mma<NTWM, 0>(a0, a1, ring, PART, l.PN, t);
            ring.wait(IC<S::I_A>{});
            {   // A^T: used by S6.  Copied out of the ring into VGPRs - an opaque copy: if `atile` merely aliased the slot's register,
                // the slot's next request would get ANOTHER register and hipcc would copy that in-flight register back at the
                // loop latch (ISA check, G = 2)
                const f32x4 tsrc = ring.r[ring.slot(IC<S::I_A>{})][0];
                float ax = tsrc.x, ay = tsrc.y, az = tsrc.z, aw = tsrc.w;
                asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(aw));
                atile = (f32x4){ax, ay, az, aw};
            }
            ring.refill(IC<S::I_A>{});
            r4f_prefetch_l2<S::TL>(pf_junk, pf_voff, pf_img, pf_slot++, 2 * f.K);
            float bv[G];
#pragma unroll
            for (int g = 0; g < G; ++g) bv[g] = 0.f;
            r4_barrier();
            r4_epilogue_rf<G, 2>(PART, bv, HA, l.WS, mk + NTHREADS, t);
            r4_barrier();
        }
        if (tl) FAB_TL(f, 18);
        if constexpr (FAST) r4f_dense_wide_bf16<NTWM, 2>(HA, l.WS, ring, nullptr, HB, l.WS, mk, PART, t);
        else r4f_dense_wide<NTWM, 2>(HA, l.WS, ring, nullptr, HB, l.WS, mk, PART, l.PN, t);
        if (tl) FAB_TL(f, 19);
        {   // S6: g_y = dh1 W1'^T + g_z A^T, then the coupling cotangents of layer + 1
            R4FAcc p;
            r4f_narrow_mma<NTWM>(HB, l.WS, ring, p, t);
            const float4 ag = *reinterpret_cast<const float4*>(X + t.arow * R4_DS + 8 * t.wave + 4 * sblk);
            p.template tile<0>(ag, atile);
            p.store(PZ, t);
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
    ring.drain();
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
    // S1 of virtual layer v: h1 = relu(z W1'' + b1''), y = z W'^-1(layer v - 1) + at   (v = 0: y = z; v = K: the last affine map
    // alone, its h1 discarded).  Called from the loop and once behind it: a loop with an exit in the middle gets rotated by hipcc,
    // and the copies of S1 then disagree about the ring's registers (found by the build's ISA check)
    auto s1 = [&](const float* bt) {
        const float* xr = X + t.arow * R4_DS + 8 * t.wave;
        const float4 a0 = *reinterpret_cast<const float4*>(xr), a1 = *reinterpret_cast<const float4*>(xr + 4);
        r4f_short_mma<NTWM, 0>(a0, a1, ring, PART, l.PN, t);
        R4FAcc z;
        ring.wait(IC<S::I_A>{});
        z.template tile<0>(sblk ? a1 : a0, ring.r[ring.slot(IC<S::I_A>{})][0]);
        ring.refill(IC<S::I_A>{});
        z.store(PZ, t);
        float bv[G];
        r4_bias_rf<G>(bv, bt, t);
        r4_barrier();
        float zv = 0.f;                                // (read before the wide epilogue's writes: one LDS round trip less)
        if (t.tid < 128) zv = r4f_tree8(PZ, zrow, zc) + bt[2 * f.Wp + zc];
        r4_epilogue_rf<G, 1>(PART, bv, HA, l.WS, mk, t);
        if (t.tid < 128) X[zrow * R4_DS + zc] = zv;
        r4_barrier();
        logq -= -bt[2 * f.Wp + 64];
    };
#pragma unroll 1                       // (an unrolled copy gets other ring registers, joined by copies of in-flight slots: ISA check)
    for (int v = 0; v < f.K; ++v) {
        const float* bt = BT + (size_t)v * BS;
        s1(bt);
        r4f_dense_wide<NTWM, 1>(HA, l.WS, ring, bt + f.Wp, HB, l.WS, mk + NTHREADS, PART, l.PN, t);
        {   // S3 + AffineCoupling.forward: z2 <- z2 exp(s) + shift, log_det = sum(s)
            R4FAcc p;
            r4f_narrow_mma<NTWM>(HB, l.WS, ring, p, t);
            p.store(PZ, t);
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
    s1(BT + (size_t)f.K * BS);
    ring.drain();
    *x_off = l.o_X0;
    return logq;
}

// host side: the fused-stage variant is chosen where its image exists, FABHIP_OPT_R4_STREAM >= 2 (default) and the bias blocks of
// all layers fit the CU's LDS next to the tile state (the transition kernel adds 4 x [4][D] floats of HMC state)
int option(int key);                                   // (launch.h)
bool r4f_lds_fits(const FlowDims& f);                  // (flow_kernels.hip)
static inline bool use_r4_fused(const FlowDims& f) {
    if (f.o_r4f < 0 || f.NTW / 4 < 2 || f.NTW / 4 > 5 || option(FABHIP_OPT_R4_STREAM) < 2) return false;
    return r4f_lds_fits(f);
}

}  // namespace fab
