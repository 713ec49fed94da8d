// 4-chain tiles (R4): the flow density + d/dx for FOUR chains per workgroup on v_mfma_f32_4x4x1_16b_f32.
//
// Why a second tile shape: the 16-chain tile code (flow_device.h) gives B / 16 workgroups - 64 of 256 CUs at the
// headline batch of 1024 chains, and its time per launch is flat from 16 to 4096 chains (per-workgroup dependent
// chain).  With 4 chains per workgroup 1024 chains fill the chip; a workgroup then does a quarter of the MFMA work but
// still streams ALL the weights, so the stages are bound by the L2 -> VGPR weight stream (tools/ubench/stream.hip:
// 56 B/clk per CU with 256 workgroups reading the same 10 MB in lockstep) instead of by the matrix pipe.
//
// GEMM shape: OUT[4][N] = ACT[4][K] @ B[K][N].  v_mfma_f32_4x4x1_16b computes 16 independent 4x4 outer products:
// lane l = 4 b + j supplies A[b][i = l % 4] and B[b][j], and holds D[b][i = VGPR r][j].  Block b of column group g is
// columns 64 g + 4 b .. + 3, so lane l owns column 64 g + l of the group: ONE instruction = 4 chains x 64 columns x 1 k.
// K is split over the 4 waves (wave w: k-quads [w Q/4, (w+1) Q/4), Q = Kp / 4); the four partial [4][N] products go
// through LDS and are added in a fixed order by the epilogue (bias, ReLU + sign bits / sign-bit mask), which every
// thread runs for N / 64 of the 4 N outputs.
// Packed image of a matrix (k_pack_r4): float4 tile (q, g), lane l = { B[4 q + kk][64 g + l] } kk < 4, tiles ordered
// q-major: every load instruction of a wave is one contiguous 1-KiB block, a wave's K range is one contiguous stream.
#pragma once
#include "flow_device.h"
#include "target_device.h"

namespace fab {

constexpr int R4 = 4;                  // chains per workgroup
constexpr int R4_DS = MAX_DIM + 4;     // leading dim of the state / parameter buffers (D <= 64)

// geometry of the r4 weight image of one layer (floats), appended to the packed flow image at FlowDims::o_r4
struct R4Dims {
    int Kd, Ko, KD;                    // padded K extents: conditioner input d, coupling parameters 2 DOp, state D (Kw = Wp)
    int G;                             // hidden column groups: Wp / 64
    int o_AW, o_AWT, o_W1, o_W2, o_W3, o_W3T, o_W2T, o_W1T, layer_stride;
};

FAB_HD R4Dims make_r4_dims(const FlowDims& f) {
    R4Dims r;
    r.Kd = pad16(f.d); r.Ko = pad16(2 * f.DOp); r.KD = pad16(f.D);
    r.G = f.Wp / 64;
    int o = 0;
    r.o_AW = o; o += r.KD * 64;        // [D -> D]   (N padded to 64)
    r.o_AWT = o; o += r.KD * 64;
    r.o_W1 = o; o += r.Kd * f.Wp;      // [d -> W]
    r.o_W2 = o; o += f.Wp * f.Wp;      // [W -> W]
    r.o_W3 = o; o += f.Wp * 2 * f.DOp; // [W -> shift | scale]   (dense tiles, see r4_dense_n16)
    r.o_W3T = o; o += r.Ko * f.Wp;     // [shift | scale -> W]
    r.o_W2T = o; o += f.Wp * f.Wp;
    r.o_W1T = o; o += f.Wp * pad16(f.d); // [W -> d]            (16-column tiles)
    r.layer_stride = o;
    return r;
}

// LDS plan of an r4 workgroup (floats)
struct R4Lds {
    int WS, PN;                        // leading dims: hidden activations, partial products
    int o_X0, o_X1, o_HA, o_HB, o_PRM, o_DP, o_PART, o_ES, o_V2, o_MASK, total;
    int o_PZ, o_BIAS;                  // fused stages (flow_r4f.h): partials of the dense narrow products, bias blocks of all layers
    int o_PF;                          // ... and 256 bytes per wave that the L2 prefetch's LDS-DMA loads land in (never read)
};

FAB_HD R4Lds make_r4_lds(const FlowDims& f, bool fused = false) {
    R4Lds l;
    // (Wp + 4) * 4 bytes = 16 mod 128: the 4 rows' 16-byte A reads hit 4 different bank groups; fused stages: Wp + 16 (64 bytes
    // mod 256) - their epilogues store a wave's outputs as 16 columns x 4 rows, which then fall into 64 different banks
    l.WS = fused ? f.Wp + 16 : f.Wp + 4;
    l.PN = f.Wp;
    int o = 0;
    l.o_X0 = o; o += R4 * R4_DS;
    l.o_X1 = o; o += R4 * R4_DS;
    l.o_HA = o; o += R4 * l.WS;
    l.o_HB = o; o += R4 * l.WS;
    l.o_PRM = o; o += R4 * R4_DS;
    l.o_DP = o; o += R4 * R4_DS;
    l.o_PART = o; o += NWAVE * R4 * l.PN;
    l.o_ES = o; o += f.K * R4 * f.DOp;
    l.o_V2 = o; o += f.K * R4 * f.DOp;
    l.o_MASK = o; o += f.K * 2 * NTHREADS;
    l.o_PZ = l.o_BIAS = l.o_PF = 0;
    if (fused) {
        o = (o + 3) & ~3;
        l.o_PZ = o; o += 2 * NWAVE * R4 * 32;
        l.o_BIAS = o; o += (f.K + 1) * r4f_bias_stride(f.Wp);
        o = (o + 3) & ~3;
        l.o_PF = o; o += NWAVE * 64;
    }
    l.total = (o + 3) & ~3;
    return l;
}

__device__ __forceinline__ f32x4 mfma44(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// thread roles: GEMMs use all 4 waves; the element-wise stages run on wave 0 as 4 rows x 16 lanes (row = tid >> 4,
// c = tid & 15: the 16-chain code's mapping restricted to its first 4 rows, so row16_sum / target_tile apply as they are)
struct Tid4 {
    int tid, wave, lane, arow;
    __device__ __forceinline__ Tid4() {
        tid = threadIdx.x;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        lane = tid & 63;
        arow = lane & 3;               // the chain whose activations this lane feeds to the MFMA
    }
};

// ---- weight tiles requested ahead of their stage (per-stage schedule: flow_log_prob_r4) -------------------------------
// A stage that requests its own weights starts with a cold L2 / MALL access (2 - 3 k cycles, 8 stages per layer pair), so
// tiles are requested earlier and parked in registers (R4Pre / R4PreT) until their stage runs.
template <int NQ, int G>
struct R4Pre {
    float4 b[NQ][G];                   // tiles (q, g) of the first NQ k-quads of this wave's K range
};

// nq_valid <= NQ quads starting at this wave's quad `qbase`
template <int NQ, int G>
__device__ __forceinline__ void r4_preload(R4Pre<NQ, G>& p, const float4* __restrict__ Bm, int qbase, int nq_valid,
                                           const Tid4& t) {
    const float4* bw = Bm + ((size_t)qbase * G) * 64 + t.lane;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (q < nq_valid) p.b[q][g] = bw[(size_t)(q * G + g) * 64];
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() carries an all-address-space fence, for which hipcc
// emits s_waitcnt vmcnt(0) - every weight request in flight would be waited for at every barrier.  The stages exchange
// data through LDS only, so lgkmcnt(0) + s_barrier is sufficient; the "memory" clobber keeps the compiler from moving
// LDS accesses across it, and the registers of pending (compiler-tracked) global loads are still waited for at first use.
__device__ __forceinline__ void r4_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// quads [Q0, Q1) of a preload (the ring of the next W x W GEMM is requested a few quads per short stage: a burst of
// requests blocks the issuing wave until the memory pipe has taken it, ~16 cycles per KiB and CU)
template <int Q0, int Q1, int NQ, int G>
__device__ __forceinline__ void r4_preload_part(R4Pre<NQ, G>& p, const float4* __restrict__ Bm, int qbase, const Tid4& t) {
    const float4* bw = Bm + ((size_t)qbase * G) * 64 + t.lane;
#pragma unroll
    for (int q = Q0; q < Q1 && q < NQ; ++q)
#pragma unroll
        for (int g = 0; g < G; ++g) p.b[q][g] = bw[(size_t)(q * G + g) * 64];
}

struct R4NoNext {
    __device__ __forceinline__ void operator()() const {}
};

template <int G, class BT>
__device__ __forceinline__ void r4_quad(const float4& a, const BT (&b)[G], f32x4 (&acc)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma44(a.x, b[g].x, acc[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma44(a.y, b[g].y, acc[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma44(a.z, b[g].z, acc[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma44(a.w, b[g].w, acc[g]);
}

// ---- wide-K product (K = Wp), G >= 2 column groups: this wave's 4 NTWM k-quads stream through a ring of RD quads
// (G tiles each): the first RD quads come preloaded (requested during the short stages before this one, so the weight
// stream does not stop there), quad q + RD is requested as soon as quad q is multiplied.  Straight-line code: hipcc
// counts vmcnt exactly.
template <int NTWM>
struct R4Ring {
    static constexpr int NQ = 4 * NTWM, RD = NQ < 6 ? NQ : 6;
};

template <int NTWM, int G, class Next>
__device__ __forceinline__ void r4_mma_wide(const float* __restrict__ act, int lda, const float4* __restrict__ Bm,
                                            const R4Pre<R4Ring<NTWM>::RD, G>& pre, const Tid4& t, f32x4 (&acc)[G], Next next) {
    constexpr int NQ = R4Ring<NTWM>::NQ, RD = R4Ring<NTWM>::RD;
    const float* arow = act + t.arow * lda + 16 * NTWM * t.wave;                  // this wave's K range starts at 16 NTWM w
    const float4* bw = Bm + ((size_t)(4 * NTWM * t.wave) * G) * 64 + t.lane;      // tile (q, g) at (q G + g) * 64
    float4 b[RD][G];
#pragma unroll
    for (int q = 0; q < RD; ++q)
#pragma unroll
        for (int g = 0; g < G; ++g) b[q][g] = pre.b[q][g];
    next();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(arow + 4 * q);
        r4_quad<G>(a, b[q % RD], acc);
        if (q + RD < NQ) {
#pragma unroll
            for (int g = 0; g < G; ++g) b[q % RD][g] = bw[(size_t)((q + RD) * G + g) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// this wave's partial [4][64 G] product -> PART[wave][row][col]
template <int G>
__device__ __forceinline__ void r4_store_part(const f32x4 (&acc)[G], float* __restrict__ part, int PN, const Tid4& t) {
    float* pw = part + (size_t)t.wave * R4 * PN + t.lane;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[r * PN + 64 * g] = acc[g][r];
}

// bias of this thread's G outputs (output o = 256 i + tid -> col o % (64 G)), requested at the start of the stage
template <int G>
__device__ __forceinline__ void r4_bias_load(float (&bv)[G], const float* __restrict__ bias, const Tid4& t) {
    constexpr int N = 64 * G;
#pragma unroll
    for (int i = 0; i < G; ++i) bv[i] = bias ? bias[(256 * i + t.tid) % N] : 0.f;
}

// The 4 partials of this thread's G outputs (output o = 256 i + tid; PN = 64 G, so partial w of output o sits at
// part[w 4 PN + o]) summed as (P0 + P1) + (P2 + P3).  All 2 G reads are issued back to back from ONE address register
// (ds_read2st64_b32: two dwords 64-dword strides apart) and waited for once: written as compiler-visible loads the epilogue
// came out as G dependent LDS round trips (read 4, wait, add, read 4, ...: hipcc's scheduler at 255 live VGPRs), ~1 k cycles
// per wide stage in the round-5 stage stamps.  The waits name the destination registers; "memory" clobbers keep the
// compiler's own LDS accesses on their side of the block.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int O0, int O1>
__device__ __forceinline__ f32x2 lds_read2st64(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1) : "memory");
    return v;
}
template <int G>
__device__ __forceinline__ void r4_read_partials(const float* part, int PN, int tid, float (&v)[G]) {
    (void)PN;                                              // (== 64 G: make_r4_lds)
    const unsigned addr = (unsigned)(size_t)(part + tid);  // LDS byte address (low half of the generic pointer)
    f32x2 r[2 * G];
    static_for<0, G>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        r[2 * i] = lds_read2st64<4 * i, 4 * i + 8 * G>(addr);                // (P0, P2)
        r[2 * i + 1] = lds_read2st64<4 * i + 4 * G, 4 * i + 12 * G>(addr);   // (P1, P3): r[2 i] + r[2 i + 1] = (P0 + P1, P2 + P3)
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 2 * G; ++k) asm volatile("" : "+v"(r[k]));
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const f32x2 h = r[2 * i] + r[2 * i + 1];
        v[i] = h.x + h.y;
    }
}

// epilogue: OUT[row][col] = f(bias[col] + ((P0 + P1) + (P2 + P3))) for this thread's G of the 4 x 64 G outputs
// (row o / (64 G), col o % (64 G)).  EP 0: plain, 1: ReLU, sign bit i kept in *mask, 2: multiplied by sign bit i.
struct R4NoPost {
    __device__ __forceinline__ float operator()(int, int, float v) const { return v; }
};

// `post(row, col, v)` maps the finished value before it is stored (the D x D map of the reverse sweep hands its result
// straight to the next layer's coupling cotangents)
template <int G, int EP, class Post = R4NoPost>
__device__ __forceinline__ void r4_epilogue(const float* __restrict__ part, int PN, const float (&bv)[G],
                                            float* __restrict__ out, int ldo, unsigned* mask, const Tid4& t,
                                            Post post = Post()) {
    constexpr int N = 64 * G;
    unsigned m = EP == 2 ? mask[t.tid] : 0u;
    // every partial is read before the first output is written: `part` and `out` are both LDS, and hipcc keeps a read behind an
    // earlier write it cannot prove disjoint - with the reads and writes of one output adjacent the epilogue was G dependent LDS
    // round trips (~1 k cycles per wide stage, round 5 stage stamps); same sums, same order
    float v[G];
    if constexpr (G >= 2) {                                // (wide outputs: PN == 64 G)
        r4_read_partials<G>(part, PN, t.tid, v);
#pragma unroll
        for (int i = 0; i < G; ++i) v[i] += bv[i];
    } else {                                               // the D x D maps of the unfused variants: 64 columns, partials PN apart
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int o = 256 * i + t.tid, row = o / N, col = o - row * N;
            const float* p = part + row * PN + col;
            v[i] = ((p[0] + p[R4 * PN]) + (p[2 * R4 * PN] + p[3 * R4 * PN])) + bv[i];
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int o = 256 * i + t.tid, row = o / N, col = o - row * N;
        float w = v[i];
        if (EP == 1) { const bool pos = w > 0.f; m |= (pos ? 1u : 0u) << i; w = pos ? w : 0.f; }
        if (EP == 2) w = ((m >> i) & 1u) ? w : 0.f;
        out[row * ldo + col] = post(row, col, w);
    }
    if (EP == 1) mask[t.tid] = m;
}

// ---- row-fastest partials (fused stages, flow_r4f.h): PART[wave][col][row] - a lane's accumulator of a column group IS the
// float4 (rows 0 .. 3) of its column, so the partial product goes out as G 16-byte stores instead of 4 G dword stores (and the
// compiler does not first move 4 G accumulation registers into VGPRs).  Output o = 256 i + tid of a thread is then
// (col = o / 4, row = o % 4); its partials still sit at part[w 4 N + o], so r4_read_partials applies unchanged.
template <int G>
__device__ __forceinline__ void r4_store_part_rf(const f32x4 (&acc)[G], float* __restrict__ part, const Tid4& t) {
    f32x4* pw = reinterpret_cast<f32x4*>(part + (size_t)t.wave * R4 * 64 * G) + t.lane;
#pragma unroll
    for (int g = 0; g < G; ++g) pw[64 * g] = acc[g];
}
template <int G>
__device__ __forceinline__ void r4_bias_rf(float (&bv)[G], const float* __restrict__ b, const Tid4& t) {
#pragma unroll
    for (int i = 0; i < G; ++i) bv[i] = b[(256 * i + t.tid) >> 2];
}
template <int G, int EP>
__device__ __forceinline__ void r4_epilogue_rf(const float* __restrict__ part, const float (&bv)[G], float* __restrict__ out, int ldo,
                                               unsigned* mask, const Tid4& t) {
    unsigned m = EP == 2 ? mask[t.tid] : 0u;
    float v[G];
    r4_read_partials<G>(part, 64 * G, t.tid, v);
    const int row = t.tid & 3, col0 = t.tid >> 2;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        float w = v[i] + bv[i];
        if (EP == 1) { const bool pos = w > 0.f; m |= (pos ? 1u : 0u) << i; w = pos ? w : 0.f; }
        if (EP == 2) w = ((m >> i) & 1u) ? w : 0.f;
        out[row * ldo + col0 + 64 * i] = w;
    }
    if (EP == 1) mask[t.tid] = m;
}

// OUT[4][64 G] = epilogue(ACT[4][Wp] @ B)   (two workgroup barriers: partials visible / outputs visible).
// `bv`: the bias of this thread's outputs, already in registers (r4_bias_load one W x W stage earlier).
template <int NTWM, int G, int EP, class Next = R4NoNext>
__device__ __forceinline__ void r4_dense_wide(const float* act, int lda, const float4* Bm,
                                              const R4Pre<R4Ring<NTWM>::RD, G>& pre,
                                              const float (&bv)[G], float* out, int ldo, unsigned* mask, float* part, int PN,
                                              const Tid4& t, Next next = Next()) {
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    r4_mma_wide<NTWM, G>(act, lda, Bm, pre, t, acc, next);
    r4_store_part<G>(acc, part, PN, t);
    r4_barrier();
    r4_epilogue<G, EP>(part, PN, bv, out, ldo, mask, t);
    r4_barrier();
}

// ---- narrow outputs (the coupling parameters, N = 2 DOp; the d-wide input gradient, N = pad16(d)) of a K = Wp
// product: with 64-column groups half to three quarters of every weight tile would be padding, and these stages are
// bound by the bytes they stream.  Their tiles are dense instead: with CW = 16 NT output columns (NT = 1, 2, 4) a tile
// holds NSUB = 64 / CW consecutive k-quads side by side - lanes [s CW, (s + 1) CW) carry quad NSUB T + s - so the 16
// blocks of one v_mfma_f32_4x4x1 work on NSUB different k's at once and every lane sub-range accumulates its own partial
// product (4 NSUB partials per workgroup, added in a fixed order by the epilogue).  Tile T of wave w, lane l:
// { B[4 (4 NTWM w + NSUB T + l / CW) + j][l % CW] } j < 4; the wave owns NTWM NT tiles (kept as b[T / NT][T % NT]).
template <int NTWM, int NT>
struct R4PreT {
    float4 b[NTWM][NT];
};

template <int NTWM, int NT>
__device__ __forceinline__ void r4_preload_n16(R4PreT<NTWM, NT>& p, const float4* __restrict__ Bm, const Tid4& t) {
    const float4* bw = Bm + ((size_t)(NTWM * t.wave) * NT) * 64 + t.lane;
#pragma unroll
    for (int Q = 0; Q < NTWM; ++Q)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) p.b[Q][ct] = bw[(size_t)(Q * NT + ct) * 64];
}

// ACC: the result is added to `out` (the d-wide input gradient joins the cotangent it belongs to: no separate stage)
template <int NTWM, int NT, bool ACC = false, class Next = R4NoNext>
__device__ __forceinline__ void r4_dense_n16(const float* act, int lda, const R4PreT<NTWM, NT>& pre, float* out, int ldo,
                                             float* part, int PN, const Tid4& t, Next next = Next()) {
    constexpr int NSUB = 4 / NT, CW = 16 * NT, NTILE = NTWM * NT;
    static_assert(NT == 1 || NT == 2 || NT == 4, "16, 32 or 64 output columns");
    const int sblk = t.lane / CW, col = t.lane % CW;
    const float* arow = act + t.arow * lda + 16 * NTWM * t.wave + 4 * sblk;
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;                        // two chains: no back-to-back dependence
#pragma unroll
    for (int T = 0; T < NTILE; ++T) {
        const float4 a = *reinterpret_cast<const float4*>(arow + 4 * NSUB * T);
        const float4 w = pre.b[T / NT][T % NT];
        acc0 = mfma44(a.x, w.x, acc0);
        acc1 = mfma44(a.y, w.y, acc1);
        acc0 = mfma44(a.z, w.z, acc0);
        acc1 = mfma44(a.w, w.w, acc1);
    }
    next();
    acc0 += acc1;
    (void)PN;
    float* pw = part + ((size_t)(t.wave * NSUB + sblk) * R4) * CW + col;            // partial p = wave NSUB + sblk: [p][row][CW]
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[r * CW] = acc0[r];
    r4_barrier();
    if (t.tid < R4 * CW) {
        const int row = t.tid / CW, c = t.tid - row * CW;
        const float* p = part + row * CW + c;
        float v[4 * NSUB];
#pragma unroll
        for (int i = 0; i < 4 * NSUB; ++i) v[i] = p[(size_t)i * R4 * CW];
#pragma unroll
        for (int n = 4 * NSUB; n > 1; n >>= 1)                                      // fixed pairwise tree
#pragma unroll
            for (int i = 0; i < n / 2; ++i) v[i] = v[2 * i] + v[2 * i + 1];
        if (ACC) out[row * ldo + c] += v[0];
        else out[row * ldo + c] = v[0];
    }
    r4_barrier();
}

// short K (nqw <= NQ quads per wave), all tiles preloaded
template <int NQ, int G, int EP, class Next = R4NoNext, class Post = R4NoPost>
__device__ __forceinline__ void r4_dense_short(const float* act, int lda, int kmax, int nqw, const R4Pre<NQ, G>& pre,
                                               const float (&bv)[G], float* out, int ldo, unsigned* mask, float* part, int PN,
                                               const Tid4& t, Next next = Next(), Post post = Post()) {
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int q0 = nqw * t.wave;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q < nqw) {
            const int k0 = 4 * (q0 + q);
            float4 a = *reinterpret_cast<const float4*>(act + t.arow * lda + k0);
            if (k0 + 0 >= kmax) a.x = 0.f;
            if (k0 + 1 >= kmax) a.y = 0.f;
            if (k0 + 2 >= kmax) a.z = 0.f;
            if (k0 + 3 >= kmax) a.w = 0.f;
            r4_quad<G>(a, pre.b[q], acc);
        }
    }
    next();
    r4_store_part<G>(acc, part, PN, t);
    r4_barrier();
    r4_epilogue<G, EP>(part, PN, bv, out, ldo, mask, t, post);
    r4_barrier();
}

// ------------------------------------------------------------------------------------------------
// log q(x) and d log q / dx for the 4 rows in X0 (columns >= D zero); the gradient is left in the state buffer whose
// offset is returned through *grad_off.  Same arithmetic as flow_log_prob_tile<GRAD = true> (flow_device.h) up to the
// summation order inside the GEMMs.  Returns log q of row `tid >> 4` on wave 0 (other waves: undefined).
// ------------------------------------------------------------------------------------------------
// Per-stage request schedule (used for D > 32 or a hidden width of 64; the stream variant below covers the rest): the
// weights of the stages between two W x W GEMMs are requested at the START of the preceding W x W stage - forward W2(layer)
// -> { W3(layer), AW / W1 of layer - 1 }, reverse W2T(layer) -> { W1T, AWT of the layer, W3T of layer + 1 } - and the first
// RD quads of every W x W GEMM a few per short stage before it.  NQS / NQA: k-quads per wave of the short GEMMs into the
// hidden width / of the D x D maps (2 for D <= 32, 4 above); NT3 / NT1: 16-column units of the narrow outputs.
template <int NTWM, int NQS, int NQA, int NT3, int NT1>
__device__ float flow_log_prob_r4(const FlowDims& f, const R4Dims& rd, const R4Lds& l, const float* __restrict__ packed,
                                  float* lds, const Tid4& t, int* grad_off) {
    constexpr int G = NTWM;
    const float* r4base = packed + f.o_r4;
    int cur = l.o_X0, nxt = l.o_X1;
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
    float* PRM = lds + l.o_PRM;
    float* DP = lds + l.o_DP;
    float* PART = lds + l.o_PART;
    const bool ew = t.tid < 64;                       // element-wise stages: wave 0
    const int row = t.tid >> 4, c = t.tid & 15;
    const int nqD = rd.KD / 16, nqd = rd.Kd / 16, nqo = rd.Ko / 16;     // quads per wave of the short GEMMs
    const int qW = 4 * NTWM * t.wave;                                    // this wave's first quad of a K = Wp GEMM
    R4Pre<NQA, 1> preA;                // affine maps (AW, AWT)
    R4PreT<NTWM, NT3> preN3;           // W3 (16-column tiles), whole K range of the wave
    R4PreT<NTWM, NT1> preN1;           // W1T
    R4Pre<NQS, G> preS;                // short GEMMs into the hidden width (W1, W3T)
    constexpr int RD = R4Ring<NTWM>::RD;
    R4Pre<RD, G> preW;                 // first RD quads of the W x W GEMMs (W2, W2T), requested at the top of the layer
    float bvA[1], bv1[G], bv2[G], bv0[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bv0[g] = 0.f;
    float logq = 0.f;
    {                                  // the first layer's pre-W2 stages: requested here (one exposed latency per call)
        const float* Lp = packed + (size_t)(f.K - 1) * f.layer_stride;
        const float* Rp = r4base + (size_t)(f.K - 1) * rd.layer_stride;
        r4_preload<NQA, 1>(preA, reinterpret_cast<const float4*>(Rp + rd.o_AW), nqD * t.wave, nqD, t);
        r4_preload<NQS, G>(preS, reinterpret_cast<const float4*>(Rp + rd.o_W1), nqd * t.wave, nqd, t);
        r4_bias_load<1>(bvA, Lp + f.o_ac, t);
        r4_bias_load<G>(bv1, Lp + f.o_b1, t);
        r4_preload_part<0, 3>(preW, reinterpret_cast<const float4*>(Rp + rd.o_W2), qW, t);
    }
    for (int layer = f.K - 1; layer >= 0; --layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        const float* Rp = r4base + (size_t)layer * rd.layer_stride;
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        const bool tl = layer == f.K - 2;
        if (tl) FAB_TL(f, 0);
        // InvertibleAffine.inverse (+ folded ActNorm): z <- z @ W' + ac   (+ quads 3, 4 of this layer's W2 ring)
        r4_dense_short<NQA, 1, 0>(lds + cur, R4_DS, f.D, nqD, preA, bvA, lds + nxt, R4_DS, nullptr, PART, l.PN, t, [&] {
            r4_preload_part<3, 5>(preW, reinterpret_cast<const float4*>(Rp + rd.o_W2), qW, t);
        });
        logq += Lp[f.o_logS];
        float* Z = lds + nxt;
        if (tl) FAB_TL(f, 1);
        // conditioner
        r4_dense_short<NQS, G, 1>(Z, R4_DS, f.d, nqd, preS, bv1, HA, l.WS, mk, PART, l.PN, t, [&] {
            r4_preload_part<5, RD>(preW, reinterpret_cast<const float4*>(Rp + rd.o_W2), qW, t);
        });
        if (tl) FAB_TL(f, 2);
        float b3s[2] = {0.f, 0.f}, b3c[2] = {0.f, 0.f};      // coupling biases of this thread's columns (D - d <= 32)
        r4_bias_load<G>(bv2, Lp + f.o_b2, t);
        r4_dense_wide<NTWM, G, 1>(HA, l.WS, reinterpret_cast<const float4*>(Rp + rd.o_W2), preW, bv2, HB, l.WS,
                                  mk + NTHREADS, PART, l.PN, t, [&] {
            r4_preload_n16<NTWM, NT3>(preN3, reinterpret_cast<const float4*>(Rp + rd.o_W3), t);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int j = c + 16 * it;
                if (ew && j < f.DO) { b3s[it] = Lp[f.o_b3 + j]; b3c[it] = Lp[f.o_b3 + f.DOp + j]; }
            }
            if (layer > 0) {
                const float* Ln = Lp - f.layer_stride;
                const float* Rn = Rp - rd.layer_stride;
                r4_preload<NQA, 1>(preA, reinterpret_cast<const float4*>(Rn + rd.o_AW), nqD * t.wave, nqD, t);
                r4_preload<NQS, G>(preS, reinterpret_cast<const float4*>(Rn + rd.o_W1), nqd * t.wave, nqd, t);
                r4_bias_load<1>(bvA, Ln + f.o_ac, t);
                r4_bias_load<G>(bv1, Ln + f.o_b1, t);
            } else {                   // the reverse sweep starts at layer 0 with the (shift | scale) -> hidden GEMM
                r4_preload<NQS, G>(preS, reinterpret_cast<const float4*>(Rp + rd.o_W3T), nqo * t.wave, nqo, t);
            }
        });
        if (tl) FAB_TL(f, 3);
        const float4* Wnext = reinterpret_cast<const float4*>(layer > 0 ? Rp - rd.layer_stride + rd.o_W2 : Rp + rd.o_W2T);
        r4_dense_n16<NTWM, NT3>(HB, l.WS, preN3, PRM, R4_DS, PART, l.PN, t, [&] { r4_preload_part<0, 2>(preW, Wnext, qW, t); });
        if (tl) FAB_TL(f, 4);
        r4_preload_part<2, 3>(preW, Wnext, qW, t);
        // AffineCoupling.inverse: z2 <- (z2 - shift) exp(-s), log_det = -sum(s)
        if (ew) {
            float ssum = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int j = c + 16 * it;
                if (j < f.DO) {
                    const float shift = PRM[row * R4_DS + j] + b3s[it];
                    const float s = PRM[row * R4_DS + f.DOp + j] + b3c[it];
                    const float es = expf(-s);
                    const float v2 = (Z[row * R4_DS + f.d + j] - shift) * es;
                    Z[row * R4_DS + f.d + j] = v2;
                    lds[l.o_ES + ((size_t)layer * R4 + row) * f.DOp + j] = es;
                    lds[l.o_V2 + ((size_t)layer * R4 + row) * f.DOp + j] = v2;
                    ssum += s;
                }
            }
            logq += -row16_sum(ssum);
        }
        r4_barrier();
        if (tl) FAB_TL(f, 5);
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    // DiagGaussian.log_prob, and the seed of the reverse sweep
    if (ew) {
        const float* base = packed + f.o_base;
        float* Zc = lds + cur;
        float bsum = 0.f;
        for (int j = c; j < f.D; j += 16) {
            const float ls = base[f.Dp + j];
            const float sc = expf(ls);
            const float zn = (Zc[row * R4_DS + j] - base[j]) / sc;
            bsum += ls + 0.5f * (zn * zn);
            Zc[row * R4_DS + j] = -(zn / sc);
        }
        logq += -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
    }
    r4_barrier();
    // reverse sweep: g = d log q / d(state), layers 0 .. K-1
    bvA[0] = 0.f;
    for (int layer = 0; layer < f.K; ++layer) {
        const float* Rp = r4base + (size_t)layer * rd.layer_stride;
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        float* Gs = lds + cur;
        const bool tl = layer == 1;
        if (tl) FAB_TL(f, 16);
        r4_preload_part<3, 5>(preW, reinterpret_cast<const float4*>(Rp + rd.o_W2T), qW, t);
        if (layer == 0) {                             // (layers > 0: done by the previous layer's D x D epilogue, `couple` below)
            if (ew) {
                for (int j = c; j < f.DO; j += 16) {
                    const float g2 = Gs[row * R4_DS + f.d + j];
                    const float es = lds[l.o_ES + (size_t)row * f.DOp + j];
                    const float v2 = lds[l.o_V2 + (size_t)row * f.DOp + j];
                    DP[row * R4_DS + j] = -(g2 * es);
                    DP[row * R4_DS + f.DOp + j] = -(g2 * v2) - 1.f;
                    Gs[row * R4_DS + f.d + j] = g2 * es;
                }
            }
            r4_barrier();
        }
        // cotangents of the NEXT layer's coupling parameters, formed where its input gradient is produced
        auto couple = [&](int r, int col, float v) -> float {
            if (layer + 1 < f.K && col >= f.d && col < f.d + f.DO) {
                const int j = col - f.d;
                const float es = lds[l.o_ES + ((size_t)(layer + 1) * R4 + r) * f.DOp + j];
                const float v2 = lds[l.o_V2 + ((size_t)(layer + 1) * R4 + r) * f.DOp + j];
                DP[r * R4_DS + j] = -(v * es);
                DP[r * R4_DS + f.DOp + j] = -(v * v2) - 1.f;
                return v * es;
            }
            return v;
        };
        if (tl) FAB_TL(f, 17);
        r4_dense_short<NQS, G, 2>(DP, R4_DS, 2 * f.DOp, nqo, preS, bv0, HA, l.WS, mk + NTHREADS, PART, l.PN, t, [&] {
            r4_preload_part<5, RD>(preW, reinterpret_cast<const float4*>(Rp + rd.o_W2T), qW, t);
        });
        if (tl) FAB_TL(f, 18);
        r4_dense_wide<NTWM, G, 2>(HA, l.WS, reinterpret_cast<const float4*>(Rp + rd.o_W2T), preW, bv0, HB, l.WS, mk, PART,
                                  l.PN, t, [&] {
            r4_preload_n16<NTWM, NT1>(preN1, reinterpret_cast<const float4*>(Rp + rd.o_W1T), t);
            r4_preload<NQA, 1>(preA, reinterpret_cast<const float4*>(Rp + rd.o_AWT), nqD * t.wave, nqD, t);
            if (layer + 1 < f.K) {
                const float* Rn = Rp + rd.layer_stride;
                r4_preload<NQS, G>(preS, reinterpret_cast<const float4*>(Rn + rd.o_W3T), nqo * t.wave, nqo, t);
            }
        });
        if (tl) FAB_TL(f, 19);
        const float4* Wnext = reinterpret_cast<const float4*>(Rp + rd.layer_stride + rd.o_W2T);
        const bool more = layer + 1 < f.K;
        r4_dense_n16<NTWM, NT1, true>(HB, l.WS, preN1, Gs, R4_DS, PART, l.PN, t, [&] {
            if (more) r4_preload_part<0, 2>(preW, Wnext, qW, t);
        });                                           // g[:, :d] += (conditioner input gradient)
        if (tl) FAB_TL(f, 20);
        if (tl) FAB_TL(f, 21);
        r4_dense_short<NQA, 1, 0>(Gs, R4_DS, f.D, nqD, preA, bvA, lds + nxt, R4_DS, nullptr, PART, l.PN, t, [&] {
            if (more) r4_preload_part<2, 3>(preW, Wnext, qW, t);
        }, couple);
        if (tl) FAB_TL(f, 22);
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    *grad_off = cur;
    return logq;
}


// ================================================================================================
// One continuous weight stream per wave (shapes with D <= 32 and Wp >= 128).
// The per-stage request groups above leave the memory pipe idle in the short stages.  Here the tiles of ALL matrices
// are laid out once more in the order a wave consumes them (k_pack_r4s: "items" of G float4 per lane = one k-quad of a
// 64 G-wide matrix, or G tiles of a narrow one; forward layers K-1 .. 0, then reverse layers 0 .. K-1), and every wave
// keeps a ring of RD items: whenever an item has been consumed, the item RD places further down the stream is requested
// into its slot - across stage and layer boundaries - so RD - 1 items (25 KB per wave) are in flight at all times.
// Per layer and direction the stream has C = 4 NTWM + 4 items (forward: AW | W1 | 4 NTWM quads of W2 | 2 of W3;
// reverse: 2 of W3T | 4 NTWM quads of W2T | W1T | AWT); RD divides C, so the slot of every item is a compile-time
// constant inside the layer body.
// ================================================================================================
template <int NTWM>
struct R4Stream {
    static constexpr int G = NTWM;
    static constexpr int C = 4 * NTWM + 4;
    static constexpr int RD = C % 6 == 0 ? 6 : (C % 8 == 0 ? 8 : 5);
    static constexpr int IS = 4 * G * 64;                  // float4 between consecutive items of one wave
    // forward items                                          reverse items
    static constexpr int IA = 0, IW1 = 1, IW2 = 2, IW3 = 2 + 4 * NTWM;
    static constexpr int IW3T = 0, IW2T = 2, IW1T = 2 + 4 * NTWM, IAT = 3 + 4 * NTWM;
    static_assert(C % RD == 0, "ring size must divide the items per layer");
};

// W x W stage on the shared ring: quad qq of the stage is item I0 + qq
template <int NTWM, int EP, int I0, class Ring, class Refill>
__device__ __forceinline__ void r4s_dense_wide(const float* act, int lda, Ring& ring, Refill refill,
                                               const float (&bv)[NTWM], float* out, int ldo, unsigned* mask, float* part,
                                               int PN, const Tid4& t) {
    constexpr int G = NTWM, NQ = 4 * NTWM, RD = R4Stream<NTWM>::RD;
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow = act + t.arow * lda + 16 * NTWM * t.wave;
    static_for<0, NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float4 a = *reinterpret_cast<const float4*>(arow + 4 * q);
        r4_quad<G>(a, ring[(I0 + q) % RD], acc);
        refill(IC<I0 + q>{});
        __builtin_amdgcn_sched_barrier(0);
    });
    r4_store_part<G>(acc, part, PN, t);
    r4_barrier();
    r4_epilogue<G, EP>(part, PN, bv, out, ldo, mask, t);
    r4_barrier();
}

template <int NTWM>
__device__ float flow_log_prob_r4s(const FlowDims& f, const R4Dims& rd, const R4Lds& l, const float* __restrict__ packed,
                                   float* lds, const Tid4& t, int* grad_off) {
    using S = R4Stream<NTWM>;
    constexpr int G = NTWM, RD = S::RD, IS = S::IS;
    int cur = l.o_X0, nxt = l.o_X1;
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
    float* PRM = lds + l.o_PRM;
    float* DP = lds + l.o_DP;
    float* PART = lds + l.o_PART;
    const bool ew = t.tid < 64;
    const int row = t.tid >> 4, c = t.tid & 15;
    const int nqD = rd.KD / 16;                        // 1 or 2 k-quads per wave in the D x D maps (D <= 32)
    const float4* sp = reinterpret_cast<const float4*>(packed + f.o_r4s) + ((size_t)t.wave * G) * 64 + t.lane;
    float4 ring[RD][G];
#pragma unroll
    for (int i = 0; i < RD; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) ring[i][g] = sp[(size_t)i * IS + g * 64];
    auto refill = [&](auto ic) {                       // item I of the current layer was consumed: request item I + RD
        constexpr int I = decltype(ic)::value;
#pragma unroll
        for (int g = 0; g < G; ++g) ring[I % RD][g] = sp[(size_t)(I + RD) * IS + g * 64];
    };
    float bvA[1], bv1[G], bv2[G], bv0[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bv0[g] = 0.f;
    float logq = 0.f;
    {
        const float* Lp = packed + (size_t)(f.K - 1) * f.layer_stride;
        r4_bias_load<1>(bvA, Lp + f.o_ac, t);
        r4_bias_load<G>(bv1, Lp + f.o_b1, t);
    }
    for (int layer = f.K - 1; layer >= 0; --layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        const bool tl = layer == f.K - 2;
        if (tl) FAB_TL(f, 0);
        {   // InvertibleAffine.inverse (+ folded ActNorm): z <- z @ W' + ac
            R4Pre<2, 1> pa;
            pa.b[0][0] = ring[S::IA % RD][0]; pa.b[1][0] = ring[S::IA % RD][1];
            refill(IC<S::IA>{});
            r4_dense_short<2, 1, 0>(lds + cur, R4_DS, f.D, nqD, pa, bvA, lds + nxt, R4_DS, nullptr, PART, l.PN, t);
        }
        logq += Lp[f.o_logS];
        float* Z = lds + nxt;
        if (tl) FAB_TL(f, 1);
        {   // conditioner, first layer (K = 16: one k-quad per wave)
            R4Pre<1, G> p1;
#pragma unroll
            for (int g = 0; g < G; ++g) p1.b[0][g] = ring[S::IW1 % RD][g];
            refill(IC<S::IW1>{});
            r4_dense_short<1, G, 1>(Z, R4_DS, f.d, 1, p1, bv1, HA, l.WS, mk, PART, l.PN, t);
        }
        if (tl) FAB_TL(f, 2);
        float b3s[2] = {0.f, 0.f}, b3c[2] = {0.f, 0.f};
        r4_bias_load<G>(bv2, Lp + f.o_b2, t);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int j = c + 16 * it;
            if (ew && j < f.DO) { b3s[it] = Lp[f.o_b3 + j]; b3c[it] = Lp[f.o_b3 + f.DOp + j]; }
        }
        if (layer > 0) {
            const float* Ln = Lp - f.layer_stride;
            r4_bias_load<1>(bvA, Ln + f.o_ac, t);
            r4_bias_load<G>(bv1, Ln + f.o_b1, t);
        }
        r4s_dense_wide<NTWM, 1, S::IW2>(HA, l.WS, ring, refill, bv2, HB, l.WS, mk + NTHREADS, PART, l.PN, t);
        if (tl) FAB_TL(f, 3);
        {   // coupling parameters: N NTWM k-tiles x 2 column tiles = the 2 items after W2
            R4PreT<NTWM, 2> p3;
#pragma unroll
            for (int Q = 0; Q < NTWM; ++Q)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) p3.b[Q][ct] = ring[(S::IW3 + (2 * Q + ct) / G) % RD][(2 * Q + ct) % G];
            refill(IC<S::IW3>{});
            refill(IC<S::IW3 + 1>{});
            r4_dense_n16<NTWM, 2>(HB, l.WS, p3, PRM, R4_DS, PART, l.PN, t);
        }
        if (tl) FAB_TL(f, 4);
        if (ew) {
            float ssum = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int j = c + 16 * it;
                if (j < f.DO) {
                    const float shift = PRM[row * R4_DS + j] + b3s[it];
                    const float s = PRM[row * R4_DS + f.DOp + j] + b3c[it];
                    const float es = expf(-s);
                    const float v2 = (Z[row * R4_DS + f.d + j] - shift) * es;
                    Z[row * R4_DS + f.d + j] = v2;
                    lds[l.o_ES + ((size_t)layer * R4 + row) * f.DOp + j] = es;
                    lds[l.o_V2 + ((size_t)layer * R4 + row) * f.DOp + j] = v2;
                    ssum += s;
                }
            }
            logq += -row16_sum(ssum);
        }
        r4_barrier();
        if (tl) FAB_TL(f, 5);
        sp += (size_t)S::C * IS;
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    if (ew) {
        const float* base = packed + f.o_base;
        float* Zc = lds + cur;
        float bsum = 0.f;
        for (int j = c; j < f.D; j += 16) {
            const float ls = base[f.Dp + j];
            const float sc = expf(ls);
            const float zn = (Zc[row * R4_DS + j] - base[j]) / sc;
            bsum += ls + 0.5f * (zn * zn);
            Zc[row * R4_DS + j] = -(zn / sc);
        }
        logq += -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
    }
    r4_barrier();
    bvA[0] = 0.f;
    for (int layer = 0; layer < f.K; ++layer) {
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        float* Gs = lds + cur;
        const bool tl = layer == 1;
        if (tl) FAB_TL(f, 16);
        if (layer == 0) {                             // (layers > 0: done by the previous layer's D x D epilogue, `couple` below)
            if (ew) {
                for (int j = c; j < f.DO; j += 16) {
                    const float g2 = Gs[row * R4_DS + f.d + j];
                    const float es = lds[l.o_ES + (size_t)row * f.DOp + j];
                    const float v2 = lds[l.o_V2 + (size_t)row * f.DOp + j];
                    DP[row * R4_DS + j] = -(g2 * es);
                    DP[row * R4_DS + f.DOp + j] = -(g2 * v2) - 1.f;
                    Gs[row * R4_DS + f.d + j] = g2 * es;
                }
            }
            r4_barrier();
        }
        // cotangents of the NEXT layer's coupling parameters, formed where its input gradient is produced
        auto couple = [&](int r, int col, float v) -> float {
            if (layer + 1 < f.K && col >= f.d && col < f.d + f.DO) {
                const int j = col - f.d;
                const float es = lds[l.o_ES + ((size_t)(layer + 1) * R4 + r) * f.DOp + j];
                const float v2 = lds[l.o_V2 + ((size_t)(layer + 1) * R4 + r) * f.DOp + j];
                DP[r * R4_DS + j] = -(v * es);
                DP[r * R4_DS + f.DOp + j] = -(v * v2) - 1.f;
                return v * es;
            }
            return v;
        };
        if (tl) FAB_TL(f, 17);
        {   // (shift | scale) -> hidden: K = 32, two k-quads per wave = items 0, 1
            R4Pre<2, G> ps;
#pragma unroll
            for (int g = 0; g < G; ++g) { ps.b[0][g] = ring[S::IW3T % RD][g]; ps.b[1][g] = ring[(S::IW3T + 1) % RD][g]; }
            refill(IC<S::IW3T>{});
            refill(IC<S::IW3T + 1>{});
            r4_dense_short<2, G, 2>(DP, R4_DS, 2 * f.DOp, 2, ps, bv0, HA, l.WS, mk + NTHREADS, PART, l.PN, t);
        }
        if (tl) FAB_TL(f, 18);
        r4s_dense_wide<NTWM, 2, S::IW2T>(HA, l.WS, ring, refill, bv0, HB, l.WS, mk, PART, l.PN, t);
        if (tl) FAB_TL(f, 19);
        {   // hidden -> d: NTWM k-tiles x 1 column tile = one item
            R4PreT<NTWM, 1> p1t;
#pragma unroll
            for (int Q = 0; Q < NTWM; ++Q) p1t.b[Q][0] = ring[S::IW1T % RD][Q];
            refill(IC<S::IW1T>{});
            r4_dense_n16<NTWM, 1, true>(HB, l.WS, p1t, Gs, R4_DS, PART, l.PN, t);     // g[:, :d] += (conditioner input gradient)
        }
        if (tl) FAB_TL(f, 20);
        if (tl) FAB_TL(f, 21);
        {
            R4Pre<2, 1> pa;
            pa.b[0][0] = ring[S::IAT % RD][0]; pa.b[1][0] = ring[S::IAT % RD][1];
            refill(IC<S::IAT>{});
            r4_dense_short<2, 1, 0>(Gs, R4_DS, f.D, nqD, pa, bvA, lds + nxt, R4_DS, nullptr, PART, l.PN, t, R4NoNext(), couple);
        }
        if (tl) FAB_TL(f, 22);
        sp += (size_t)S::C * IS;
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    *grad_off = cur;
    return logq;
}

// ================================================================================================
// x, log q = flow.sample(eps) for the 4 rows whose base noise is in X0 (columns >= D zero): the SAMPLING direction on the same
// 4-chain tiles and stage code (r4: the flow sample of a <= 1152-chain AIS call ran on 64 workgroups of 16 chains for 110 us).
// Stream section 3 of the r4s image (k_pack_r4s: after the forward and reverse sections and their 8 padding items), layers
// 0 .. K-1, C items per layer in consumption order: W1 | 4 NTWM quads of W2 | 2 of W3 | W'^-1 (the forward D x D map).
// Arithmetic of flow_sample_tile (flow_device.h) stage by stage; the sums inside the products are the 4-chain tiles' (4 waves
// K-split, ((P0 + P1) + (P2 + P3)) + bias), so x agrees with the 16-chain kernel to fp32 rounding, not bit for bit.
// Leaves x in the state buffer at *x_off (leading dimension R4_DS) and returns log q of row tid >> 4 (threads < 64).
// ================================================================================================
template <int NTWM>
__device__ float flow_sample_r4s(const FlowDims& f, const R4Dims& rd, const R4Lds& l, const float* __restrict__ packed,
                                 float* lds, const Tid4& t, int* x_off) {
    using S = R4Stream<NTWM>;
    constexpr int G = NTWM, RD = S::RD, IS = S::IS;
    constexpr int JW1 = 0, JW2 = 1, JW3 = 1 + 4 * NTWM, JA = 3 + 4 * NTWM;      // items of a sampling layer
    static_assert(JA + 1 == S::C, "a sampling layer has as many items as a density layer");
    int cur = l.o_X0, nxt = l.o_X1;
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
    float* PRM = lds + l.o_PRM;
    float* PART = lds + l.o_PART;
    const bool ew = t.tid < 64;
    const int row = t.tid >> 4, c = t.tid & 15;
    const int nqD = rd.KD / 16;
    const float4* sp = reinterpret_cast<const float4*>(packed + f.o_r4s) +
                       ((size_t)(2 * f.K * S::C + 8) * 4 * G + (size_t)t.wave * G) * 64 + t.lane;
    float4 ring[RD][G];
#pragma unroll
    for (int i = 0; i < RD; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) ring[i][g] = sp[(size_t)i * IS + g * 64];
    auto refill = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
#pragma unroll
        for (int g = 0; g < G; ++g) ring[I % RD][g] = sp[(size_t)(I + RD) * IS + g * 64];
    };
    float bvA[1], bv1[G], bv2[G];
    float logq = 0.f;
    if (ew) {                                          // z = loc + exp(log_scale) eps ; log N(eps)
        const float* base = packed + f.o_base;
        float* Z = lds + cur;
        float bsum = 0.f;
        for (int j = c; j < f.D; j += 16) {
            const float e = Z[row * R4_DS + j];
            const float ls = base[f.Dp + j];
            Z[row * R4_DS + j] = base[j] + expf(ls) * e;
            bsum += ls + 0.5f * (e * e);
        }
        logq = -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
    }
    r4_bias_load<G>(bv1, packed + f.o_b1, t);
    r4_barrier();
    for (int layer = 0; layer < f.K; ++layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;   // (ReLU signs: scratch here)
        float* Z = lds + cur;
        {   // conditioner, first layer (K = 16: one k-quad per wave)
            R4Pre<1, G> p1;
#pragma unroll
            for (int g = 0; g < G; ++g) p1.b[0][g] = ring[JW1 % RD][g];
            refill(IC<JW1>{});
            r4_dense_short<1, G, 1>(Z, R4_DS, f.d, 1, p1, bv1, HA, l.WS, mk, PART, l.PN, t);
        }
        float b3s[2] = {0.f, 0.f}, b3c[2] = {0.f, 0.f};
        r4_bias_load<G>(bv2, Lp + f.o_b2, t);
        r4_bias_load<1>(bvA, Lp + f.o_at, t);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int j = c + 16 * it;
            if (ew && j < f.DO) { b3s[it] = Lp[f.o_b3 + j]; b3c[it] = Lp[f.o_b3 + f.DOp + j]; }
        }
        if (layer + 1 < f.K) r4_bias_load<G>(bv1, Lp + f.layer_stride + f.o_b1, t);
        r4s_dense_wide<NTWM, 1, JW2>(HA, l.WS, ring, refill, bv2, HB, l.WS, mk + NTHREADS, PART, l.PN, t);
        {   // coupling parameters
            R4PreT<NTWM, 2> p3;
#pragma unroll
            for (int Q = 0; Q < NTWM; ++Q)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) p3.b[Q][ct] = ring[(JW3 + (2 * Q + ct) / G) % RD][(2 * Q + ct) % G];
            refill(IC<JW3>{});
            refill(IC<JW3 + 1>{});
            r4_dense_n16<NTWM, 2>(HB, l.WS, p3, PRM, R4_DS, PART, l.PN, t);
        }
        if (ew) {
            float ssum = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int j = c + 16 * it;
                if (j < f.DO) {
                    const float shift = PRM[row * R4_DS + j] + b3s[it];
                    const float s = PRM[row * R4_DS + f.DOp + j] + b3c[it];
                    Z[row * R4_DS + f.d + j] = Z[row * R4_DS + f.d + j] * expf(s) + shift;
                    ssum += s;
                }
            }
            logq -= row16_sum(ssum);
        }
        r4_barrier();
        {   // InvertibleAffine.forward (+ folded ActNorm): z <- z @ W'^-1 + at, log_det = -sum(log_S)
            R4Pre<2, 1> pa;
            pa.b[0][0] = ring[JA % RD][0]; pa.b[1][0] = ring[JA % RD][1];
            refill(IC<JA>{});
            r4_dense_short<2, 1, 0>(Z, R4_DS, f.D, nqD, pa, bvA, lds + nxt, R4_DS, nullptr, PART, l.PN, t);
        }
        logq -= -Lp[f.o_logS];
        sp += (size_t)S::C * IS;
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    *x_off = cur;
    return logq;
}

}  // namespace fab
