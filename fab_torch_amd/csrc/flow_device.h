// Device-side RealNVP evaluation on a 16-chain tile held by one 256-thread workgroup.
//
// Every linear map is a [16 x K] @ [K x N] product on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, bit-identical to an fmaf chain): activations live in LDS
// (row-major, leading dimension padded by 4 floats), weights are streamed from the L2-resident
// packed image straight into VGPRs (each weight is used by exactly one wave of the workgroup, so an
// LDS round trip would be pure overhead) through a software prefetch ring.
//
// Packed B-operand tile (c, S) of a K x N matrix: 64 lanes x float4, lane l = (q = l>>4, n = l&15)
// holds  B[16S + 4q + t][16c + n], t = 0..3.  MFMA step t of k-block S therefore multiplies
// A[row][16S + 4q + t] (one ds_read_b128 per lane per k-block) with that register.
//
// Two kinds of loads: (1) the weight ring of a main loop uses loads hipcc does not track (inline asm +
// hand-counted vmcnt); registers that are the target of such an in-flight load must not be live across code
// hipcc is free to re-allocate / spill (it copies them before the data lands), so a ring never outlives its
// loop.  (2) Everything that has to arrive BEFORE its stage starts (short-K weights, D x D maps, biases, the
// first k-block of the next ring, the first K-split tiles) uses plain compiler-tracked loads, which may stay live
// across any code; they are issued a few at a time behind the MFMAs of the previous W x W main loop, never as a
// burst (the CU's vector-memory path takes ~16 cycles per 1-KiB wave load).
#pragma once
#include <type_traits>
#include "fabhip_common.h"

namespace fab {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// lane i <- lane (i + N) mod 16 of its 16-lane row: one DPP move (row_ror), no LDS crossbar (ds_bpermute, which is what
// __shfl_xor compiles to, costs an LDS round trip per step)
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    // sum over the 16 lanes that share a chain row (tid = row*16 + c); fixed tree => deterministic.  The xor-butterfly 8, 4, 2, 1
    // as rotations: after the first step the values have period 8, so "+ 4 mod 16" reads the lane "xor 4" would (and so on):
    // bit for bit the sums of the __shfl_xor form.
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}

// ReLU epilogue helpers.  The sign pattern of a lane's 4*NTWM outputs is kept as one 32-bit word, first output in
// bit 31: built with add-with-carry (m <- 2m + [v > 0], three VALU ops per value sharing one compare) and consumed
// in the reverse sweep with add-carry-out (carry = top bit, m <- 2m: two ops per value).
__device__ __forceinline__ float relu_bit(float v, unsigned& m) {
    float o;
    asm("v_cmp_lt_f32 vcc, 0, %2\n\tv_cndmask_b32 %0, 0, %2, vcc\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
        : "=&v"(o), "+v"(m)
        : "v"(v)
        : "vcc");
    return o;                                   // v > 0 ? v : 0  (NaN -> 0, like the plain expression)
}
__device__ __forceinline__ float mask_bit(float v, unsigned& m) {
    float o;
    asm("v_add_co_u32 %1, vcc, %1, %1\n\tv_cndmask_b32 %0, 0, %2, vcc" : "=&v"(o), "+v"(m) : "v"(v) : "vcc");
    return o;                                   // top bit of m ? v : 0, m <<= 1
}

// dev-only timeline: workgroup 0 / thread 0 records s_memtime at stage boundaries of one layer
#define FAB_TL(f, idx) do { if ((f).timeline && blockIdx.x == 0 && threadIdx.x == 0) (f).timeline[idx] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

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

// ---- loads hidden from hipcc's s_waitcnt bookkeeping ---------------------------------------------
// hipcc drains vmcnt(0) at the head of every loop iteration that carries loads in flight (and in front of
// every barrier), which serialises a register prefetch ring.  Weight / bias loads are therefore issued
// through inline asm and waited for with hand-counted s_waitcnt vmcnt(N) (loads return in order, so N = the
// number of hidden loads issued after the one needed; any other load in flight only makes the wait more
// conservative).  Every wait names the destination registers as operands so that hipcc keeps them allocated
// until the data has landed.
__device__ __forceinline__ void gload16s(f32x4& dst, unsigned voff, const float4* sbase) {   // SGPR base + lane offset
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
// First load of a group that uses a freshly computed scalar base: hipcc pads no hazards for instructions
// inside an asm statement, and an SGPR written by a VALU op (v_readlane / v_readfirstlane, e.g. an SGPR-spill
// restore) needs 5 wait states before a VMEM instruction reads it as its base.
__device__ __forceinline__ void gload16s_first(f32x4& dst, unsigned voff, const float4* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
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

// depth of the weight ring of the W x W GEMMs (must divide KBW = 4 * NTWM)
template <int NTWM>
__host__ __device__ constexpr int depth_w() { return NTWM == 8 ? 2 : 4; }

// ---- N-split GEMM: wave w owns column tiles c_i = w + 4 i, i < NTWM (the tile count is padded to
// 4*NTWM with zero tiles at pack time, so the hot loop has no predication at all).
// acc[i] (+)= A[16 x 16*KB] @ B[:, tile c_i];  KB % DEPTH == 0 (K padded at pack time).
// WRing = the wave's register prefetch ring (DEPTH k-blocks x NTWM tiles) + the bias of its columns.
template <int NTWM, int DEPTH>
struct WRing {
    f32x4 r[DEPTH][NTWM];
    float bv[NTWM];
};

template <int NTWM>
__device__ __forceinline__ void tile_offsets(unsigned (&voff)[NTWM], int KB, const Tid& t) {
#pragma unroll
    for (int i = 0; i < NTWM; ++i) voff[i] = (unsigned)(((size_t)i * 4 * KB * 64 + t.lane) * 16);
}

// issue the ring prologue (k-blocks 0 .. DEPTH-1) and, with BIAS, the bias loads (issued first = oldest)
template <int NTWM, int DEPTH, bool BIAS>
__device__ __forceinline__ void ring_issue(WRing<NTWM, DEPTH>& w, const float4* __restrict__ Bp, int KB,
                                           const float* __restrict__ bias, const Tid& t) {
    const float4* bw = Bp + (size_t)t.wave * KB * 64;              // wave-uniform base of this wave's tile 0
    unsigned voff[NTWM];
    tile_offsets<NTWM>(voff, KB, t);
    if (BIAS) {
#pragma unroll
        for (int i = 0; i < NTWM; ++i) gload4(w.bv[i], bias + 16 * (t.wave + 4 * i) + t.n);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const float4* sb = bw + (size_t)d * 64;
        gload16s_first(w.r[d][0], voff[0], sb);
#pragma unroll
        for (int i = 1; i < NTWM; ++i) gload16s(w.r[d][i], voff[i], sb);
    }
}

// k-block 0 of a W x W GEMM (+ its bias), requested with plain loads one stage early: the main loop (ring_run_pre)
// then starts on data that is already there and issues the hidden loads of blocks 1 .. DEPTH-1 behind its MFMAs.
template <int NTWM>
struct RingPre {
    float4 b[NTWM];
    float bv[NTWM];
};

template <int NTWM, bool BIAS>
__device__ __forceinline__ void ringpre_load(RingPre<NTWM>& p, const float4* __restrict__ Bp, int KB,
                                             const float* __restrict__ bias, const Tid& t) {
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        p.b[i] = Bp[((size_t)(t.wave + 4 * i) * KB) * 64 + t.lane];
        p.bv[i] = BIAS ? bias[16 * (t.wave + 4 * i) + t.n] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
}

// main loop on a ring whose prologue was issued earlier and after which NO other hidden load was issued.
// Loop schedule per k-block S (ring slot d = S mod DEPTH), pinned with sched_barrier:
//   wait slot d  |  NTWM MFMA (a.x)  |  refill the slot of block S-1 with block S-1+DEPTH  |  3 NTWM MFMA
// so the refill's address arithmetic and load issue run in the shadow of the matrix pipe.
template <int NTWM, int DEPTH, bool MASKK, bool BIAS>
__device__ __forceinline__ void ring_run(WRing<NTWM, DEPTH>& w, const float* __restrict__ A, int lda, int kmax,
                                         int KB, const float4* __restrict__ Bp, const Tid& t, f32x4 (&acc)[NTWM]) {
    static_assert(DEPTH >= 2, "ring depth");
    const float* arow = A + t.n * lda + 4 * t.q;
    const float4* bw = Bp + (size_t)t.wave * KB * 64;
    unsigned voff[NTWM];
    tile_offsets<NTWM>(voff, KB, t);
    float4 a_nxt = *reinterpret_cast<const float4*>(arow);
    if (BIAS) {
        wait_vals<DEPTH * NTWM, NTWM>(w.bv);                       // the ring (younger) may stay in flight
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){w.bv[i], w.bv[i], w.bv[i], w.bv[i]};
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
            // slot d was (re)filled during block S-DEPTH+1; DEPTH-2 younger refill groups may stay in flight
            wait_tiles<(DEPTH - 2) * NTWM, NTWM>(w.r[d]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.x, w.r[d][i].x, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
            // refill the slot consumed by block S-1 with block S-1+DEPTH (clamped: the tail re-reads the last
            // block, which keeps the in-flight count constant so the hand-counted vmcnt stays exact) - ONE load
            // behind each MFMA of the a.y group, so that no gap between two MFMAs carries more than the matrix
            // pipe's 32-cycle shadow hides (a block of 5 loads + address arithmetic stalled it ~100 cycles/k-block)
            {
                const int dp = (d + DEPTH - 1) % DEPTH;      // static after unrolling
                const int Sp = (S - 1 + DEPTH < KB) ? S - 1 + DEPTH : KB - 1;
                const float4* sb = bw + (size_t)Sp * 64;
#pragma unroll
                for (int i = 0; i < NTWM; ++i) {
                    acc[i] = mfma4(a.y, w.r[d][i].y, acc[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (S > 0) {
                        if (i == 0) gload16s_first(w.r[dp][0], voff[0], sb);
                        else gload16s(w.r[dp][i], voff[i], sb);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.z, w.r[d][i].z, acc[i]);
#pragma unroll
            for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.w, w.r[d][i].w, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // Refills (incl. the clamped tail) are still in flight.  Drain them with the ring registers as operands
    // of the wait: otherwise hipcc, which believes they are dead, may re-allocate them above the wait and the
    // landing loads would clobber live values.
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) wait_tiles<0, NTWM>(w.r[d]);
    __builtin_amdgcn_sched_barrier(0);
}

// ---- W x W main loop, software-pipelined across stage boundaries -------------------------------------------------
// Used when k-block 0 (+ bias) is already in ring slot 0 (RingPre, plain loads of the previous stage).  The vector
// memory path of a CU takes ~16 cycles per 1-KiB wave load, so a burst of 15-25 loads costs ~1000 cycles when all
// four waves issue it at once with the matrix pipe idle (measured: ring prologue 1.0k, post-loop requests 1.15k).
// Here nothing is issued in a burst:
//   * block 0 runs straight away and the hidden loads of blocks 1 .. DEPTH-1 go out one behind each of its MFMAs;
//   * the requests of LATER stages (compiler-tracked plain loads, `inj(integral_constant<j>)`, j < NI) go out PER
//     k-block a few at a time behind the MFMAs of the last k-blocks (fully unrolled tail: static register targets).
// Extra loads in flight only make the hand-counted vmcnt waits more conservative (loads return in order).
struct NoInject {
    template <class J>
    __device__ __forceinline__ void operator()(J) const {}
};

template <int V>
struct IC {
    static constexpr int value = V;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>());
        static_for<I + 1, N>(f);
    }
}

// one k-block.  MODE 1: first block (data present, no wait; issues the prologue of slots 1 .. DEPTH-1);
// MODE 0: steady state (wait slot D, refill slot D-1 with block S-1+DEPTH, NJ injected requests from index JB)
template <int NTWM, int DEPTH, int D, int MODE, int NJ, int JB, class Inject>
__device__ __forceinline__ void kblock_pre(WRing<NTWM, DEPTH>& w, float4& a_nxt, const float* __restrict__ arow, int S,
                                           int KB, const float4* __restrict__ bw, const unsigned (&voff)[NTWM],
                                           f32x4 (&acc)[NTWM], Inject& inj) {
    const float4 a = a_nxt;
    const int Sn = (S + 1 < KB) ? S + 1 : S;
    a_nxt = *reinterpret_cast<const float4*>(arow + 16 * Sn);
    if (MODE == 0) wait_tiles<(DEPTH - 2) * NTWM, NTWM>(w.r[D]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {                      // group x
        acc[i] = mfma4(a.x, w.r[D][i].x, acc[i]);
        if (MODE == 1) {
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) gload16s_first(w.r[1][0], voff[0], bw + 64);
            else gload16s(w.r[1][i], voff[i], bw + 64);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        constexpr int dp = (D + DEPTH - 1) % DEPTH;
        const int Sp = (S - 1 + DEPTH < KB) ? S - 1 + DEPTH : KB - 1;
        const float4* sb = bw + (size_t)Sp * 64;
#pragma unroll
        for (int i = 0; i < NTWM; ++i) {                  // group y (+ refill / prologue slot 2)
            acc[i] = mfma4(a.y, w.r[D][i].y, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) {
                if (i == 0) gload16s_first(w.r[dp][0], voff[0], sb);
                else gload16s(w.r[dp][i], voff[i], sb);
            } else if (DEPTH > 2) {
                if (i == 0) gload16s_first(w.r[2 % DEPTH][0], voff[0], bw + 2 * 64);
                else gload16s(w.r[2 % DEPTH][i], voff[i], bw + 2 * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {                      // group z (+ prologue slot 3 / injected requests)
        acc[i] = mfma4(a.z, w.r[D][i].z, acc[i]);
        if (MODE == 1 && DEPTH > 3) {
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) gload16s_first(w.r[3 % DEPTH][0], voff[0], bw + 3 * 64);
            else gload16s(w.r[3 % DEPTH][i], voff[i], bw + 3 * 64);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static_assert(DEPTH == 2 || DEPTH == 4, "prologue interleave is written for ring depths 2 and 4");
    if (NJ > 0) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NJ > 0) inj(IC<JB>());
        if constexpr (NJ > 1) inj(IC<JB + 1>());
        if constexpr (NJ > 2) inj(IC<JB + 2>());
        if constexpr (NJ > 3) inj(IC<JB + 3>());
        if constexpr (NJ > 4) inj(IC<JB + 4>());
        if constexpr (NJ > 5) inj(IC<JB + 5>());
        static_assert(NJ <= 6, "at most 6 injected requests per k-block");
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.w, w.r[D][i].w, acc[i]);   // group w
    __builtin_amdgcn_sched_barrier(0);
}

template <int NTWM, int DEPTH, int G, int PER, bool INJECT, class Inject, int D = 0>
__device__ __forceinline__ void kgroup_pre(WRing<NTWM, DEPTH>& w, float4& a_nxt, const float* __restrict__ arow, int S0,
                                           int KB, const float4* __restrict__ bw, const unsigned (&voff)[NTWM],
                                           f32x4 (&acc)[NTWM], Inject& inj) {
    // steady-state group: slots D .. DEPTH-1; INJECT: tail group G (requests (G*DEPTH + d)*PER ..)
    if constexpr (D < DEPTH) {
        kblock_pre<NTWM, DEPTH, D, 0, INJECT ? PER : 0, (G * DEPTH + D) * PER>(w, a_nxt, arow, S0 + D, KB, bw, voff, acc,
                                                                              inj);
        kgroup_pre<NTWM, DEPTH, G, PER, INJECT, Inject, D + 1>(w, a_nxt, arow, S0, KB, bw, voff, acc, inj);
    }
}

template <int NTWM, int DEPTH, int PER, class Inject, int D = 1>
__device__ __forceinline__ void kfirst_rest(WRing<NTWM, DEPTH>& w, float4& a_nxt, const float* __restrict__ arow, int KB,
                                            const float4* __restrict__ bw, const unsigned (&voff)[NTWM],
                                            f32x4 (&acc)[NTWM], Inject& inj) {
    // blocks 1 .. DEPTH-1 of the first group; with a single group (KB == DEPTH) they carry the injected requests
    if constexpr (D < DEPTH) {
        kblock_pre<NTWM, DEPTH, D, 0, PER, (D - 1) * PER>(w, a_nxt, arow, D, KB, bw, voff, acc, inj);
        kfirst_rest<NTWM, DEPTH, PER, Inject, D + 1>(w, a_nxt, arow, KB, bw, voff, acc, inj);
    }
}

template <int NTWM, int DEPTH, int TG, int PER, class Inject, int G = 0>
__device__ __forceinline__ void ktail_pre(WRing<NTWM, DEPTH>& w, float4& a_nxt, const float* __restrict__ arow, int KB,
                                          const float4* __restrict__ bw, const unsigned (&voff)[NTWM],
                                          f32x4 (&acc)[NTWM], Inject& inj) {
    if constexpr (G < TG) {
        kgroup_pre<NTWM, DEPTH, G, PER, true>(w, a_nxt, arow, KB - (TG - G) * DEPTH, KB, bw, voff, acc, inj);
        ktail_pre<NTWM, DEPTH, TG, PER, Inject, G + 1>(w, a_nxt, arow, KB, bw, voff, acc, inj);
    }
}

// acc (pre-set by the caller) += A[16 x 64 NTWM] @ B, K = 64 NTWM (KB = 4 NTWM k-blocks), NI injected requests
template <int NTWM, int DEPTH, int NI, class Inject>
__device__ __forceinline__ void ring_run_pre(WRing<NTWM, DEPTH>& w, const float* __restrict__ A, int lda,
                                             const float4* __restrict__ Bp, const Tid& t, f32x4 (&acc)[NTWM],
                                             Inject& inj) {
    constexpr int KB = 4 * NTWM, NG = KB / DEPTH;
    static_assert(KB % DEPTH == 0, "ring depth must divide the k-block count");
    constexpr int TG = NG - 1 < 12 / DEPTH ? NG - 1 : 12 / DEPTH;   // fully unrolled tail groups: up to 12 k-blocks
    constexpr int TB = NG == 1 ? DEPTH - 1 : TG * DEPTH;  // k-blocks that carry injected requests
    constexpr int PER = (NI + TB - 1) / TB;
    const float* arow = A + t.n * lda + 4 * t.q;
    const float4* bw = Bp + (size_t)t.wave * KB * 64;
    unsigned voff[NTWM];
    tile_offsets<NTWM>(voff, KB, t);
    float4 a_nxt = *reinterpret_cast<const float4*>(arow);
    kblock_pre<NTWM, DEPTH, 0, 1, 0, 0>(w, a_nxt, arow, 0, KB, bw, voff, acc, inj);
    kfirst_rest<NTWM, DEPTH, NG == 1 ? PER : 0>(w, a_nxt, arow, KB, bw, voff, acc, inj);
    for (int S0 = DEPTH; S0 < KB - TG * DEPTH; S0 += DEPTH)
        kgroup_pre<NTWM, DEPTH, 0, 0, false>(w, a_nxt, arow, S0, KB, bw, voff, acc, inj);
    ktail_pre<NTWM, DEPTH, TG, PER>(w, a_nxt, arow, KB, bw, voff, acc, inj);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) wait_tiles<0, NTWM>(w.r[d]);
    __builtin_amdgcn_sched_barrier(0);
}

// ---- K-split GEMM for narrow outputs (N = 16*NT <= 64, K = 64*KW): wave w sums k-blocks S = w, w+4, ... of
// every column tile; the four partial [16 x 16*NT] products go to LDS part[w][row][PN] and are added by the
// caller.  Plain (compiler-tracked) loads in straight-line code: the weights of tile i+1 are requested before
// tile i is multiplied.  (An inline-asm double buffer was tried here and is unsafe: hipcc copies registers
// that are the target of an in-flight hidden load when their live ranges are split.)
template <int KW>
__device__ __forceinline__ void ksplit_load(float4 (&b)[KW], const float4* __restrict__ Bp, int tile, const Tid& t) {
    const float4* bt = Bp + ((size_t)tile * 4 * KW + t.wave) * 64 + t.lane;
#pragma unroll
    for (int s = 0; s < KW; ++s) b[s] = bt[(size_t)s * 4 * 64];
}

template <int KW>
__device__ __forceinline__ void ksplit_mul(const float4 (&b)[KW], const float* __restrict__ arow,
                                           float* __restrict__ pw, int PN, int tile, const Tid& t) {
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;               // two independent chains
#pragma unroll
    for (int s = 0; s < KW; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(arow + 16 * (t.wave + 4 * s));
        if (s & 1) {
            acc1 = mfma4(a.x, b[s].x, acc1); acc1 = mfma4(a.y, b[s].y, acc1);
            acc1 = mfma4(a.z, b[s].z, acc1); acc1 = mfma4(a.w, b[s].w, acc1);
        } else {
            acc0 = mfma4(a.x, b[s].x, acc0); acc0 = mfma4(a.y, b[s].y, acc0);
            acc0 = mfma4(a.z, b[s].z, acc0); acc0 = mfma4(a.w, b[s].w, acc0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[(4 * t.q + r) * PN + 16 * tile + t.n] = acc0[r] + acc1[r];
}

// weights of the first column tiles of a K-split GEMM, requested from inside the producer's main loop
// (plain, compiler-tracked loads: safe to keep live across any code)
template <int KW>
struct KsplitPre {
    float4 b0[KW], b1[KW];                      // b1 only where the producer injects it (PRE2)
};

template <int KW, bool PRE = false, bool PRE2 = false>
__device__ __forceinline__ void gemm_ksplit(const float* __restrict__ A, int lda, const float4* __restrict__ Bp,
                                            int NT, float* __restrict__ part, int PN, const Tid& t,
                                            const KsplitPre<KW>* pre = nullptr) {
    const float* arow = A + t.n * lda + 4 * t.q;
    float* pw = part + (size_t)t.wave * ROWS * PN;
    float4 b0[KW], b1[KW];
    if (PRE) {
#pragma unroll
        for (int s = 0; s < KW; ++s) b0[s] = pre->b0[s];
    } else {
        ksplit_load<KW>(b0, Bp, 0, t);
    }
    if (PRE2) {
#pragma unroll
        for (int s = 0; s < KW; ++s) b1[s] = pre->b1[s];
    } else if (NT > 1) {
        ksplit_load<KW>(b1, Bp, 1, t);
    }
    __builtin_amdgcn_sched_barrier(0);
    ksplit_mul<KW>(b0, arow, pw, PN, 0, t);
    if (NT > 1) {
        if (NT > 2) ksplit_load<KW>(b0, Bp, 2, t);
        __builtin_amdgcn_sched_barrier(0);
        ksplit_mul<KW>(b1, arow, pw, PN, 1, t);
        if (NT > 2) {
            if (NT > 3) ksplit_load<KW>(b1, Bp, 3, t);
            __builtin_amdgcn_sched_barrier(0);
            ksplit_mul<KW>(b0, arow, pw, PN, 2, t);
            if (NT > 3) ksplit_mul<KW>(b1, arow, pw, PN, 3, t);
        }
    }
}

__device__ __forceinline__ float part_sum(const float* part, int PN, int row, int col) {
    float s = part[row * PN + col];
    s += part[(ROWS + row) * PN + col];
    s += part[(2 * ROWS + row) * PN + col];
    s += part[(3 * ROWS + row) * PN + col];
    return s;
}

// ---- short-K GEMMs (K = 16 KB, KB <= 4: the first conditioner layer and the first reverse GEMM): all of the
// wave's weights fit in registers, so they are requested with plain (compiler-tracked) loads one stage EARLY and
// stay live across the stage in between and its barrier; the stage itself is then MFMAs + epilogue only.
template <int NTWM, int KB>
struct SmallW {
    float4 b[KB][NTWM];
    float bv[NTWM];
};

template <int NTWM, int KB, bool BIAS>
__device__ __forceinline__ void smallw_load(SmallW<NTWM, KB>& w, const float4* __restrict__ Bp,
                                            const float* __restrict__ bias, const Tid& t) {
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const float4* bt = Bp + ((size_t)(t.wave + 4 * i) * KB) * 64 + t.lane;
#pragma unroll
        for (int S = 0; S < KB; ++S) w.b[S][i] = bt[(size_t)S * 64];
        w.bv[i] = BIAS ? bias[16 * (t.wave + 4 * i) + t.n] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);          // the requests stay ahead of whatever follows
}

template <int NTWM, int KB, bool MASKK>
__device__ __forceinline__ void smallw_mul(const SmallW<NTWM, KB>& w, const float* __restrict__ A, int lda, int kmax,
                                           const Tid& t, f32x4 (&acc)[NTWM]) {
    const float* arow = A + t.n * lda + 4 * t.q;
#pragma unroll
    for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){w.bv[i], w.bv[i], w.bv[i], w.bv[i]};
#pragma unroll
    for (int S = 0; S < KB; ++S) {
        float4 a = *reinterpret_cast<const float4*>(arow + 16 * S);
        if (MASKK) {
            const int k0 = 16 * S + 4 * t.q;
            a.x = (k0 + 0 < kmax) ? a.x : 0.f;
            a.y = (k0 + 1 < kmax) ? a.y : 0.f;
            a.z = (k0 + 2 < kmax) ? a.z : 0.f;
            a.w = (k0 + 3 < kmax) ? a.w : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.x, w.b[S][i].x, acc[i]);
#pragma unroll
        for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.y, w.b[S][i].y, acc[i]);
#pragma unroll
        for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.z, w.b[S][i].z, acc[i]);
#pragma unroll
        for (int i = 0; i < NTWM; ++i) acc[i] = mfma4(a.w, w.b[S][i].w, acc[i]);
    }
}

// smallw_mul with ONE injected request behind each of its 4 KB NTWM MFMAs (`inj(IC<m>)`, m = running MFMA index):
// fast mode streams the bf16 weight slice of the W x W GEMM that FOLLOWS this short stage this way - a burst of 50
// loads blocks the issuing wave for ~3 k cycles (measured: the stage grew from 2.9 k to 6.2 k), one load per MFMA
// issues in the shadow of the matrix pipe.

template <int NTWM, int KB, bool MASKK, class Inject>
__device__ __forceinline__ void smallw_mul_inj(const SmallW<NTWM, KB>& w, const float* __restrict__ A, int lda, int kmax,
                                               const Tid& t, f32x4 (&acc)[NTWM], Inject& inj) {
    const float* arow = A + t.n * lda + 4 * t.q;
#pragma unroll
    for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){w.bv[i], w.bv[i], w.bv[i], w.bv[i]};
    float a[KB][4];
#pragma unroll
    for (int S = 0; S < KB; ++S) {
        const float4 v = *reinterpret_cast<const float4*>(arow + 16 * S);
        const int k0 = 16 * S + 4 * t.q;
        a[S][0] = (!MASKK || k0 + 0 < kmax) ? v.x : 0.f;
        a[S][1] = (!MASKK || k0 + 1 < kmax) ? v.y : 0.f;
        a[S][2] = (!MASKK || k0 + 2 < kmax) ? v.z : 0.f;
        a[S][3] = (!MASKK || k0 + 3 < kmax) ? v.w : 0.f;
    }
    static_for<0, 4 * KB * NTWM>([&](auto mc) {
        constexpr int m = decltype(mc)::value, S = m / (4 * NTWM), sub = (m / NTWM) % 4, i = m % NTWM;
        const float bb = sub == 0 ? w.b[S][i].x : (sub == 1 ? w.b[S][i].y : (sub == 2 ? w.b[S][i].z : w.b[S][i].w));
        acc[i] = mfma4(a[S][sub], bb, acc[i]);
        __builtin_amdgcn_sched_barrier(0);
        inj(mc);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// OUT = relu(A @ B + bias) with pre-loaded weights (same epilogue as dense_relu)
template <int NTWM, int KB, bool MASK, bool TAPE, class Inject = NoInject>
__device__ __forceinline__ void dense_relu_small(const SmallW<NTWM, KB>& w, const float* A, int lda, int kmax,
                                                 float* OUT, int ldo, unsigned* mask, const Tid& t,
                                                 float* __restrict__ gout, int ldg, Inject inj = Inject()) {
    f32x4 acc[NTWM];
    if constexpr (std::is_same<Inject, NoInject>::value) smallw_mul<NTWM, KB, true>(w, A, lda, kmax, t, acc);
    else smallw_mul_inj<NTWM, KB, true>(w, A, lda, kmax, t, acc, inj);
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const int c = t.wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[i][r];
            const float o = MASK ? relu_bit(v, m) : (v > 0.f ? v : 0.f);
            OUT[(4 * t.q + r) * ldo + 16 * c + t.n] = o;
            if (TAPE) gout[(long)(4 * t.q + r) * ldg + 16 * c + t.n] = o;
        }
    }
    if (MASK) mask[t.tid] = m << (32 - 4 * NTWM);
}

// OUT = (A @ B) * mask with pre-loaded weights (same epilogue as dense_masked)
template <int NTWM, int KB, bool TAPE, class Inject = NoInject>
__device__ __forceinline__ void dense_masked_small(const SmallW<NTWM, KB>& w, const float* A, int lda, float* OUT,
                                                   int ldo, const unsigned* mask, const Tid& t,
                                                   float* __restrict__ gout, int ldg, Inject inj = Inject()) {
    f32x4 acc[NTWM];
    unsigned m = mask[t.tid];
    if constexpr (std::is_same<Inject, NoInject>::value) smallw_mul<NTWM, KB, false>(w, A, lda, 0, t, acc);
    else smallw_mul_inj<NTWM, KB, false>(w, A, lda, 0, t, acc, inj);
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const int c = t.wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float o = mask_bit(acc[i][r], m);
            OUT[(4 * t.q + r) * ldo + 16 * c + t.n] = o;
            if (TAPE) gout[(long)(4 * t.q + r) * ldg + 16 * c + t.n] = o;
        }
    }
}

// OUT[16 x 16*NT] = A @ B for the D x D affine maps with pre-loaded weights (tile = wave, KB = 2: D <= 32)
__device__ __forceinline__ void dense_small_pre(const SmallW<1, 2>& w, const float* A, int lda, int kmax, int NT,
                                                float* OUT, int ldo, const Tid& t) {
    if (t.wave < NT) {
        f32x4 acc[1];
        smallw_mul<1, 2, true>(w, A, lda, kmax, t, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) OUT[(4 * t.q + r) * ldo + 16 * t.wave + t.n] = acc[0][r];
    }
}

struct NoPost {
    __device__ __forceinline__ void operator()() const {}
};

// hidden layer: OUT = relu(A @ B + bias).  With MASK the ReLU sign pattern of this lane's 4*NTWM outputs is kept
// as one 32-bit word per thread (first output in bit 31, see relu_bit / mask_bit) for the reverse sweep: the same
// lane of the same wave owns the same (tile, register) there, so no cross-lane exchange and a single LDS store
// per GEMM.  The K-split GEMM that consumes OUT reads only the columns this wave writes here (its k-blocks
// S = wave + 4 s are this wave's column tiles), so it follows without a workgroup barrier.
// `post` runs between the main loop and the epilogue (unused hook); PRE: ring_run_pre with injected requests.
template <int NTWM, int DEPTH, bool MASKK, bool MASK, bool TAPE = false, class Post = NoPost, bool PRE = false,
          int NI = 0, class Inject = NoInject>
__device__ __forceinline__ void dense_relu(const float* A, int lda, int kmax, int KB, const float4* Bp,
                                           const float* __restrict__ bias, float* OUT, int ldo, unsigned* mask,
                                           const Tid& t, float* __restrict__ gout = nullptr, int ldg = 0,
                                           Post post = Post(), const RingPre<NTWM>* pre = nullptr,
                                           long long* tlp = nullptr, Inject inj = Inject()) {
    f32x4 acc[NTWM];
    WRing<NTWM, DEPTH> w;
    if (PRE) {
#pragma unroll
        for (int i = 0; i < NTWM; ++i) {
            w.r[0][i] = (f32x4){pre->b[i].x, pre->b[i].y, pre->b[i].z, pre->b[i].w};
            acc[i] = (f32x4){pre->bv[i], pre->bv[i], pre->bv[i], pre->bv[i]};
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tlp && t.tid == 0) tlp[40] = (long long)__builtin_amdgcn_s_memtime();  // dev-only stage-internal stamps
        ring_run_pre<NTWM, DEPTH, NI>(w, A, lda, Bp, t, acc, inj);
    } else {
        ring_issue<NTWM, DEPTH, true>(w, Bp, KB, bias, t);
        ring_run<NTWM, DEPTH, MASKK, true>(w, A, lda, kmax, KB, Bp, t, acc);
    }
    if (tlp && t.tid == 0) tlp[41] = (long long)__builtin_amdgcn_s_memtime();
    post();
    if (tlp && t.tid == 0) tlp[42] = (long long)__builtin_amdgcn_s_memtime();
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const int c = t.wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[i][r];
            const float o = MASK ? relu_bit(v, m) : (v > 0.f ? v : 0.f);
            OUT[(4 * t.q + r) * ldo + 16 * c + t.n] = o;
            if (TAPE) gout[(long)(4 * t.q + r) * ldg + 16 * c + t.n] = o;       // training tape (HBM)
        }
    }
    if (MASK) mask[t.tid] = m << (32 - 4 * NTWM);
}

// backward of a hidden layer: OUT = (A @ B) * mask
template <int NTWM, int DEPTH, bool TAPE = false, class Post = NoPost, bool PRE = false, int NI = 0,
          class Inject = NoInject>
__device__ __forceinline__ void dense_masked(const float* A, int lda, int KB, const float4* Bp, float* OUT,
                                             int ldo, const unsigned* mask, const Tid& t,
                                             float* __restrict__ gout = nullptr, int ldg = 0, Post post = Post(),
                                             const RingPre<NTWM>* pre = nullptr, Inject inj = Inject()) {
    f32x4 acc[NTWM];
#pragma unroll
    for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    WRing<NTWM, DEPTH> w;
    unsigned m = mask[t.tid];
    if (PRE) {
#pragma unroll
        for (int i = 0; i < NTWM; ++i) w.r[0][i] = (f32x4){pre->b[i].x, pre->b[i].y, pre->b[i].z, pre->b[i].w};
        __builtin_amdgcn_sched_barrier(0);
        ring_run_pre<NTWM, DEPTH, NI>(w, A, lda, Bp, t, acc, inj);
    } else {
        ring_issue<NTWM, DEPTH, false>(w, Bp, KB, nullptr, t);
        ring_run<NTWM, DEPTH, false, false>(w, A, lda, 0, KB, Bp, t, acc);
    }
    post();
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const int c = t.wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float o = mask_bit(acc[i][r], m);
            OUT[(4 * t.q + r) * ldo + 16 * c + t.n] = o;
            if (TAPE) gout[(long)(4 * t.q + r) * ldg + 16 * c + t.n] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FAST MODE (fabhip_set_fast_mode(1); NOT the parity path): the two W x W GEMMs of a coupling layer - 87 % of its
// flops - on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation).  Weights come from a second, bf16
// image of W2 / W2^T (FlowDims::o_W2h / o_W2Th, packed by k_pack_bf16); activations stay fp32 in LDS and are rounded
// to bf16 (v_cvt_pk_bf16_f32, round to nearest even) as they are fetched.  Everything else (first / last conditioner
// layer, affine maps, coupling arithmetic, accumulation, the reverse sweep's structure) is unchanged.
// Packed bf16 B-operand tile (c, S) of a K x N matrix: 64 lanes x 16 bytes, lane l = (g = l>>4, n = l&15) holds
// B[32S + 8g + j][16c + n], j = 0..7; the A fragment of lane (g, row) is A[row][32S + 8g + j]: the same (g, j) -> k
// map on both sides, which is all the instruction's dot product needs.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// v_mfma_f32_16x16x32_bf16 whose destination never overlaps its A / B operands.  With the accumulators in architectural registers
// (_build.py: -amdgpu-mfma-vgpr-form) hipcc hands a B tile that dies in the MFMA to it as the destination (v_mfma v[92:95],
// v[160:163], v[92:95], v[148:151]); gfx950 then returns a wrong first row of every 4-row group (round 5 met the same with the
// 4x4x4 bf16 form and a ring slot as the destination, flow_r4f.h).  The fp32 forms are unaffected (the whole GPU suite compares them).
// The empty asm keeps both operands alive until the result exists, so the allocator cannot reuse their registers for it.
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    asm volatile("" : "+v"(d) : "v"(a), "v"(b));       // (reads the RESULT too: the scheduler cannot lift it above the MFMA)
    return d;
}

__device__ __forceinline__ f32x4 mfma_bf16(const float4& a0, const float4& a1, const uint4& b, f32x4 c) {
    union { unsigned u[4]; bf16x8 v; } ua, ub;
    ua.u[0] = cvt_pk_bf16(a0.x, a0.y); ua.u[1] = cvt_pk_bf16(a0.z, a0.w);
    ua.u[2] = cvt_pk_bf16(a1.x, a1.y); ua.u[3] = cvt_pk_bf16(a1.z, a1.w);
    ub.u[0] = b.x; ub.u[1] = b.y; ub.u[2] = b.z; ub.u[3] = b.w;
    return mfma_bf16_16x16x32(ua.v, ub.v, c);
}

// acc[i] += A[16 x 64 NTWM] (fp32, LDS) @ B[:, tile wave + 4 i] (bf16 image), K = 64 NTWM = 2 NTWM blocks of 32.
// Plain compiler-tracked loads in straight-line code, two chunks of CH k-blocks in flight (CH NTWM <= 25 loads each:
// the vector-memory counter is 6 bits - with more than 63 loads outstanding waits release early, measured).  `mid`
// runs once chunk 0 has been multiplied: the requests of the stages after this GEMM go out there (they stay in flight
// behind chunk 1), not in front of it.  Never more than 63 loads in flight.
// KB2T: k-blocks per column-tile strip of the packed matrix (its K / 32); a call multiplies 2 NTWM of them (K = 64 NTWM).
// Chunk C is multiplied out of buffer C & 1 while the loads of chunk C + 1 go out into the other buffer ONE BEHIND EACH
// MFMA (a burst of loads blocks the issuing wave: see Bf16Injector); at most one chunk (<= 25 loads) + the later stages'
// requests in flight.  b0 holds chunk 0 on entry.
template <int NTWM, int CH, int C, class Mid>
__device__ __forceinline__ void gemm_bf16_chunk(uint4 (&b0)[CH][NTWM], uint4 (&b1)[CH][NTWM], const float* __restrict__ arow,
                                                const uint4* __restrict__ bw, f32x4 (&acc)[NTWM], Mid& mid, int KB2T) {
    constexpr int KB2 = 2 * NTWM, NCH = (KB2 + CH - 1) / CH;
    if constexpr (C < NCH) {
        uint4 (&b)[CH][NTWM] = (C & 1) ? b1 : b0;
        uint4 (&bn)[CH][NTWM] = (C & 1) ? b0 : b1;
        static_for<0, CH>([&](auto cc) {
            constexpr int c = decltype(cc)::value, S = C * CH + c;
            if constexpr (S < KB2) {
                const float4 a0 = *reinterpret_cast<const float4*>(arow + 32 * S);
                const float4 a1 = *reinterpret_cast<const float4*>(arow + 32 * S + 4);
                union { unsigned u[4]; bf16x8 v; } ua;
                ua.u[0] = cvt_pk_bf16(a0.x, a0.y); ua.u[1] = cvt_pk_bf16(a0.z, a0.w);
                ua.u[2] = cvt_pk_bf16(a1.x, a1.y); ua.u[3] = cvt_pk_bf16(a1.z, a1.w);
                static_for<0, NTWM>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    union { uint4 q; bf16x8 v; } ub;
                    ub.q = b[c][i];
                    acc[i] = mfma_bf16_16x16x32(ua.v, ub.v, acc[i]);
                    // block (C + 1) CH + c, tile i of the next chunk.  (No sched_barrier pins around this load: pinned, the
                    // spline kernel's reverse sweep gave run-to-run different gradients at NTWM = 4 - hipcc's own
                    // placement is deterministic over 30 x 4096-chain repeats and as fast.)
                    if constexpr ((C + 1) * CH + c < KB2) bn[c][i] = bw[((size_t)i * 4 * KB2T + (C + 1) * CH + c) * 64];
                });
            }
        });
        // the later stages' requests: two chunks (NTWM <= 5) - behind chunk 0; more chunks (NTWM = 8, up to 46 requests) -
        // only after the last chunk, when nothing else is in flight (never more than 63 loads outstanding)
        if constexpr (C == (NCH == 2 ? 0 : NCH - 1)) mid();
        gemm_bf16_chunk<NTWM, CH, C + 1>(b0, b1, arow, bw, acc, mid, KB2T);
    }
}

template <int NTWM>
__host__ __device__ constexpr int bf16_chunk() { return NTWM <= 5 ? NTWM : 24 / NTWM; }     // NTWM = 8: 3 k-blocks x 8 tiles

// blocks requested one short stage EARLY with plain loads: for NTWM <= 5 the wave's whole weight slice (2 NTWM^2 <= 50
// loads, 200 registers - a wave runs alone on its SIMD) so that the stream overlaps the short stage in front of the
// GEMM (a pure stream reaches 45 - 60 B/clk per CU, tools/ubench/stream.hip; issued at the GEMM's own start the same
// bytes cost their latency + transfer on top of its MFMAs); for NTWM = 8 chunk 0 only (the rest is streamed in chunks).
template <int NTWM>
__host__ __device__ constexpr int bf16_pre_blocks() { return NTWM <= 5 ? 2 * NTWM : bf16_chunk<NTWM>(); }

template <int NTWM>
struct Bf16Pre {
    uint4 b[bf16_pre_blocks<NTWM>()][NTWM];
    float bv[NTWM];
};

template <int NTWM, bool BIAS>
__device__ __forceinline__ void bf16pre_load(Bf16Pre<NTWM>& p, const uint4* __restrict__ Bh, const float* __restrict__ bias,
                                             const Tid& t) {
    constexpr int KB2 = 2 * NTWM, PB = bf16_pre_blocks<NTWM>();
    const uint4* bw = Bh + ((size_t)t.wave * KB2) * 64 + t.lane;
#pragma unroll
    for (int i = 0; i < NTWM; ++i) p.bv[i] = BIAS ? bias[16 * (t.wave + 4 * i) + t.n] : 0.f;
#pragma unroll
    for (int c = 0; c < PB; ++c)
#pragma unroll
        for (int i = 0; i < NTWM; ++i) p.b[c][i] = bw[((size_t)i * 4 * KB2 + c) * 64];
    __builtin_amdgcn_sched_barrier(0);
}

// request m of the preload (m < NTWM: bias of tile m, then block (m - NTWM) / NTWM, tile (m - NTWM) % NTWM), one per call
template <int NTWM, bool BIAS>
struct Bf16Injector {
    Bf16Pre<NTWM>& p;
    const uint4* Bh;
    const float* bias;
    const Tid& t;
    static constexpr int KB2 = 2 * NTWM, PB = bf16_pre_blocks<NTWM>(), TOTAL = NTWM + PB * NTWM;
    template <class M>
    __device__ __forceinline__ void operator()(M) const {
        constexpr int m = M::value;
        if constexpr (m < NTWM) {
            p.bv[m] = BIAS ? bias[16 * (t.wave + 4 * m) + t.n] : 0.f;
        } else if constexpr (m < TOTAL) {
            constexpr int c = (m - NTWM) / NTWM, i = (m - NTWM) % NTWM;
            p.b[c][i] = Bh[((size_t)t.wave * KB2 + (size_t)i * 4 * KB2 + c) * 64 + t.lane];
        }
    }
    template <int DONE>
    __device__ __forceinline__ void rest() const {            // the requests with index >= DONE
        static_for<DONE, (TOTAL > DONE ? TOTAL : DONE)>(*this);
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <int NTWM, class Mid>
__device__ __forceinline__ void gemm_bf16(const float* __restrict__ A, int lda, const uint4* __restrict__ Bh, const Tid& t,
                                          f32x4 (&acc)[NTWM], Mid& mid, const Bf16Pre<NTWM>& pre) {
    constexpr int KB2 = 2 * NTWM;
    constexpr int CH = bf16_chunk<NTWM>();
    const float* arow = A + t.n * lda + 8 * t.q;
    if constexpr (bf16_pre_blocks<NTWM>() == KB2) {          // everything is already on its way / here
        static_for<0, KB2>([&](auto Sc) {
            constexpr int S = decltype(Sc)::value;
            const float4 a0 = *reinterpret_cast<const float4*>(arow + 32 * S);
            const float4 a1 = *reinterpret_cast<const float4*>(arow + 32 * S + 4);
            union { unsigned u[4]; bf16x8 v; } ua;
            ua.u[0] = cvt_pk_bf16(a0.x, a0.y); ua.u[1] = cvt_pk_bf16(a0.z, a0.w);
            ua.u[2] = cvt_pk_bf16(a1.x, a1.y); ua.u[3] = cvt_pk_bf16(a1.z, a1.w);
            static_for<0, NTWM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                union { uint4 q; bf16x8 v; } ub;
                ub.q = pre.b[S][i];
                acc[i] = mfma_bf16_16x16x32(ua.v, ub.v, acc[i]);
                mid.template step<S * NTWM + i>();            // one of the later stages' requests behind each MFMA
            });
        });
        mid.template rest<KB2 * NTWM>();
        return;
    }
    const uint4* bw = Bh + ((size_t)t.wave * KB2) * 64 + t.lane;
    uint4 b0[CH][NTWM], b1[CH][NTWM];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < NTWM; ++i) b0[c][i] = pre.b[c][i];
    gemm_bf16_chunk<NTWM, CH, 0>(b0, b1, arow, bw, acc, mid, KB2);
}

// the same without an early chunk 0, for a K = 64 NTWM slice (k-blocks S0 .. S0 + 2 NTWM - 1) of a matrix with KB2T
// k-blocks per strip (the spline conditioner's K = NFP reverse GEMM is NCH such slices): acc += A[:, slice] @ B[slice, :]
struct NoMid {
    __device__ __forceinline__ void operator()() const {}
    template <int M> __device__ __forceinline__ void step() const {}
    template <int DONE> __device__ __forceinline__ void rest() const {}
};
template <int NTWM>
__device__ __forceinline__ void gemm_bf16_slice(const float* __restrict__ A, int lda, const uint4* __restrict__ Bh, int KB2T,
                                                int S0, const Tid& t, f32x4 (&acc)[NTWM]) {
    constexpr int KB2 = 2 * NTWM;
    constexpr int CH = bf16_chunk<NTWM>();
    const float* arow = A + t.n * lda + 8 * t.q + 32 * S0;
    const uint4* bw = Bh + ((size_t)t.wave * KB2T + S0) * 64 + t.lane;
    uint4 b0[CH][NTWM], b1[CH][NTWM];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < NTWM; ++i) b0[c][i] = bw[((size_t)i * 4 * KB2T + c) * 64];
    NoMid mid;
    gemm_bf16_chunk<NTWM, CH, 0>(b0, b1, arow, bw, acc, mid, KB2T);
}

// fast-mode twin of dense_relu (K = N = 64 NTWM): OUT = relu(A @ B + bias), ReLU sign words as there
template <int NTWM, bool MASK, bool TAPE, class Mid>
__device__ __forceinline__ void dense_relu_bf16(const float* A, int lda, const uint4* Bh, const float* __restrict__ bias,
                                                float* OUT, int ldo, unsigned* mask, const Tid& t, Mid& mid,
                                                const Bf16Pre<NTWM>& pre, float* __restrict__ gout = nullptr, int ldg = 0) {
    f32x4 acc[NTWM];
#pragma unroll
    for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){pre.bv[i], pre.bv[i], pre.bv[i], pre.bv[i]};
    gemm_bf16<NTWM>(A, lda, Bh, t, acc, mid, pre);
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const int c = t.wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[i][r];
            const float o = MASK ? relu_bit(v, m) : (v > 0.f ? v : 0.f);
            OUT[(4 * t.q + r) * ldo + 16 * c + t.n] = o;
            if (TAPE) gout[(long)(4 * t.q + r) * ldg + 16 * c + t.n] = o;
        }
    }
    if (MASK) mask[t.tid] = m << (32 - 4 * NTWM);
}

// fast-mode twin of dense_masked (K = N = 64 NTWM): OUT = (A @ B) * mask
template <int NTWM, bool TAPE, class Mid>
__device__ __forceinline__ void dense_masked_bf16(const float* A, int lda, const uint4* Bh, float* OUT, int ldo,
                                                  const unsigned* mask, const Tid& t, Mid& mid, const Bf16Pre<NTWM>& pre,
                                                  float* __restrict__ gout = nullptr, int ldg = 0) {
    f32x4 acc[NTWM];
#pragma unroll
    for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned m = mask[t.tid];
    gemm_bf16<NTWM>(A, lda, Bh, t, acc, mid, pre);
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const int c = t.wave + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float o = mask_bit(acc[i][r], m);
            OUT[(4 * t.q + r) * ldo + 16 * c + t.n] = o;
            if (TAPE) gout[(long)(4 * t.q + r) * ldg + 16 * c + t.n] = o;
        }
    }
}

// the requests a fp32 W x W main loop issues behind its MFMAs (`inj(IC<j>)`, j < N), all at once
template <int J, int N, class Inject>
__device__ __forceinline__ void inject_all(Inject& inj) {
    if constexpr (J < N) {
        inj(IC<J>());
        inject_all<J + 1, N>(inj);
    }
}

// hook of the bf16 GEMMs: `step(m)` behind MFMA m (a runtime index into a compile-time switch: the loops around it are
// fully unrolled, so it folds), `rest(done)` for the requests the MFMAs did not carry, `operator()` = all at once
template <int NI, class Inject>
struct MidInject {
    Inject& inj;
    template <int M>
    __device__ __forceinline__ void step() const {
        if constexpr (M < NI) { __builtin_amdgcn_sched_barrier(0); inj(IC<M>()); __builtin_amdgcn_sched_barrier(0); }
    }
    template <int DONE>
    __device__ __forceinline__ void rest() const {
        if constexpr (DONE < NI) inject_all<DONE, NI>(inj);
    }
    __device__ __forceinline__ void operator()() const { inject_all<0, NI>(inj); }
};

// OUT[16 x 16*NT] = A[16 x K] @ B   (NT <= 4: one column tile per wave), used for the D x D affine maps.
// `add` (nullable): per-column additive term [16 * NT] (the ActNorm shift folded into the affine map).
__device__ __forceinline__ void dense_small(const float* A, int lda, int kmax, int KB, const float4* Bp, int NT,
                                            float* OUT, int ldo, const Tid& t, const float* __restrict__ add = nullptr) {
    if (t.wave < NT) {
        f32x4 acc[1];
        const float av = add ? add[16 * t.wave + t.n] : 0.f;
        acc[0] = (f32x4){av, av, av, av};
        WRing<1, 2> w;
        ring_issue<1, 2, false>(w, Bp, KB, nullptr, t);
        ring_run<1, 2, true, false>(w, A, lda, kmax, KB, Bp, t, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) OUT[(4 * t.q + r) * ldo + 16 * t.wave + t.n] = acc[0][r];
    }
}

// ------------------------------------------------------------------------------------------------
// log q(x) (and d log q / dx) for the 16 rows in U0 (columns >= D must be zero).
// Returns log q of this thread's row (replicated over the row's 16 lanes).  With GRAD the gradient
// is left in the state buffer whose LDS offset is returned through *grad_off.
// normflows NormalizingFlow.log_prob: inverses in reversed layer order, log-dets added, base last.
// ------------------------------------------------------------------------------------------------
// copy a 16 x w block of an LDS matrix (leading dim ld) to the tape rows of this tile (coalesced)
__device__ __forceinline__ void tape_copy(float* __restrict__ dst, int w, const float* src, int ld, const Tid& t) {
    for (int e = t.tid; e < ROWS * w; e += NTHREADS) {
        const int r = e / w, j = e - r * w;
        dst[(long)r * w + j] = src[r * ld + j];
    }
}

// With TAPE (requires GRAD) the quantities the parameter-gradient GEMMs need are also written to `tape`
// (TapeDims layout, rows row0 .. row0+15): see fabhip_common.h.
template <int NTWM, bool GRAD, bool TAPE = false, bool FAST = false>
__device__ float flow_log_prob_tile(const FlowDims& f, const FlowLds& l, const float* __restrict__ packed,
                                    float* lds, const Tid& t, int* grad_off, const TapeDims* td = nullptr,
                                    float* __restrict__ tape = nullptr, long row0 = 0) {
    static_assert(!TAPE || GRAD, "the tape is written by the forward + reverse sweep");
    constexpr int DW = depth_w<NTWM>();
    int cur = l.o_U0, nxt = l.o_U1;
    float logq = 0.f;
    float* PART = lds + l.o_PART;
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
    // the first conditioner layer's weights (K = dp = 32 always: D <= 64) live in registers and are requested a
    // whole long stage early (ahead of the previous layer's W x W GEMM), so that neither their latency nor
    // hipcc's vmcnt(0) in front of the barriers in between costs anything
    SmallW<NTWM, 2> w1r;
    SmallW<1, 2> awr;                                     // the layer's D x D affine map (register path: D <= 32)
    const bool kbd2 = f.KBD == 2;
    {
        const float* Lp = packed + (size_t)(f.K - 1) * f.layer_stride;
        smallw_load<NTWM, 2, true>(w1r, reinterpret_cast<const float4*>(Lp + f.o_W1), Lp + f.o_b1, t);
        // (its bias slot carries the ActNorm term `ac` of the layer, zero without ActNorm)
        if (kbd2) smallw_load<1, 2, true>(awr, reinterpret_cast<const float4*>(Lp + f.o_AW), Lp + f.o_ac, t);
    }
    for (int layer = f.K - 1; layer >= 0; --layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        const float4* W2 = reinterpret_cast<const float4*>(Lp + f.o_W2);
        const float4* W3 = reinterpret_cast<const float4*>(Lp + f.o_W3);
        unsigned* mk = reinterpret_cast<unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        const bool tl = (layer == f.K - 2);
        if (tl) FAB_TL(f, 0);
        float* tl_layer = TAPE ? tape + (size_t)layer * td->layer_stride : nullptr;
        if (TAPE) tape_copy(tl_layer + td->o_ZA + row0 * td->wz, td->wz, lds + cur, l.DS, t);
        // ---- InvertibleAffine.inverse: z <- z @ (P L U), log_det = +sum(log_S) --------------------------
        if (kbd2) dense_small_pre(awr, lds + cur, l.DS, f.D, f.NTD, lds + nxt, l.DS, t);
        else dense_small(lds + cur, l.DS, f.D, f.KBD, reinterpret_cast<const float4*>(Lp + f.o_AW), f.NTD, lds + nxt,
                         l.DS, t, Lp + f.o_ac);
        logq += Lp[f.o_logS];
        if (tl) FAB_TL(f, 1);
        __syncthreads();
        if (tl) FAB_TL(f, 2);
        float* Z = lds + nxt;
        if (TAPE) {                                   // z1 | ones column, and the ones columns of H1 / H2
            float* Z1 = tl_layer + td->o_Z1 + row0 * td->w1;
            const int w1 = td->w1, c1 = w1 - 16;
            for (int e = t.tid; e < ROWS * w1; e += NTHREADS) {
                const int r = e / w1, j = e - r * w1;
                Z1[(long)r * w1 + j] = j < f.d ? Z[r * l.DS + j] : (j == c1 ? 1.f : 0.f);
            }
            const int r = t.tid >> 4, j = t.tid & 15;
            const float one = j == 0 ? 1.f : 0.f;
            tl_layer[td->o_H1 + (row0 + r) * td->wh + f.Wp + j] = one;
            tl_layer[td->o_H2 + (row0 + r) * td->wh + f.Wp + j] = one;
        }
        // ---- conditioner MLP: relu(relu(z1 W1 + b1) W2 + b2) W3' ----------------------------------------
        RingPre<NTWM> rp;                                 // block 0 + bias of the W x W GEMM below: lands during this stage
        Bf16Pre<NTWM> bp;                                 // fast mode: the bf16 slice of the W x W GEMM (+ bias) instead,
        if constexpr (!FAST) ringpre_load<NTWM, true>(rp, W2, f.KBW, Lp + f.o_b2, t);      // requested behind the MFMAs below
        auto inj_w2h = Bf16Injector<NTWM, true>{bp, reinterpret_cast<const uint4*>(Lp + f.o_W2h), Lp + f.o_b2, t};
        if constexpr (FAST) {
            dense_relu_small<NTWM, 2, GRAD, TAPE>(w1r, Z, l.DS, f.d, HA, l.WS, mk, t,
                                                  TAPE ? tl_layer + td->o_H1 + row0 * td->wh : nullptr, TAPE ? td->wh : 0,
                                                  inj_w2h);
            inj_w2h.template rest<8 * NTWM>();                     // what the 8 NTWM MFMAs did not carry
        } else {
            dense_relu_small<NTWM, 2, GRAD, TAPE>(w1r, Z, l.DS, f.d, HA, l.WS, mk, t,
                                                  TAPE ? tl_layer + td->o_H1 + row0 * td->wh : nullptr, TAPE ? td->wh : 0);
        }
        if (tl) FAB_TL(f, 3);
        __syncthreads();
        if (tl) FAB_TL(f, 4);
        KsplitPre<NTWM> kp;
        float b3s[2], b3c[2];                             // coupling biases of this thread's columns (D - d <= 32)
        // requests of the stages after this W x W GEMM, issued a few per k-block behind its last MFMAs:
        //   [0, N)   first tile of the K-split GEMM that follows      [N, 3N)  next layer's W1 (2 k-blocks x N tiles)
        //   [3N, 4N) next layer's b1      4N, 4N+1  next layer's affine map      4N+2 .. 4N+5  coupling biases
        const float* Ln = packed + (size_t)(layer > 0 ? layer - 1 : 0) * f.layer_stride;
        const bool nxt_layer = layer > 0;
        auto inj_fwd = [&](auto jc) {
            constexpr int j = decltype(jc)::value, N = NTWM;
            if constexpr (j < N) {
                kp.b0[j] = W3[((size_t)t.wave + 4 * j) * 64 + t.lane];
            } else if constexpr (j < 3 * N) {
                constexpr int S = (j - N) / N, i = (j - N) % N;
                if (nxt_layer)
                    w1r.b[S][i] = reinterpret_cast<const float4*>(Ln + f.o_W1)[((size_t)(t.wave + 4 * i) * 2 + S) * 64 + t.lane];
            } else if constexpr (j < 4 * N) {
                constexpr int i = j - 3 * N;
                if (nxt_layer) w1r.bv[i] = Ln[f.o_b1 + 16 * (t.wave + 4 * i) + t.n];
            } else if constexpr (j < 4 * N + 2) {
                constexpr int S = j - 4 * N;
                if (nxt_layer && kbd2) {
                    awr.b[S][0] = reinterpret_cast<const float4*>(Ln + f.o_AW)[((size_t)t.wave * 2 + S) * 64 + t.lane];
                    if (S == 0) awr.bv[0] = Ln[f.o_ac + 16 * t.wave + t.n];     // (64 floats, zero beyond D)
                }
            } else if constexpr (j < 4 * N + 6) {
                constexpr int it = (j - 4 * N - 2) & 1, sc = (j - 4 * N - 2) >> 1;
                const int col = t.c + 16 * it;
                const float v = col < f.DO ? Lp[f.o_b3 + sc * f.DOp + col] : 0.f;
                if (sc) b3c[it] = v; else b3s[it] = v;
            } else if constexpr (j < 5 * N + 6) {         // second tile of the K-split GEMM (its scale columns)
                constexpr int sidx = j - 4 * N - 6;
                kp.b1[sidx] = W3[((size_t)4 * N + t.wave + 4 * sidx) * 64 + t.lane];
            }
        };
        if constexpr (FAST) {
            MidInject<5 * NTWM + 6, decltype(inj_fwd)> mid{inj_fwd};
            dense_relu_bf16<NTWM, GRAD, TAPE>(HA, l.WS, reinterpret_cast<const uint4*>(Lp + f.o_W2h), Lp + f.o_b2, HB, l.WS,
                                              mk + NTHREADS, t, mid, bp,
                                              TAPE ? tl_layer + td->o_H2 + row0 * td->wh : nullptr, TAPE ? td->wh : 0);
        } else {
            dense_relu<NTWM, DW, false, GRAD, TAPE, NoPost, true, 5 * NTWM + 6, decltype(inj_fwd)>(
                HA, l.WS, f.Wp, f.KBW, W2, Lp + f.o_b2, HB, l.WS, mk + NTHREADS, t,
                TAPE ? tl_layer + td->o_H2 + row0 * td->wh : nullptr, TAPE ? td->wh : 0, NoPost(), &rp,
                (tl && blockIdx.x == 0) ? f.timeline : nullptr, inj_fwd);
        }
        if (tl) FAB_TL(f, 5);
        // no workgroup barrier: the K-split GEMM reads only this wave's own columns of HB (LDS is in-order per wave)
        __builtin_amdgcn_wave_barrier();
        if (tl) FAB_TL(f, 6);
        if (f.NTO == 2) gemm_ksplit<NTWM, true, true>(HB, l.WS, W3, 2, PART, l.PN, t, &kp);   // D - d <= 16: [shift | scale]
        else gemm_ksplit<NTWM, true>(HB, l.WS, W3, f.NTO, PART, l.PN, t, &kp);
        if (tl) FAB_TL(f, 7);
        __syncthreads();
        if (tl) FAB_TL(f, 8);
        // ---- AffineCoupling.inverse: z2 <- (z2 - shift) * exp(-s), log_det = -sum(s) ----------------------
        float ssum = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int j = t.c + 16 * it;
            if (j < f.DO) {
                const float shift = part_sum(PART, l.PN, t.row, j) + b3s[it];
                const float s = part_sum(PART, l.PN, t.row, f.DOp + j) + b3c[it];
                const float es = expf(-s);
                const float v2 = (Z[t.row * l.DS + f.d + j] - shift) * es;
                Z[t.row * l.DS + f.d + j] = v2;
                if (GRAD) {
                    lds[l.o_ES + ((size_t)layer * ROWS + t.row) * f.DOp + j] = es;
                    lds[l.o_V2 + ((size_t)layer * ROWS + t.row) * f.DOp + j] = v2;
                }
                ssum += s;
            }
        }
        logq += -row16_sum(ssum);
        if (tl) FAB_TL(f, 10);
        __syncthreads();
        if (tl) FAB_TL(f, 11);
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    // ---- DiagGaussian.log_prob -----------------------------------------------------------------------
    const float* base = packed + f.o_base;
    float* Zc = lds + cur;
    float bsum = 0.f;
    for (int j = t.c; j < f.D; j += 16) {
        const float ls = base[f.Dp + j];
        const float sc = expf(ls);
        const float zn = (Zc[t.row * l.DS + j] - base[j]) / sc;
        bsum += ls + 0.5f * (zn * zn);
        if (TAPE) {                                       // what d/dloc and d/dlog_scale reduce over the batch
            float* TB = tape + td->o_TB + (row0 + t.row) * td->wb;
            TB[j] = zn / sc;
            TB[td->wz + j] = zn * zn - 1.f;
        }
        if (GRAD) Zc[t.row * l.DS + j] = -(zn / sc);      // d/dz of -0.5 ((z - loc)/sc)^2
    }
    logq += -0.5f * (float)f.D * 1.8378770664093453f - row16_sum(bsum);
    if (TAPE) tape[td->o_TB + (row0 + t.row) * td->wb + 2 * td->wz + t.c] = t.c == 0 ? 1.f : 0.f;
    if (!GRAD) return logq;

    // ---- reverse sweep: g = d log q / d(state), layers 0 .. K-1 -----------------------------------
    float* DP = lds + l.o_DP;
    // first reverse GEMM (K = 2 DOp): with 2 k-blocks (D <= 32) its weights are register-resident and requested
    // behind the previous layer's W x W main loop; with 4 k-blocks (D > 32) the streaming version is used (keeping
    // a second, 85-register variant alive across the loop costs more than it hides)
    SmallW<NTWM, 2> w3a;
    const bool kbo2 = f.KBO == 2;
    if (kbo2) smallw_load<NTWM, 2, false>(w3a, reinterpret_cast<const float4*>(packed + f.o_W3T), nullptr, t);
    for (int layer = 0; layer < f.K; ++layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        const float4* W2T = reinterpret_cast<const float4*>(Lp + f.o_W2T);
        const float4* W1T = reinterpret_cast<const float4*>(Lp + f.o_W1T);
        const float4* AWT = reinterpret_cast<const float4*>(Lp + f.o_AWT);
        float* G = lds + cur;
        const bool tl = (layer == 1);
        if (tl) FAB_TL(f, 16);
        for (int j = t.c; j < f.DO; j += 16) {
            const float g2 = G[t.row * l.DS + f.d + j];
            const float es = lds[l.o_ES + ((size_t)layer * ROWS + t.row) * f.DOp + j];
            const float v2 = lds[l.o_V2 + ((size_t)layer * ROWS + t.row) * f.DOp + j];
            DP[t.row * l.PS + j] = -(g2 * es);                    // d/d shift
            DP[t.row * l.PS + f.DOp + j] = -(g2 * v2) - 1.f;       // d/d s  (incl. the -sum(s) log-det)
            G[t.row * l.DS + f.d + j] = g2 * es;                  // d/d z2
        }
        if (tl) FAB_TL(f, 17);
        __syncthreads();
        if (tl) FAB_TL(f, 18);
        const unsigned* mk = reinterpret_cast<const unsigned*>(lds + l.o_MASK) + (size_t)layer * 2 * NTHREADS;
        float* tl_layer = TAPE ? tape + (size_t)layer * td->layer_stride : nullptr;
        if (TAPE) tape_copy(tl_layer + td->o_DP + row0 * td->wp, td->wp, DP, l.PS, t);
        RingPre<NTWM> rpb;                                // block 0 of the W x W GEMM below: lands during this stage
        Bf16Pre<NTWM> bpb;
        if constexpr (!FAST) ringpre_load<NTWM, false>(rpb, W2T, f.KBW, nullptr, t);
        auto inj_w2th = Bf16Injector<NTWM, false>{bpb, reinterpret_cast<const uint4*>(Lp + f.o_W2Th), nullptr, t};
        if (kbo2) {
            if constexpr (FAST) {
                dense_masked_small<NTWM, 2, TAPE>(w3a, DP, l.PS, HA, l.WS, mk + NTHREADS, t,
                                                  TAPE ? tl_layer + td->o_E2 + row0 * td->we : nullptr, TAPE ? td->we : 0,
                                                  inj_w2th);
                inj_w2th.template rest<8 * NTWM>();
            } else {
                dense_masked_small<NTWM, 2, TAPE>(w3a, DP, l.PS, HA, l.WS, mk + NTHREADS, t,
                                                  TAPE ? tl_layer + td->o_E2 + row0 * td->we : nullptr, TAPE ? td->we : 0);
            }
        } else {
            if constexpr (FAST) bf16pre_load<NTWM, false>(bpb, reinterpret_cast<const uint4*>(Lp + f.o_W2Th), nullptr, t);
            dense_masked<NTWM, 2, TAPE>(DP, l.PS, f.KBO, reinterpret_cast<const float4*>(Lp + f.o_W3T), HA, l.WS,
                                        mk + NTHREADS, t, TAPE ? tl_layer + td->o_E2 + row0 * td->we : nullptr,
                                        TAPE ? td->we : 0);
        }
        if (tl) FAB_TL(f, 19);
        __syncthreads();
        if (tl) FAB_TL(f, 20);
        KsplitPre<NTWM> kp;
        SmallW<1, 2> awtr;                                // this layer's affine^T, used three short stages below
        awtr.bv[0] = 0.f;
        //   [0, N) first tile of the K-split GEMM that follows   [N, 3N) next layer's W3'^T   3N, 3N+1 this affine^T
        const float4* W3Tn = reinterpret_cast<const float4*>(
            packed + (size_t)(layer + 1 < f.K ? layer + 1 : layer) * f.layer_stride + f.o_W3T);
        const bool nxt_w3 = kbo2 && layer + 1 < f.K;
        auto inj_bwd = [&](auto jc) {
            constexpr int j = decltype(jc)::value, N = NTWM;
            if constexpr (j < N) {
                kp.b0[j] = W1T[((size_t)t.wave + 4 * j) * 64 + t.lane];
            } else if constexpr (j < 3 * N) {
                constexpr int S = (j - N) / N, i = (j - N) % N;
                if (nxt_w3) w3a.b[S][i] = W3Tn[((size_t)(t.wave + 4 * i) * 2 + S) * 64 + t.lane];
            } else if constexpr (j < 3 * N + 2) {
                constexpr int S = j - 3 * N;
                if (kbd2) awtr.b[S][0] = AWT[((size_t)t.wave * 2 + S) * 64 + t.lane];
            }
        };
        if constexpr (FAST) {
            MidInject<3 * NTWM + 2, decltype(inj_bwd)> mid{inj_bwd};
            dense_masked_bf16<NTWM, TAPE>(HA, l.WS, reinterpret_cast<const uint4*>(Lp + f.o_W2Th), HB, l.WS, mk, t, mid, bpb,
                                          TAPE ? tl_layer + td->o_E1 + row0 * td->we : nullptr, TAPE ? td->we : 0);
        } else {
            dense_masked<NTWM, DW, TAPE, NoPost, true, 3 * NTWM + 2, decltype(inj_bwd)>(
                HA, l.WS, f.KBW, W2T, HB, l.WS, mk, t, TAPE ? tl_layer + td->o_E1 + row0 * td->we : nullptr,
                TAPE ? td->we : 0, NoPost(), &rpb, inj_bwd);
        }
        if (tl) FAB_TL(f, 21);
        __builtin_amdgcn_wave_barrier();              // as in the forward sweep: own columns only, no barrier
        if (tl) FAB_TL(f, 22);
        gemm_ksplit<NTWM, true>(HB, l.WS, W1T, f.NTd, PART, l.PN, t, &kp);
        if (tl) FAB_TL(f, 23);
        __syncthreads();
        if (tl) FAB_TL(f, 24);
        for (int j = t.c; j < f.d; j += 16) G[t.row * l.DS + j] += part_sum(PART, l.PN, t.row, j);
        if (tl) FAB_TL(f, 25);
        __syncthreads();
        if (tl) FAB_TL(f, 26);
        if (TAPE) tape_copy(tl_layer + td->o_GZ + row0 * td->wz, td->wz, G, l.DS, t);
        // ---- through InvertibleAffine.inverse: g <- g @ W^T -----------------------------------------------
        if (kbd2) dense_small_pre(awtr, G, l.DS, f.D, f.NTD, lds + nxt, l.DS, t);
        else dense_small(G, l.DS, f.D, f.KBD, AWT, f.NTD, lds + nxt, l.DS, t);
        if (tl) FAB_TL(f, 27);
        __syncthreads();
        if (tl) FAB_TL(f, 28);
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    *grad_off = cur;
    return logq;
}

// ------------------------------------------------------------------------------------------------
// x, log q = flow.sample given base noise in U0 (NormalizingFlow.sample): forward maps in layer order.
// Leaves x in the buffer at *x_off and returns log q of this thread's row.  (Runs once per AIS call:
// stages issue their own prologues, no cross-stage pipelining.)
// ------------------------------------------------------------------------------------------------
template <int NTWM>
__device__ float flow_sample_tile(const FlowDims& f, const FlowLds& l, const float* __restrict__ packed,
                                  float* lds, const Tid& t, int* x_off) {
    constexpr int DW = depth_w<NTWM>();
    int cur = l.o_U0, nxt = l.o_U1;
    const float* base = packed + f.o_base;
    float* PART = lds + l.o_PART;
    float* HA = lds + l.o_HA;
    float* HB = lds + l.o_HB;
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
        const float4* W1 = reinterpret_cast<const float4*>(Lp + f.o_W1);
        const float4* W2 = reinterpret_cast<const float4*>(Lp + f.o_W2);
        const float4* W3 = reinterpret_cast<const float4*>(Lp + f.o_W3);
        const float4* AWI = reinterpret_cast<const float4*>(Lp + f.o_AWI);
        float* Z = lds + cur;
        dense_relu<NTWM, 2, true, false>(Z, l.DS, f.d, f.KBd, W1, Lp + f.o_b1, HA, l.WS, nullptr, t);
        __syncthreads();
        dense_relu<NTWM, DW, false, false>(HA, l.WS, f.Wp, f.KBW, W2, Lp + f.o_b2, HB, l.WS, nullptr, t);
        __syncthreads();
        gemm_ksplit<NTWM>(HB, l.WS, W3, f.NTO, PART, l.PN, t);
        __syncthreads();
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
        dense_small(Z, l.DS, f.D, f.KBD, AWI, f.NTD, lds + nxt, l.DS, t, Lp + f.o_at);
        logq -= -Lp[f.o_logS];
        __syncthreads();
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    *x_off = cur;
    return logq;
}

}  // namespace fab
