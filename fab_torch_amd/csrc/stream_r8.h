// The weight stream + GEMM step shared by the 4x4x1 stream kernels (spline_r8.h: spline conditioner, flow_r8.h: RealNVP):
// every wave reads ITS 1-KiB tiles (4 k x 64 columns, lane = column, float4 = 4 k) as one contiguous stream in the order it
// consumes them, through a ring of S8_RD tiles in ACCUMULATION registers that inline-asm loads fill behind hipcc's back
// (hand-counted s_waitcnt; hipcc never sees a load it could wait for, copy or re-allocate - the AGPR file has no other
// tenant, and v_mfma reads its B operand from it directly).  A k-quad of a row block = 4 v_mfma_f32_4x4x1_16b_f32 on four
// accumulators (k mod 4: the instruction's ~54-cycle dependent latency), added as (a0 + a1) + (a2 + a3) at the end.
// Rules for the code around it (tools/check_r8_isa.py verifies them on the ISA): no loop may carry ring slots whose load
// is in flight EXCEPT a loop whose latch re-requests the ring with s8_prologue (hipcc copies loop-carried slots at the back
// edge otherwise); compiler-tracked loads issued while the ring is in flight only make the hand-counted waits conservative.
#pragma once
#include "flow_device.h"

namespace fab {

constexpr int S8_RD = 32;              // default ring depth: 1-KiB tiles in flight per wave (S8StreamT<RD>)
constexpr int S8_INF = 1 << 20;

__device__ __forceinline__ void s8_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int I, int N, class F>
__device__ __forceinline__ void s8_for(F&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        s8_for<I + 1, N>(fn);
    }
}

// ---- the wave's weight stream ---------------------------------------------------------------------------------------
template <int RD_>
struct S8StreamT {
    static constexpr int RD = RD_;
    f32x4 r[RD_];                      // ring slot of stream tile k: k % RD   ("a" registers)
    static constexpr int NV = RD_ > 32 ? (RD_ + 3) / 4 : 8;
    unsigned voff[NV];                 // lane * 16 + 4096 j: with the 4 immediate offsets, 4 NV tiles from one scalar base
    const float4* next;                // tile that step 0 of the next iteration requests
};
using S8Stream = S8StreamT<S8_RD>;

template <int IMM>
__device__ __forceinline__ void s8_load(f32x4& dst, unsigned voff, const float4* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(sbase), "n"(IMM));
}
template <int IMM>
__device__ __forceinline__ void s8_load_first(f32x4& dst, unsigned voff, const float4* sbase) {   // fresh scalar base: see gload16s_first
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(sbase), "n"(IMM));
}
template <int N>
__device__ __forceinline__ void s8_wait(f32x4& r) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%1)" : "+a"(r) : "n"(N));
}

template <class ST>
__device__ __forceinline__ void s8_stream_init(ST& s, int lane) {
#pragma unroll
    for (int j = 0; j < ST::NV; ++j) s.voff[j] = (unsigned)(lane * 16 + 4096 * j);
}

// layer top: request tiles 0 .. RD-2 of the wave's stream at `base`
template <class ST>
__device__ __forceinline__ void s8_prologue(ST& s, const float4* base) {
    constexpr int S8_RD = ST::RD;
    s8_for<0, S8_RD - 1>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if constexpr (d == 0) s8_load_first<0>(s.r[0], s.voff[0], base);
        else s8_load<(d % 4) * 1024>(s.r[d], s.voff[d / 4], base);
    });
    s.next = base + (size_t)(S8_RD - 1) * 64;
}

// wait for everything the ring has in flight (end of a never-drained stream: the last RD - 1 requests read past its end)
template <class ST>
__device__ __forceinline__ void s8_drain(ST& s) {
    s8_for<0, ST::RD>([&](auto dc) { s8_wait<0>(s.r[decltype(dc)::value]); });
}

// NSTEP k-quads (stream tiles T0 .. T0 + NSTEP - 1, T0 % RD == PHASE) of which the first USE are multiplied:
//   acc[k % 4][rb] += A[rb][4 q + k] (x) B[4 q + k][64 w + lane]
// REMAIN = stream tiles of this layer after T0 (S8_INF: more than 2 RD): a refill is issued only for a tile that exists and
// the wait counts only loads that were issued (the last stages of a layer drain the ring).
// `ap`: this lane's row of the activation tile at the iteration's first quad; `rb1`: float offset of row block 1.
template <int RB>
struct S8Acc {
    static constexpr int KI = 4;                                           // accumulators per row block (k mod 4): the same sums for every RB
    f32x4 a[KI][RB];
};

// KSTEP: floats between the A operands of consecutive tiles (4: one k-quad per tile; 8 / 16: DENSE tiles of a narrow output -
// 2 / 4 k-quads side by side in the tile's lane halves / quarters, the lane's sub-block offset is part of `ap`)
template <int KSTEP, int NSTEP, int USE, int PHASE, int REMAIN, int RB, class ST>
__device__ __forceinline__ void s8_iter_k(ST& s, const float* ap, int rb1, S8Acc<RB>& acc) {
    constexpr int KI = S8Acc<RB>::KI;
    constexpr int S8_RD = ST::RD;
    float4 an[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ap + rb * rb1);
    s8_for<0, NSTEP>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        constexpr int slot = (PHASE + d) % S8_RD;
        constexpr int left = REMAIN - d;                                   // tiles after this one
        constexpr int N = left < S8_RD - 2 ? (left < 0 ? 0 : left) : S8_RD - 2;
        s8_wait<N>(s.r[slot]);
        __builtin_amdgcn_sched_barrier(0);
        float4 a[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a[rb] = an[rb];
        if constexpr (d < USE) {
            if constexpr (d + 1 < USE) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) an[rb] = *reinterpret_cast<const float4*>(ap + rb * rb1 + KSTEP * (d + 1));
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc.a[0][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].x, s.r[slot].x, acc.a[0][rb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (d - 1 + S8_RD <= REMAIN) {                          // top up: tile T0 + d - 1 + RD into the slot of T0 + d - 1
            constexpr int dp = (slot + S8_RD - 1) % S8_RD;
            if constexpr (d == 0) s8_load_first<0>(s.r[dp], s.voff[0], s.next);
            else s8_load<(d % 4) * 1024>(s.r[dp], s.voff[d / 4], s.next);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (d < USE) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc.a[1 % KI][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].y, s.r[slot].y, acc.a[1 % KI][rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc.a[2 % KI][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].z, s.r[slot].z, acc.a[2 % KI][rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc.a[3 % KI][rb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rb].w, s.r[slot].w, acc.a[3 % KI][rb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    s.next += (size_t)NSTEP * 64;
}

template <int NSTEP, int USE, int PHASE, int REMAIN, int RB, class ST>
__device__ __forceinline__ void s8_iter(ST& s, const float* ap, int rb1, S8Acc<RB>& acc) {
    s8_iter_k<4, NSTEP, USE, PHASE, REMAIN>(s, ap, rb1, acc);
}

template <int RB>
__device__ __forceinline__ void s8_zero(S8Acc<RB>& acc) {
#pragma unroll
    for (int k = 0; k < S8Acc<RB>::KI; ++k)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc.a[k][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
template <int RB>
__device__ __forceinline__ void s8_fold(const S8Acc<RB>& acc, f32x4 (&o)[RB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        if constexpr (S8Acc<RB>::KI == 4) o[rb] = (acc.a[0][rb] + acc.a[1][rb]) + (acc.a[2][rb] + acc.a[3][rb]);
        else o[rb] = acc.a[0][rb] + acc.a[1][rb];
    }
}

// NQ k-quads from stream tile T0 of a layer stream of TOTAL tiles, in iterations of at most 32 (static ring slots)
template <int KSTEP, int T0, int NQ, int TOTAL, int RB, class ST>
__device__ __forceinline__ void s8_run_k(ST& s, const float* ap, int rb1, S8Acc<RB>& acc) {
    constexpr int N0 = NQ < 32 ? NQ : 32;
    s8_iter_k<KSTEP, N0, N0, T0 % ST::RD, TOTAL - 1 - T0>(s, ap, rb1, acc);
    if constexpr (NQ > N0) s8_run_k<KSTEP, T0 + N0, NQ - N0, TOTAL>(s, ap + KSTEP * N0, rb1, acc);
}
template <int T0, int NQ, int TOTAL, int RB, class ST>
__device__ __forceinline__ void s8_run(ST& s, const float* ap, int rb1, S8Acc<RB>& acc) {
    s8_run_k<4, T0, NQ, TOTAL>(s, ap, rb1, acc);
}

}  // namespace fab
