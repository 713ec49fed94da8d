// the forward half of one coupling layer on ONE 16-chain tile shared by FOUR CUs of an XCD, built from the
// product's own stream code (csrc/stream_r8.h, RB = 4) and the tagged-data exchange tools/ubench/xcu.hip priced - the first
// building block of the shared-tile kernel, as a timing prototype (random weights, hidden width 256, no parity claim).
//
// Member c of a group {xcd + 8 (4 g + k)} per layer:
//   S1  W1 (16 -> 256), redundant: wave w multiplies columns 64 w .. 64 w + 63 (4 tiles), ReLU -> H1 [16][256] in LDS
//   S2  W2 (256 x 256), N-split over the MEMBERS: member c owns column group c; its four waves split K (16 tiles each),
//       partial sums through LDS, ReLU -> H2 slice [16][64]
//   S3  W3 (256 -> 64 padded), K-split: the member's own 64 rows of K, 4 k-quads per wave (4 tiles), partials through LDS ->
//       the member's partial [16][64]; EXCHANGE: written as 256 tagged float4, every thread re-reads its element of the four
//       members' slabs (sc1) until all carry this layer, adds them in member order -> next layer's input [16][16]
// 24 tiles per wave and layer = the ring depth (static slots).  SOLO = 1: the same stages without the exchange (one member's
// timing alone: what the exchange costs on top).  The object passes the build's ISA check
// (fab_torch_amd/_isa_check.py) - see the note at the ring request below for what did not.  Prints cycles per layer; the 4-chain product kernel spends ~10 k cycles on the
// forward half of a width-256 layer for its 4 chains (0.470 ms per transition / 50 layer pairs / 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../fab_torch_amd/csrc/stream_r8.h"

using namespace fab;

constexpr int XG = 4, XRD = 24, XRB = 4;
constexpr int WS_A = 20, WS_H = 260, WS_S = 68;
constexpr unsigned XPOLL_LIMIT = 1u << 17;

__device__ __forceinline__ void x_ld4_sc1(f32x4& v, const f32x4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
}

struct XOut {
    long long cycles;
    unsigned xcc, fail, pad0, pad1;
    long long stage[6];                       // TL = 1: cycles summed per stage (S1, S2 GEMM, S2 sum, S3, exchange + request, update)
};

// ND = 1: never-drained ring (TOTAL = S8_INF, no s8_prologue at all): iteration 0 runs the SAME body on an empty ring - its
// top-ups are the initial fill (s.next starts one tile early so that tile j lands in slot j mod RD), its results are discarded by
// a select - so the loop carries in-flight slots but has no second request site for hipcc to give other registers.
template <int SOLO, int ORDER, int TL, int ND>
__global__ __launch_bounds__(256) void k_xcu_mlp(const float4* __restrict__ src, size_t wave_f4, f32x4* __restrict__ xbuf,
                                                 int n_layers, float* __restrict__ sink, XOut* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int abort_flag;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3, group = slot / XG, member = slot % XG;
    const int gid = xcd * (32 / XG) + group;
    f32x4* gdata = xbuf + (size_t)gid * 2 * XG * 256;                         // [parity][member][256]
    float* XA = lds;                                                          // [16][WS_A]
    float* H1 = XA + 16 * WS_A;                                               // [16][WS_H]
    float* H2 = H1 + 16 * WS_H;                                               // [16][WS_S]
    float* PART = H2 + 16 * WS_S;                                             // [4 waves][16][64]
    for (int e = tid; e < 16 * WS_A; e += 256) XA[e] = 0.01f * (float)(e % 13);
    if (tid == 0) abort_flag = 0;
    __syncthreads();
    S8StreamT<XRD> s;
    s8_stream_init(s, lane);
    constexpr int XT = ND ? S8_INF : XRD;                                    // tiles of a "layer stream": never drained / one ring
    const float4* wbase = src + (size_t)wave * wave_f4;
    // a layer's 24 tiles = the ring: drained at the layer's end and re-requested at the loop latch (the ISA rule of stream_r8.h).
    // ONE request site: iteration 0 only requests (a second s8_prologue in front of the loop made hipcc give the ring other
    // registers inside the loop - as soon as the loop contains any global access - and copy the in-flight slots at its entry)
    const int arow = lane & 3;
    unsigned fail = 0, timed_out = 0;
    long long tl[6] = {0, 0, 0, 0, 0, 0}, tp = 0;
#define XSTAMP(i) if constexpr (TL) { const long long tn = __builtin_amdgcn_s_memtime(); tl[i] += tn - tp; tp = tn; }
    if constexpr (ND) s.next = wbase - 64;                                   // (one tile early: see ND above; the buffer has slack in front)
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it <= n_layers; ++it) {
        f32x4 o[XRB];
        f32x4 mine = {0.f, 0.f, 0.f, 0.f};
        if constexpr (TL) tp = __builtin_amdgcn_s_memtime();
        if (ND || it > 0) {
        {   // S1: W1, redundant
            S8Acc<XRB> acc;
            s8_zero(acc);
            s8_run_k<4, 0, 4, XT>(s, XA + arow * WS_A, 4 * WS_A, acc);
            s8_fold(acc, o);
#pragma unroll
            for (int rb = 0; rb < XRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) H1[(4 * rb + r) * WS_H + 64 * wave + lane] = fmaxf(o[rb][r], 0.f) * 1e-2f;
            s8_barrier();
            XSTAMP(0)
        }
        {   // S2: W2, the member's column group, K-split over the waves
            S8Acc<XRB> acc;
            s8_zero(acc);
            s8_run_k<4, 4, 16, XT>(s, H1 + arow * WS_H + 64 * wave, 4 * WS_H, acc);
            s8_fold(acc, o);
#pragma unroll
            for (int rb = 0; rb < XRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) PART[(wave * 16 + 4 * rb + r) * 64 + lane] = o[rb][r];
            s8_barrier();
            XSTAMP(1)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = tid + 256 * i, row = e >> 6, col = e & 63;
                const float v = (PART[e] + PART[1024 + e]) + (PART[2048 + e] + PART[3072 + e]);
                H2[row * WS_S + col] = fmaxf(v, 0.f) * 1e-2f;
            }
            s8_barrier();
            XSTAMP(2)
        }
        {   // S3: W3, K-split over the members (own 64 rows of K), over the waves inside the member
            S8Acc<XRB> acc;
            s8_zero(acc);
            s8_run_k<4, 20, 4, XT>(s, H2 + arow * WS_S + 16 * wave, 4 * WS_S, acc);
            s8_fold(acc, o);
#pragma unroll
            for (int rb = 0; rb < XRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) PART[(wave * 16 + 4 * rb + r) * 64 + lane] = o[rb][r];
            s8_barrier();
        }
        {
            const f32x4* P = reinterpret_cast<const f32x4*>(PART);
            mine = (P[tid] + P[256 + tid]) + (P[512 + tid] + P[768 + tid]);    // this member's partial, 4 outputs per thread
        }
        XSTAMP(3)
        }
        if constexpr (ORDER == 1 && !ND) s8_prologue(s, wbase + (size_t)it * XRD * 64);   // the ring fills behind the exchange
        f32x4 total;
        if constexpr (SOLO) {
            total = mine;
            if constexpr (ORDER == 0 && !ND) s8_prologue(s, wbase + (size_t)it * XRD * 64);
        } else {
            mine.x = (float)it;                                              // the tag (a product kernel packs 3 values + tag, or 128-bit LL)
            gdata[((size_t)(it & 1) * XG + member) * 256 + tid] = mine;
            // ONE asm statement, no compiler-visible control flow (hipcc copies in-flight ring registers around an inner loop or a
            // branch - the ISA check of the first two versions found v_accvgpr_mov of slots whose load had not landed): poll the
            // four TAG words of this thread's elements (sc1: served by the L2), then read the four float4.
            f32x4 v0, v1, v2, v3;
            const f32x4* q = gdata + (size_t)(it & 1) * XG * 256 + tid;
            {
                unsigned t0, t1, t2, t3, cnt;
                unsigned long long m;
                const unsigned tag = __float_as_uint((float)it), lim = XPOLL_LIMIT;
                asm volatile(
                    "s_mov_b32 %[cnt], 0\n"
                    "1:\n\t"
                    "global_load_dword %[t0], %[p0], off sc1\n\t"
                    "global_load_dword %[t1], %[p1], off sc1\n\t"
                    "global_load_dword %[t2], %[p2], off sc1\n\t"
                    "global_load_dword %[t3], %[p3], off sc1\n\t"
                    "s_waitcnt vmcnt(0)\n\t"
                    "v_cmp_ne_u32_e64 %[m], %[t0], %[tag]\n\t"
                    "v_cmp_ne_u32_e32 vcc, %[t1], %[tag]\n\t"
                    "s_or_b64 %[m], %[m], vcc\n\t"
                    "v_cmp_ne_u32_e32 vcc, %[t2], %[tag]\n\t"
                    "s_or_b64 %[m], %[m], vcc\n\t"
                    "v_cmp_ne_u32_e32 vcc, %[t3], %[tag]\n\t"
                    "s_or_b64 %[m], %[m], vcc\n\t"
                    "s_cmp_eq_u64 %[m], 0\n\t"
                    "s_cbranch_scc1 2f\n\t"
                    "s_add_u32 %[cnt], %[cnt], 1\n\t"
                    "s_cmp_lt_u32 %[cnt], %[lim]\n\t"
                    "s_cbranch_scc1 1b\n"
                    "2:\n\t"
                    "global_load_dwordx4 %[v0], %[p0], off sc1\n\t"
                    "global_load_dwordx4 %[v1], %[p1], off sc1\n\t"
                    "global_load_dwordx4 %[v2], %[p2], off sc1\n\t"
                    "global_load_dwordx4 %[v3], %[p3], off sc1\n"
                    : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [m] "=&s"(m), [cnt] "=&s"(cnt),
                      [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3)
                    : [p0] "v"(q), [p1] "v"(q + 256), [p2] "v"(q + 512), [p3] "v"(q + 768), [tag] "v"(tag), [lim] "s"(lim)
                    : "vcc", "scc", "memory");
                timed_out |= cnt >= lim;
            }
            // the next layer's ring is requested behind the four data loads (loads return in order: the data first)
            if constexpr (ORDER == 0 && !ND) {
                s8_prologue(s, wbase + (size_t)it * XRD * 64);
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "n"(XRD - 1));
            } else if constexpr (ND) {                                       // the four data loads are the youngest requests
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            } else {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            }
            total = (v0 + v1) + (v2 + v3);
        }
        XSTAMP(4)
        if constexpr (ND) {                                                  // iteration 0 multiplied an empty ring: discard
            const float keepf = it > 0 ? 1.f : 0.f;
            total.y = it > 0 ? total.y : 0.f; total.z = it > 0 ? total.z : 0.f; total.w = it > 0 ? total.w : 0.f;
            (void)keepf;
        }
        // "coupling": the next layer's input, one element per thread
        XA[(tid >> 4) * WS_A + (tid & 15)] = 0.5f * XA[(tid >> 4) * WS_A + (tid & 15)] + 1e-3f * (total.y + total.z + total.w) + 0.01f;
        s8_barrier();
        XSTAMP(5)
    }
    fail = timed_out;
    const long long t1 = __builtin_amdgcn_s_memtime();
    s8_drain(s);
    sink[(size_t)wg * 256 + tid] = XA[tid];
    if (tid == 0) {
        out[wg].cycles = t1 - t0;
        out[wg].xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));
        out[wg].fail = fail;
        for (int i = 0; i < 6; ++i) out[wg].stage[i] = tl[i];
    }
}

template <int SOLO, int ORDER = 0, int TL = 0, int ND = 0>
static void run(const char* name, const float4* src, size_t region, f32x4* xbuf, float* sink, XOut* out) {
    const size_t wave_bytes = region / 4;
    const int n_layers = (int)(wave_bytes / 1024 / XRD) - 2;
    auto kern = k_xcu_mlp<SOLO, ORDER, TL, ND>;
    const size_t lds = 96 * 1024;                                            // one workgroup per CU
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(xbuf, 0, (size_t)64 * 2 * XG * 256 * 16);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, src, wave_bytes / 16, xbuf, n_layers, sink, out);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%s failed: %s\n", name, hipGetErrorString(e)); return; }
    }
    std::vector<XOut> h(256);
    (void)hipMemcpy(h.data(), out, 256 * sizeof(XOut), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    int failed = 0;
    for (int wg = 0; wg < 256; ++wg) {
        mean += (double)h[wg].cycles;
        mx = h[wg].cycles > mx ? (double)h[wg].cycles : mx;
        failed += h[wg].fail != 0;
    }
    mean /= 256;
    printf("%-44s %3d layers: %7.0f cycles per forward layer of 16 chains (slowest WG %7.0f), %d WGs timed out\n", name, n_layers,
           mean / (n_layers + 1), mx / (n_layers + 1), failed);
    if (TL) {
        printf("    stage cycles per layer (WG 0): W1 %lld | W2 GEMM %lld | W2 sum %lld | W3 + partial %lld | exchange + ring request %lld | "
               "update + barrier %lld\n", h[0].stage[0] / n_layers, h[0].stage[1] / n_layers, h[0].stage[2] / n_layers,
               h[0].stage[3] / n_layers, h[0].stage[4] / n_layers, h[0].stage[5] / n_layers);
    }
    fflush(stdout);
}

int main() {
    const size_t region = 10u << 20;
    float4* src; f32x4* xbuf; float* sink; XOut* out;
    float4* src_alloc;
    (void)hipMalloc((void**)&src_alloc, region + (3u << 20));
    src = src_alloc + (1u << 20) / 16;                                       // 1 MiB of slack in front (ND reads one tile early)
    std::vector<float> hw((region + (2u << 20)) / 4);
    unsigned st = 12345u;
    for (auto& w : hw) { st = st * 1664525u + 1013904223u; w = ((float)(st >> 8) / 16777216.f - 0.5f) * 0.2f; }
    (void)hipMemcpy(src, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc((void**)&xbuf, (size_t)64 * 2 * XG * 256 * 16);
    (void)hipMalloc((void**)&sink, 256 * 256 * 4);
    (void)hipMalloc((void**)&out, 256 * sizeof(XOut));
    run<1, 0>("one member alone (no exchange)", src, region, xbuf, sink, out);
    run<0, 0>("four members, ring requested after exchange", src, region, xbuf, sink, out);
    run<0, 1>("four members, ring requested before exchange", src, region, xbuf, sink, out);
    run<1, 0>("one member alone (no exchange)", src, region, xbuf, sink, out);
    run<0, 0>("four members, ring requested after exchange", src, region, xbuf, sink, out);
    run<0, 1>("four members, ring requested before exchange", src, region, xbuf, sink, out);
    run<1, 0, 0, 1>("one member alone, never-drained ring", src, region, xbuf, sink, out);
    run<0, 0, 0, 1>("four members, never-drained ring", src, region, xbuf, sink, out);
    run<1, 0, 0, 1>("one member alone, never-drained ring", src, region, xbuf, sink, out);
    run<0, 0, 0, 1>("four members, never-drained ring", src, region, xbuf, sink, out);
    run<0, 0, 1, 1>("four members, never-drained, stage stamps", src, region, xbuf, sink, out);
    run<1, 0, 1>("one member alone, stage stamps", src, region, xbuf, sink, out);
    run<0, 0, 1>("four members, stage stamps", src, region, xbuf, sink, out);
    return 0;
}
