// AIS inner loop: point creation, HMC / Metropolis transitions with on-device step-size adaptation,
// NaN/inf compaction, log-weight accumulation, and the one-call AIS driver.
//
// One workgroup (256 threads, one wave per SIMD) owns 16 chains for a whole transition: every
// leapfrog's flow log-density + gradient (fp32 MFMA, flow_device.h), target log-density + gradient,
// momentum/position updates, the Metropolis test, the in-place commit and the AIS log-weight
// increment happen inside ONE launch; the only cross-workgroup quantity (mean acceptance, needed
// for the shared `common_epsilon`) is reduced by a one-wave follow-up kernel from per-block partials
// in a fixed order (deterministic).
#include "flow_device.h"
#include "target_device.h"
#include "flow_r4.h"
#include "flow_r4f.h"
#include "flow_r8.h"
#include "launch.h"

#pragma clang fp contract(off)   // keep a*b+c un-fused in the elementwise code, like the eager CPU reference

namespace fab {

struct PointDev {
    float *x, *lq, *lp, *gq, *gp;
};
static inline PointDev make_point_dev(const fabhip_point& p) { return PointDev{p.x, p.log_q, p.log_p, p.grad_log_q, p.grad_log_p}; }

struct ExtraLds {   // per-tile HMC state appended after the flow's LDS plan (floats)
    int o_XP, o_P, o_GU, o_GP, o_ROW, total;
};
static inline ExtraLds make_extra_lds(const FlowLds& l, int D) {
    ExtraLds e;
    int o = l.total;
    e.o_XP = o; o += ROWS * D;
    e.o_P = o; o += ROWS * D;
    e.o_GU = o; o += ROWS * D;
    e.o_GP = o; o += ROWS * D;
    e.o_ROW = o; o += 2 * ROWS;
    e.total = (o + 3) & ~3;
    return e;
}

__device__ __forceinline__ void load_state_to_u0(const FlowLds& l, int D, float* lds, const float* XP, const Tid& t) {
    for (int e = t.tid; e < ROWS * l.DS; e += NTHREADS) {
        const int r = e / l.DS, j = e % l.DS;
        lds[l.o_U0 + e] = j < D ? XP[r * D + j] : 0.f;
    }
}

__device__ __forceinline__ void zero_dp(const FlowLds& l, float* lds, const Tid& t) {
    for (int e = t.tid; e < ROWS * l.PS; e += NTHREADS) lds[l.o_DP + e] = 0.f;
}

__device__ __forceinline__ float clamp_nan0(float g, float mg) {
    // torch.nan_to_num(torch.clamp(g, -mg, mg), nan=0): NaN -> 0, +-inf -> +-mg   (hmc.py:194-199)
    return (g != g) ? 0.f : fminf(fmaxf(g, -mg), mg);
}

// ------------------------------------------------------------------------------------------------
// create_point: log q (+grad), log p (+grad) at point.x      (fab/sampling_methods/base.py:59-72)
// ------------------------------------------------------------------------------------------------
template <int NTWM, bool GRAD, bool FAST>
__device__ __forceinline__ void create_point_body(const FlowDims& f, const FlowLds& l, const ExtraLds& x,
                                                  const float* __restrict__ packed, const TargetDev& tg, const PointDev& pt,
                                                  long B, float* lds) {
    Tid t;
    const int D = f.D;
    const long row0 = (long)blockIdx.x * ROWS;
    float* XP = lds + x.o_XP;
    float* GP = lds + x.o_GP;
    zero_dp(l, lds, t);
    for (int e = t.tid; e < ROWS * D; e += NTHREADS) {
        const long g = row0 + e / D;
        XP[e] = g < B ? pt.x[g * D + e % D] : 0.f;
    }
    __syncthreads();
    load_state_to_u0(l, D, lds, XP, t);
    __syncthreads();
    int goff = 0;
    const float lq = flow_log_prob_tile<NTWM, GRAD, false, FAST>(f, l, packed, lds, t, &goff);
    const float lp = target_tile<GRAD>(tg, XP, D, GP, D, t);
    const long g = row0 + t.row;
    if (g < B) {
        if (t.c == 0) { pt.lq[g] = lq; pt.lp[g] = lp; }
        if (GRAD) {
            for (int j = t.c; j < D; j += 16) {
                pt.gq[g * D + j] = lds[goff + t.row * l.DS + j];
                pt.gp[g * D + j] = GP[t.row * D + j];
            }
        }
    }
}

template <int NTWM, bool GRAD>
__global__ __launch_bounds__(NTHREADS) void k_create_point(FlowDims f, FlowLds l, ExtraLds x, const float* __restrict__ packed,
                                                           TargetDev tg, PointDev pt, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    create_point_body<NTWM, GRAD, false>(f, l, x, packed, tg, pt, B, lds);
}
// fast mode (bf16 W x W GEMMs, flow_device.h): same kernel, not the parity path
template <int NTWM, bool GRAD>
__global__ __launch_bounds__(NTHREADS) void k_create_point_fast(FlowDims f, FlowLds l, ExtraLds x,
                                                                const float* __restrict__ packed, TargetDev tg, PointDev pt,
                                                                long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    create_point_body<NTWM, GRAD, true>(f, l, x, packed, tg, pt, B, lds);
}

// ------------------------------------------------------------------------------------------------
// AIS chain initialisation (ais.py:55-64): x, log_q0 = flow.sample(eps0); point = create_point(x);
// log_w = pi_beta1(point) - log_q0.   With GRAD (HMC) log q is re-evaluated through log_prob, as the
// reference does (base.py:65-68 ignores the supplied log_q_x).
// ------------------------------------------------------------------------------------------------
template <int NTWM, bool GRAD, bool FAST>
__device__ __forceinline__ void ais_init_body(const FlowDims& f, const FlowLds& l, const ExtraLds& x,
                                              const float* __restrict__ packed, const TargetDev& tg,
                                              const float* __restrict__ eps0, const PointDev& pt, float* __restrict__ log_w,
                                              float* __restrict__ base_log_w, const fabhip_anneal& an, long B, float* lds) {
    Tid t;
    const int D = f.D;
    const long row0 = (long)blockIdx.x * ROWS;
    float* XP = lds + x.o_XP;
    float* GP = lds + x.o_GP;
    zero_dp(l, lds, t);
    for (int e = t.tid; e < ROWS * l.DS; e += NTHREADS) {
        const int r = e / l.DS, j = e % l.DS;
        const long g = row0 + r;
        lds[l.o_U0 + e] = (j < D && g < B) ? eps0[g * D + j] : 0.f;
    }
    __syncthreads();
    int xoff = 0;
    const float lq0 = flow_sample_tile<NTWM>(f, l, packed, lds, t, &xoff);
    for (int j = t.c; j < D; j += 16) XP[t.row * D + j] = lds[xoff + t.row * l.DS + j];
    __syncthreads();
    float lq = lq0;
    int goff = 0;
    if (GRAD) {
        load_state_to_u0(l, D, lds, XP, t);
        __syncthreads();
        lq = flow_log_prob_tile<NTWM, true, false, FAST>(f, l, packed, lds, t, &goff);
    }
    const float lp = target_tile<GRAD>(tg, XP, D, GP, D, t);
    const long g = row0 + t.row;
    if (g < B) {
        for (int j = t.c; j < D; j += 16) {
            pt.x[g * D + j] = XP[t.row * D + j];
            if (GRAD) {
                pt.gq[g * D + j] = lds[goff + t.row * l.DS + j];
                pt.gp[g * D + j] = GP[t.row * D + j];
            }
        }
        if (t.c == 0) {
            pt.lq[g] = lq;
            pt.lp[g] = lp;
            log_w[g] = (an.c_q * lq + an.c_p * lp) - lq0;
            if (base_log_w) base_log_w[g] = lp - lq0;          // ais.py:160 (log q of the sampling pass)
        }
    }
}

template <int NTWM, bool GRAD>
__global__ __launch_bounds__(NTHREADS) void k_ais_init(FlowDims f, FlowLds l, ExtraLds x, const float* __restrict__ packed,
                                                       TargetDev tg, const float* __restrict__ eps0, PointDev pt,
                                                       float* __restrict__ log_w, float* __restrict__ base_log_w,
                                                       fabhip_anneal an, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ais_init_body<NTWM, GRAD, false>(f, l, x, packed, tg, eps0, pt, log_w, base_log_w, an, B, lds);
}
// fast mode: the sampling pass stays fp32; the density the transitions continue from is the bf16-GEMM one
template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_ais_init_fast(FlowDims f, FlowLds l, ExtraLds x,
                                                            const float* __restrict__ packed, TargetDev tg,
                                                            const float* __restrict__ eps0, PointDev pt,
                                                            float* __restrict__ log_w, float* __restrict__ base_log_w,
                                                            fabhip_anneal an, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ais_init_body<NTWM, true, true>(f, l, x, packed, tg, eps0, pt, log_w, base_log_w, an, B, lds);
}

// ------------------------------------------------------------------------------------------------
// One HMC outer step (hmc.py:129-160) for 16 chains per workgroup.
// ------------------------------------------------------------------------------------------------
struct HmcK {
    PointDev cur, start, prop_out;   // prop_out.x == nullptr: do not store the proposal
    long B;
    const int* n_valid;
    fabhip_anneal c, nx;
    float* log_w;                    // nullptr: no log-weight update in this launch
    const float* noise_p;            // [B][D] for this outer step
    const float* noise_e;            // [B]
    const float* eps_ptr;            // epsilons[i-1][n]
    const float* ceps_ptr;           // common_epsilon
    const float* mass;
    int L;
    float max_grad;
    float* part_acc;                 // [nblk] sum of min(1, acceptance prob)
    float* part_dist;                // [nblk] sum of the store_info distance
    float* row_acc;                  // 4-chain tiles: the same two quantities per chain [16 nblk]; k_hmc_adapt adds them
    float* row_dist;                 //   in the 16-chain kernel's order (16 rows, then blocks)
    // step-size rule inside the transition kernel (4- / 8-chain tiles of a fused AIS call; hmc_adapt_last): ticket == nullptr
    // leaves it to k_hmc_adapt
    int* ticket;                     // zero before the launch; the last wave to finish resets it
    float* eps_w;                    // = eps_ptr / ceps_ptr, writable
    float* ceps_w;
    float target_p_accept;
    int tune;
    float* p_accept_out;
    float* dist_out;
    int nblk;
};

__device__ __forceinline__ void hmc_adapt_rule(float s, float d, long nv, float* eps_ptr, float* ceps_ptr,
                                               float target_p_accept, int tune, float* p_accept_out, float* dist_out);

// k_hmc_adapt's work (per-chain values -> blocks of 16 rows in row order -> blocks in order -> the rule: the same additions, bit
// for bit) done by the LAST wave of the launch to finish, instead of a launch of its own after every transition kernel (6 us + the
// launch gap, eight times per AIS call).  Every wave that stores row_acc / row_dist calls this once, after its stores
// (`n_waves` = such waves per workgroup).  The L2s of the 8 XCDs are not coherent with each other, and a device-scope release
// fence (buffer_wbl2) costs as much as the launch it would save (measured: +6 us per kernel): the per-chain values are written
// and read with DEVICE-SCOPE relaxed atomics instead (sc1: written through / read past the XCD's L2), the wave waits for its
// stores (vmcnt) before it draws its ticket, and the wave that draws the last one sums and applies the rule.  Nobody reads the
// step sizes any more by then (every workgroup loads them at its top); the next launch sees the new ones.
// `scratch`: 2 nblk floats of LDS that no wave of this workgroup still reads.
__device__ __forceinline__ void hmc_store_row_stats(const HmcK& a, long g, float contrib, float dist) {
    if (a.ticket) {
        __hip_atomic_store(a.row_acc + g, contrib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.row_dist + g, dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        a.row_acc[g] = contrib; a.row_dist[g] = dist;
    }
}
__device__ __forceinline__ void hmc_adapt_last(const HmcK& a, float* scratch, int lane, int n_waves) {
    if (!a.ticket) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk = __builtin_amdgcn_readfirstlane(tk);
    if (tk != (int)gridDim.x * n_waves - 1) return;
    const int nblk = a.nblk;
    for (int b = lane; b < nblk; b += 64) {
        float s = 0.f, d = 0.f;
        for (int r = 0; r < ROWS; ++r) {
            s += __hip_atomic_load(a.row_acc + (long)b * ROWS + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d += __hip_atomic_load(a.row_dist + (long)b * ROWS + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        scratch[b] = s; scratch[nblk + b] = d;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // one wave: its LDS writes are done before lane 0 reads them
    if (lane == 0) {
        const long nv = a.n_valid ? (long)*a.n_valid : a.B;
        if (nv > 0) {
            float s = 0.f, d = 0.f;
            for (int i = 0; i < nblk; ++i) { s += scratch[i]; d += scratch[nblk + i]; }
            hmc_adapt_rule(s, d, nv, a.eps_w, a.ceps_w, a.target_p_accept, a.tune, a.p_accept_out, a.dist_out);
        }
        __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NTWM, bool FAST>
__device__ __forceinline__ void hmc_step_body(const FlowDims& f, const FlowLds& l, const ExtraLds& x,
                                              const float* __restrict__ packed, const TargetDev& tg, const HmcK& a,
                                              float* lds) {
    Tid t;
    const int D = f.D;
    const long nv = a.n_valid ? (long)*a.n_valid : a.B;
    const long row0 = (long)blockIdx.x * ROWS;
    if (row0 >= nv) {
        if (t.tid == 0) { a.part_acc[blockIdx.x] = 0.f; a.part_dist[blockIdx.x] = 0.f; }
        return;
    }
    float* XP = lds + x.o_XP;
    float* P = lds + x.o_P;
    float* GU = lds + x.o_GU;
    float* GP = lds + x.o_GP;
    float* ROWB = lds + x.o_ROW;
    const long g = row0 + t.row;
    const bool active = g < nv;
    const float eps = *a.eps_ptr + *a.ceps_ptr;                     // get_epsilon (hmc.py:90-100)
    zero_dp(l, lds, t);
    float k0 = 0.f;
    for (int j = t.c; j < D; j += 16) {
        const float m = a.mass[j];
        float xs = 0.f, p = 0.f, gu = 0.f;
        if (active) {
            xs = a.start.x[g * D + j];
            p = a.noise_p[g * D + j] * m;                           // hmc.py:134 (x mass, not sqrt)
            const float gr = -(a.c.g_q * a.start.gq[g * D + j] + a.c.g_p * a.start.gp[g * D + j]);
            gu = clamp_nan0(gr, a.max_grad);
        }
        XP[t.row * D + j] = xs; P[t.row * D + j] = p; GU[t.row * D + j] = gu;
        k0 += p * p / m;
    }
    k0 = row16_sum(k0) / 2.f;
    float lq_c = 0.f, lp_c = 0.f;
    if (active) { lq_c = a.cur.lq[g]; lp_c = a.cur.lp[g]; }
    const float logp_cur = (a.c.c_q * lq_c + a.c.c_p * lp_c) - k0;  // -U(current) - K(p0)
    float lq = 0.f, lp = 0.f;
    int goff = 0;
    for (int step = 0; step < a.L; ++step) {
        for (int j = t.c; j < D; j += 16) {
            const float m = a.mass[j];
            float p = P[t.row * D + j] - eps * GU[t.row * D + j] / 2.f;
            const float xn = XP[t.row * D + j] + eps / m * p;
            P[t.row * D + j] = p; XP[t.row * D + j] = xn;
        }
        __syncthreads();
        load_state_to_u0(l, D, lds, XP, t);
        __syncthreads();
        lq = flow_log_prob_tile<NTWM, true, false, FAST>(f, l, packed, lds, t, &goff);
        lp = target_tile<true>(tg, XP, D, GP, D, t);
        for (int j = t.c; j < D; j += 16) {
            const float gr = -(a.c.g_q * lds[goff + t.row * l.DS + j] + a.c.g_p * GP[t.row * D + j]);
            const float gu = clamp_nan0(gr, a.max_grad);
            GU[t.row * D + j] = gu;
            P[t.row * D + j] = P[t.row * D + j] - eps * gu / 2.f;
        }
    }
    // Metropolis test (hmc.py:105-124)
    float k1 = 0.f, dist2 = 0.f;
    for (int j = t.c; j < D; j += 16) {
        const float p = P[t.row * D + j];
        k1 += p * p / a.mass[j];
        if (active) { const float dx = a.cur.x[g * D + j] - XP[t.row * D + j]; dist2 += dx * dx; }
    }
    k1 = row16_sum(k1) / 2.f;
    dist2 = row16_sum(dist2);
    const float logp_prop = (a.c.c_q * lq + a.c.c_p * lp) - k1;
    const float delta = logp_prop - logp_cur;
    const bool valid = isfinite(delta);
    const float dd = valid ? delta : -INFINITY;
    bool accept = false;
    float contrib = 0.f, dist = 0.f;
    if (active) {
        accept = valid && (dd > -a.noise_e[g]);
        contrib = expf(fminf(dd, 0.f));
        dist = accept ? 0.f : sqrtf(dist2);      // store_info sees the already-committed point (hmc.py:154-156)
    }
    if (a.prop_out.x && active) {                 // n_outer > 1: the next outer step starts from the PROPOSAL
        for (int j = t.c; j < D; j += 16) {
            a.prop_out.x[g * D + j] = XP[t.row * D + j];
            a.prop_out.gq[g * D + j] = lds[goff + t.row * l.DS + j];
            a.prop_out.gp[g * D + j] = GP[t.row * D + j];
        }
        if (t.c == 0) { a.prop_out.lq[g] = lq; a.prop_out.lp[g] = lp; }
    }
    if (accept) {                                 // current_point[accept] = point[accept]
        for (int j = t.c; j < D; j += 16) {
            a.cur.x[g * D + j] = XP[t.row * D + j];
            a.cur.gq[g * D + j] = lds[goff + t.row * l.DS + j];
            a.cur.gp[g * D + j] = GP[t.row * D + j];
        }
        if (t.c == 0) { a.cur.lq[g] = lq; a.cur.lp[g] = lp; }
    }
    if (a.log_w && active && t.c == 0) {          // ais.py:93-100
        const float lqf = accept ? lq : lq_c, lpf = accept ? lp : lp_c;
        const float num = a.nx.c_q * lqf + a.nx.c_p * lpf;
        const float den = a.c.c_q * lqf + a.c.c_p * lpf;
        a.log_w[g] = a.log_w[g] + (num - den);
    }
    if (t.c == 0) { ROWB[t.row] = contrib; ROWB[ROWS + t.row] = dist; }
    __syncthreads();
    if (t.tid == 0) {
        float s = 0.f, d = 0.f;
        for (int r = 0; r < ROWS; ++r) { s += ROWB[r]; d += ROWB[ROWS + r]; }
        a.part_acc[blockIdx.x] = s;
        a.part_dist[blockIdx.x] = d;
    }
}

template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_hmc_step(FlowDims f, FlowLds l, ExtraLds x, const float* __restrict__ packed,
                                                       TargetDev tg, HmcK a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    hmc_step_body<NTWM, false>(f, l, x, packed, tg, a, lds);
}
// fast mode (fabhip_set_fast_mode): the W x W GEMMs of every coupling layer on the bf16 matrix cores
template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_hmc_step_fast(FlowDims f, FlowLds l, ExtraLds x,
                                                            const float* __restrict__ packed, TargetDev tg, HmcK a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    hmc_step_body<NTWM, true>(f, l, x, packed, tg, a, lds);
}

// ------------------------------------------------------------------------------------------------
// The same HMC outer step for FOUR chains per workgroup (flow_r4.h): used when 16-chain tiles would leave most of
// the chip idle (B <= 1152: 1024 chains = 256 workgroups instead of 64).  Element-wise work runs on wave 0 with the
// 16-chain code's (row = tid >> 4, c = tid & 15) mapping; acceptance / distance contributions go out per chain.
// ------------------------------------------------------------------------------------------------
struct ExtraLds4 {
    int o_XP, o_P, o_GU, o_GP, total;
};
static inline ExtraLds4 make_extra_lds4(const R4Lds& l, int D) {
    ExtraLds4 e;
    int o = l.total;
    e.o_XP = o; o += R4 * D;
    e.o_P = o; o += R4 * D;
    e.o_GU = o; o += R4 * D;
    e.o_GP = o; o += R4 * D;
    e.total = (o + 3) & ~3;
    return e;
}

// MODE: 0 = per-stage weight requests (flow_log_prob_r4), 1 = one stream per wave (flow_log_prob_r4s), 2 = fused stages on
// their own stream (flow_r4f.h: flow_log_prob_r4f; the bias blocks of all layers are copied to LDS once per launch), 3 = the
// fused stages in FAST mode (bf16 W x W tiles: never the parity path), 4 = the fused stages with the last R4F_NS items of every
// W x W stage prefetched into LDS during the short stages (flow_r4f.h "stash": same arithmetic, bit-identical to MODE 2)
constexpr int R4F_NS = 3;
template <int NTWM, bool BIGD, int MODE>
__global__ __launch_bounds__(NTHREADS) void k_hmc_step_r4(FlowDims f, R4Dims rd, R4Lds l, ExtraLds4 x,
                                                          const float* __restrict__ packed, TargetDev tg, HmcK a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid4 t4;
    Tid t;                                           // (row, c) of the element-wise mapping; valid rows: tid < 64
    const int D = f.D;
    const long nv = a.n_valid ? (long)*a.n_valid : a.B;
    const long row0 = (long)blockIdx.x * R4;
    const bool ew = t.tid < 64;
    const long g = row0 + t.row;
    if (row0 >= nv) {
        if (ew && t.c == 0 && g < (a.B + 15) / 16 * 16) hmc_store_row_stats(a, g, 0.f, 0.f);
        if (ew) hmc_adapt_last(a, lds, t.tid, 1);
        return;
    }
    float* XP = lds + x.o_XP;
    float* P = lds + x.o_P;
    float* GU = lds + x.o_GU;
    float* GP = lds + x.o_GP;
    const bool active = ew && g < nv;
    const float eps = *a.eps_ptr + *a.ceps_ptr;
    for (int e = t.tid; e < R4 * R4_DS; e += NTHREADS) { lds[l.o_DP + e] = 0.f; lds[l.o_PRM + e] = 0.f; }
    if constexpr (MODE >= 2) r4f_load_bias(packed + f.o_r4fb, lds + l.o_BIAS, f.K * r4f_bias_stride(f.Wp), t.tid);
    float k0 = 0.f;
    if (ew) {
        for (int j = t.c; j < D; j += 16) {
            const float m = a.mass[j];
            float xs = 0.f, p = 0.f, gu = 0.f;
            if (active) {
                xs = a.start.x[g * D + j];
                p = a.noise_p[g * D + j] * m;
                const float gr = -(a.c.g_q * a.start.gq[g * D + j] + a.c.g_p * a.start.gp[g * D + j]);
                gu = clamp_nan0(gr, a.max_grad);
            }
            XP[t.row * D + j] = xs; P[t.row * D + j] = p; GU[t.row * D + j] = gu;
            k0 += p * p / m;
        }
        k0 = row16_sum(k0) / 2.f;
    }
    float lq_c = 0.f, lp_c = 0.f;
    if (active) { lq_c = a.cur.lq[g]; lp_c = a.cur.lp[g]; }
    const float logp_cur = (a.c.c_q * lq_c + a.c.c_p * lp_c) - k0;
    float lq = 0.f, lp = 0.f;
    int goff = 0;
    for (int step = 0; step < a.L; ++step) {
        if (ew) {
            for (int j = t.c; j < D; j += 16) {
                const float m = a.mass[j];
                float p = P[t.row * D + j] - eps * GU[t.row * D + j] / 2.f;
                const float xn = XP[t.row * D + j] + eps / m * p;
                P[t.row * D + j] = p; XP[t.row * D + j] = xn;
            }
        }
        __syncthreads();
        for (int e = t.tid; e < R4 * R4_DS; e += NTHREADS) {
            const int r = e / R4_DS, j = e % R4_DS;
            lds[l.o_X0 + e] = j < D ? XP[r * D + j] : 0.f;
        }
        __syncthreads();
        if constexpr (MODE == 4) lq = flow_log_prob_r4f<NTWM, false, R4F_NS>(f, l, packed, lds, t4, &goff, lds + x.total);
        else if constexpr (MODE == 3) lq = flow_log_prob_r4f<NTWM, true>(f, l, packed, lds, t4, &goff);
        else if constexpr (MODE == 2) lq = flow_log_prob_r4f<NTWM>(f, l, packed, lds, t4, &goff);
        else if constexpr (MODE == 1) lq = flow_log_prob_r4s<NTWM>(f, rd, l, packed, lds, t4, &goff);
        else lq = flow_log_prob_r4<NTWM, BIGD ? 4 : 2, BIGD ? 4 : 2, BIGD ? 4 : 2, BIGD ? 2 : 1>(f, rd, l, packed, lds, t4, &goff);
        if (ew) {
            lp = target_tile<true>(tg, XP, D, GP, D, t);
            for (int j = t.c; j < D; j += 16) {
                const float gr = -(a.c.g_q * lds[goff + t.row * R4_DS + j] + a.c.g_p * GP[t.row * D + j]);
                const float gu = clamp_nan0(gr, a.max_grad);
                GU[t.row * D + j] = gu;
                P[t.row * D + j] = P[t.row * D + j] - eps * gu / 2.f;
            }
        }
    }
    if (!ew) return;
    // Metropolis test (hmc.py:105-124)
    float k1 = 0.f, dist2 = 0.f;
    for (int j = t.c; j < D; j += 16) {
        const float p = P[t.row * D + j];
        k1 += p * p / a.mass[j];
        if (active) { const float dx = a.cur.x[g * D + j] - XP[t.row * D + j]; dist2 += dx * dx; }
    }
    k1 = row16_sum(k1) / 2.f;
    dist2 = row16_sum(dist2);
    const float logp_prop = (a.c.c_q * lq + a.c.c_p * lp) - k1;
    const float delta = logp_prop - logp_cur;
    const bool valid = isfinite(delta);
    const float dd = valid ? delta : -INFINITY;
    bool accept = false;
    float contrib = 0.f, dist = 0.f;
    if (active) {
        accept = valid && (dd > -a.noise_e[g]);
        contrib = expf(fminf(dd, 0.f));
        dist = accept ? 0.f : sqrtf(dist2);
    }
    // (first: the latency of the write-through stores of the in-kernel step-size rule hides behind the commit stores)
    if (t.c == 0 && g < (a.B + 15) / 16 * 16) hmc_store_row_stats(a, g, contrib, dist);
    if (a.prop_out.x && active) {
        for (int j = t.c; j < D; j += 16) {
            a.prop_out.x[g * D + j] = XP[t.row * D + j];
            a.prop_out.gq[g * D + j] = lds[goff + t.row * R4_DS + j];
            a.prop_out.gp[g * D + j] = GP[t.row * D + j];
        }
        if (t.c == 0) { a.prop_out.lq[g] = lq; a.prop_out.lp[g] = lp; }
    }
    if (accept) {
        for (int j = t.c; j < D; j += 16) {
            a.cur.x[g * D + j] = XP[t.row * D + j];
            a.cur.gq[g * D + j] = lds[goff + t.row * R4_DS + j];
            a.cur.gp[g * D + j] = GP[t.row * D + j];
        }
        if (t.c == 0) { a.cur.lq[g] = lq; a.cur.lp[g] = lp; }
    }
    if (a.log_w && active && t.c == 0) {
        const float lqf = accept ? lq : lq_c, lpf = accept ? lp : lp_c;
        const float num = a.nx.c_q * lqf + a.nx.c_p * lpf;
        const float den = a.c.c_q * lqf + a.c.c_p * lpf;
        a.log_w[g] = a.log_w[g] + (num - den);
    }
    hmc_adapt_last(a, lds, t.tid, 1);                 // (wave 0 is the only wave left)
}

// Chain initialisation on 4-chain tiles (HMC runs of <= 1152 chains): the flow SAMPLE stays on the 16-chain kernel
// (fabhip_flow_sample: no 4-chain tile code for that direction), this kernel re-evaluates log q + d/dx at the samples (the
// reference does, base.py:65-68), the target and the initial log-weight - two of the three flow passes of k_ais_init on
// 256 instead of 64 workgroups.
template <int NTWM, int MODE>
__global__ __launch_bounds__(NTHREADS) void k_ais_init_r4(FlowDims f, R4Dims rd, R4Lds l, ExtraLds4 x,
                                                          const float* __restrict__ packed, TargetDev tg,
                                                          const float* __restrict__ lq0, const float* __restrict__ eps0,
                                                          PointDev pt, float* __restrict__ log_w,
                                                          float* __restrict__ base_log_w, fabhip_anneal an, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid4 t4;
    Tid t;
    const int D = f.D;
    const long row0 = (long)blockIdx.x * R4;
    const bool ew = t.tid < 64;
    const long g = row0 + t.row;
    float* XP = lds + x.o_XP;
    float* GP = lds + x.o_GP;
    float q0s = 0.f;                                   // log q of the sampling pass (eps0 given: the flow SAMPLE runs here as well)
    bool sampled = false;
    if constexpr (MODE >= 1) {
        if (eps0) {                                    // x, log q0 = flow.sample(eps0) on this tile (k_flow_sample_r4's work)
            for (int e = t.tid; e < R4 * R4_DS; e += NTHREADS) {
                const int r = e / R4_DS, j = e % R4_DS;
                lds[l.o_X0 + e] = (j < D && row0 + r < B) ? eps0[(row0 + r) * D + j] : 0.f;
            }
            if constexpr (MODE >= 2)                   // the sampling direction's bias blocks (K + 1 virtual layers)
                r4f_load_bias(packed + f.o_r4fb + (size_t)f.K * r4f_bias_stride(f.Wp), lds + l.o_BIAS,
                              (f.K + 1) * r4f_bias_stride(f.Wp), t.tid);
            __syncthreads();
            int xoff = 0;
            if constexpr (MODE >= 2) q0s = flow_sample_r4f<NTWM>(f, l, packed, lds, t4, &xoff);       // (the sample stays fp32)
            else q0s = flow_sample_r4s<NTWM>(f, rd, l, packed, lds, t4, &xoff);
            for (int e = t.tid; e < R4 * D; e += NTHREADS) {
                const int r = e / D, j = e % D;
                const float v = lds[xoff + r * R4_DS + j];
                XP[r * D + j] = v;
                if (row0 + r < B) pt.x[(row0 + r) * D + j] = v;
            }
            __syncthreads();
            sampled = true;
        }
    }
    for (int e = t.tid; e < R4 * R4_DS; e += NTHREADS) { lds[l.o_DP + e] = 0.f; lds[l.o_PRM + e] = 0.f; }
    for (int e = t.tid; e < R4 * R4_DS; e += NTHREADS) {
        const int r = e / R4_DS, j = e % R4_DS;
        float v;
        if (sampled) v = (j < D && row0 + r < B) ? XP[r * D + j] : 0.f;
        else v = (j < D && row0 + r < B) ? pt.x[(row0 + r) * D + j] : 0.f;
        lds[l.o_X0 + e] = v;
        if (j < D) XP[r * D + j] = v;
    }
    if constexpr (MODE >= 2) r4f_load_bias(packed + f.o_r4fb, lds + l.o_BIAS, f.K * r4f_bias_stride(f.Wp), t.tid);
    __syncthreads();
    int goff = 0;
    float lq;
    if constexpr (MODE == 3) lq = flow_log_prob_r4f<NTWM, true>(f, l, packed, lds, t4, &goff);
    else if constexpr (MODE == 2) lq = flow_log_prob_r4f<NTWM>(f, l, packed, lds, t4, &goff);
    else if constexpr (MODE == 1) lq = flow_log_prob_r4s<NTWM>(f, rd, l, packed, lds, t4, &goff);
    else lq = flow_log_prob_r4<NTWM, 2, 2, 2, 1>(f, rd, l, packed, lds, t4, &goff);
    if (!ew) return;
    const float lp = target_tile<true>(tg, XP, D, GP, D, t);
    if (g < B) {
        for (int j = t.c; j < D; j += 16) {
            pt.gq[g * D + j] = lds[goff + t.row * R4_DS + j];
            pt.gp[g * D + j] = GP[t.row * D + j];
        }
        if (t.c == 0) {
            const float q0 = sampled ? q0s : lq0[g];
            pt.lq[g] = lq;
            pt.lp[g] = lp;
            log_w[g] = (an.c_q * lq + an.c_p * lp) - q0;
            if (base_log_w) base_log_w[g] = lp - q0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same HMC outer step for EIGHT chains per workgroup (flow_r8.h; G = hidden width / 64 = 4 or 5): half the weight
// bytes per chain of the 4-chain kernel.  Element-wise work on threads < 128 with the 16-chain code's (row = tid >> 4,
// c = tid & 15) mapping; acceptance / distance contributions go out per chain, as from the 4-chain kernel.
// ------------------------------------------------------------------------------------------------
struct ExtraLds8 {
    int o_XP, o_P, o_GU, o_GP, total;
};
static inline ExtraLds8 make_extra_lds8(const R8Lds& l, int D) {
    ExtraLds8 e;
    int o = l.total;
    e.o_XP = o; o += R8 * D;
    e.o_P = o; o += R8 * D;
    e.o_GU = o; o += R8 * D;
    e.o_GP = o; o += R8 * D;
    e.total = (o + 3) & ~3;
    return e;
}

template <int G, bool FUSED>
__global__ __launch_bounds__(NTHREADS) void k_hmc_step_r8(FlowDims f, R8Lds l, ExtraLds8 x, const float* __restrict__ packed,
                                                        TargetDev tg, HmcK a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = NTHREADS;
    Tid8f t8;
    Tid t;                                           // (row, c) of the element-wise mapping; valid rows: tid < 128
    const int D = f.D;
    const long nv = a.n_valid ? (long)*a.n_valid : a.B;
    const long row0 = (long)blockIdx.x * R8;
    const bool ew = t.tid < 16 * R8;
    const long g = row0 + t.row;
    if (row0 >= nv) {
        if (ew && t.c == 0 && g < (a.B + 15) / 16 * 16) hmc_store_row_stats(a, g, 0.f, 0.f);
        if (ew) hmc_adapt_last(a, lds, t.tid & 63, 16 * R8 / 64);
        return;
    }
    float* XP = lds + x.o_XP;
    float* P = lds + x.o_P;
    float* GU = lds + x.o_GU;
    float* GP = lds + x.o_GP;
    const bool active = ew && g < nv;
    const float eps = *a.eps_ptr + *a.ceps_ptr;
    if constexpr (FUSED) r8f_load_heads(f, l, packed, lds, t.tid, NT);
    else r8_load_heads(f, l, packed, lds, t.tid, NT);
    for (int e = t.tid; e < R8 * R4_DS; e += NT) { lds[l.o_DP + e] = 0.f; lds[l.o_PRM + e] = 0.f; }
    std::conditional_t<FUSED, R8FStream, R8Stream> s;
    s8_stream_init(s, t8.lane);
    float k0 = 0.f;
    if (ew) {
        for (int j = t.c; j < D; j += 16) {
            const float m = a.mass[j];
            float xs = 0.f, p = 0.f, gu = 0.f;
            if (active) {
                xs = a.start.x[g * D + j];
                p = a.noise_p[g * D + j] * m;
                const float gr = -(a.c.g_q * a.start.gq[g * D + j] + a.c.g_p * a.start.gp[g * D + j]);
                gu = clamp_nan0(gr, a.max_grad);
            }
            XP[t.row * D + j] = xs; P[t.row * D + j] = p; GU[t.row * D + j] = gu;
            k0 += p * p / m;
        }
        k0 = row16_sum(k0) / 2.f;
    }
    float lq_c = 0.f, lp_c = 0.f;
    if (active) { lq_c = a.cur.lq[g]; lp_c = a.cur.lp[g]; }
    const float logp_cur = (a.c.c_q * lq_c + a.c.c_p * lp_c) - k0;
    float lq = 0.f, lp = 0.f;
    int goff = 0;
    for (int step = 0; step < a.L; ++step) {
        if (ew) {
            for (int j = t.c; j < D; j += 16) {
                const float m = a.mass[j];
                float p = P[t.row * D + j] - eps * GU[t.row * D + j] / 2.f;
                const float xn = XP[t.row * D + j] + eps / m * p;
                P[t.row * D + j] = p; XP[t.row * D + j] = xn;
            }
        }
        __syncthreads();
        for (int e = t.tid; e < R8 * R4_DS; e += NT) {
            const int r = e / R4_DS, j = e % R4_DS;
            lds[l.o_X0 + e] = j < D ? XP[r * D + j] : 0.f;
        }
        __syncthreads();
        lq = flow_log_prob_r8<G, FUSED>(f, l, packed, lds, t8, s, &goff);
        if (ew) {
            lp = target_tile<true>(tg, XP, D, GP, D, t);
            for (int j = t.c; j < D; j += 16) {
                const float gr = -(a.c.g_q * lds[goff + t.row * R4_DS + j] + a.c.g_p * GP[t.row * D + j]);
                const float gu = clamp_nan0(gr, a.max_grad);
                GU[t.row * D + j] = gu;
                P[t.row * D + j] = P[t.row * D + j] - eps * gu / 2.f;
            }
        }
    }
    if (!ew) return;
    // Metropolis test (hmc.py:105-124)
    float k1 = 0.f, dist2 = 0.f;
    for (int j = t.c; j < D; j += 16) {
        const float p = P[t.row * D + j];
        k1 += p * p / a.mass[j];
        if (active) { const float dx = a.cur.x[g * D + j] - XP[t.row * D + j]; dist2 += dx * dx; }
    }
    k1 = row16_sum(k1) / 2.f;
    dist2 = row16_sum(dist2);
    const float logp_prop = (a.c.c_q * lq + a.c.c_p * lp) - k1;
    const float delta = logp_prop - logp_cur;
    const bool valid = isfinite(delta);
    const float dd = valid ? delta : -INFINITY;
    bool accept = false;
    float contrib = 0.f, dist = 0.f;
    if (active) {
        accept = valid && (dd > -a.noise_e[g]);
        contrib = expf(fminf(dd, 0.f));
        dist = accept ? 0.f : sqrtf(dist2);
    }
    // (first: the latency of the write-through stores of the in-kernel step-size rule hides behind the commit stores)
    if (t.c == 0 && g < (a.B + 15) / 16 * 16) hmc_store_row_stats(a, g, contrib, dist);
    if (a.prop_out.x && active) {
        for (int j = t.c; j < D; j += 16) {
            a.prop_out.x[g * D + j] = XP[t.row * D + j];
            a.prop_out.gq[g * D + j] = lds[goff + t.row * R4_DS + j];
            a.prop_out.gp[g * D + j] = GP[t.row * D + j];
        }
        if (t.c == 0) { a.prop_out.lq[g] = lq; a.prop_out.lp[g] = lp; }
    }
    if (accept) {
        for (int j = t.c; j < D; j += 16) {
            a.cur.x[g * D + j] = XP[t.row * D + j];
            a.cur.gq[g * D + j] = lds[goff + t.row * R4_DS + j];
            a.cur.gp[g * D + j] = GP[t.row * D + j];
        }
        if (t.c == 0) { a.cur.lq[g] = lq; a.cur.lp[g] = lp; }
    }
    if (a.log_w && active && t.c == 0) {
        const float lqf = accept ? lq : lq_c, lpf = accept ? lp : lp_c;
        const float num = a.nx.c_q * lqf + a.nx.c_p * lpf;
        const float den = a.c.c_q * lqf + a.c.c_p * lpf;
        a.log_w[g] = a.log_w[g] + (num - den);
    }
    hmc_adapt_last(a, lds, t.tid & 63, 16 * R8 / 64);   // (the two element-wise waves; the others have returned)
}

// Chain initialisation on 8-chain tiles (as k_ais_init_r4: the flow SAMPLE stays on the 16-chain kernel)
template <int G, bool FUSED>
__global__ __launch_bounds__(NTHREADS) void k_ais_init_r8(FlowDims f, R8Lds l, ExtraLds8 x, const float* __restrict__ packed,
                                                        TargetDev tg, const float* __restrict__ lq0, PointDev pt,
                                                        float* __restrict__ log_w, float* __restrict__ base_log_w,
                                                        fabhip_anneal an, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = NTHREADS;
    Tid8f t8;
    Tid t;
    const int D = f.D;
    const long row0 = (long)blockIdx.x * R8;
    const bool ew = t.tid < 16 * R8;
    const long g = row0 + t.row;
    float* XP = lds + x.o_XP;
    float* GP = lds + x.o_GP;
    if constexpr (FUSED) r8f_load_heads(f, l, packed, lds, t.tid, NT);
    else r8_load_heads(f, l, packed, lds, t.tid, NT);
    for (int e = t.tid; e < R8 * R4_DS; e += NT) { lds[l.o_DP + e] = 0.f; lds[l.o_PRM + e] = 0.f; }
    for (int e = t.tid; e < R8 * R4_DS; e += NT) {
        const int r = e / R4_DS, j = e % R4_DS;
        const float v = (j < D && row0 + r < B) ? pt.x[(row0 + r) * D + j] : 0.f;
        lds[l.o_X0 + e] = v;
        if (j < D) XP[r * D + j] = v;
    }
    std::conditional_t<FUSED, R8FStream, R8Stream> s;
    s8_stream_init(s, t8.lane);
    __syncthreads();
    int goff = 0;
    const float lq = flow_log_prob_r8<G, FUSED>(f, l, packed, lds, t8, s, &goff);
    if (!ew) return;
    const float lp = target_tile<true>(tg, XP, D, GP, D, t);
    if (g < B) {
        for (int j = t.c; j < D; j += 16) {
            pt.gq[g * D + j] = lds[goff + t.row * R4_DS + j];
            pt.gp[g * D + j] = GP[t.row * D + j];
        }
        if (t.c == 0) {
            const float q0 = lq0[g];
            pt.lq[g] = lq;
            pt.lp[g] = lp;
            log_w[g] = (an.c_q * lq + an.c_p * lp) - q0;
            if (base_log_w) base_log_w[g] = lp - q0;
        }
    }
}

// step-size adaptation from the block partials (hmc.py:122-123,162-170), fixed summation order
// (4-chain tiles hand over per-chain values: the 64 threads first add each block's 16 rows in row order, which is
// what a 16-chain workgroup writes, so both tile shapes give the step-size rule bit-identical sums)
__device__ __forceinline__ void hmc_adapt_rule(float s, float d, long nv, float* eps_ptr, float* ceps_ptr,
                                               float target_p_accept, int tune, float* p_accept_out, float* dist_out);

// `slab` != nullptr (chains sharded over ranks): publish [acc[nblk] | dist[nblk] | n] for the all-gather instead of
// adapting (k_hmc_adapt_gathered does that on every rank's slab, in rank order)
__global__ void k_hmc_adapt(float* __restrict__ part_acc, float* __restrict__ part_dist, int nblk,
                            const int* n_valid, long B, float* eps_ptr, float* ceps_ptr, float target_p_accept,
                            int tune, float* p_accept_out, float* dist_out, const float* __restrict__ row_acc,
                            const float* __restrict__ row_dist, float* __restrict__ slab) {
    if (row_acc) {
        for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
            float s = 0.f, d = 0.f;
            for (int r = 0; r < ROWS; ++r) { s += row_acc[(long)b * ROWS + r]; d += row_dist[(long)b * ROWS + r]; }
            part_acc[b] = s; part_dist[b] = d;
        }
        __syncthreads();
    }
    const long nv = n_valid ? (long)*n_valid : B;
    if (slab) {
        for (int b = threadIdx.x; b < nblk; b += blockDim.x) { slab[b] = part_acc[b]; slab[nblk + b] = part_dist[b]; }
        if (threadIdx.x == 0) slab[2 * nblk] = (float)(nv > 0 ? nv : 0);      // (exact: chain counts are far below 2^24)
        return;
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (nv <= 0) return;
    float s = 0.f, d = 0.f;
    for (int i = 0; i < nblk; ++i) { s += part_acc[i]; d += part_dist[i]; }
    hmc_adapt_rule(s, d, nv, eps_ptr, ceps_ptr, target_p_accept, tune, p_accept_out, dist_out);
}

// the slabs of all ranks, in rank order: the same additions, in the same order, as one device holding every chain
__global__ void k_hmc_adapt_gathered(const float* __restrict__ slabs, int n_ranks, int nblk, float* eps_ptr,
                                     float* ceps_ptr, float target_p_accept, int tune, float* p_accept_out,
                                     float* dist_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int stride = 2 * nblk + 1;
    float s = 0.f, d = 0.f;
    long nv = 0;
    for (int r = 0; r < n_ranks; ++r) {
        const float* sl = slabs + (long)r * stride;
        for (int i = 0; i < nblk; ++i) { s += sl[i]; d += sl[nblk + i]; }
        nv += (long)sl[2 * nblk];
    }
    if (nv <= 0) return;
    hmc_adapt_rule(s, d, nv, eps_ptr, ceps_ptr, target_p_accept, tune, p_accept_out, dist_out);
}

__device__ __forceinline__ void hmc_adapt_rule(float s, float d, long nv, float* eps_ptr, float* ceps_ptr,
                                               float target_p_accept, int tune, float* p_accept_out, float* dist_out) {
    const float log_mean = logf(s) - logf((float)nv);
    if (p_accept_out) *p_accept_out = expf(log_mean);
    if (dist_out) *dist_out = d / (float)nv;
    if (tune) {
        if (log_mean > logf(target_p_accept)) {
            *eps_ptr = *eps_ptr * 1.05f;
            *ceps_ptr = *ceps_ptr * 1.02f;
        } else {
            *eps_ptr = *eps_ptr / 1.05f;
            *ceps_ptr = *ceps_ptr / 1.02f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Metropolis transition: all n_updates for 16 chains per workgroup (metropolis.py:51-74).
// ------------------------------------------------------------------------------------------------
struct MetK {
    PointDev cur;
    long B;
    const int* n_valid;
    fabhip_anneal c, nx;
    float* log_w;
    const float* noise_x;            // [n_updates][B][D]
    const float* noise_u;            // [n_updates][B]
    const float* scalings;           // noise_scalings[i-1][0..n_updates)
    int n_updates;
    float* part_acc;                 // [n_updates][nblk]
    int nblk;
};

template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_metropolis(FlowDims f, FlowLds l, ExtraLds x, const float* __restrict__ packed,
                                                         TargetDev tg, MetK a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid t;
    const int D = f.D;
    const long nv = a.n_valid ? (long)*a.n_valid : a.B;
    const long row0 = (long)blockIdx.x * ROWS;
    if (row0 >= nv) {
        if (t.tid < a.n_updates) a.part_acc[t.tid * a.nblk + blockIdx.x] = 0.f;
        return;
    }
    float* XC = lds + x.o_XP;        // current positions
    float* XN = lds + x.o_P;         // proposals
    float* GP = lds + x.o_GP;
    float* ROWB = lds + x.o_ROW;
    const long g = row0 + t.row;
    const bool active = g < nv;
    zero_dp(l, lds, t);
    for (int j = t.c; j < D; j += 16) XC[t.row * D + j] = active ? a.cur.x[g * D + j] : 0.f;
    float lq_c = 0.f, lp_c = 0.f;
    if (active) { lq_c = a.cur.lq[g]; lp_c = a.cur.lp[g]; }
    const float prev_lp = a.c.c_q * lq_c + a.c.c_p * lp_c;     // computed once, never refreshed (metropolis.py:53)
    bool changed = false;
    for (int n = 0; n < a.n_updates; ++n) {
        const float sc = a.scalings[n];
        for (int j = t.c; j < D; j += 16) {
            const float nz = active ? a.noise_x[((long)n * a.B + g) * D + j] : 0.f;
            XN[t.row * D + j] = XC[t.row * D + j] + nz * sc;
        }
        __syncthreads();
        load_state_to_u0(l, D, lds, XN, t);
        __syncthreads();
        int goff;
        const float lq = flow_log_prob_tile<NTWM, false>(f, l, packed, lds, t, &goff);
        const float lp = target_tile<false>(tg, XN, D, GP, D, t);
        float acc = expf((a.c.c_q * lq + a.c.c_p * lp) - prev_lp);
        if (!isfinite(acc)) acc = 0.f;                          // nan_to_num(nan=0, posinf=0, neginf=0)
        bool accept = false;
        if (active) accept = acc > a.noise_u[(long)n * a.B + g];
        if (accept) {
            for (int j = t.c; j < D; j += 16) XC[t.row * D + j] = XN[t.row * D + j];
            lq_c = lq; lp_c = lp; changed = true;
        }
        if (t.c == 0) ROWB[t.row] = active ? fminf(acc, 1.f) : 0.f;
        __syncthreads();
        if (t.tid == 0) {
            float s = 0.f;
            for (int r = 0; r < ROWS; ++r) s += ROWB[r];
            a.part_acc[n * a.nblk + blockIdx.x] = s;
        }
        __syncthreads();
    }
    if (active) {
        if (changed) {
            for (int j = t.c; j < D; j += 16) a.cur.x[g * D + j] = XC[t.row * D + j];
            if (t.c == 0) { a.cur.lq[g] = lq_c; a.cur.lp[g] = lp_c; }
        }
        if (a.log_w && t.c == 0) {
            const float num = a.nx.c_q * lq_c + a.nx.c_p * lp_c;
            const float den = a.c.c_q * lq_c + a.c.c_p * lp_c;
            a.log_w[g] = a.log_w[g] + (num - den);
        }
    }
}

__global__ void k_metropolis_adapt(const float* __restrict__ part_acc, int nblk, int n_updates, const int* n_valid,
                                   long B, float* scalings, float target_p_accept, int tune) {
    const int n = threadIdx.x;
    if (blockIdx.x != 0 || n >= n_updates || !tune) return;
    const long nv = n_valid ? (long)*n_valid : B;
    if (nv <= 0) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += part_acc[n * nblk + i];
    const float p_accept = s / (float)nv;
    scalings[n] = (p_accept > target_p_accept) ? scalings[n] * 1.05f : scalings[n] / 1.05f;
}

// Sharded chains: a Metropolis transition whose block sums went into the caller's slab instead of part_acc
// appends the number of chains in use; the rule then runs on the slabs of ALL ranks.  noise_scalings[i - 1, n] is read by
// update n of transition i only (metropolis.py:57) and adjusted right after it (:68-73): nothing later in the SAME AIS call
// reads the adjusted value, so the adjustments of all M transitions can wait for ONE gather at the end of the call.
__global__ void k_metropolis_count(const int* n_valid, long B, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (float)(n_valid ? (long)*n_valid : B);
}

// thread (j, n) = transition j, update n: block sums over ranks, then blocks, in order - with shards that are multiples of 16
// chains the sequence k_metropolis_adapt adds on one device holding every chain
__global__ void k_metropolis_adapt_gathered(const float* __restrict__ slabs, int n_ranks, int nblk, int M, int n_updates,
                                            float* scalings, float target_p_accept, int tune) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * n_updates || !tune) return;
    const int j = t / n_updates, n = t % n_updates;
    const long per_j = (long)n_updates * nblk + 1, per_rank = (long)M * per_j;
    float s = 0.f;
    long nv = 0;
    for (int r = 0; r < n_ranks; ++r) {
        const float* sl = slabs + r * per_rank + j * per_j;
        for (int i = 0; i < nblk; ++i) s += sl[n * nblk + i];
        nv += (long)sl[per_j - 1];
    }
    if (nv <= 0) return;
    const float p_accept = s / (float)nv;
    scalings[t] = (p_accept > target_p_accept) ? scalings[t] * 1.05f : scalings[t] / 1.05f;
}

// ------------------------------------------------------------------------------------------------
// _remove_nan_and_infs (ais.py:190-213): stable compaction of the rows with finite log_p and log_q.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_valid_scan(const float* __restrict__ lq, const float* __restrict__ lp,
                                                     const int* n_in_ptr, long B, int* __restrict__ dest,
                                                     int* __restrict__ n_out) {
    __shared__ int wsum[16];
    __shared__ int running;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long n_in = n_in_ptr ? (long)*n_in_ptr : B;
    if (tid == 0) running = 0;
    __syncthreads();
    for (long base = 0; base < n_in; base += 1024) {
        const long r = base + tid;
        const bool v = r < n_in && isfinite(lq[r]) && isfinite(lp[r]);
        const unsigned long long bal = __ballot(v);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int i = 0; i < 16; ++i) { if (i < w) woff += wsum[i]; tot += wsum[i]; }
        if (r < n_in) dest[r] = v ? running + woff + pre : -1;
        __syncthreads();
        if (tid == 0) running += tot;
        __syncthreads();
    }
    if (tid == 0) *n_out = running;
}

struct CompactK {
    PointDev pt;
    float* log_w;
    float* extra;          // optional per-row scalar carried along (nullptr: none)
    float* tmp;            // [B][3D+4]
    const int* dest;
    const int* n_in_ptr;
    const int* n_out_ptr;
    long B;
    int D;
};

__global__ void k_compact_scatter(CompactK a) {
    const long n_in = a.n_in_ptr ? (long)*a.n_in_ptr : a.B;
    const long n_out = *a.n_out_ptr;
    if (n_out == n_in) return;
    const int D = a.D, RW = 3 * D + 4;
    const bool hg = a.pt.gq != nullptr;
    for (long r = blockIdx.x; r < n_in; r += gridDim.x) {
        const int d = a.dest[r];
        if (d < 0) continue;
        float* o = a.tmp + (long)d * RW;
        for (int j = threadIdx.x; j < D; j += blockDim.x) {
            o[j] = a.pt.x[r * D + j];
            if (hg) { o[D + j] = a.pt.gq[r * D + j]; o[2 * D + j] = a.pt.gp[r * D + j]; }
        }
        if (threadIdx.x == 0) {
            o[3 * D] = a.pt.lq[r]; o[3 * D + 1] = a.pt.lp[r]; o[3 * D + 2] = a.log_w[r];
            if (a.extra) o[3 * D + 3] = a.extra[r];
        }
    }
}

__global__ void k_compact_copyback(CompactK a) {
    const long n_in = a.n_in_ptr ? (long)*a.n_in_ptr : a.B;
    const long n_out = *a.n_out_ptr;
    if (n_out == n_in) return;
    const int D = a.D, RW = 3 * D + 4;
    const bool hg = a.pt.gq != nullptr;
    for (long r = blockIdx.x; r < n_out; r += gridDim.x) {
        const float* o = a.tmp + r * RW;
        for (int j = threadIdx.x; j < D; j += blockDim.x) {
            a.pt.x[r * D + j] = o[j];
            if (hg) { a.pt.gq[r * D + j] = o[D + j]; a.pt.gp[r * D + j] = o[2 * D + j]; }
        }
        if (threadIdx.x == 0) {
            a.pt.lq[r] = o[3 * D]; a.pt.lp[r] = o[3 * D + 1]; a.log_w[r] = o[3 * D + 2];
            if (a.extra) a.extra[r] = o[3 * D + 3];
        }
    }
}

__global__ void k_sub(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i] - b[i];
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int nblk_of(long B) { return (int)((B + ROWS - 1) / ROWS); }

template <int NTWM>
static int launch_create_point(const FlowDims& f, const float* packed, const TargetDev& tg, const PointDev& pt,
                               int with_grad, long B, hipStream_t st) {
    const FlowLds l = make_flow_lds(f, with_grad != 0);
    const ExtraLds x = make_extra_lds(l, f.D);
    const size_t bytes = (size_t)x.total * 4;
    const dim3 grid(nblk_of(B)), block(NTHREADS);
    if (with_grad) {
        if (f.fast) {
            FAB_TRY(set_max_lds((const void*)k_create_point_fast<NTWM, true>, bytes));
            hipLaunchKernelGGL((k_create_point_fast<NTWM, true>), grid, block, bytes, st, f, l, x, packed, tg, pt, B);
            return check_launch();
        }
        FAB_TRY(set_max_lds((const void*)k_create_point<NTWM, true>, bytes));
        hipLaunchKernelGGL((k_create_point<NTWM, true>), grid, block, bytes, st, f, l, x, packed, tg, pt, B);
    } else {
        FAB_TRY(set_max_lds((const void*)k_create_point<NTWM, false>, bytes));
        hipLaunchKernelGGL((k_create_point<NTWM, false>), grid, block, bytes, st, f, l, x, packed, tg, pt, B);
    }
    return check_launch();
}

template <int NTWM>
static int launch_ais_init(const FlowDims& f, const float* packed, const TargetDev& tg, const float* eps0,
                           const PointDev& pt, float* log_w, float* base_log_w, fabhip_anneal an, int with_grad, long B,
                           hipStream_t st) {
    const FlowLds l = make_flow_lds(f, with_grad != 0);
    const ExtraLds x = make_extra_lds(l, f.D);
    const size_t bytes = (size_t)x.total * 4;
    const dim3 grid(nblk_of(B)), block(NTHREADS);
    if (with_grad) {
        if (f.fast) {
            FAB_TRY(set_max_lds((const void*)k_ais_init_fast<NTWM>, bytes));
            hipLaunchKernelGGL((k_ais_init_fast<NTWM>), grid, block, bytes, st, f, l, x, packed, tg, eps0, pt, log_w, base_log_w, an, B);
            return check_launch();
        }
        FAB_TRY(set_max_lds((const void*)k_ais_init<NTWM, true>, bytes));
        hipLaunchKernelGGL((k_ais_init<NTWM, true>), grid, block, bytes, st, f, l, x, packed, tg, eps0, pt, log_w, base_log_w, an, B);
    } else {
        FAB_TRY(set_max_lds((const void*)k_ais_init<NTWM, false>, bytes));
        hipLaunchKernelGGL((k_ais_init<NTWM, false>), grid, block, bytes, st, f, l, x, packed, tg, eps0, pt, log_w, base_log_w, an, B);
    }
    return check_launch();
}

template <int NTWM>
static int launch_hmc_step(const FlowDims& f, const float* packed, const TargetDev& tg, const HmcK& a, hipStream_t st) {
    const FlowLds l = make_flow_lds(f, true);
    const ExtraLds x = make_extra_lds(l, f.D);
    const size_t bytes = (size_t)x.total * 4;
    if (f.fast) {
        FAB_TRY(set_max_lds((const void*)k_hmc_step_fast<NTWM>, bytes));
        hipLaunchKernelGGL((k_hmc_step_fast<NTWM>), dim3(nblk_of(a.B)), dim3(NTHREADS), bytes, st, f, l, x, packed, tg, a);
        return check_launch();
    }
    FAB_TRY(set_max_lds((const void*)k_hmc_step<NTWM>, bytes));
    hipLaunchKernelGGL((k_hmc_step<NTWM>), dim3(nblk_of(a.B)), dim3(NTHREADS), bytes, st, f, l, x, packed, tg, a);
    return check_launch();
}

// 4-chain tiles pay when 16-chain tiles cannot fill the chip: up to 288 workgroups of 4 chains (B <= 1152); off in fast
// mode (no bf16 variant of the 4-chain kernel).  FABHIP_OPT_TILE_SHAPE = 16 / 4 forces the choice (tests exercise both).
// Shapes: D <= 32 and hidden width <= 320 only - the D > 32 and the 512-wide instantiations of the 4-chain kernel spill
// registers (hipcc: 52 .. 772 VGPRs), so they are neither compiled nor selectable (16-chain tiles there).
// (use_r4_tiles: launch.h - the flow sample follows the same choice)
// 8-chain tiles (flow_r8.h: D <= 32, hidden width padded to 256 / 320, fp32): half the weight bytes per chain of the 4-chain
// kernel on one stream per wave.  FABHIP_OPT_TILE_SHAPE = 8 forces them; by default they take the batches for which
// R8_MIN_CHAINS < B <= 8 chains per CU (measured: DESIGN.md section 4).
constexpr long R8_MIN_CHAINS = 1152;
static bool r8_lds_fits(const FlowDims& f) { return (size_t)(make_r8_lds(f).total + 4 * R8 * f.D + 4) * 4 <= 160 * 1024; }
static bool use_r8_tiles(const FlowDims& f, long B) {
    if (f.fast || !r8_shape_ok(f) || !r8_lds_fits(f)) return false;
    const int shape = option(FABHIP_OPT_TILE_SHAPE);
    if (shape == 16 || shape == 4) return false;
    if (shape == 8) return true;
    return B > R8_MIN_CHAINS && B <= (long)R8 * cu_count();
}

long long* debug_timeline(hipStream_t st);           // flow_kernels.hip (dev-only stage stamps)

template <int NTWM>
static int launch_hmc_step_r4(const FlowDims& f0, const float* packed, const TargetDev& tg, const HmcK& a, hipStream_t st) {
    FlowDims f = f0;
    f.timeline = debug_timeline(st);
    const R4Dims rd = make_r4_dims(f);
    const bool fused = use_r4_fused(f);
    const R4Lds l = make_r4_lds(f, fused);
    const ExtraLds4 x = make_extra_lds4(l, f.D);
    const size_t bytes = (size_t)x.total * 4;
    // every row of the ceil(B / 16) sixteen-row blocks k_hmc_adapt sums is written by some workgroup (workgroups past
    // the last chain take the early-return branch and write zeros): the scratch is never read uninitialised
    const dim3 grid((unsigned)(nblk_of(a.B) * (ROWS / R4)));
    const bool stream = option(FABHIP_OPT_R4_STREAM) != 0;  // 0: per-stage request groups also where the stream image exists
    if constexpr (NTWM > 5) {
        return FABHIP_ENOTSUP;                              // (use_r4_tiles never selects it)
    } else if (NTWM >= 2 && fused && f.fast) {
        constexpr int NS = NTWM >= 2 ? NTWM : 2;
        FAB_TRY(set_max_lds((const void*)k_hmc_step_r4<NS, false, 3>, bytes));
        hipLaunchKernelGGL((k_hmc_step_r4<NS, false, 3>), grid, dim3(NTHREADS), bytes, st, f, rd, l, x, packed, tg, a);
    } else if (NTWM >= 4 && fused && option(FABHIP_OPT_R4_STREAM) >= 3 && bytes + (size_t)R4F_NS * NTWM * NWAVE * 1024 <= 160 * 1024) {
        constexpr int NS = NTWM >= 4 ? NTWM : 4;           // (the stash exists where a layer has no empty ring items: 4 / 5 tiles per wave)
        const size_t bytes_s = bytes + (size_t)R4F_NS * NS * NWAVE * 1024;
        FAB_TRY(set_max_lds((const void*)k_hmc_step_r4<NS, false, 4>, bytes_s));
        hipLaunchKernelGGL((k_hmc_step_r4<NS, false, 4>), grid, dim3(NTHREADS), bytes_s, st, f, rd, l, x, packed, tg, a);
    } else if (NTWM >= 2 && fused) {
        constexpr int NS = NTWM >= 2 ? NTWM : 2;
        FAB_TRY(set_max_lds((const void*)k_hmc_step_r4<NS, false, 2>, bytes));
        hipLaunchKernelGGL((k_hmc_step_r4<NS, false, 2>), grid, dim3(NTHREADS), bytes, st, f, rd, l, x, packed, tg, a);
    } else if (NTWM >= 2 && f.o_r4s >= 0 && (stream || NTWM >= 5)) {    // (the per-stage schedule spills at 5 tiles per wave)
        constexpr int NS = NTWM >= 2 ? NTWM : 2;           // (never instantiates the stream code for NTWM = 1)
        FAB_TRY(set_max_lds((const void*)k_hmc_step_r4<NS, false, 1>, bytes));
        hipLaunchKernelGGL((k_hmc_step_r4<NS, false, 1>), grid, dim3(NTHREADS), bytes, st, f, rd, l, x, packed, tg, a);
    } else if constexpr (NTWM < 5) {
        FAB_TRY(set_max_lds((const void*)k_hmc_step_r4<NTWM, false, 0>, bytes));
        hipLaunchKernelGGL((k_hmc_step_r4<NTWM, false, 0>), grid, dim3(NTHREADS), bytes, st, f, rd, l, x, packed, tg, a);
    } else {
        return FABHIP_ENOTSUP;
    }
    return check_launch();
}

template <int NTWM>
static int launch_ais_init_r4(const FlowDims& f, const float* packed, const TargetDev& tg, const float* lq0, const float* eps0,
                              const PointDev& pt, float* log_w, float* base_log_w, fabhip_anneal an, long B, hipStream_t st) {
    const R4Dims rd = make_r4_dims(f);
    const bool fused = use_r4_fused(f);
    const R4Lds l = make_r4_lds(f, fused);
    const ExtraLds4 x = make_extra_lds4(l, f.D);
    const size_t bytes = (size_t)x.total * 4;
    const dim3 grid((unsigned)((B + R4 - 1) / R4));
    if constexpr (NTWM > 5) {
        return FABHIP_ENOTSUP;
    } else if (NTWM >= 2 && fused && f.fast) {
        constexpr int NS = NTWM >= 2 ? NTWM : 2;
        FAB_TRY(set_max_lds((const void*)k_ais_init_r4<NS, 3>, bytes));
        hipLaunchKernelGGL((k_ais_init_r4<NS, 3>), grid, dim3(NTHREADS), bytes, st, f, rd, l, x, packed, tg, lq0, eps0, pt,
                           log_w, base_log_w, an, B);
    } else if (NTWM >= 2 && fused) {
        constexpr int NS = NTWM >= 2 ? NTWM : 2;
        FAB_TRY(set_max_lds((const void*)k_ais_init_r4<NS, 2>, bytes));
        hipLaunchKernelGGL((k_ais_init_r4<NS, 2>), grid, dim3(NTHREADS), bytes, st, f, rd, l, x, packed, tg, lq0, eps0, pt,
                           log_w, base_log_w, an, B);
    } else if (NTWM >= 2 && f.o_r4s >= 0) {
        constexpr int NS = NTWM >= 2 ? NTWM : 2;
        FAB_TRY(set_max_lds((const void*)k_ais_init_r4<NS, 1>, bytes));
        hipLaunchKernelGGL((k_ais_init_r4<NS, 1>), grid, dim3(NTHREADS), bytes, st, f, rd, l, x, packed, tg, lq0, eps0, pt,
                           log_w, base_log_w, an, B);
    } else if constexpr (NTWM < 5) {
        FAB_TRY(set_max_lds((const void*)k_ais_init_r4<NTWM, 0>, bytes));
        if (eps0) return FABHIP_ENOTSUP;                    // (the sampling direction exists on the stream image only)
        hipLaunchKernelGGL((k_ais_init_r4<NTWM, 0>), grid, dim3(NTHREADS), bytes, st, f, rd, l, x, packed, tg, lq0,
                           (const float*)nullptr, pt, log_w, base_log_w, an, B);
    } else {
        return FABHIP_ENOTSUP;
    }
    return check_launch();
}

static int launch_hmc_step_r8(const FlowDims& f0, const float* packed, const TargetDev& tg, const HmcK& a, hipStream_t st) {
    FlowDims f = f0;
    f.timeline = debug_timeline(st);
    const R8Lds l = make_r8_lds(f);
    const ExtraLds8 x = make_extra_lds8(l, f.D);
    const size_t bytes = (size_t)x.total * 4;
    const dim3 grid((unsigned)(nblk_of(a.B) * (ROWS / R8)));      // every row of the 16-row blocks k_hmc_adapt sums gets written
    const bool fused = use_r8_fused(f);
#define FAB_R8_STEP(G, FU)                                                                                        \
    do {                                                                                                          \
        FAB_TRY(set_max_lds((const void*)k_hmc_step_r8<G, FU>, bytes));                                           \
        hipLaunchKernelGGL((k_hmc_step_r8<G, FU>), grid, dim3(NTHREADS), bytes, st, f, l, x, packed, tg, a);      \
    } while (0)
    if (f.Wp == 320) { if (fused) FAB_R8_STEP(5, true); else FAB_R8_STEP(5, false); }
    else if (f.Wp == 256) { if (fused) FAB_R8_STEP(4, true); else FAB_R8_STEP(4, false); }
    else return FABHIP_ENOTSUP;
#undef FAB_R8_STEP
    return check_launch();
}

static int launch_ais_init_r8(const FlowDims& f, const float* packed, const TargetDev& tg, const float* lq0, const PointDev& pt,
                              float* log_w, float* base_log_w, fabhip_anneal an, long B, hipStream_t st) {
    const R8Lds l = make_r8_lds(f);
    const ExtraLds8 x = make_extra_lds8(l, f.D);
    const size_t bytes = (size_t)x.total * 4;
    const dim3 grid((unsigned)((B + R8 - 1) / R8));
    const bool fused = use_r8_fused(f);
#define FAB_R8_INIT(G, FU)                                                                                        \
    do {                                                                                                          \
        FAB_TRY(set_max_lds((const void*)k_ais_init_r8<G, FU>, bytes));                                           \
        hipLaunchKernelGGL((k_ais_init_r8<G, FU>), grid, dim3(NTHREADS), bytes, st, f, l, x, packed, tg, lq0, pt, log_w, \
                           base_log_w, an, B);                                                                    \
    } while (0)
    if (f.Wp == 320) { if (fused) FAB_R8_INIT(5, true); else FAB_R8_INIT(5, false); }
    else if (f.Wp == 256) { if (fused) FAB_R8_INIT(4, true); else FAB_R8_INIT(4, false); }
    else return FABHIP_ENOTSUP;
#undef FAB_R8_INIT
    return check_launch();
}

template <int NTWM>
static int launch_metropolis(const FlowDims& f, const float* packed, const TargetDev& tg, const MetK& a, hipStream_t st) {
    const FlowLds l = make_flow_lds(f, false);
    const ExtraLds x = make_extra_lds(l, f.D);
    const size_t bytes = (size_t)x.total * 4;
    FAB_TRY(set_max_lds((const void*)k_metropolis<NTWM>, bytes));
    hipLaunchKernelGGL((k_metropolis<NTWM>), dim3(a.nblk), dim3(NTHREADS), bytes, st, f, l, x, packed, tg, a);
    return check_launch();
}

static int hmc_transition_impl(const fabhip_hmc_args* a, hipStream_t st, int* ticket = nullptr) {
    const FlowDims f = flow_dims_of(a->flow);
    const TargetDev tg = make_target_dev(a->target);
    const int D = f.D;
    const int nblk = nblk_of(a->B);
    if (a->workspace_bytes < fabhip_hmc_workspace_bytes(a->B, D, a->n_outer)) return FABHIP_ENOSPC;
    if (a->partials && a->n_outer != 1) return FABHIP_ENOTSUP;      // (the next outer loop needs the adapted common_epsilon)
    char* ws = (char*)a->workspace;
    float* part_acc = (float*)ws; ws += align256((size_t)nblk * 4);
    float* part_dist = (float*)ws; ws += align256((size_t)nblk * 4);
    float* row_acc = (float*)ws; ws += align256((size_t)nblk * ROWS * 4);
    float* row_dist = (float*)ws; ws += align256((size_t)nblk * ROWS * 4);
    const bool r8 = use_r8_tiles(f, a->B);
    const bool r4 = !r8 && use_r4_tiles(f, a->B);
    PointDev prop{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (a->n_outer > 1) {
        float* pb = (float*)ws;
        prop.x = pb; prop.gq = pb + a->B * D; prop.gp = pb + 2 * a->B * D;
        prop.lq = pb + 3 * a->B * D; prop.lp = prop.lq + a->B;
    }
    const PointDev cur = make_point_dev(a->point);
    for (int n = 0; n < a->n_outer; ++n) {
        HmcK k;
        k.cur = cur;
        k.start = (n == 0) ? cur : prop;
        k.prop_out = (n + 1 < a->n_outer) ? prop : PointDev{nullptr, nullptr, nullptr, nullptr, nullptr};
        k.B = a->B; k.n_valid = a->n_valid; k.c = a->cur; k.nx = a->next;
        k.log_w = (n + 1 == a->n_outer) ? a->log_w : nullptr;
        k.noise_p = a->noise_p + (size_t)n * a->B * D;
        k.noise_e = a->noise_e + (size_t)n * a->B;
        k.eps_ptr = a->epsilons + n; k.ceps_ptr = a->common_epsilon; k.mass = a->mass;
        k.L = a->L; k.max_grad = a->max_grad; k.part_acc = part_acc; k.part_dist = part_dist;
        k.row_acc = row_acc; k.row_dist = row_dist;
        // the step-size rule in the transition kernel's last workgroup (4- / 8-chain tiles; `ticket` is zero: fabhip_ais_phase)
        const bool fold = ticket && (r4 || r8) && !a->partials;
        k.ticket = fold ? ticket : nullptr;
        k.eps_w = a->epsilons + n; k.ceps_w = a->common_epsilon; k.target_p_accept = a->target_p_accept; k.tune = a->tune;
        k.p_accept_out = a->p_accept ? a->p_accept + n : nullptr; k.dist_out = a->avg_distance; k.nblk = nblk;
        if (r8) FAB_TRY(launch_hmc_step_r8(f, a->flow.packed, tg, k, st));
        else if (r4) FAB_DISPATCH_NTW_NORET(f, launch_hmc_step_r4, f, a->flow.packed, tg, k, st);
        else FAB_DISPATCH_NTW_NORET(f, launch_hmc_step, f, a->flow.packed, tg, k, st);
        if (fold) { FAB_TRY(check_launch()); continue; }
        hipLaunchKernelGGL(k_hmc_adapt, dim3(1), dim3(64), 0, st, part_acc, part_dist, nblk, a->n_valid, (long)a->B,
                           a->epsilons + n, a->common_epsilon, a->target_p_accept, a->tune,
                           a->p_accept ? a->p_accept + n : nullptr, a->avg_distance,
                           (r4 || r8) ? row_acc : (const float*)nullptr, (r4 || r8) ? row_dist : (const float*)nullptr, a->partials);
        FAB_TRY(check_launch());
    }
    return FABHIP_OK;
}

static int metropolis_transition_impl(const fabhip_metropolis_args* a, hipStream_t st, float* partials = nullptr) {
    const FlowDims f = flow_dims_of(a->flow);
    const TargetDev tg = make_target_dev(a->target);
    const int nblk = nblk_of(a->B);
    if (a->workspace_bytes < fabhip_metropolis_workspace_bytes(a->B, f.D, a->n_updates)) return FABHIP_ENOSPC;
    MetK k;
    k.cur = make_point_dev(a->point); k.B = a->B; k.n_valid = a->n_valid; k.c = a->cur; k.nx = a->next;
    k.log_w = a->log_w; k.noise_x = a->noise_x; k.noise_u = a->noise_u; k.scalings = a->noise_scalings;
    k.n_updates = a->n_updates; k.part_acc = partials ? partials : (float*)a->workspace; k.nblk = nblk;
    FAB_DISPATCH_NTW_NORET(f, launch_metropolis, f, a->flow.packed, tg, k, st);
    if (partials) {                  // deferred rule (sharded chains): [n_updates][nblk] block sums | chains in use
        hipLaunchKernelGGL(k_metropolis_count, dim3(1), dim3(1), 0, st, a->n_valid, (long)a->B,
                           partials + (size_t)a->n_updates * nblk);
        return check_launch();
    }
    hipLaunchKernelGGL(k_metropolis_adapt, dim3(1), dim3(64), 0, st, k.part_acc, nblk, a->n_updates, a->n_valid,
                       (long)a->B, a->noise_scalings, a->target_p_accept, a->tune);
    return check_launch();
}

static int check_point(const fabhip_point& p, bool need_grad) {
    if (!p.x || !p.log_q || !p.log_p) return FABHIP_EINVAL;
    if (need_grad && (!p.grad_log_q || !p.grad_log_p)) return FABHIP_EINVAL;
    return FABHIP_OK;
}

}  // namespace fab

using namespace fab;

extern "C" {

void fabhip_anneal_coefs(double beta, double alpha, int32_t p_target, fabhip_anneal* out) {
    if (!out) return;
    if (!p_target) {   // base.py:93-94, 114-116
        out->c_q = (float)((1.0 - beta) + beta * (1.0 - alpha));
        out->c_p = (float)(beta * alpha);
        out->g_q = out->c_q;
        out->g_p = (float)(2.0 * beta);
    } else {           // base.py:96-97, 117-118
        out->c_q = (float)(1.0 - beta);
        out->c_p = (float)beta;
        out->g_q = out->c_q;
        out->g_p = out->c_p;
    }
}

int fabhip_create_point(const fabhip_flow* flow, const fabhip_target* target, const fabhip_point* point,
                        int32_t with_grad, int64_t B, fabhip_stream_t stream) {
    if (!flow || !flow->packed || !point || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(flow->dim, flow->n_layers, flow->width));
    FAB_TRY(check_target(target, flow->dim));
    FAB_TRY(check_point(*point, with_grad != 0));
    if (B == 0) return FABHIP_OK;
    const FlowDims f = flow_dims_of(*flow);
    FAB_DISPATCH_NTW(f, launch_create_point, f, flow->packed, make_target_dev(*target), make_point_dev(*point),
                     with_grad, (long)B, (hipStream_t)stream);
}

size_t fabhip_hmc_workspace_bytes(int64_t B, int32_t dim, int32_t n_outer) {
    const size_t nblk = (size_t)nblk_of(B);
    size_t s = 2 * align256(nblk * 4) + 2 * align256(nblk * ROWS * 4);      // block partials + per-chain values (4-chain tiles)
    if (n_outer > 1) s += align256((size_t)B * (3 * dim + 2) * 4);
    return s + 256;
}

int64_t fabhip_hmc_partials_floats(int64_t B) { return B < 0 ? -1 : 2 * (int64_t)nblk_of(B) + 1; }

int fabhip_hmc_adapt_gathered(const float* gathered, int32_t n_ranks, int64_t B_rank, float* epsilon, float* common_epsilon,
                              float target_p_accept, int32_t tune, float* p_accept, float* avg_distance,
                              fabhip_stream_t stream) {
    if (!gathered || n_ranks < 1 || B_rank < 1 || !epsilon || !common_epsilon) return FABHIP_EINVAL;
    hipLaunchKernelGGL(k_hmc_adapt_gathered, dim3(1), dim3(64), 0, (hipStream_t)stream, gathered, n_ranks, nblk_of(B_rank),
                       epsilon, common_epsilon, target_p_accept, tune, p_accept, avg_distance);
    return check_launch();
}

int64_t fabhip_metropolis_partials_floats(int64_t B, int32_t M, int32_t n_updates) {
    return (B < 0 || M < 1 || n_updates < 1) ? -1 : (int64_t)M * ((int64_t)n_updates * nblk_of(B) + 1);
}

int fabhip_metropolis_adapt_gathered(const float* gathered, int32_t n_ranks, int64_t B_rank, int32_t M, int32_t n_updates,
                                     float* noise_scalings, float target_p_accept, int32_t tune, fabhip_stream_t stream) {
    if (!gathered || n_ranks < 1 || B_rank < 1 || M < 1 || n_updates < 1 || !noise_scalings) return FABHIP_EINVAL;
    const int n = M * n_updates;
    hipLaunchKernelGGL(k_metropolis_adapt_gathered, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, gathered, n_ranks,
                       nblk_of(B_rank), M, n_updates, noise_scalings, target_p_accept, tune);
    return check_launch();
}

int fabhip_hmc_transition(const fabhip_hmc_args* a, fabhip_stream_t stream) {
    if (!a || !a->flow.packed || !a->noise_p || !a->noise_e || !a->epsilons || !a->common_epsilon || !a->mass ||
        !a->workspace || a->B < 0 || a->n_outer < 1 || a->L < 0)
        return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(a->flow.dim, a->flow.n_layers, a->flow.width));
    FAB_TRY(check_target(&a->target, a->flow.dim));
    FAB_TRY(check_point(a->point, true));
    if (a->B == 0) return FABHIP_OK;
    return hmc_transition_impl(a, (hipStream_t)stream);
}

size_t fabhip_metropolis_workspace_bytes(int64_t B, int32_t dim, int32_t n_updates) {
    (void)dim;
    return align256((size_t)nblk_of(B) * (size_t)(n_updates > 0 ? n_updates : 1) * 4) + 256;
}

int fabhip_metropolis_transition(const fabhip_metropolis_args* a, fabhip_stream_t stream) {
    if (!a || !a->flow.packed || !a->noise_x || !a->noise_u || !a->noise_scalings || !a->workspace || a->B < 0 ||
        a->n_updates < 1 || a->n_updates > 64)
        return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(a->flow.dim, a->flow.n_layers, a->flow.width));
    FAB_TRY(check_target(&a->target, a->flow.dim));
    FAB_TRY(check_point(a->point, false));
    if (a->B == 0) return FABHIP_OK;
    return metropolis_transition_impl(a, (hipStream_t)stream);
}

// ---- whole AIS call -------------------------------------------------------------------------------
size_t fabhip_ais_workspace_bytes(int64_t B, int32_t dim, int32_t n_inner) {
    size_t s = fabhip_hmc_workspace_bytes(B, dim, n_inner);
    const size_t m = fabhip_metropolis_workspace_bytes(B, dim, n_inner);
    if (m > s) s = m;
    s = align256(s);
    s += align256((size_t)B * (3 * dim + 4) * 4);   // compaction staging
    s += align256((size_t)B * 4);                    // dest ranks
    s += align256((size_t)B * 4);                    // log_p - log_q
    s += align256(fabhip_ess_workspace_bytes(B));
    s += 256;                                         // ticket of the in-kernel step-size rule
    return s + 256;
}

// the tail of a chain phase: compaction (+ log_p - log_q) + ESS / log Z - one launch for small batches, else the separate kernels
static int phase_tail(const fabhip_point& point, float* log_w, long B, int dim, const int* n_in, int* n_out, float* tmp, int* dest,
                      float* extra, float* diff, double n_norm, float* stats_out, void* ess_ws, size_t ess_bytes, float* base_x,
                      int* zero_word, float* zero_f, hipStream_t st);

static int compact_rows(const fabhip_point& point, float* log_w, long B, int dim, const int* n_in, int* n_out, float* tmp,
                        int* dest, float* extra, hipStream_t st) {
    hipLaunchKernelGGL(k_valid_scan, dim3(1), dim3(1024), 0, st, point.log_q, point.log_p, n_in, B, dest, n_out);
    CompactK c{make_point_dev(point), log_w, extra, tmp, dest, n_in, n_out, B, dim};
    const int grid = (int)(B < 4096 ? B : 4096);
    hipLaunchKernelGGL(k_compact_scatter, dim3(grid), dim3(64), 0, st, c);
    hipLaunchKernelGGL(k_compact_copyback, dim3(grid), dim3(64), 0, st, c);
    return check_launch();
}

static int phase_tail(const fabhip_point& point, float* log_w, long B, int dim, const int* n_in, int* n_out, float* tmp, int* dest,
                      float* extra, float* diff, double n_norm, float* stats_out, void* ess_ws, size_t ess_bytes, float* base_x,
                      int* zero_word, float* zero_f, hipStream_t st) {
    int rc = FABHIP_ENOTSUP;
    if (option(FABHIP_OPT_FUSED_TAIL) != 0) {
        TailArgs t;
        t.x = point.x; t.lq = point.log_q; t.lp = point.log_p; t.gq = point.grad_log_q; t.gp = point.grad_log_p;
        t.log_w = log_w; t.extra = extra; t.n_in = n_in; t.n_out = n_out; t.B = B; t.D = dim; t.diff = diff; t.n_norm = n_norm;
        t.stats_out = stats_out; t.zero_word = zero_word; t.zero_f = zero_f; t.n_zero_f = zero_f ? 10 : 0;
        rc = tail_small(t, dest, st);
        if (rc != FABHIP_OK && rc != FABHIP_ENOTSUP) return rc;
    }
    if (rc == FABHIP_ENOTSUP) {
        if (zero_word && hipMemsetAsync(zero_word, 0, 4, st) != hipSuccess) return FABHIP_ELAUNCH;
        if (zero_f && hipMemsetAsync(zero_f, 0, 10 * 4, st) != hipSuccess) return FABHIP_ELAUNCH;
        FAB_TRY(compact_rows(point, log_w, B, dim, n_in, n_out, tmp, dest, extra, st));
    }
    if (base_x &&         // the compacted starting points (rows beyond the count are don't-care)
        hipMemcpyAsync(base_x, point.x, (size_t)B * dim * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return FABHIP_ELAUNCH;
    if (rc == FABHIP_ENOTSUP) {
        if (diff) hipLaunchKernelGGL(k_sub, dim3(ceil_div((int)B, 256)), dim3(256), 0, st, point.log_p, point.log_q, diff, B);
        FAB_TRY(fabhip_ess_logz(diff ? diff : log_w, B, n_out, n_norm, stats_out, ess_ws, ess_bytes, (fabhip_stream_t)st));
    }
    return check_launch();
}

// ---- spline flow as base distribution (csrc/spline_kernels.hip) ------------------------------------------------------
__global__ void k_spline_init_logw(const float* __restrict__ lq, const float* __restrict__ lp, const float* __restrict__ lq0,
                                   fabhip_anneal an, float* __restrict__ log_w, float* __restrict__ base_log_w, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        log_w[i] = (an.c_q * lq[i] + an.c_p * lp[i]) - lq0[i];              // ais.py:62-64
        if (base_log_w) base_log_w[i] = lp[i] - lq0[i];                     // ais.py:160
    }
}

size_t fabhip_spline_hmc_workspace_bytes(int32_t dim, int32_t n_layers, int32_t hidden, int64_t B) {
    return align256(fabhip_generic_workspace_bytes(B, dim)) +
           align256(fabhip_spline_workspace_bytes(dim, n_layers, hidden, B, 1)) + align256((size_t)B * (3 * dim + 2) * 4) +
           align256(spline_fold_scratch_floats(B) * 4) + 256;
}

// `ticket_zeroed`: the caller has already zeroed the fold's ticket word in this workspace (fabhip_spline_ais_run: once per call;
// the wave that draws the last ticket of a launch resets it)
static int spline_hmc_transition_impl(const fabhip_spline_hmc_args* a, fabhip_stream_t stream, bool ticket_zeroed);
static int* spline_fold_ticket(const fabhip_spline_hmc_args* a) {
    char* ws = (char*)a->workspace;
    ws += align256(fabhip_generic_workspace_bytes(a->B, a->flow.dim));
    ws += align256(fabhip_spline_workspace_bytes(a->flow.dim, a->flow.n_layers, a->flow.hidden, a->B, 1));
    ws += align256((size_t)a->B * (3 * a->flow.dim + 2) * 4);
    return (int*)((float*)ws + 32 * ((a->B + 15) / 16));
}

int fabhip_spline_hmc_transition(const fabhip_spline_hmc_args* a, fabhip_stream_t stream) {
    return spline_hmc_transition_impl(a, stream, false);
}

static int spline_hmc_transition_impl(const fabhip_spline_hmc_args* a, fabhip_stream_t stream, bool ticket_zeroed) {
    if (!a || !a->flow.packed || !a->noise_p || !a->noise_e || !a->epsilons || !a->common_epsilon || !a->mass ||
        !a->workspace || a->B < 0 || a->n_outer < 1 || a->L < 1)
        return FABHIP_EINVAL;
    FAB_TRY(check_target(&a->target, a->flow.dim));
    FAB_TRY(check_point(a->point, true));
    const int D = a->flow.dim;
    const long B = a->B;
    if (a->workspace_bytes < fabhip_spline_hmc_workspace_bytes(D, a->flow.n_layers, a->flow.hidden, B)) return FABHIP_ENOSPC;
    if (B == 0) return FABHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)a->workspace;
    const size_t gb = fabhip_generic_workspace_bytes(B, D);
    void* gws = ws; ws += align256(gb);
    const size_t sb = fabhip_spline_workspace_bytes(D, a->flow.n_layers, a->flow.hidden, B, 1);
    void* sws = ws; ws += align256(sb);
    float* pb = (float*)ws; ws += align256((size_t)B * (3 * D + 2) * 4);
    float* fold_ws = (float*)ws;                                   // row_acc, row_dist [16 nblk], ticket
    fabhip_point prop{pb, pb + 3 * B * D, pb + 3 * B * D + B, pb + B * D, pb + 2 * B * D};
    const fabhip_point cur = a->point;
    fabhip_point start = cur;
    // round 5: begin / accept / step-size rule inside the first / last leapfrog launch (L launches per outer step instead of
    // L + 3) where the one-launch leapfrog applies - the same arithmetic in the same order, bit for bit (spline_r8.h)
    const bool fold = spline_leap_fold_supported(&a->flow, B);
    const int nblk = (int)((B + 15) / 16);
    if (fold && !ticket_zeroed && hipMemsetAsync(spline_fold_ticket(a), 0, 4, st) != hipSuccess) return FABHIP_ELAUNCH;
    for (int n = 0; n < a->n_outer; ++n) {
        const bool last = n + 1 == a->n_outer;
        if (!fold)
            FAB_TRY(gen_hmc_begin(&start, &cur, B, D, a->cur, a->noise_p + (size_t)n * B * D, a->mass, a->max_grad, gws, a->n_valid, st));
        for (int l = 0; l < a->L; ++l) {
            {   // one launch per leapfrog where the 4x4x1 spline kernel applies (launch.h: SplineLeap)
                float *XPw, *Pw, *GUw;
                gen_hmc_state(gws, B, D, &XPw, &Pw, &GUw);
                SplineLeap lp;
                lp.fold = SplineFold{};
                if (fold) {
                    SplineFold& fd = lp.fold;
                    fd.flags = (l == 0 ? 1 : 0) | (l + 1 == a->L ? 2 : 0);
                    fd.start_x = start.x; fd.start_gq = start.grad_log_q; fd.start_gp = start.grad_log_p;
                    fd.noise_p = a->noise_p + (size_t)n * B * D;
                    fd.logp_cur = GUw + (size_t)B * D;              // the generic workspace's per-chain row (generic_kernels.hip: split_ws)
                    fd.cur_x = cur.x; fd.cur_lq = cur.log_q; fd.cur_lp = cur.log_p; fd.cur_gq = cur.grad_log_q; fd.cur_gp = cur.grad_log_p;
                    fd.noise_e = a->noise_e + (size_t)n * B; fd.nx = a->next; fd.log_w = last ? a->log_w : nullptr;
                    fd.n_valid = a->n_valid; fd.row_acc = fold_ws; fd.row_dist = fold_ws + 16 * (size_t)nblk;
                    fd.ticket = (int*)(fold_ws + 32 * (size_t)nblk);
                    fd.eps_w = a->epsilons + n; fd.ceps_w = a->common_epsilon; fd.target_p_accept = a->target_p_accept; fd.tune = a->tune;
                    fd.p_accept_out = a->p_accept ? a->p_accept + n : nullptr; fd.dist_out = a->avg_distance; fd.nblk = nblk;
                }
                lp.XP = XPw; lp.x_out = prop.x; lp.P = Pw; lp.GU = GUw; lp.eps_ptr = a->epsilons + n; lp.ceps_ptr = a->common_epsilon; lp.mass = a->mass;
                lp.c = a->cur; lp.max_grad = a->max_grad; lp.tg = a->target; lp.prop_lp = prop.log_p; lp.prop_gp = prop.grad_log_p;
                const int rc = spline_log_prob_leap(&a->flow, lp, prop.log_q, prop.grad_log_q, B, sws, sb, st);
                if (rc == FABHIP_OK) continue;
                if (rc != FABHIP_ENOTSUP || fold) return rc == FABHIP_ENOTSUP ? FABHIP_EINVAL : rc;
            }
            FAB_TRY(fabhip_hmc_generic_leap_pre(B, D, a->epsilons + n, a->common_epsilon, a->mass, prop.x, gws, gb, stream));
            FAB_TRY(fabhip_spline_log_prob(&a->flow, prop.x, prop.log_q, prop.grad_log_q, B, sws, sb, stream));
            FAB_TRY(fabhip_target_log_prob(&a->target, prop.x, prop.log_p, prop.grad_log_p, B, stream));
            FAB_TRY(fabhip_hmc_generic_leap_post(B, D, prop.grad_log_q, prop.grad_log_p, a->cur, a->max_grad, a->epsilons + n,
                                                 a->common_epsilon, gws, gb, stream));
        }
        if (!fold)
            FAB_TRY(gen_hmc_accept(&prop, &cur, B, D, a->cur, a->next, last ? a->log_w : nullptr, a->noise_e + (size_t)n * B, a->mass,
                                   a->epsilons + n, a->common_epsilon, a->target_p_accept, a->tune,
                                   a->p_accept ? a->p_accept + n : nullptr, a->avg_distance, gws, a->n_valid, st));
        start = prop;                                  // the reference continues from the PROPOSAL (hmc.py:133-142)
    }
    return FABHIP_OK;
}

size_t fabhip_spline_ais_workspace_bytes(int32_t dim, int32_t n_layers, int32_t hidden, int64_t B) {
    size_t s = align256(fabhip_spline_hmc_workspace_bytes(dim, n_layers, hidden, B));
    s += align256((size_t)B * (3 * dim + 4) * 4);    // compaction staging
    s += 3 * align256((size_t)B * 4);                // dest ranks, log_p - log_q, log q of the sampling pass
    s += align256(fabhip_ess_workspace_bytes(B));
    return s + 256;
}

int fabhip_spline_ais_run(const fabhip_spline_ais_args* a, fabhip_stream_t stream) {
    if (!a || !a->flow.packed || !a->betas || !a->u0 || !a->eps0 || !a->noise_p || !a->noise_e || !a->epsilons ||
        !a->common_epsilon || !a->mass || !a->log_w || !a->n_valid || !a->stats || !a->workspace || a->B < 1 || a->M < 1 ||
        a->n_outer < 1 || a->L < 1)
        return FABHIP_EINVAL;
    FAB_TRY(check_target(&a->target, a->flow.dim));
    FAB_TRY(check_point(a->point, true));
    const int D = a->flow.dim;
    const long B = a->B;
    if (a->workspace_bytes < fabhip_spline_ais_workspace_bytes(D, a->flow.n_layers, a->flow.hidden, B)) return FABHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)a->workspace;
    const size_t tws = align256(fabhip_spline_hmc_workspace_bytes(D, a->flow.n_layers, a->flow.hidden, B));
    void* trans_ws = ws; ws += tws;
    float* tmp = (float*)ws; ws += align256((size_t)B * (3 * D + 4) * 4);
    int* dest = (int*)ws; ws += align256((size_t)B * 4);
    float* lwb = (float*)ws; ws += align256((size_t)B * 4);
    float* lq0 = (float*)ws; ws += align256((size_t)B * 4);
    void* ess_ws = ws;
    const size_t ess_bytes = fabhip_ess_workspace_bytes(B);
    // the spline kernels' own scratch: the transition workspace is free until the first transition
    const size_t sb = fabhip_spline_workspace_bytes(D, a->flow.n_layers, a->flow.hidden, B, 1);
    // 1. chain initialisation: x, log q0 = flow.sample ; point = create_point(x) ; log_w = pi_beta1(point) - log q0
    FAB_TRY(fabhip_spline_sample(&a->flow, a->u0, a->eps0, a->point.x, lq0, B, trans_ws, sb, stream));
    FAB_TRY(fabhip_spline_log_prob(&a->flow, a->point.x, a->point.log_q, a->point.grad_log_q, B, trans_ws, sb, stream));
    FAB_TRY(fabhip_target_log_prob(&a->target, a->point.x, a->point.log_p, a->point.grad_log_p, B, stream));
    fabhip_anneal a1;
    fabhip_anneal_coefs(a->betas[1], a->alpha, a->p_target, &a1);
    hipLaunchKernelGGL(k_spline_init_logw, dim3(ceil_div((int)B, 256)), dim3(256), 0, st, a->point.log_q, a->point.log_p, lq0,
                       a1, a->log_w, a->base_log_w, B);
    // 2. "chain init" filter, 3. base ESS
    FAB_TRY(phase_tail(a->point, a->log_w, B, D, nullptr, a->n_valid, tmp, dest, a->base_log_w, lwb, 1.0, a->stats + 0, ess_ws,
                       ess_bytes, a->base_x, nullptr, a->stats + 6, st));
    // 4. transitions
    for (int j = 1; j <= a->M; ++j) {
        fabhip_spline_hmc_args h;
        h.flow = a->flow; h.target = a->target; h.point = a->point; h.B = B; h.n_valid = a->n_valid;
        fabhip_anneal_coefs(a->betas[j], a->alpha, a->p_target, &h.cur);
        fabhip_anneal_coefs(a->betas[j + 1], a->alpha, a->p_target, &h.next);
        h.log_w = (a->betas[j + 1] != a->betas[j]) ? a->log_w : nullptr;      // ais.py:93
        const size_t nslab = (size_t)(j - 1) * a->n_outer;
        h.noise_p = a->noise_p + nslab * B * D; h.noise_e = a->noise_e + nslab * B;
        h.epsilons = a->epsilons + nslab; h.common_epsilon = a->common_epsilon; h.mass = a->mass;
        h.n_outer = a->n_outer; h.L = a->L; h.max_grad = a->max_grad; h.target_p_accept = a->target_p_accept; h.tune = a->tune;
        h.p_accept = nullptr; h.avg_distance = nullptr;
        if (j == 1) { h.p_accept = a->p_accept_first; h.avg_distance = a->avg_distance_first; }
        else if (j == a->M) { h.p_accept = a->p_accept_last; h.avg_distance = a->avg_distance_last; }
        h.workspace = trans_ws; h.workspace_bytes = tws;
        if (j == 1 && spline_leap_fold_supported(&a->flow, B) && hipMemsetAsync(spline_fold_ticket(&h), 0, 4, st) != hipSuccess)
            return FABHIP_ELAUNCH;
        FAB_TRY(spline_hmc_transition_impl(&h, stream, true));
    }
    // 5. "chain end" filter, 6. ESS / log Z
    FAB_TRY(phase_tail(a->point, a->log_w, B, D, a->n_valid, a->n_valid + 1, tmp, dest, nullptr, nullptr, (double)B, a->stats + 3,
                       ess_ws, ess_bytes, nullptr, nullptr, nullptr, st));
    return check_launch();
}

int fabhip_ais_run(const fabhip_ais_args* a, fabhip_stream_t stream) {
    if (!a) return FABHIP_EINVAL;
    return fabhip_ais_phase(a, FABHIP_AIS_INIT | FABHIP_AIS_FINISH, 1, a->M, nullptr, stream);
}

int fabhip_ais_phase(const fabhip_ais_args* a, int32_t phases, int32_t j_begin, int32_t j_end, float* partials,
                     fabhip_stream_t stream) {
    if (!a || !a->flow.packed || !a->betas || !a->step_state || !a->log_w ||
        !a->n_valid || !a->stats || !a->workspace || a->B < 1 || a->M < 1 || a->n_inner < 1)
        return FABHIP_EINVAL;
    if (j_begin <= j_end && (!a->noise_a || !a->noise_b)) return FABHIP_EINVAL;       // (read by the transitions only)
    const bool do_init = (phases & FABHIP_AIS_INIT) != 0, do_finish = (phases & FABHIP_AIS_FINISH) != 0;
    const bool ws_kept = do_init || (phases & FABHIP_AIS_CONTINUE) != 0;              // the ticket word of this workspace is zero
    if (do_init && !a->eps0) return FABHIP_EINVAL;
    if (j_begin <= j_end && (j_begin < 1 || j_end > a->M)) return FABHIP_EINVAL;
    if (partials && a->transition == FABHIP_TRANSITION_HMC && j_begin != j_end) return FABHIP_ENOTSUP;
    const bool hmc = a->transition == FABHIP_TRANSITION_HMC;
    if (!hmc && a->transition != FABHIP_TRANSITION_METROPOLIS) return FABHIP_EINVAL;
    if (hmc && (!a->common_epsilon || !a->mass)) return FABHIP_EINVAL;
    FAB_TRY(check_flow_shape(a->flow.dim, a->flow.n_layers, a->flow.width));
    FAB_TRY(check_target(&a->target, a->flow.dim));
    FAB_TRY(check_point(a->point, hmc));
    if (a->workspace_bytes < fabhip_ais_workspace_bytes(a->B, a->flow.dim, a->n_inner)) return FABHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    const FlowDims f = flow_dims_of(a->flow);
    const TargetDev tg = make_target_dev(a->target);
    const int D = f.D;
    const long B = a->B;

    char* ws = (char*)a->workspace;
    size_t tws = fabhip_hmc_workspace_bytes(B, D, a->n_inner);
    const size_t mws = fabhip_metropolis_workspace_bytes(B, D, a->n_inner);
    if (mws > tws) tws = mws;
    tws = align256(tws);
    void* trans_ws = ws; ws += tws;
    float* tmp = (float*)ws; ws += align256((size_t)B * (3 * D + 4) * 4);
    int* dest = (int*)ws; ws += align256((size_t)B * 4);
    float* lwb = (float*)ws; ws += align256((size_t)B * 4);
    void* ess_ws = ws; ws += align256(fabhip_ess_workspace_bytes(B));
    const size_t ess_bytes = fabhip_ess_workspace_bytes(B);
    // step-size rule inside the transition kernels (hmc_adapt_last): only where the ticket word is known to be zero (this call's or,
    // with FABHIP_AIS_CONTINUE, an earlier call's init phase zeroed it; every transition kernel leaves it zero)
    int* ticket = (hmc && !partials && option(FABHIP_OPT_ADAPT_FOLD) != 0 && nblk_of(B) <= 2048 &&
                   (use_r8_tiles(f, B) || use_r4_tiles(f, B))) ? (int*)ws : nullptr;      // (2 nblk floats of LDS scratch)
    // (round 6: a call that runs transitions on a workspace nobody zeroed - the second piece of a call whose INIT piece was its own
    //  fabhip_ais_phase call, fab_torch_amd/ais.py: repeated calls - zeroes the word itself: one 4-byte fill instead of a
    //  k_hmc_adapt launch per transition)
    if (ticket && !ws_kept) {
        if (j_begin > j_end) ticket = nullptr;
        else if (hipMemsetAsync(ticket, 0, 4, st) != hipSuccess) return FABHIP_ELAUNCH;
    }

    if (do_init) {
    // 1. chain initialisation
    fabhip_anneal a1;
    fabhip_anneal_coefs(a->betas[1], a->alpha, a->p_target, &a1);
    {
        const PointDev pt = make_point_dev(a->point);
        if (hmc && use_r8_tiles(f, B)) {
            FAB_TRY(fabhip_flow_sample(&a->flow, a->eps0, a->point.x, lwb, B, stream));
            FAB_TRY(launch_ais_init_r8(f, a->flow.packed, tg, lwb, pt, a->log_w, a->base_log_w, a1, B, st));
        } else if (hmc && use_r4_tiles(f, B) && option(FABHIP_OPT_R4_STREAM) != 0) {
            // 4-chain tiles: x, log q0 = flow.sample(eps0), then log q + d/dx (the reference re-evaluates: base.py:65-68), target,
            // log w - one kernel where the stream image exists (f.o_r4s), else the flow-sample kernel first
            if (f.o_r4s >= 0 && f.NTW / 4 >= 2) {
                FAB_DISPATCH_NTW_NORET(f, launch_ais_init_r4, f, a->flow.packed, tg, (const float*)nullptr, a->eps0, pt, a->log_w,
                                       a->base_log_w, a1, B, st);
            } else {
                FAB_TRY(fabhip_flow_sample(&a->flow, a->eps0, a->point.x, lwb, B, stream));
                FAB_DISPATCH_NTW_NORET(f, launch_ais_init_r4, f, a->flow.packed, tg, lwb, (const float*)nullptr, pt, a->log_w,
                                       a->base_log_w, a1, B, st);
            }
        } else {
            FAB_DISPATCH_NTW_NORET(f, launch_ais_init, f, a->flow.packed, tg, a->eps0, pt, a->log_w, a->base_log_w, a1,
                                   hmc ? 1 : 0, B, st);
        }
    }
    // 2. remove nan/inf ("chain init"), 3. ESS over the base samples (ais.py:68-71) -> stats[0..2]
    FAB_TRY(phase_tail(a->point, a->log_w, B, D, nullptr, a->n_valid, tmp, dest, a->base_log_w, lwb, 1.0, a->stats + 0, ess_ws,
                       ess_bytes, a->base_x, ticket, a->stats + 6, st));
    }
    // 4. transitions
    for (int j = j_begin; j <= j_end; ++j) {
        fabhip_anneal cj, cn;
        fabhip_anneal_coefs(a->betas[j], a->alpha, a->p_target, &cj);
        fabhip_anneal_coefs(a->betas[j + 1], a->alpha, a->p_target, &cn);
        float* lw = (a->betas[j + 1] != a->betas[j]) ? a->log_w : nullptr;      // ais.py:93
        const size_t nslab = (size_t)(j - 1) * a->n_inner;
        if (hmc) {
            fabhip_hmc_args h;
            h.flow = a->flow; h.target = a->target; h.point = a->point; h.B = B; h.n_valid = a->n_valid;
            h.cur = cj; h.next = cn; h.log_w = lw;
            h.noise_p = a->noise_a + nslab * B * D; h.noise_e = a->noise_b + nslab * B;
            h.epsilons = a->step_state + nslab; h.common_epsilon = a->common_epsilon; h.mass = a->mass;
            h.n_outer = a->n_inner; h.L = a->L; h.max_grad = a->max_grad; h.target_p_accept = a->target_p_accept;
            h.tune = a->tune;
            h.p_accept = nullptr; h.avg_distance = nullptr;
            h.workspace = trans_ws; h.workspace_bytes = tws;
            h.partials = partials;
            // logging slots of the first / last distribution (hmc.py:173-183), one acceptance per outer loop
            if (j == 1) { h.p_accept = a->p_accept_first; h.avg_distance = a->avg_distance_first; }
            else if (j == a->M) { h.p_accept = a->p_accept_last; h.avg_distance = a->avg_distance_last; }
            FAB_TRY(hmc_transition_impl(&h, st, ticket));
        } else {
            fabhip_metropolis_args m;
            m.flow = a->flow; m.target = a->target; m.point = a->point; m.B = B; m.n_valid = a->n_valid;
            m.cur = cj; m.next = cn; m.log_w = lw;
            m.noise_x = a->noise_a + nslab * B * D; m.noise_u = a->noise_b + nslab * B;
            m.noise_scalings = a->step_state + nslab; m.n_updates = a->n_inner;
            m.target_p_accept = a->target_p_accept; m.tune = a->tune;
            m.workspace = trans_ws; m.workspace_bytes = tws;
            // deferred rule: transition j's block sums + count at slab offset (j - 1) (n_updates nblk + 1)
            FAB_TRY(metropolis_transition_impl(&m, st, partials ? partials + (size_t)(j - 1) * ((size_t)a->n_inner * nblk_of(B) + 1)
                                                                 : nullptr));
        }
    }
    // 5. remove nan/inf ("chain end"), 6. ESS / log Z over the survivors (ais.py:77-86)
    if (do_finish)
        FAB_TRY(phase_tail(a->point, a->log_w, B, D, a->n_valid, a->n_valid + 1, tmp, dest, nullptr, nullptr, (double)B,
                           a->stats + 3, ess_ws, ess_bytes, nullptr, nullptr, nullptr, st));
    return check_launch();
}

}  // extern "C"
