// 4x4x1 stream kernels (S8) for the one-launch spline density: hidden width padded to 256 (the shape of BASELINE cfg 3 and of
// the alanine-dipeptide flow), 4 RB chains per workgroup on v_mfma_f32_4x4x1_16b_f32 - RB = 1 / 2 row blocks (4 / 8 chains)
// while that leaves at most one workgroup per CU, RB = 4 (16 chains) for larger batches; same arithmetic (bit-identical).
//
// Why: k_spline_logprob (16 chains per workgroup, 16x16x4 MFMAs) gives B / 16 workgroups - 128 of 256 CUs at cfg 3's 2048
// chains - and its element-wise stages (splines, reverse mode, tile reloads: a third of its time) run on those 128 CUs only.
// With 8 chains per workgroup 2048 chains fill the chip (0.43 ms per density + gradient instead of 0.74).
//
// GEMM shape: OUT[4 RB][256] = ACT[4 RB][K] @ B[K][256], N-split: wave w owns columns 64 w .. 64 w + 63 for all of K.  One
// v_mfma_f32_4x4x1_16b = 16 independent 4x4 outer products: lane l = 4 b + j supplies A[i = l % 4] and B[64 w + l] and holds
// D[i = VGPR r][64 w + l]; the row blocks (chains 0-3, 4-7, ..) share every weight register.  The instruction has a ~54-cycle
// dependent latency (and issues every ~11 cycles, not 8: measured), so k mod 4 goes to four accumulators per row block
// (>= 8 independent chains), added as (a0 + a1) + (a2 + a3) at the end: no partial sums through LDS, ONE workgroup barrier per
// GEMM stage.  Measured: 98 cycles per k-quad (8 MFMAs + one 1-KiB tile per wave) at RB = 2 = 0.9 of the instruction's
// issue rate, 47 B/clk per CU of weight stream; 186 at RB = 4.
//
// Weight stream: every wave reads ITS tiles (1 KiB = 4 k x 64 columns) of a layer and direction as one contiguous stream in
// the order it consumes them ([W0 | Wa | Wb | Wf chunks] forward, [WfT | WbT | WaT | W0T] reverse; k_spline_pack_r8),
// through a ring of S8_RD = 32 tiles that stays full ACROSS the stages of a layer (requested at the layer top, topped up one
// tile per k-quad, drained by the layer's last stage): the stream does not stop at stage boundaries.  The loads are issued
// from inline asm with hand-counted s_waitcnt (flow_device.h) into ACCUMULATION registers ("=a"): hipcc copies / re-allocates
// VGPRs it believes idle - an in-flight VGPR ring that outlives its loop got corrupted that way in round 1 - but the AGPR
// file has no other tenant here, and v_mfma reads its B operand from it directly.  tools/check_r8_isa.py walks the ISA of
// every instantiation and fails if any instruction touches an AGPR whose load may still be in flight.
// Everything else a layer needs (metadata rows, unconditional spline parameters, periodic-feature weights, biases) is one
// contiguous head block per layer, copied to LDS at the layer top with plain loads BEFORE the ring is requested, so no
// compiler-tracked load ever waits behind the stream.  ReLU decisions: one 64-bit ballot per (wave, chain), kept in LDS.
// Included from spline_kernels.hip (namespace fab, after the rqs_* helpers); the stream / GEMM-step machinery is stream_r8.h.

constexpr int S8_AS = 64 + 4;          // leading dim of the identity-feature tile (K = 64)
constexpr int S8_WS = 256 + 4;         // leading dim of the hidden tiles

// head block of a layer in the r8 image (floats); the first three regions sit where SplineDims puts them in a layer image
constexpr int S8H_META = 0, S8H_UNC = (SP_META_ROWS + 2) * 64, S8H_PFW = S8H_UNC + 1664, S8H_B0 = S8H_PFW + 128,
              S8H_BA = S8H_B0 + 256, S8H_BB = S8H_BA + 256, S8H_NXT = S8H_BB + 256, S8H_BF = S8H_NXT + 128;

static_assert(S8H_BF == (SP_META_ROWS + 2) * 64 + 1664 + 128 + 3 * 256 + 128, "make_spline_dims: r8_head");
FAB_HD int s8_head_floats(const SplineDims& f) { return f.r8_head; }                      // S8H_BF + NFP: a multiple of 128
FAB_HD int s8_tiles_per_wave(const SplineDims& f) { return f.r8_tpl; }                    // per layer and direction: 16 + 64 (2 + NCH)
FAB_HD long s8_layer_floats(const SplineDims& f) { return f.r8_layer; }

struct S8Lds {
    int PS;                            // leading dim of the conditioner-output tile
    int o_A0, o_X1, o_X2, o_T, o_PT, o_ZT, o_GT, o_HD, o_MASK, total;      // PART (K-split partials of W0T, [4][rows][AS]) lies over X1 | X2
};

FAB_HD S8Lds make_s8_lds(const SplineDims& f, bool grad, int rows) {
    S8Lds l;
    l.PS = f.NFP + 4;
    int o = 0;
    l.o_A0 = o; o += rows * S8_AS;
    l.o_X1 = o; o += rows * S8_WS;
    l.o_X2 = o; o += rows * S8_WS;
    l.o_T = o; o += rows * S8_WS;
    l.o_PT = o; o += rows * l.PS;
    l.o_ZT = o; o += rows * 64;
    l.o_GT = o; if (grad) o += rows * 64;
    l.o_HD = o; o += s8_head_floats(f);
    l.o_MASK = o; if (grad) o += f.L * NWAVE * 64;                    // ReLU decisions: one word per layer and thread
    l.total = (o + 3) & ~3;
    return l;
}

struct Tid8 {
    int tid, wave, lane, arow, row, c;
    __device__ __forceinline__ Tid8() {
        tid = threadIdx.x;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        lane = tid & 63;
        arow = lane & 3;               // the chain (of a row block) whose activations this lane feeds to the MFMA
        row = tid >> 5;                // element-wise stages: 8 chains x 32 coordinate lanes
        c = tid & 31;
    }
};

__device__ __forceinline__ float row32_sum(float v) {
    {   // v + v(lane ^ 16): v_permlane16_swap of the register with itself gives [r0 r0 r2 r2] and [r1 r1 r3 r3] (16-lane rows)
        const unsigned u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}

// OUT[4 RB][64 w ..] = ACT[4 RB][256] @ B: two iterations of 32 k-quads; LAST = REMAIN of the second one.
// Straight-line on purpose: around a loop hipcc carries the ring as loop variables and COPIES slots at the back edge
// (v_accvgpr_mov of a register whose load is still in flight; seen in the ISA of a first version) - every iteration of
// every stage of a layer is therefore unrolled (NCH is a template parameter of the kernel).
template <int PHASE, int RB, int LAST = S8_INF>
__device__ __forceinline__ void s8_gemm64(S8Stream& s, const float* act, int lda, const Tid8& t, f32x4 (&o)[RB]) {
    S8Acc<RB> acc;
    s8_zero(acc);
    const float* ap = act + t.arow * lda;
    s8_iter<32, 32, PHASE, S8_INF>(s, ap, 4 * lda, acc);
    s8_iter<32, 32, PHASE, LAST>(s, ap + 128, 4 * lda, acc);
    s8_fold(acc, o);
}

// ---- image -----------------------------------------------------------------------------------------------------------
// Layer block of the r8 image: [head | forward tiles: wave 0 .. 3 | reverse tiles: wave 0 .. 3], a wave's tiles in stream order.
__global__ __launch_bounds__(256) void k_spline_pack_r8(SplineDims f, SplineSrc s, const float* __restrict__ prev_meta,
                                                        int layer, float* __restrict__ packed) {
    const int H = s8_head_floats(f), TPL = s8_tiles_per_wave(f);
    const long total = s8_layer_floats(f);
    float* __restrict__ dst = packed + f.o_r8 + (size_t)layer * total;
    const int n_id = (int)s.meta[M_CNT * 64 + 0], n_tr = (int)s.meta[M_CNT * 64 + 1], n_pf = (int)s.meta[M_CNT * 64 + 2];
    const int W = f.W, nout = n_tr * SP_NP;
    for (long off = (long)blockIdx.x * blockDim.x + threadIdx.x; off < total; off += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (off < H) {
            const int e = (int)off;
            if (e < SP_META_ROWS * 64) v = s.meta[e];
            else if (e < S8H_UNC) {                                            // M_POSID / M_POSTR (k_spline_pack_layer)
                const int q = e - SP_META_ROWS * 64, row = q >> 6, j = q & 63;
                const int cnt = row == 0 ? n_id : n_tr;
                const float* feats = s.meta + (row == 0 ? M_IDF : M_TRF) * 64;
                v = -1.f;
                for (int i = 0; i < cnt; ++i) if ((int)feats[i] == j) v = (float)i;
            } else if (e < S8H_PFW) {
                const int q = e - S8H_UNC, i = q / SP_NP, p = q % SP_NP;
                if (i < n_id && q < SP_MD * SP_NP)
                    v = p < SP_K ? s.uw[i * SP_K + p] : (p < 2 * SP_K ? s.uh[i * SP_K + p - SP_K] : s.ud[i * (SP_K + 1) + p - 2 * SP_K]);
            } else if (e < S8H_B0) { const int q = e - S8H_PFW; if (q < 2 * n_pf) v = s.pfw[q]; }
            else if (e < S8H_BA) { const int j = e - S8H_B0; if (j < W) v = s.b0[j]; }
            else if (e < S8H_BB) { const int j = e - S8H_BA; if (j < W) v = s.ba[j]; }
            else if (e < S8H_NXT) { const int j = e - S8H_BB; if (j < W) v = s.bb[j]; }
            else if (e < S8H_BF) {                                             // pre-shift of the NEXT stage (layer - 1): shift | on
                const int q = e - S8H_NXT;
                if (prev_meta) v = prev_meta[(q < 64 ? M_PRESH : M_PREON) * 64 + (q & 63)];
            } else { const int j = e - S8H_BF; if (j < nout) v = s.bf[j]; }
        } else {
            const long e = off - H;
            const int kk = (int)(e & 3), lane = (int)((e >> 2) & 63);
            const long tl = e >> 8;
            const int tile = (int)(tl % TPL), wave = (int)((tl / TPL) % NWAVE), dir = (int)(tl / ((long)TPL * NWAVE));
            const int n = 64 * wave + lane;
            // round 5 (f.r8_trim): the stream without zero tiles - W0 as K0Q = 4 k-quads, WfT as KFQ = 100, W0T as 4 dense tiles
            const int K0Q = f.r8_trim ? 4 : 16, KFQ = f.r8_trim ? 100 : 64 * f.NCH;
            if (dir == 0) {
                if (tile < K0Q) {                                              // W0: B[k][n] = w0[n][k]
                    const int k = 4 * tile + kk;
                    if (k < n_id && n < W) v = s.w0[n * n_id + k];
                } else if (tile < K0Q + 64 * (2 + f.NCH)) {
                    const int t2 = tile - K0Q, m = t2 >> 6, k = 4 * (t2 & 63) + kk;
                    if (m == 0) { if (k < W && n < W) v = s.wa[n * W + k]; }
                    else if (m == 1) { if (k < W && n < W) v = s.wb[n * W + k]; }
                    else { const int col = (m - 2) * 256 + n; if (k < W && col < nout) v = s.wf[col * W + k]; }
                }
            } else {
                if (tile < KFQ) {                                              // WfT: B[k][n] = wf[k][n]
                    const int k = 4 * tile + kk;
                    if (k < nout && n < W) v = s.wf[k * W + n];
                } else {
                    const int t2 = tile - KFQ;
                    if (t2 < 64) { const int k = 4 * t2 + kk; if (k < W && n < W) v = s.wb[k * W + n]; }
                    else if (t2 < 128) { const int k = 4 * (t2 - 64) + kk; if (k < W && n < W) v = s.wa[k * W + n]; }
                    else if (!f.r8_trim) {                                     // W0T, K split over the waves: B[k][i] = w0[k][i]
                        if (t2 < 144) {
                            const int k = 64 * wave + 4 * (t2 - 128) + kk;
                            if (k < W && lane < n_id) v = s.w0[k * n_id + lane];
                        }
                    } else if (t2 < 132) {                                     // ... as dense tiles: 4 k-quads x 16 identity features
                        const int k = 64 * wave + 4 * (4 * (t2 - 128) + (lane >> 4)) + kk, c = lane & 15;
                        if (k < W && c < n_id) v = s.w0[k * n_id + c];
                    }
                }
            }
        }
        dst[off] = v;
    }
}

// optional leapfrog around the density evaluation (launch.h: SplineLeap; XP == nullptr: a plain density call)
// Round 5: the REST of an HMC outer step (hmc.py:129-160) inside the leapfrog launches - `flags` & 1: this launch is the first
// leapfrog and does k_gen_hmc_begin's work at its top (p0 = noise x mass, grad U of the start point, -U(current) - K(p0));
// & 2: it is the last one and does k_gen_hmc_accept's (accept / reject, commit, AIS log-weight increment) and k_gen_hmc_adapt's
// (step-size rule, by the last wave of the launch to finish: the mechanism of ais_kernels.hip's hmc_adapt_last) at its end:
// a transition is L launches instead of L + 3.  Every sum is formed in the order of the kernels it replaces (16 lanes per chain:
// lane c adds coordinates c, c + 16, ..; xor butterfly 8, 4, 2, 1; 16 chains per block in row order; blocks in order): bit-identical.
using SplineFoldDev = SplineFold;     // launch.h
struct SplineLeapDev {
    float *XP, *P, *GU, *x_out;
    const float *eps_ptr, *ceps_ptr, *mass;
    fabhip_anneal c;
    float max_grad;
    TargetDev tg;
    float *prop_lp, *prop_gp;
    SplineFoldDev fold;
};
__device__ __forceinline__ float g_clamp_nan0_s8(float g, float mg) { return (g != g) ? 0.f : fminf(fmaxf(g, -mg), mg); }   // hmc.py:194-199
__device__ __forceinline__ float s8_row16_sum_xor(float v) {                   // generic_kernels.hip: g_row16_sum
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
// k_gen_hmc_adapt's work by the last wave of the launch to finish.  Cross-workgroup ordering as in ais_kernels.hip
// (hmc_store_row_stats / hmc_adapt_last; _isa_check.py pins the lowering): per-chain values written through (sc1) by
// device-scope relaxed atomic stores, vmcnt(0), ticket; the last ticket's wave reads them past its XCD's L2 (sc1 loads).
// Every wave of every workgroup calls this once, after its stores.  `scratch`: 2 nblk floats of LDS nobody else touches.
__device__ __forceinline__ void s8_fold_adapt_last(const SplineFoldDev& a, long B, float* scratch, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk = __builtin_amdgcn_readfirstlane(tk);
    if (tk != (int)gridDim.x * NWAVE - 1) return;
    const int nblk = a.nblk;
    for (int b = lane; b < nblk; b += 64) {
        float s = 0.f, d = 0.f;
        for (int r = 0; r < 16; ++r) {
            s += __hip_atomic_load(a.row_acc + (long)b * 16 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d += __hip_atomic_load(a.row_dist + (long)b * 16 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        scratch[b] = s; scratch[nblk + b] = d;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // one wave: its LDS writes are done before lane 0 reads them
    if (lane == 0) {
        long nv = a.n_valid ? (long)*a.n_valid : B;
        nv = nv < B ? nv : B;
        if (nv > 0) {
            float s = 0.f, d = 0.f;
            for (int i = 0; i < nblk; ++i) { s += scratch[i]; d += scratch[nblk + i]; }
            const float log_mean = logf(s) - logf((float)nv);                 // hmc.py:122-123,162-170
            if (a.p_accept_out) *a.p_accept_out = expf(log_mean);
            if (a.dist_out) *a.dist_out = d / (float)nv;
            if (a.tune) {
                if (log_mean > logf(a.target_p_accept)) { *a.eps_w = *a.eps_w * 1.05f; *a.ceps_w = *a.ceps_w * 1.02f; }
                else { *a.eps_w = *a.eps_w / 1.05f; *a.ceps_w = *a.ceps_w / 1.02f; }
            }
        }
        __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the kernel ------------------------------------------------------------------------------------------------------
// TRIM (SplineDims::r8_trim: D <= 32, two output chunks): the layer's stream without its zero tiles - see k_spline_pack_r8
template <int NCH, int RB, bool GRAD, bool TRIM = false>
__global__ __launch_bounds__(NTHREADS) void k_spline_logprob_r8(SplineDims f, S8Lds l, const float* __restrict__ packed,
                                                                const float* __restrict__ x, float* __restrict__ log_q,
                                                                float* __restrict__ grad_x, long B, float* __restrict__ Zsave,
                                                                float* __restrict__ Psave, long long* tlp, SplineLeapDev lp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid8 t;
    constexpr int R8 = 4 * RB;                                                 // chains of this workgroup
    constexpr int K0Q = TRIM ? 4 : 16;                                         // k-quads of W0 (identity features / 4)
    constexpr int KFQ = TRIM ? 100 : 64 * NCH;                                 // k-quads of WfT (conditioner outputs / 4)
    constexpr int PHF = K0Q % S8_RD, PHR = KFQ % S8_RD;                        // ring phase behind W0 / behind WfT
    static_assert(!TRIM || NCH == 2, "the trimmed stream exists for two output chunks");
#define S8_TL(idx) do { if (tlp && blockIdx.x == 0 && threadIdx.x == 0) tlp[idx] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    const long row0 = (long)blockIdx.x * R8;
    float* A0 = lds + l.o_A0; float* X1 = lds + l.o_X1; float* X2 = lds + l.o_X2; float* T = lds + l.o_T;
    float* PT = lds + l.o_PT; float* ZT = lds + l.o_ZT; float* GT = lds + l.o_GT; float* HD = lds + l.o_HD;
    float* PART = X1;
    // ReLU decisions of the two hidden stages, kept per THREAD (r6, as flow_r8.h): bit 4 rb + r of the low / high half of the
    // thread's word of a layer = its output row of h0 / t; the reverse sweep's products have the same thread-to-output mapping
    static_assert(4 * RB <= 16, "one halfword of decisions per stage");
    float* MASK = lds + l.o_MASK;
    const float* meta = HD + S8H_META;
    const float isq = 1.f / sqrtf((float)f.W);
    const int H = s8_head_floats(f), TPL = s8_tiles_per_wave(f);
    const long lfl = s8_layer_floats(f);
    const float* img = packed + f.o_r8;
    const size_t zs = (size_t)B * f.D, ps = (size_t)B * f.NFP;
    const int col = 64 * t.wave + t.lane;                                      // this lane's output column of a GEMM stage
    S8Stream s;
    s8_stream_init(s, t.lane);
    // The next layer's head block (and, in the reverse sweep, its state and conditioner-output tiles) is requested with plain
    // loads at the start of a mid-layer GEMM stage - the requests queue up between the ring's tiles and have landed long before
    // the layer ends - and committed to LDS at the next layer top: a layer top then starts on data that is already there.
    constexpr int HP = (S8H_BF + 256 * NCH + 1023) / 1024;                     // float4 per thread of a head block
    constexpr bool TPRE = RB * NCH <= 8;                                       // else the tile prefetch does not fit the VGPRs: loaded at the layer top
    float4 hpre[HP], ppre[TPRE ? RB : 1][TPRE ? NCH : 1];
    float zpre[RB];
    auto head_fetch = [&](const float* Lr) {
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            const int e = t.tid + NTHREADS * i;
            hpre[i] = e < H / 4 ? reinterpret_cast<const float4*>(Lr)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto head_commit = [&]() {
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            const int e = t.tid + NTHREADS * i;
            if (e < H / 4) reinterpret_cast<float4*>(HD)[e] = hpre[i];
        }
    };
    const int w4max = (f.n_tr_max * SP_NP + 3) >> 2;
    auto tile_fetch = [&](int layer) {                                         // reverse sweep: layer input state + conditioner output
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const long g = row0 + t.wave + NWAVE * i;
#pragma unroll
            for (int k = 0; k < (TPRE ? NCH : 0); ++k) {
                const int c4 = t.lane + 64 * k;
                ppre[i][k] = (g < B && c4 < w4max) ? *reinterpret_cast<const float4*>(Psave + (size_t)layer * ps + g * f.NFP + 4 * c4)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const int e = t.tid + NTHREADS * i, r = e >> 6, j = e & 63;
            zpre[i] = (j < f.D && row0 + r < B) ? Zsave[(size_t)layer * zs + (row0 + r) * f.D + j] : 0.f;
        }
    };
    auto tile_commit = [&](int layer) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                float4 v;
                if constexpr (TPRE) v = ppre[i][k];
                else {
                    const long g = row0 + t.wave + NWAVE * i;
                    const int c4 = t.lane + 64 * k;
                    v = (g < B && c4 < w4max) ? *reinterpret_cast<const float4*>(Psave + (size_t)layer * ps + g * f.NFP + 4 * c4)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *reinterpret_cast<float4*>(PT + (t.wave + NWAVE * i) * l.PS + 4 * (t.lane + 64 * k)) = v;
            }
            ZT[t.tid + NTHREADS * i] = zpre[i];
        }
    };
    head_fetch(img + (size_t)(f.L - 1) * lfl);
    // x <- wrap(x - pre-shift of the top layer)
    {
        const float* mt = packed + (size_t)(f.L - 1) * f.layer_stride + f.o_meta;
        const bool first = GRAD && lp.XP && (lp.fold.flags & 1);               // + k_gen_hmc_begin (hmc.py:134)
        long nvb = B;
        if (first && lp.fold.n_valid) { const long nvd = (long)*lp.fold.n_valid; nvb = nvd < B ? nvd : B; }
        for (int e = t.tid; e < R8 * 64; e += NTHREADS) {
            const int r = e >> 6, j = e & 63;
            const long g = row0 + r;
            float v = 0.f, kt = 0.f;
            if (j < f.D && g < B) {
                if (GRAD && lp.XP) {                                            // first half of a leapfrog (k_gen_leap_pre, hmc.py:140-142)
                    const long i = g * f.D + j;
                    const float eps = *lp.eps_ptr + *lp.ceps_ptr;
                    const float m = lp.mass[j];
                    float p0, gu0, x0;
                    if (first) {                                                // rows of dropped chains: a defined (zero) state
                        const bool on = g < nvb;
                        p0 = on ? lp.fold.noise_p[i] * m : 0.f;
                        const float gr = on ? -(lp.c.g_q * lp.fold.start_gq[i] + lp.c.g_p * lp.fold.start_gp[i]) : 0.f;
                        gu0 = g_clamp_nan0_s8(gr, lp.max_grad);
                        x0 = on ? lp.fold.start_x[i] : 0.f;
                        kt = p0 * p0 / m;
                    } else {
                        p0 = lp.P[i]; gu0 = lp.GU[i]; x0 = lp.XP[i];
                    }
                    const float p = p0 - eps * gu0 / 2.f;
                    lp.P[i] = p;
                    v = x0 + eps / m * p;
                    lp.XP[i] = v;
                    lp.x_out[i] = v;
                } else {
                    v = x[g * f.D + j];
                }
                if (mt[M_PREON * 64 + j] != 0.f) v = sp_wrap(v - mt[M_PRESH * 64 + j], mt[M_TB * 64 + j]);
            }
            ZT[e] = v;
            if (first) {                    // K(p0) in k_gen_hmc_begin's order: lane c < 16 adds coordinates c, c + 16, c + 32, c + 48
                const int c16 = t.lane & 15;
                float k0 = kt;
                k0 += __shfl(kt, c16 + 16);
                k0 += __shfl(kt, c16 + 32);
                k0 += __shfl(kt, c16 + 48);
                k0 = s8_row16_sum_xor(k0) / 2.f;
                if (t.lane == 0 && g < nvb)
                    lp.fold.logp_cur[g] = (lp.c.c_q * lp.fold.cur_lq[g] + lp.c.c_p * lp.fold.cur_lp[g]) - k0;
            }
        }
    }
    constexpr int NRG = (RB + 1) / 2;                                          // groups of 8 chains (RB = 1: half a group)
    float ld_acc[NRG];                                                      // element-wise stages: thread = (chain 8 i + tid / 32, coordinate tid % 32)
#pragma unroll
    for (int i = 0; i < NRG; ++i) ld_acc[i] = 0.f;
    for (int layer = f.L - 1; layer >= 0; --layer) {
        const float* Lr = img + (size_t)layer * lfl;
        const bool tl = layer == 1;
        __syncthreads();                                                       // ZT complete; HD / PT free
        if (tl) S8_TL(0);
        head_commit();
        if (GRAD) {
            for (int e = t.tid; e < R8 * f.D; e += NTHREADS) {
                const int r = e / f.D, j = e % f.D;
                if (row0 + r < B) Zsave[(size_t)layer * zs + (row0 + r) * f.D + j] = ZT[r * 64 + j];
            }
        }
        __syncthreads();
        s8_prologue(s, reinterpret_cast<const float4*>(Lr + H) + (size_t)t.wave * TPL * 64);
        const int n_id = (int)meta[M_CNT * 64];
        for (int e = t.tid; e < R8 * S8_AS; e += NTHREADS) {                   // identity coordinates + periodic features
            const int r = e / S8_AS, i = e % S8_AS;
            float v = 0.f;
            if (i < n_id) {
                v = ZT[r * 64 + (int)meta[M_IDF * 64 + i]];
                if (meta[M_PFON * 64 + i] != 0.f) {
                    const int k = (int)meta[M_PFK * 64 + i];
                    const float sc = meta[M_PFS * 64 + i];
                    v = HD[S8H_PFW + 2 * k] * sinf(sc * v) + HD[S8H_PFW + 2 * k + 1] * cosf(sc * v);
                }
            }
            A0[e] = v;
        }
        s8_barrier();
        if (tl) S8_TL(1);
        f32x4 o[RB];
        float h0[RB][4];
        unsigned short* mk = GRAD ? reinterpret_cast<unsigned short*>(MASK + (size_t)layer * (NWAVE * 64) + t.tid) : nullptr;
        {   // h0 = A0 W0 + b0; X2 = relu(h0)
            S8Acc<RB> acc;
            s8_zero(acc);
            s8_iter<K0Q, K0Q, 0, S8_INF>(s, A0 + t.arow * S8_AS, 4 * S8_AS, acc);
            s8_fold(acc, o);
            const float bv = HD[S8H_B0 + col];
            unsigned bits = 0;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = o[rb][r] + bv;
                    h0[rb][r] = v;
                    X2[(4 * rb + r) * S8_WS + col] = v > 0.f ? v : 0.f;
                    if (GRAD) bits |= v > 0.f ? (1u << (4 * rb + r)) : 0u;
                }
            if (GRAD) mk[0] = (unsigned short)bits;
        }
        s8_barrier();
        {   // t = relu(h0) Wa + ba; X1 = relu(t)
            if (layer > 0) head_fetch(Lr - lfl);
            s8_gemm64<PHF, RB>(s, X2, S8_WS, t, o);
            const float bv = HD[S8H_BA + col];
            unsigned bits = 0;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = o[rb][r] + bv;
                    X1[(4 * rb + r) * S8_WS + col] = v > 0.f ? v : 0.f;
                    if (GRAD) bits |= v > 0.f ? (1u << (4 * rb + r)) : 0u;
                }
            if (GRAD) mk[1] = (unsigned short)bits;
        }
        s8_barrier();
        {   // h1 = h0 + relu(t) Wb + bb -> T
            s8_gemm64<PHF, RB>(s, X1, S8_WS, t, o);
            const float bv = HD[S8H_BB + col];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) T[(4 * rb + r) * S8_WS + col] = h0[rb][r] + (o[rb][r] + bv);
        }
        s8_barrier();
        if (tl) S8_TL(2);
        s8_for<0, NCH>([&](auto cc) {                                          // P = h1 Wf + bf, chunks of 256 outputs
            constexpr int c = decltype(cc)::value;
            if constexpr (c + 1 < NCH) s8_gemm64<PHF, RB>(s, T, S8_WS, t, o);
            else s8_gemm64<PHF, RB, 31>(s, T, S8_WS, t, o);                        // the layer's last tiles: the ring drains
            const float bv = HD[S8H_BF + c * 256 + col];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) PT[(4 * rb + r) * l.PS + c * 256 + col] = o[rb][r] + bv;
        });
        __syncthreads();
        if (tl) S8_TL(3);
        const int n_tr = (int)meta[M_CNT * 64 + 1];
        if (GRAD) {                                                            // conditioner output, for the reverse sweep
            const int w4 = (n_tr * SP_NP + 3) >> 2;
            for (int r = t.wave; r < R8; r += NWAVE)
                if (row0 + r < B)
                    for (int c4 = t.lane; c4 < w4; c4 += 64)
                        *reinterpret_cast<float4*>(Psave + (size_t)layer * ps + (row0 + r) * f.NFP + 4 * c4) =
                            *reinterpret_cast<const float4*>(PT + r * l.PS + 4 * c4);
        }
#pragma unroll 1
        for (int ri = 0; ri < NRG; ++ri) {
        const int row = 8 * ri + t.row;
        float ldv = 0.f;
        for (int j = t.c; j < f.D && row < R8; j += 32) {
            float p[SP_NP];
            int pos;
            const int kind = sp_coord_params(f, HD, meta, PT, l.PS, row, j, isq, p, pos);
            const float tb = meta[M_TB * 64 + j];
            float out = ZT[row * 64 + j];
            if (kind) {
                Rqs sp;
                rqs_setup(p, meta[M_CIRC * 64 + j] != 0.f, tb, sp);
                float l1;
                rqs_forward(sp, out, tb, out, l1);
                ldv += l1;
            }
            if (layer > 0 && HD[S8H_NXT + 64 + j] != 0.f) out = sp_wrap(out - HD[S8H_NXT + j], tb);   // next stage's shift
            ZT[row * 64 + j] = out;
        }
        if (NRG == 1 || ri == 0) ld_acc[0] += ldv;
        else ld_acc[NRG - 1] += ldv;
        }
        if (tl) S8_TL(4);
    }
    __syncthreads();
    if constexpr (GRAD) { head_fetch(img); tile_fetch(0); }
    // base UniformGaussian
#pragma unroll
    for (int ri = 0; ri < NRG; ++ri) {
        const int row = 8 * ri + t.row;
        for (int j = t.c; j < f.D && row < R8; j += 32) {
            const float sc = packed[f.o_base + j];
            const float z = ZT[row * 64 + j];
            if (packed[f.o_base + 64 + j] != 0.f) { ld_acc[ri] += -logf(sc); if (GRAD) GT[row * 64 + j] = 0.f; }
            else {
                ld_acc[ri] += -0.5f * 1.8378770664093453f - logf(sc) - 0.5f * ((z / sc) * (z / sc));
                if (GRAD) GT[row * 64 + j] = -(z / sc) / sc;
            }
        }
        const float lq = row32_sum(ld_acc[ri]);
        if (t.c == 0 && row < R8 && row0 + row < B) log_q[row0 + row] = lq;
    }
    if constexpr (GRAD) {
        // ---- reverse sweep: GT = d log q / d(state), layers 0 .. L-1 -------------------------------------------------
        for (int layer = 0; layer < f.L; ++layer) {
            const float* Lr = img + (size_t)layer * lfl;
            const bool tl = layer == 1;
            __syncthreads();
            if (tl) S8_TL(8);
            head_commit();
            tile_commit(layer);
            __syncthreads();
            s8_prologue(s, reinterpret_cast<const float4*>(Lr + H) + (size_t)(NWAVE + t.wave) * TPL * 64);
            if (tl) S8_TL(9);
            const int n_id = (int)meta[M_CNT * 64], n_tr = (int)meta[M_CNT * 64 + 1];
#pragma unroll 1
            for (int ri = 0; ri < NRG; ++ri) {
            const int row = 8 * ri + t.row;
            for (int j = t.c; j < f.D && row < R8; j += 32) {
                float p[SP_NP];
                int pos;
                const int kind = sp_coord_params(f, HD, meta, PT, l.PS, row, j, isq, p, pos);
                if (kind) {
                    const float tb = meta[M_TB * 64 + j];
                    const bool circ = meta[M_CIRC * 64 + j] != 0.f;
                    Rqs sp;
                    rqs_setup(p, circ, tb, sp);
                    const float z = ZT[row * 64 + j], gy = GT[row * 64 + j];
                    if (kind == 2) {
                        float dp[SP_NP];
                        GT[row * 64 + j] = rqs_backward(sp, p, circ, z, tb, gy, isq, dp);
#pragma unroll
                        for (int k = 0; k < SP_NP; ++k) PT[row * l.PS + pos * SP_NP + k] = dp[k];
                    } else {
                        GT[row * 64 + j] = rqs_backward(sp, p, circ, z, tb, gy, 1.f, nullptr);
                    }
                }
            }
            // columns of dP no coordinate owns: the stored tile has the conditioner's padding there, Psave beyond w4 was never written
            for (int c = n_tr * SP_NP + t.c; c < f.NFP && row < R8; c += 32) PT[row * l.PS + c] = 0.f;
            }
            s8_barrier();
            if (tl) S8_TL(10);
            const unsigned mbits = *reinterpret_cast<const unsigned*>(MASK + (size_t)layer * (NWAVE * 64) + t.tid);
            f32x4 o[RB];
            float dh1[RB][4];
            {   // dh1 = dP WfT   (K = NFP)
                S8Acc<RB> acc;
                s8_zero(acc);
                const float* ap = PT + t.arow * l.PS;
                s8_for<0, KFQ / 32>([&](auto ic) { s8_iter<32, 32, 0, S8_INF>(s, ap + 128 * decltype(ic)::value, 4 * l.PS, acc); });
                if constexpr (KFQ % 32 != 0) s8_iter<KFQ % 32, KFQ % 32, 0, S8_INF>(s, ap + 128 * (KFQ / 32), 4 * l.PS, acc);
                s8_fold(acc, o);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { dh1[rb][r] = o[rb][r]; X1[(4 * rb + r) * S8_WS + col] = o[rb][r]; }
            }
            s8_barrier();
            if (tl) S8_TL(11);
            {   // d relu(t) = dh1 WbT, masked by t > 0
                if (layer + 1 < f.L) { head_fetch(Lr + lfl); tile_fetch(layer + 1); }
                s8_gemm64<PHR, RB>(s, X1, S8_WS, t, o);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        X2[(4 * rb + r) * S8_WS + col] = (mbits & (0x10000u << (4 * rb + r))) ? o[rb][r] : 0.f;
            }
            s8_barrier();
            {   // dh0 = dh1 + (d t WaT masked by h0 > 0)
                s8_gemm64<PHR, RB, (TRIM ? 4 : 16) + 31>(s, X2, S8_WS, t, o);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        T[(4 * rb + r) * S8_WS + col] = dh1[rb][r] + ((mbits & (1u << (4 * rb + r))) ? o[rb][r] : 0.f);
            }
            s8_barrier();
            if (tl) S8_TL(12);
            {   // dA0 = dh0 W0T: K split over the waves (wave w: k = 64 w .. 64 w + 63), partial [8][64] products through LDS
                S8Acc<RB> acc;
                s8_zero(acc);
                if constexpr (!TRIM) {
                    s8_iter<16, 16, 0, 15>(s, T + t.arow * S8_WS + 64 * t.wave, 4 * S8_WS, acc);
                    s8_fold(acc, o);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) PART[(t.wave * R8 + 4 * rb + r) * S8_AS + t.lane] = o[rb][r];
                } else {        // dense tiles: lane quarter h4 multiplies k-quad 4 T + h4 of the wave's 16; 16 partials per output
                    const int h4 = t.lane >> 4;
                    s8_iter_k<16, 4, 4, PHR, 3>(s, T + t.arow * S8_WS + 64 * t.wave + 4 * h4, 4 * S8_WS, acc);
                    s8_fold(acc, o);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) PART[((4 * t.wave + h4) * R8 + 4 * rb + r) * 16 + (t.lane & 15)] = o[rb][r];
                }
            }
            s8_barrier();
            if (tl) S8_TL(13);
            for (int e = t.tid; e < R8 * 64; e += NTHREADS) {
                const int r = e >> 6, i = e & 63;
                if (i < n_id) {
                    float d;
                    if constexpr (!TRIM) {
                        const float* pp = PART + r * S8_AS + i;
                        d = (pp[0] + pp[R8 * S8_AS]) + (pp[2 * R8 * S8_AS] + pp[3 * R8 * S8_AS]);
                    } else {                                                   // (n_id <= 16) fixed balanced tree over the 16 partials
                        const float* pp = PART + r * 16 + i;
                        float v16[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) v16[q] = pp[q * R8 * 16];
#pragma unroll
                        for (int w = 1; w < 16; w *= 2)
#pragma unroll
                            for (int q = 0; q < 16; q += 2 * w) v16[q] = v16[q] + v16[q + w];
                        d = v16[0];
                    }
                    const int feat = (int)meta[M_IDF * 64 + i];
                    if (meta[M_PFON * 64 + i] != 0.f) {
                        const int k = (int)meta[M_PFK * 64 + i];
                        const float sc = meta[M_PFS * 64 + i], zz = ZT[r * 64 + feat];
                        d = d * (sc * (HD[S8H_PFW + 2 * k] * cosf(sc * zz) - HD[S8H_PFW + 2 * k + 1] * sinf(sc * zz)));
                    }
                    GT[r * 64 + feat] += d;
                }
            }
            if (tl) S8_TL(14);
        }
        __syncthreads();
        for (int e = t.tid; e < R8 * f.D; e += NTHREADS) {
            const int r = e / f.D, j = e % f.D;
            if (row0 + r < B) grad_x[(row0 + r) * f.D + j] = GT[r * 64 + j];
        }
        if (lp.XP) {
            // target + gradient of the moved point (k_target) and the second half step (k_gen_leap_post, hmc.py:145-147) for this
            // workgroup's chains, on the 16-row mapping of target_tile (rows >= R8: zeros, not stored)
            const int D = f.D;
            float* XH = X1;                                                     // [16][D]; X1 / X2 (the K-split partials) are free now
            float* GPH = X2;
            Tid tt;
            for (int e = t.tid; e < 16 * D; e += NTHREADS) {
                const int r = e / D, j = e - r * D;
                const long g = row0 + r;
                XH[e] = (r < R8 && g < B) ? lp.XP[g * D + j] : 0.f;
            }
            __syncthreads();
            const float lpv = target_tile<true>(lp.tg, XH, D, GPH, D, tt);
            const long g = row0 + tt.row;
            const bool last = (lp.fold.flags & 2) != 0;                         // + k_gen_hmc_accept, k_gen_hmc_adapt
            long nv = B;
            if (last && lp.fold.n_valid) { const long nvd = (long)*lp.fold.n_valid; nv = nvd < B ? nvd : B; }
            const bool active = last && tt.row < R8 && g < nv;
            float k1 = 0.f, dist2 = 0.f;
            if (tt.row < R8 && g < B) {
                if (tt.c == 0) lp.prop_lp[g] = lpv;
                const float eps = *lp.eps_ptr + *lp.ceps_ptr;
                for (int j = tt.c; j < D; j += 16) {
                    const long i = g * D + j;
                    const float gpv = GPH[tt.row * D + j];
                    lp.prop_gp[i] = gpv;
                    const float gu = g_clamp_nan0_s8(-(lp.c.g_q * GT[tt.row * 64 + j] + lp.c.g_p * gpv), lp.max_grad);
                    lp.GU[i] = gu;
                    const float p = lp.P[i] - eps * gu / 2.f;
                    lp.P[i] = p;
                    if (active) {
                        k1 += p * p / lp.mass[j];
                        const float dx = lp.fold.cur_x[i] - XH[tt.row * D + j];
                        dist2 += dx * dx;
                    }
                }
            }
            if (last) {
                k1 = s8_row16_sum_xor(k1) / 2.f;
                dist2 = s8_row16_sum_xor(dist2);
                float contrib = 0.f, dist = 0.f;
                if (active) {
                    const float lq = log_q[g], lpp = lpv;
                    const float lq_c = lp.fold.cur_lq[g], lp_c = lp.fold.cur_lp[g];
                    const float delta = ((lp.c.c_q * lq + lp.c.c_p * lpp) - k1) - lp.fold.logp_cur[g];
                    const bool valid = isfinite(delta);
                    const float dd = valid ? delta : -INFINITY;
                    const bool accept = valid && (dd > -lp.fold.noise_e[g]);            // hmc.py:105-124
                    contrib = expf(fminf(dd, 0.f));
                    dist = accept ? 0.f : sqrtf(dist2);                          // store_info sees the committed point
                    if (accept) {
                        for (int j = tt.c; j < D; j += 16) {
                            const long i = g * D + j;
                            lp.fold.cur_x[i] = XH[tt.row * D + j];
                            lp.fold.cur_gq[i] = GT[tt.row * 64 + j];
                            lp.fold.cur_gp[i] = GPH[tt.row * D + j];
                        }
                    }
                    if (tt.c == 0) {
                        if (accept) { lp.fold.cur_lq[g] = lq; lp.fold.cur_lp[g] = lpp; }
                        if (lp.fold.log_w) {                                      // ais.py:93-100
                            const float lqf = accept ? lq : lq_c, lpf = accept ? lpp : lp_c;
                            lp.fold.log_w[g] = lp.fold.log_w[g] + ((lp.fold.nx.c_q * lqf + lp.fold.nx.c_p * lpf) -
                                                                    (lp.c.c_q * lqf + lp.c.c_p * lpf));
                        }
                    }
                }
                // per-chain statistics of EVERY row of the 16-row blocks (rows past the grid: the last workgroup's)
                const long nrow = 16L * lp.fold.nblk;
                if (tt.c == 0) {
                    if (tt.row < R8 && g < nrow) {
                        __hip_atomic_store(lp.fold.row_acc + g, contrib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(lp.fold.row_dist + g, dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const long g2 = (long)R8 * gridDim.x + tt.row;
                    if (blockIdx.x == gridDim.x - 1 && g2 < nrow) {
                        __hip_atomic_store(lp.fold.row_acc + g2, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(lp.fold.row_dist + g2, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                s8_fold_adapt_last(lp.fold, B, PT, t.lane);
            }
        }
    }
#undef S8_TL
}
