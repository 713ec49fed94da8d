// Circular / linear-tail rational-quadratic spline COUPLING flow (R9s): the flow family the reference builds for
// alanine dipeptide - normflows `CircularCoupledRationalQuadraticSpline` x n_layers (+ PeriodicShift / PeriodicWrap,
// UniformGaussian base), experiments/make_flow/make_aldp_model.py:57-71,121-134,146-167.  Arithmetic restated in
// oracle/spline.py (normflows is absent from the reference tree: parity unpinned, oracle = specification).
//
// Structure (one C-ABI call enqueues all launches of a density / sampling pass, no host synchronisation):
//   k_spline_net_fwd   conditioner of one coupling layer for a 16-chain tile: periodic features of the identity
//                      coordinates -> ResidualNet (Linear, pre-activation residual block, Linear) on the fp32 matrix
//                      cores (flow_device.h ring GEMMs, weights streamed in MFMA B-operand tiles) -> 3K+1 spline
//                      parameters per transformed coordinate, written to HBM
//   k_spline_apply     element-wise: spline evaluation (log_prob direction) or inversion (sampling direction) of the
//                      transformed coordinates with the conditioner's parameters and of the identity coordinates with
//                      the layer's unconditional parameters; log-det row sums; periodic shift / wrap of the next stage
//   k_spline_apply_bwd reverse-mode through the same maps: d log q / d(input) and the cotangents of the 3K+1
//                      conditioner outputs
//   k_spline_net_bwd   recomputes the conditioner's activations for the tile and back-propagates the parameter
//                      cotangents to the identity coordinates (transposed weight tiles on the matrix cores)
// One wave handles one chain in the element-wise kernels (lane = coordinate, D <= 64).
#include "flow_device.h"
#include "target_device.h"
#include "stream_r8.h"
#include "launch.h"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace fab {

constexpr int SP_K = 8;                 // bins
constexpr int SP_NP = 3 * SP_K + 1;     // parameters per coordinate: K widths, K heights, K + 1 knot derivatives
constexpr int SP_MD = 64;               // max dim
constexpr int SP_META_ROWS = 12;        // per-layer metadata rows of 64 floats (see fabhip.h)
constexpr float SP_MIN_W = 1e-3f, SP_MIN_H = 1e-3f, SP_MIN_D = 1e-3f;

enum { M_IDF = 0, M_TRF = 1, M_CIRC = 2, M_TB = 3, M_PFON = 4, M_PFS = 5, M_PFK = 6, M_PRESH = 7, M_PREON = 8,
       M_POSTSH = 9, M_POSTON = 10, M_CNT = 11,
       M_POSID = 12, M_POSTR = 13 };      // derived at pack time: position of coordinate j among the identity / transformed
                                          // features of the layer (-1: not one), so the element-wise kernels need no search

struct SplineDims {
    int D, L, W, Wp, NTWM, KBW;          // hidden width, padded to 64 * tiles-per-wave
    int n_tr_max, NCH, NFP;              // chunks of Wp conditioner outputs: NFP = NCH * Wp >= n_tr_max * SP_NP
    int o_meta, o_unc, o_pfw, o_W0, o_b0, o_Wa, o_ba, o_Wb, o_bb, o_Wf, o_bf, o_WfT, o_WbT, o_WaT, o_W0T;
    int o_h;                             // fast mode: bf16 images [Wa | Wb | Wf (NCH chunks) | WfT | WbT | WaT], Wp^2/2 floats per Wp x Wp
    int layer_stride, o_base, total;
    // 8-chain-tile image (spline_r8.h; Wp == 256 only, else o_r8 == 0): per layer [head | forward tiles | reverse tiles]
    int o_r8, r8_head, r8_tpl, r8_layer; // head floats, 1-KiB tiles per wave and direction, floats per layer
    int r8_trim;                         // round 5: the r8 stream without its zero tiles (D <= 32 with two output chunks: <= 16 identity
                                         // features -> W0 is 4 k-quads, not 16; <= 400 conditioner outputs -> WfT is 100 k-quads, not
                                         // 128; W0T as 4 dense tiles, not 16): 260 + 232 instead of 272 + 272 tiles per wave and layer
};

FAB_HD SplineDims make_spline_dims(int D, int L, int W) {
    SplineDims f;
    f.D = D; f.L = L; f.W = W;
    f.NTWM = ntw_variant(W); f.Wp = 64 * f.NTWM; f.KBW = f.Wp / 16;
    f.n_tr_max = (D + 1) / 2;
    f.NCH = ceil_div(f.n_tr_max * SP_NP, f.Wp);
    f.NFP = f.NCH * f.Wp;
    int o = 0;
    f.o_meta = o; o += (SP_META_ROWS + 2) * 64;
    f.o_unc = o; o += SP_MD * SP_NP + 32;                 // [64][25] (+ pad to a multiple of 64 floats below)
    o = (o + 63) & ~63;
    f.o_pfw = o; o += 2 * SP_MD;                           // [64][2] periodic-feature weights
    const int T = f.KBW * 256;                             // floats per column-tile strip of K = Wp
    f.o_W0 = o; o += 4 * (4 * f.NTWM) * 256;              // K = 64 (identity coordinates, padded), N = Wp
    f.o_b0 = o; o += f.Wp;
    f.o_Wa = o; o += (4 * f.NTWM) * T;
    f.o_ba = o; o += f.Wp;
    f.o_Wb = o; o += (4 * f.NTWM) * T;
    f.o_bb = o; o += f.Wp;
    f.o_Wf = o; o += f.NCH * (4 * f.NTWM) * T;            // NCH chunks of [Wp x Wp]
    f.o_bf = o; o += f.NFP;
    f.o_WfT = o; o += (4 * f.NTWM) * (f.NFP / 16) * 256;  // K = NFP, N = Wp
    f.o_WbT = o; o += (4 * f.NTWM) * T;
    f.o_WaT = o; o += (4 * f.NTWM) * T;
    f.o_W0T = o; o += 4 * T;                               // K = Wp, N = 64
    f.o_h = o; o += (4 + 2 * f.NCH) * (f.Wp * f.Wp / 2);
    f.layer_stride = o;
    f.o_base = L * f.layer_stride;                         // scale[64], circ[64]
    f.total = f.o_base + 128;
    f.o_r8 = 0;
    f.r8_head = ((SP_META_ROWS + 2) * 64 + 1664 + 128) + 3 * 256 + 128 + f.NFP;   // spline_r8.h: S8H_*
    f.r8_tpl = 16 + 64 * (2 + f.NCH);
    f.r8_trim = (f.NCH == 2 && (D + 1) / 2 <= 16) ? 1 : 0;
    f.r8_layer = f.r8_head + 2 * NWAVE * f.r8_tpl * 256;
    if (f.NTWM == 4) {
        f.o_r8 = (f.total + 255) & ~255;
        f.total = f.o_r8 + L * f.r8_layer;
    }
    return f;
}

static inline int check_spline_shape(int D, int L, int W) {
    if (D < 2 || L < 1 || W < 1) return FABHIP_EINVAL;
    if (D > SP_MD || L > FABHIP_MAX_LAYERS || W > 256) return FABHIP_ENOTSUP;       // NTWM in {1, 2, 4}
    return FABHIP_OK;
}

// Training tape of one log_prob call (fabhip_spline_log_prob_tape): per coupling layer, row-major [B][width] matrices
// of the conditioner's activations and of the reverse sweep's cotangents (seed 1 per sample).  The weight gradients
// are sum_b c_b (cotangent row)^T (activation row): GEMMs over these matrices, done by the caller (rocBLAS through
// torch.mm in fab_torch_amd/spline_flow.py) with the per-sample coefficient c_b folded into the cotangent rows.
struct SplineTape {
    float* base;                          // nullptr: no tape
    long o_XI, o_A0, o_dA0;               // [B][64]: raw identity coordinates, after the periodic features, cotangent of A0
    long o_R0, o_R1, o_H1;                // [B][Wp]: relu(h0), relu(t), h1
    long o_dH1, o_dT, o_dH0;              // [B][Wp]: cotangents of h1, of t (masked), of h0
    long o_dP;                            // [B][NFP]: cotangent of the conditioner output
    long o_dU;                            // [B][64 * 25]: cotangent of the unconditional parameters per identity position
    long layer_stride;
};

static inline SplineTape make_spline_tape(const SplineDims& f, long B, float* base) {
    SplineTape t;
    t.base = base;
    long o = 0;
    t.o_XI = o; o += B * 64; t.o_A0 = o; o += B * 64; t.o_dA0 = o; o += B * 64;
    t.o_R0 = o; o += B * f.Wp; t.o_R1 = o; o += B * f.Wp; t.o_H1 = o; o += B * f.Wp;
    t.o_dH1 = o; o += B * f.Wp; t.o_dT = o; o += B * f.Wp; t.o_dH0 = o; o += B * f.Wp;
    t.o_dP = o; o += B * f.NFP;
    t.o_dU = o; o += B * (long)(SP_MD * SP_NP);
    t.layer_stride = o;
    return t;
}

// ------------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------------
struct SplineSrc {
    const float *meta, *w0, *b0, *wa, *ba, *wb, *bb, *wf, *bf, *pfw, *uw, *uh, *ud;
};

__device__ __forceinline__ void sp_tile_kn(int off, int KB, int& k, int& n) {
    const int tile = off >> 8, within = off & 255;
    const int lane = within >> 2, tt = within & 3;
    const int c = tile / KB, S = tile % KB;
    k = 16 * S + 4 * (lane >> 4) + tt;
    n = 16 * c + (lane & 15);
}

__global__ __launch_bounds__(256) void k_spline_pack_layer(SplineDims f, SplineSrc s, int layer, float* __restrict__ packed) {
    float* __restrict__ dst = packed + (size_t)layer * f.layer_stride;
    const int n_id = (int)s.meta[M_CNT * 64 + 0], n_tr = (int)s.meta[M_CNT * 64 + 1], n_pf = (int)s.meta[M_CNT * 64 + 2];
    const int W = f.W, Wp = f.Wp, KBW = f.KBW, nout = n_tr * SP_NP;
    for (int off = blockIdx.x * blockDim.x + threadIdx.x; off < f.o_h; off += gridDim.x * blockDim.x) {
        float v = 0.f;
        int k, n;
        if (off < f.o_meta + SP_META_ROWS * 64) {
            v = s.meta[off - f.o_meta];
        } else if (off < f.o_meta + (SP_META_ROWS + 2) * 64) {
            const int e = off - f.o_meta - SP_META_ROWS * 64, row = e >> 6, j = e & 63;
            const int cnt = row == 0 ? n_id : n_tr;
            const float* feats = s.meta + (row == 0 ? M_IDF : M_TRF) * 64;
            v = -1.f;
            for (int i = 0; i < cnt; ++i) if ((int)feats[i] == j) v = (float)i;
        } else if (off < f.o_unc) {
            v = 0.f;
        } else if (off < f.o_pfw) {
            const int e = off - f.o_unc, i = e / SP_NP, p = e % SP_NP;
            if (i < n_id && e < SP_MD * SP_NP)
                v = p < SP_K ? s.uw[i * SP_K + p] : (p < 2 * SP_K ? s.uh[i * SP_K + p - SP_K] : s.ud[i * (SP_K + 1) + p - 2 * SP_K]);
        } else if (off < f.o_W0) {
            const int e = off - f.o_pfw;
            if (e < 2 * n_pf) v = s.pfw[e];
        } else if (off < f.o_b0) {                 // W0: B[k][n] = w0[n][k]   (k identity position, n hidden)
            sp_tile_kn(off - f.o_W0, 4, k, n);
            if (k < n_id && n < W) v = s.w0[n * n_id + k];
        } else if (off < f.o_Wa) {
            const int j = off - f.o_b0; if (j < W) v = s.b0[j];
        } else if (off < f.o_ba) {                 // Wa: B[k][n] = wa[n][k]
            sp_tile_kn(off - f.o_Wa, KBW, k, n);
            if (k < W && n < W) v = s.wa[n * W + k];
        } else if (off < f.o_Wb) {
            const int j = off - f.o_ba; if (j < W) v = s.ba[j];
        } else if (off < f.o_bb) {
            sp_tile_kn(off - f.o_Wb, KBW, k, n);
            if (k < W && n < W) v = s.wb[n * W + k];
        } else if (off < f.o_Wf) {
            const int j = off - f.o_bb; if (j < W) v = s.bb[j];
        } else if (off < f.o_bf) {                 // Wf chunk c: B[k][n] = wf[c Wp + n][k]
            const int e = off - f.o_Wf, per = 4 * f.NTWM * KBW * 256;
            const int c = e / per;
            sp_tile_kn(e % per, KBW, k, n);
            const int col = c * Wp + n;
            if (k < W && col < nout) v = s.wf[col * W + k];
        } else if (off < f.o_WfT) {
            const int j = off - f.o_bf; if (j < nout) v = s.bf[j];
        } else if (off < f.o_WbT) {                // WfT: B[k][n] = wf[k][n]   (k conditioner output, n hidden)
            sp_tile_kn(off - f.o_WfT, f.NFP / 16, k, n);
            if (k < nout && n < W) v = s.wf[k * W + n];
        } else if (off < f.o_WaT) {                // WbT: B[k][n] = wb[k][n]
            sp_tile_kn(off - f.o_WbT, KBW, k, n);
            if (k < W && n < W) v = s.wb[k * W + n];
        } else if (off < f.o_W0T) {
            sp_tile_kn(off - f.o_WaT, KBW, k, n);
            if (k < W && n < W) v = s.wa[k * W + n];
        } else {                                    // W0T: B[k][n] = w0[k][n]   (k hidden, n identity position)
            sp_tile_kn(off - f.o_W0T, KBW, k, n);
            if (k < W && n < n_id) v = s.w0[k * n_id + n];
        }
        dst[off] = v;
    }
}

__global__ void k_spline_pack_base(SplineDims f, const float* __restrict__ scale, const float* __restrict__ circ,
                                   float* __restrict__ packed) {
    const int j = threadIdx.x;
    if (j < 64) {
        packed[f.o_base + j] = j < f.D ? scale[j] : 1.f;
        packed[f.o_base + 64 + j] = j < f.D ? circ[j] : 0.f;
    }
}

// fast mode: bf16 images of the conditioner's Wp x Wp GEMM operands in v_mfma_f32_16x16x32_bf16 B-operand tiles
// (flow_device.h).  Image m of a layer, HALF = Wp^2/2 floats each, K x N:
//   0 Wa (Wp x Wp)   1 Wb   2 .. 2+NCH-1 Wf chunk c   then WfT (K = NFP: NCH HALFs, KB2T = NFP/32)   WbT   WaT
__device__ __forceinline__ unsigned sp_bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__global__ __launch_bounds__(256) void k_spline_pack_bf16(SplineDims f, SplineSrc s, int layer, float* __restrict__ packed) {
    const int W = f.W, Wp = f.Wp, HALF = Wp * Wp / 2, n_tr = (int)s.meta[M_CNT * 64 + 1], nout = n_tr * SP_NP;
    const int total = (4 + 2 * f.NCH) * HALF;
    unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(packed + (size_t)layer * f.layer_stride + f.o_h);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        int m = e / HALF, off = e % HALF;
        int KB2T = Wp / 32;
        const bool wft = m >= 2 + f.NCH && m < 2 + 2 * f.NCH;
        if (wft) { off = e - (2 + f.NCH) * HALF; KB2T = f.NFP / 32; }      // one K = NFP image spanning NCH HALFs
        const int tile = off >> 8, lane = (off >> 2) & 63, j = 2 * (off & 3);
        const int c = tile / KB2T, S = tile % KB2T;
        const int k = 32 * S + 8 * (lane >> 4) + j, n = 16 * c + (lane & 15);
        float v[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kk = k + q;
            if (m == 0) { if (kk < W && n < W) v[q] = s.wa[n * W + kk]; }
            else if (m == 1) { if (kk < W && n < W) v[q] = s.wb[n * W + kk]; }
            else if (m < 2 + f.NCH) { const int col = (m - 2) * Wp + n; if (kk < W && col < nout) v[q] = s.wf[col * W + kk]; }
            else if (wft) { if (kk < nout && n < W) v[q] = s.wf[kk * W + n]; }
            else if (m == 2 + 2 * f.NCH) { if (kk < W && n < W) v[q] = s.wb[kk * W + n]; }
            else { if (kk < W && n < W) v[q] = s.wa[kk * W + n]; }
        }
        dst[e] = sp_bf16_rne(v[0]) | (sp_bf16_rne(v[1]) << 16);
    }
}

// ------------------------------------------------------------------------------------------------
// conditioner (ResidualNet) on a 16-chain tile
// ------------------------------------------------------------------------------------------------
struct NetLds {
    int AS, WS, PS;                       // leading dims: identity inputs (64 + 4), hidden (Wp + 4), dparams (NFP + 4)
    int o_A0, o_H0, o_T, o_X1, o_X2, o_DP, o_PART, total;
    int o_ZT, o_GT;                       // persistent kernel: state tile and cotangent tile [16][64]
};
// persist: the one-launch density kernel (k_spline_logprob): + state / cotangent tiles; its conditioner output P lives
// in the DP region (the reverse sweep turns it into dP in place)
FAB_HD NetLds make_net_lds(const SplineDims& f, bool bwd, bool persist = false) {
    NetLds l;
    l.AS = 64 + 4; l.WS = f.Wp + 4; l.PS = f.NFP + 4;
    int o = 0;
    l.o_A0 = o; o += ROWS * l.AS;
    l.o_H0 = o; o += ROWS * l.WS;
    l.o_T = o; o += ROWS * l.WS;
    l.o_X1 = o; o += ROWS * l.WS;
    l.o_X2 = o; o += ROWS * l.WS;
    l.o_DP = o; if (bwd || persist) o += ROWS * l.PS;
    l.o_PART = o; if (bwd) o += NWAVE * ROWS * l.AS;
    l.o_ZT = o; if (persist) o += ROWS * 64;
    l.o_GT = o; if (persist) o += ROWS * 64;
    l.total = (o + 3) & ~3;
    return l;
}

// plain GEMM on the tile: acc[i] = A[16 x 16 KB] @ B[:, tile wave + 4 i] (+ bias)
template <int NTWM, int DEPTH, bool BIAS>
__device__ __forceinline__ void sp_gemm(const float* A, int lda, int KB, const float4* Bp, const float* bias, const Tid& t,
                                        f32x4 (&acc)[NTWM]) {
    WRing<NTWM, DEPTH> w;
    if (!BIAS) {
#pragma unroll
        for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    ring_issue<NTWM, DEPTH, BIAS>(w, Bp, KB, bias, t);
    ring_run<NTWM, DEPTH, false, BIAS>(w, A, lda, 0, KB, Bp, t, acc);
}

// fast-mode twin of sp_gemm for the Wp x Wp operands: image `m` of the layer's bf16 block (see k_spline_pack_bf16)
template <int NTWM, bool BIAS>
__device__ __forceinline__ void sp_gemm_bf16(const SplineDims& f, const float* A, int lda, const float* __restrict__ Lp, int m,
                                             const float* bias, const Tid& t, f32x4 (&acc)[NTWM]) {
#pragma unroll
    for (int i = 0; i < NTWM; ++i) {
        const float bv = BIAS ? bias[16 * (t.wave + 4 * i) + t.n] : 0.f;
        acc[i] = (f32x4){bv, bv, bv, bv};
    }
    const uint4* Bh = reinterpret_cast<const uint4*>(Lp + f.o_h + (size_t)m * (f.Wp * f.Wp / 2));
    gemm_bf16_slice<NTWM>(A, lda, Bh, f.Wp / 32, 0, t, acc);
}

// identity coordinates of the tile with their periodic features -> A0 (columns >= n_id zero); returns nothing
__device__ __forceinline__ void sp_load_identity(const SplineDims& f, const float* __restrict__ Lp, const float* __restrict__ Z,
                                                 long row0, long B, float* A0, int AS, const Tid& t) {
    const float* meta = Lp + f.o_meta;
    const int n_id = (int)meta[M_CNT * 64];
    for (int e = t.tid; e < ROWS * AS; e += NTHREADS) {
        const int r = e / AS, i = e % AS;
        const long g = row0 + r;
        float v = 0.f;
        if (i < n_id && g < B) {
            const int feat = (int)meta[M_IDF * 64 + i];
            v = Z[g * f.D + feat];
            if (meta[M_PFON * 64 + i] != 0.f) {
                const int k = (int)meta[M_PFK * 64 + i];
                const float s = meta[M_PFS * 64 + i];
                v = Lp[f.o_pfw + 2 * k] * sinf(s * v) + Lp[f.o_pfw + 2 * k + 1] * cosf(s * v);
            }
        }
        A0[e] = v;
    }
}

// rows row0 .. row0+15 of a row-major [B][width] matrix (width % 4 == 0, 16-byte aligned rows) -> LDS tile [16][ldd]:
// one wave per row (4 rows each), float4 per lane, no index division (the element-wise version of this load cost 27 %
// of the reverse sweep)
__device__ __forceinline__ void sp_tile_load4(float* dst, int ldd, const float* __restrict__ src, long lds_src, int width,
                                              long row0, long B, const Tid& t) {
    for (int r = t.wave; r < ROWS; r += NWAVE) {
        const long g = row0 + r;
        for (int c4 = t.lane; 4 * c4 < width; c4 += 64) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < B) v = *reinterpret_cast<const float4*>(src + g * lds_src + 4 * c4);
            *reinterpret_cast<float4*>(dst + r * ldd + 4 * c4) = v;
        }
    }
}

// the reverse: LDS tile [16][lds_src] -> rows row0 .. of a row-major [B][ldd] matrix, float4 per lane
__device__ __forceinline__ void sp_tile_store4(float* __restrict__ dst, long ldd, const float* src, int lds_src, int width,
                                               long row0, long B, const Tid& t) {
    for (int r = t.wave; r < ROWS; r += NWAVE) {
        const long g = row0 + r;
        if (g < B)
            for (int c4 = t.lane; 4 * c4 < width; c4 += 64)
                *reinterpret_cast<float4*>(dst + g * ldd + 4 * c4) = *reinterpret_cast<const float4*>(src + r * lds_src + 4 * c4);
    }
}

// tape: rows of an LDS tile [16][ld] (first `width` columns) -> dst [B][width]
__device__ __forceinline__ void sp_tape_rows(float* __restrict__ dst, int width, const float* src, int ld, long row0, long B,
                                             const Tid& t) {
    for (int e = t.tid; e < ROWS * width; e += NTHREADS) {
        const int r = e / width, c = e % width;
        if (row0 + r < B) dst[(row0 + r) * width + c] = src[r * ld + c];
    }
}

// h0 = A0 W0 + b0 (kept raw in H0), t = relu(h0) Wa + ba (kept raw in T), h1 = h0 + relu(t) Wb + bb -> X1
// `bits` (nullable, the one-launch kernel): the ReLU decisions of h0 and t as one 64-bit ballot per (wave, tile, r) -
// 2 * 4 * NTWM * 4 words per 16-chain tile - for the reverse sweep (which then needs no activations at all).
// `act` (nullable): relu(h0) | relu(t) of the tile's rows are kept ([B][2 Wp], rows row0..) for k_spline_net_bwd, which
// then only needs their signs (the ReLU decisions) and skips this recomputation.
template <int NTWM, bool FAST = false>
__device__ __forceinline__ void sp_net_hidden(const SplineDims& f, const NetLds& l, const float* __restrict__ Lp, float* lds,
                                              const Tid& t, float* __restrict__ act = nullptr, long row0 = 0, long B = 0,
                                              unsigned long long* __restrict__ bits = nullptr) {
    constexpr int DW = depth_w<NTWM>();
    float* A0 = lds + l.o_A0; float* H0 = lds + l.o_H0; float* T = lds + l.o_T;
    float* X1 = lds + l.o_X1; float* X2 = lds + l.o_X2;
    f32x4 acc[NTWM];
    sp_gemm<NTWM, 2, true>(A0, l.AS, 4, reinterpret_cast<const float4*>(Lp + f.o_W0), Lp + f.o_b0, t, acc);
#pragma unroll
    for (int i = 0; i < NTWM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n;
            H0[o] = acc[i][r];
            X2[o] = acc[i][r] > 0.f ? acc[i][r] : 0.f;
            if (act && row0 + 4 * t.q + r < B)
                act[(row0 + 4 * t.q + r) * (2 * f.Wp) + 16 * (t.wave + 4 * i) + t.n] = X2[o];
            if (bits) {                                  // ReLU decisions of (rows 4q + r, tile wave + 4 i): one word per wave
                const unsigned long long m = __ballot(acc[i][r] > 0.f);
                if (t.lane == 0) bits[((0 * NWAVE + t.wave) * NTWM + i) * 4 + r] = m;
            }
        }
    __syncthreads();
    if constexpr (FAST) sp_gemm_bf16<NTWM, true>(f, X2, l.WS, Lp, 0, Lp + f.o_ba, t, acc);
    else sp_gemm<NTWM, DW, true>(X2, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_Wa), Lp + f.o_ba, t, acc);
#pragma unroll
    for (int i = 0; i < NTWM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n;
            T[o] = acc[i][r];
            X1[o] = acc[i][r] > 0.f ? acc[i][r] : 0.f;
            if (act && row0 + 4 * t.q + r < B)
                act[(row0 + 4 * t.q + r) * (2 * f.Wp) + f.Wp + 16 * (t.wave + 4 * i) + t.n] = X1[o];
            if (bits) {
                const unsigned long long m = __ballot(acc[i][r] > 0.f);
                if (t.lane == 0) bits[((1 * NWAVE + t.wave) * NTWM + i) * 4 + r] = m;
            }
        }
    __syncthreads();
    if constexpr (FAST) sp_gemm_bf16<NTWM, true>(f, X1, l.WS, Lp, 1, Lp + f.o_bb, t, acc);
    else sp_gemm<NTWM, DW, true>(X1, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_Wb), Lp + f.o_bb, t, acc);
    __syncthreads();                                   // everybody is done reading X1 (relu(t)) before it is overwritten
#pragma unroll
    for (int i = 0; i < NTWM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n;
            X1[o] = H0[o] + acc[i][r];
        }
    __syncthreads();
}

template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_spline_net_fwd(SplineDims f, NetLds l, const float* __restrict__ packed,
                                                             int layer, const float* __restrict__ Z,
                                                             float* __restrict__ P, long B, float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid t;
    constexpr int DW = depth_w<NTWM>();
    const float* Lp = packed + (size_t)layer * f.layer_stride;
    const long row0 = (long)blockIdx.x * ROWS;
    sp_load_identity(f, Lp, Z, row0, B, lds + l.o_A0, l.AS, t);
    __syncthreads();
    sp_net_hidden<NTWM>(f, l, Lp, lds, t, act, row0, B);
    const float* X1 = lds + l.o_X1;
    const int per = 4 * NTWM * f.KBW * 256;
    for (int c = 0; c < f.NCH; ++c) {
        f32x4 acc[NTWM];
        sp_gemm<NTWM, DW, true>(X1, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_Wf + (size_t)c * per),
                                Lp + f.o_bf + c * f.Wp, t, acc);
#pragma unroll
        for (int i = 0; i < NTWM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long g = row0 + 4 * t.q + r;
                if (g < B) P[g * f.NFP + c * f.Wp + 16 * (t.wave + 4 * i) + t.n] = acc[i][r];
            }
    }
}

// backward of the conditioner: dP [B][NFP] -> contribution to d log q / d(identity coordinates), ADDED into G [B][D]
template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_spline_net_bwd(SplineDims f, NetLds l, const float* __restrict__ packed,
                                                             int layer, const float* __restrict__ Z,
                                                             const float* __restrict__ dP, float* __restrict__ G, long B,
                                                             SplineTape tp, const float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid t;
    constexpr int DW = depth_w<NTWM>();
    const float* Lp = packed + (size_t)layer * f.layer_stride;
    const float* meta = Lp + f.o_meta;
    const long row0 = (long)blockIdx.x * ROWS;
    float* tl = tp.base ? tp.base + (size_t)layer * tp.layer_stride : nullptr;
    float* A0 = lds + l.o_A0; float* H0 = lds + l.o_H0; float* T = lds + l.o_T;
    float* X1 = lds + l.o_X1; float* X2 = lds + l.o_X2; float* DP = lds + l.o_DP; float* PART = lds + l.o_PART;
    if (!act) sp_load_identity(f, Lp, Z, row0, B, A0, l.AS, t);
    sp_tile_load4(DP, l.PS, dP, f.NFP, f.NFP, row0, B, t);
    if (act) {                                          // the forward kept relu(h0) | relu(t): only their signs matter here
        sp_tile_load4(H0, l.WS, act, 2 * f.Wp, f.Wp, row0, B, t);
        sp_tile_load4(T, l.WS, act + f.Wp, 2 * f.Wp, f.Wp, row0, B, t);
    }
    __syncthreads();
    if (!act) sp_net_hidden<NTWM>(f, l, Lp, lds, t);    // recompute h0 (H0) and t (T): the ReLU decisions
    if (tl) {
        sp_tape_rows(tl + tp.o_A0, 64, A0, l.AS, row0, B, t);
        sp_tape_rows(tl + tp.o_R0, f.Wp, X2, l.WS, row0, B, t);
        sp_tape_rows(tl + tp.o_H1, f.Wp, X1, l.WS, row0, B, t);
        for (int e = t.tid; e < ROWS * f.Wp; e += NTHREADS) {
            const int r = e / f.Wp, c = e % f.Wp;
            if (row0 + r < B) { const float v = T[r * l.WS + c]; tl[tp.o_R1 + (row0 + r) * f.Wp + c] = v > 0.f ? v : 0.f; }
        }
        __syncthreads();
    }
    f32x4 acc[NTWM];
    // dh1 = dP WfT  -> X1
    sp_gemm<NTWM, DW, false>(DP, l.PS, f.NFP / 16, reinterpret_cast<const float4*>(Lp + f.o_WfT), nullptr, t, acc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NTWM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) X1[(4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n] = acc[i][r];
    __syncthreads();
    if (tl) sp_tape_rows(tl + tp.o_dH1, f.Wp, X1, l.WS, row0, B, t);
    // d relu(t) = dh1 WbT, masked by t > 0 -> X2
    sp_gemm<NTWM, DW, false>(X1, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_WbT), nullptr, t, acc);
#pragma unroll
    for (int i = 0; i < NTWM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n;
            X2[o] = T[o] > 0.f ? acc[i][r] : 0.f;
        }
    __syncthreads();
    if (tl) sp_tape_rows(tl + tp.o_dT, f.Wp, X2, l.WS, row0, B, t);
    // dh0 = dh1 + (dt WaT) masked by h0 > 0 -> T
    sp_gemm<NTWM, DW, false>(X2, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_WaT), nullptr, t, acc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NTWM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n;
            T[o] = X1[o] + (H0[o] > 0.f ? acc[i][r] : 0.f);
        }
    __syncthreads();
    if (tl) sp_tape_rows(tl + tp.o_dH0, f.Wp, T, l.WS, row0, B, t);
    // dA0 = dh0 W0T (N = 64: K-split over the waves)
    gemm_ksplit<NTWM>(T, l.WS, reinterpret_cast<const float4*>(Lp + f.o_W0T), 4, PART, l.AS, t);
    __syncthreads();
    const int n_id = (int)meta[M_CNT * 64];
    for (int e = t.tid; e < ROWS * 64; e += NTHREADS) {
        const int r = e >> 6, i = e & 63;
        const long g = row0 + r;
        if (tl && g < B && i >= n_id) { tl[tp.o_XI + g * 64 + i] = 0.f; tl[tp.o_dA0 + g * 64 + i] = 0.f; }
        if (i < n_id && g < B) {
            float d = part_sum(PART, l.AS, r, i);
            const int feat = (int)meta[M_IDF * 64 + i];
            if (tl) { tl[tp.o_XI + g * 64 + i] = Z[g * f.D + feat]; tl[tp.o_dA0 + g * 64 + i] = d; }
            if (meta[M_PFON * 64 + i] != 0.f) {
                const int k = (int)meta[M_PFK * 64 + i];
                const float s = meta[M_PFS * 64 + i], x = Z[g * f.D + feat];
                d = d * (s * (Lp[f.o_pfw + 2 * k] * cosf(s * x) - Lp[f.o_pfw + 2 * k + 1] * sinf(s * x)));
            }
            G[g * f.D + feat] += d;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rational-quadratic spline (Durkan et al. 2019, eqs. 4-8; nflows / normflows rational_quadratic_spline)
// ------------------------------------------------------------------------------------------------
// The spline arithmetic below uses the hardware transcendental forms (__expf = v_exp_f32 on a scaled argument, __logf =
// v_log_f32 scaled: ~1-2 ulp) - the dependent exp / log chains of a coordinate are what bounds the element-wise part
// of the kernels, and 2e-7 relative is far inside the 1e-4 parity bar (softplus keeps log1pf where exp(u) is tiny).
// Round 5: quotients by a shared denominator are products with its hardware reciprocal (v_rcp_f32, 1 ulp; every
// denominator here is a softmax sum >= 1, a bin width / height >= 2 tb MIN, or a positive rational-quadratic term): an IEEE
// fp32 division is ~10 dependent instructions, a coordinate had 17 in the forward and 29 in the reverse pass - 2 + 2 and 2 + 6
// reciprocals now.  Shared by every spline kernel (16- / 8-chain tiles, sampling, tape), so they stay bit-compatible.
__device__ __forceinline__ float sp_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
struct Rqs {
    float cw[SP_K + 1], ch[SP_K + 1];                    // knot positions / values
    float ud[SP_K + 1];                                  // effective unnormalised knot derivatives (end knots fixed / tied)
    float pw[SP_K], ph[SP_K];                            // softmax probabilities (backward)
    bool circ;
};
// derivative min + softplus(u) of knot j and (backward) its d/du = sigmoid(u), 0 for a fixed end knot - evaluated only
// for the two knots of the bin that holds x (9 softplus + 9 sigmoid per coordinate otherwise)
__device__ __forceinline__ float sp_softplus(float x);
__device__ __forceinline__ void rqs_knot_pair(const Rqs& s, int b, float& d0, float& d1, float* sg0, float* sg1) {
    float u0 = s.ud[0], u1 = s.ud[1];
#pragma unroll
    for (int j = 1; j < SP_K; ++j)
        if (j == b) { u0 = s.ud[j]; u1 = s.ud[j + 1]; }
    d0 = SP_MIN_D + sp_softplus(u0);
    d1 = SP_MIN_D + sp_softplus(u1);
    if (sg0) {
        const bool f0 = !s.circ && b == 0, f1 = !s.circ && b == SP_K - 1;
        *sg0 = f0 ? 0.f : sp_rcp(1.f + __expf(-u0));
        *sg1 = f1 ? 0.f : sp_rcp(1.f + __expf(-u1));
    }
}

__device__ __forceinline__ float sp_softplus(float x) { return x > 20.f ? x : (x < -5.f ? log1pf(expf(x)) : __logf(1.f + __expf(x))); }

// p[0..K) widths, [K..2K) heights (both already divided by sqrt(hidden) when conditional), [2K..3K] derivatives
__device__ __forceinline__ void rqs_setup(const float* p, bool circ, float tb, Rqs& s) {
    float mw = p[0], mh = p[SP_K];
#pragma unroll
    for (int j = 1; j < SP_K; ++j) { mw = fmaxf(mw, p[j]); mh = fmaxf(mh, p[SP_K + j]); }
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int j = 0; j < SP_K; ++j) { s.pw[j] = __expf(p[j] - mw); sw += s.pw[j]; s.ph[j] = __expf(p[SP_K + j] - mh); sh += s.ph[j]; }
    float cw = 0.f, chh = 0.f;
    s.cw[0] = -tb; s.ch[0] = -tb;
    const float isw = sp_rcp(sw), ish = sp_rcp(sh);
#pragma unroll
    for (int j = 0; j < SP_K; ++j) {
        s.pw[j] = s.pw[j] * isw; s.ph[j] = s.ph[j] * ish;
        cw += SP_MIN_W + (1.f - SP_MIN_W * SP_K) * s.pw[j];
        chh += SP_MIN_H + (1.f - SP_MIN_H * SP_K) * s.ph[j];
        s.cw[j + 1] = (2.f * tb) * cw + (-tb);
        s.ch[j + 1] = (2.f * tb) * chh + (-tb);
    }
    s.cw[SP_K] = tb; s.ch[SP_K] = tb;
    const float cst = __logf(__expf(1.f - SP_MIN_D) - 1.f);
    s.circ = circ;
#pragma unroll
    for (int j = 0; j <= SP_K; ++j) {
        float u = p[2 * SP_K + j];
        if (!circ && (j == 0 || j == SP_K)) u = cst;
        if (circ && j == SP_K) u = p[2 * SP_K];
        s.ud[j] = u;
    }
}

__device__ __forceinline__ int rqs_bin(const float* knots, float x) {
    int b = 0;
#pragma unroll
    for (int j = 1; j < SP_K; ++j) b += (x >= knots[j]) ? 1 : 0;
    return b;                                             // clamp(sum(x >= knots (last + eps)) - 1, 0, K - 1), x in [-B, B]
}

// y = f(x), logabsdet; identity outside [-tb, tb]
__device__ __forceinline__ void rqs_forward(const Rqs& s, float x, float tb, float& y, float& ld) {
    if (!(x >= -tb && x <= tb)) { y = x; ld = 0.f; return; }
    const int b = rqs_bin(s.cw, x);
    float xk = 0.f, w = 1.f, yk = 0.f, h = 1.f, d0, d1;
#pragma unroll
    for (int j = 0; j < SP_K; ++j)
        if (j == b) { xk = s.cw[j]; w = s.cw[j + 1] - s.cw[j]; yk = s.ch[j]; h = s.ch[j + 1] - s.ch[j]; }
    rqs_knot_pair(s, b, d0, d1, nullptr, nullptr);
    const float iw = sp_rcp(w);
    const float th = (x - xk) * iw, t1 = th * (1.f - th), dl = h * iw;
    const float num = h * (dl * (th * th) + d0 * t1);
    const float den = dl + (d0 + d1 - 2.f * dl) * t1;
    y = yk + num * sp_rcp(den);
    const float dn = (dl * dl) * (d1 * (th * th) + 2.f * dl * t1 + d0 * ((1.f - th) * (1.f - th)));
    ld = __logf(dn) - 2.f * __logf(den);
}

// x = f^-1(y), logabsdet of the INVERSE map
__device__ __forceinline__ void rqs_inverse(const Rqs& s, float y, float tb, float& x, float& ld) {
    if (!(y >= -tb && y <= tb)) { x = y; ld = 0.f; return; }
    const int b = rqs_bin(s.ch, y);
    float xk = 0.f, w = 1.f, yk = 0.f, h = 1.f, d0, d1;
#pragma unroll
    for (int j = 0; j < SP_K; ++j)
        if (j == b) { xk = s.cw[j]; w = s.cw[j + 1] - s.cw[j]; yk = s.ch[j]; h = s.ch[j + 1] - s.ch[j]; }
    rqs_knot_pair(s, b, d0, d1, nullptr, nullptr);
    const float dl = h * sp_rcp(w), dy = y - yk, A = d0 + d1 - 2.f * dl;
    const float a = dy * A + h * (dl - d0);
    const float bq = h * d0 - dy * A;
    const float c = -dl * dy;
    const float disc = bq * bq - 4.f * a * c;
    const float root = (2.f * c) * sp_rcp(-bq - sqrtf(disc));
    x = root * w + xk;
    const float t1 = root * (1.f - root);
    const float den = dl + A * t1;
    const float dn = (dl * dl) * (d1 * (root * root) + 2.f * dl * t1 + d0 * ((1.f - root) * (1.f - root)));
    ld = -(__logf(dn) - 2.f * __logf(den));
}

// reverse mode of rqs_forward with cotangents (gy, 1 on logabsdet): returns d/dx and, if dp != nullptr, the cotangents
// of the 3K+1 unnormalised parameters (scaled by `wh_scale` for widths / heights).  LD = false: cotangents (gy, 0) - the map
// alone, without its log-derivative (k_spline_vsweep)
template <bool LD = true>
__device__ __forceinline__ float rqs_backward(const Rqs& s, const float* p, bool circ, float x, float tb, float gy,
                                              float wh_scale, float* dp) {
    if (dp) {
#pragma unroll
        for (int j = 0; j < SP_NP; ++j) dp[j] = 0.f;
    }
    if (!(x >= -tb && x <= tb)) return gy;
    const int b = rqs_bin(s.cw, x);
    float xk = 0.f, w = 1.f, h = 1.f, d0, d1, sg0 = 0.f, sg1 = 0.f;
#pragma unroll
    for (int j = 0; j < SP_K; ++j)
        if (j == b) { xk = s.cw[j]; w = s.cw[j + 1] - s.cw[j]; h = s.ch[j + 1] - s.ch[j]; }
    rqs_knot_pair(s, b, d0, d1, dp ? &sg0 : nullptr, dp ? &sg1 : nullptr);
    const float iw = sp_rcp(w), ih = sp_rcp(h);
    const float th = (x - xk) * iw, t1 = th * (1.f - th), dl = h * iw, A = d0 + d1 - 2.f * dl;
    const float num = h * (dl * (th * th) + d0 * t1);
    const float den = dl + A * t1;
    const float e = d1 * (th * th) + 2.f * dl * t1 + d0 * ((1.f - th) * (1.f - th));
    const float iden = sp_rcp(den);
    const float nb = gy * iden;                                    // cotangent of num
    const float db = LD ? -gy * num * (iden * iden) - 2.f * iden : -gy * num * (iden * iden);       // of den
    const float eb = LD ? sp_rcp(e) : 0.f;                         // of e
    const float sb = (LD ? 2.f * (w * ih) : 0.f) + eb * 2.f * t1 + db * (1.f - 2.f * t1) + nb * h * (th * th);   // (2 / dl = 2 w / h)
    const float t1b = eb * 2.f * dl + db * A + nb * h * d0;
    const float d0b = eb * ((1.f - th) * (1.f - th)) + db * t1 + nb * h * t1;
    const float d1b = eb * (th * th) + db * t1;
    const float thb = eb * (2.f * d1 * th - 2.f * d0 * (1.f - th)) + nb * 2.f * h * dl * th + t1b * (1.f - 2.f * th);
    const float hb = nb * (num * ih) + sb * iw;
    const float wb = -sb * h * (iw * iw) - thb * th * iw;
    const float xb = thb * iw;
    if (dp) {
        // knots: W_b += -xb - wb, W_{b+1} += wb ; H_b += gy - hb, H_{b+1} += hb ; D_b += d0b, D_{b+1} += d1b
        const float cWb = -xb - wb, cW1 = wb, cHb = gy - hb, cH1 = hb;
        // W_j = -tb + 2 tb (j MIN + (1 - K MIN) P_j), P_j = sum_{i<j} p_i  ->  d uw_m = c p_m sum_j cW_j (1[m < j] - P_j)
        const float cw_ = 2.f * tb * (1.f - SP_MIN_W * SP_K), chh = 2.f * tb * (1.f - SP_MIN_H * SP_K);
        float Pw = 0.f, Ph = 0.f, Pwb = 0.f, Pw1 = 0.f, Phb = 0.f, Ph1 = 0.f;
#pragma unroll
        for (int j = 0; j <= SP_K; ++j) {
            if (j == b) { Pwb = Pw; Phb = Ph; }
            if (j == b + 1) { Pw1 = Pw; Ph1 = Ph; }
            if (j < SP_K) { Pw += s.pw[j]; Ph += s.ph[j]; }
        }
        const bool fb = b >= 1, f1 = b + 1 <= SP_K - 1;           // the end knots W_0, W_K are fixed
#pragma unroll
        for (int m = 0; m < SP_K; ++m) {
            float aw = 0.f, ah = 0.f;
            if (fb) { aw += cWb * ((m < b ? 1.f : 0.f) - Pwb); ah += cHb * ((m < b ? 1.f : 0.f) - Phb); }
            if (f1) { aw += cW1 * ((m < b + 1 ? 1.f : 0.f) - Pw1); ah += cH1 * ((m < b + 1 ? 1.f : 0.f) - Ph1); }
            dp[m] = cw_ * s.pw[m] * aw * wh_scale;
            dp[SP_K + m] = chh * s.ph[m] * ah * wh_scale;
        }
#pragma unroll
        for (int j = 0; j <= SP_K; ++j) {
            float g = 0.f;
            if (j == b) g += d0b * sg0;
            if (j == b + 1) g += d1b * sg1;
            if (circ && j == SP_K) { dp[2 * SP_K] += g; g = 0.f; }            // last knot tied to the first (same u)
            dp[2 * SP_K + j] += g;
        }
    }
    return xb;
}

__device__ __forceinline__ float sp_wrap(float z, float bound) {       // torch.remainder(z + b, 2 b) - b
    const float p = 2.f * bound;
    float r = fmodf(z + bound, p);
    if (r < 0.f) r += p;
    return r - bound;
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// One wave per chain, lane = coordinate.  MODE 0: log_prob direction (evaluate), 1: sampling direction, identity
// coordinates only (unconditional inverse, before the conditioner runs), 2: sampling direction, transformed
// coordinates (needs P) + post shift (Ypre, nullable: the state before that shift, kept for fabhip_spline_sample_vjp_tape).
template <int MODE>
__global__ __launch_bounds__(256) void k_spline_apply(SplineDims f, const float* __restrict__ packed, int layer,
                                                      const float* __restrict__ Zin, const float* __restrict__ P,
                                                      float* __restrict__ Zout, float* __restrict__ log_q, float ld_sign,
                                                      long B, float* __restrict__ Ypre) {
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;
    const float* Lp = packed + (size_t)layer * f.layer_stride;
    const float* meta = Lp + f.o_meta;
    const int n_id = (int)meta[M_CNT * 64], n_tr = (int)meta[M_CNT * 64 + 1];
    const float isq = 1.f / sqrtf((float)f.W);
    float ld = 0.f;
    if (lane < f.D) {
        // role of this coordinate
        const int pos_id = (int)meta[M_POSID * 64 + lane], pos_tr = (int)meta[M_POSTR * 64 + lane];
        const bool circ = meta[M_CIRC * 64 + lane] != 0.f;
        const float tb = meta[M_TB * 64 + lane];
        float z = Zin[g * f.D + lane], out = z;
        float p[SP_NP];
        Rqs s;
        if (pos_id >= 0 && MODE != 2) {
#pragma unroll
            for (int j = 0; j < SP_NP; ++j) p[j] = Lp[f.o_unc + pos_id * SP_NP + j];
            rqs_setup(p, circ, tb, s);
            float l1;
            if (MODE == 0) rqs_forward(s, z, tb, out, l1); else rqs_inverse(s, z, tb, out, l1);
            ld = l1;
        } else if (pos_tr >= 0 && MODE != 1) {
#pragma unroll
            for (int j = 0; j < SP_NP; ++j) {
                const float v = P[g * f.NFP + pos_tr * SP_NP + j];
                p[j] = j < 2 * SP_K ? v * isq : v;
            }
            rqs_setup(p, circ, tb, s);
            float l1;
            if (MODE == 0) rqs_forward(s, z, tb, out, l1); else rqs_inverse(s, z, tb, out, l1);
            ld = l1;
        }
        // stage boundary: MODE 0 -> the NEXT (lower) layer's pre-shift; MODE 2 -> this layer's post-shift
        if (MODE == 0 && layer > 0) {
            const float* mn = packed + (size_t)(layer - 1) * f.layer_stride + f.o_meta;
            if (mn[M_PREON * 64 + lane] != 0.f) out = sp_wrap(out - mn[M_PRESH * 64 + lane], tb);
        }
        if (MODE == 2 && Ypre) Ypre[g * f.D + lane] = out;         // the layer's output before the shift = the log_prob direction's input
        if (MODE == 2 && meta[M_POSTON * 64 + lane] != 0.f) out = sp_wrap(out + meta[M_POSTSH * 64 + lane], tb);
        Zout[g * f.D + lane] = out;
    }
    ld = wave_sum64(ld);
    if (lane == 0) log_q[g] += ld_sign * ld;
}

// x <- wrap(x - pre_shift of the top layer) : the first stage of the log_prob direction (PeriodicWrap.inverse)
__global__ void k_spline_first_stage(SplineDims f, const float* __restrict__ packed, const float* __restrict__ x,
                                     float* __restrict__ Z, float* __restrict__ log_q, long B) {
    const float* meta = packed + (size_t)(f.L - 1) * f.layer_stride + f.o_meta;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < B * f.D; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e % f.D);
        float v = x[e];
        if (meta[M_PREON * 64 + j] != 0.f) v = sp_wrap(v - meta[M_PRESH * 64 + j], meta[M_TB * 64 + j]);
        Z[e] = v;
        if (j == 0) log_q[e / f.D] = 0.f;
    }
}

// base UniformGaussian: log_q += log p0(z); G = d log p0 / dz (if G)
__global__ __launch_bounds__(256) void k_spline_base(SplineDims f, const float* __restrict__ packed,
                                                     const float* __restrict__ Z, float* __restrict__ log_q,
                                                     float* __restrict__ G, long B) {
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;
    float lp = 0.f;
    if (lane < f.D) {
        const float sc = packed[f.o_base + lane];
        const bool circ = packed[f.o_base + 64 + lane] != 0.f;
        const float z = Z[g * f.D + lane];
        if (circ) { lp = -logf(sc); if (G) G[g * f.D + lane] = 0.f; }
        else {
            lp = -0.5f * 1.8378770664093453f - logf(sc) - 0.5f * ((z / sc) * (z / sc));
            if (G) G[g * f.D + lane] = -(z / sc) / sc;
        }
    }
    lp = wave_sum64(lp);
    if (lane == 0) log_q[g] += lp;
}

// sampling: z0 = base sample from (u, eps), log_q = log p0(z0)
__global__ __launch_bounds__(256) void k_spline_base_sample(SplineDims f, const float* __restrict__ packed,
                                                            const float* __restrict__ u, const float* __restrict__ eps,
                                                            float* __restrict__ Z, float* __restrict__ log_q, long B) {
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;
    float lp = 0.f;
    if (lane < f.D) {
        const float sc = packed[f.o_base + lane];
        const bool circ = packed[f.o_base + 64 + lane] != 0.f;
        const float z = (circ ? u[g * f.D + lane] - 0.5f : eps[g * f.D + lane]) * sc;
        Z[g * f.D + lane] = z;
        lp = circ ? -logf(sc) : -0.5f * 1.8378770664093453f - logf(sc) - 0.5f * ((z / sc) * (z / sc));
    }
    lp = wave_sum64(lp);
    if (lane == 0) log_q[g] = lp;
}

// reverse mode through k_spline_apply<0> of one layer: Gout (cotangent of the layer's output state, BEFORE the stage
// boundary shift, which has unit derivative) -> Gin (w.r.t. the layer's input state, conditioner path excluded) and dP
__global__ __launch_bounds__(256) void k_spline_apply_bwd(SplineDims f, const float* __restrict__ packed, int layer,
                                                          const float* __restrict__ Zin, const float* __restrict__ P,
                                                          const float* __restrict__ Gout, float* __restrict__ Gin,
                                                          float* __restrict__ dP, long B, float* __restrict__ dU) {
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;
    const float* Lp = packed + (size_t)layer * f.layer_stride;
    const float* meta = Lp + f.o_meta;
    const int n_id = (int)meta[M_CNT * 64], n_tr = (int)meta[M_CNT * 64 + 1];
    const float isq = 1.f / sqrtf((float)f.W);
    // zero the padding columns of this chain's dP row (the conditioner backward reads all NFP of them)
    for (int j = n_tr * SP_NP + lane; j < f.NFP; j += 64) dP[g * f.NFP + j] = 0.f;
    if (lane >= f.D) return;
    const int pos_id = (int)meta[M_POSID * 64 + lane], pos_tr = (int)meta[M_POSTR * 64 + lane];
    const bool circ = meta[M_CIRC * 64 + lane] != 0.f;
    const float tb = meta[M_TB * 64 + lane];
    const float z = Zin[g * f.D + lane], gy = Gout[g * f.D + lane];
    float p[SP_NP];
    Rqs s;
    float gx = gy;
    if (pos_id >= 0) {
#pragma unroll
        for (int j = 0; j < SP_NP; ++j) p[j] = Lp[f.o_unc + pos_id * SP_NP + j];
        rqs_setup(p, circ, tb, s);
        if (dU) {
            float dp[SP_NP];
            gx = rqs_backward(s, p, circ, z, tb, gy, 1.f, dp);
#pragma unroll
            for (int j = 0; j < SP_NP; ++j) dU[g * (SP_MD * SP_NP) + pos_id * SP_NP + j] = dp[j];
        } else {
            gx = rqs_backward(s, p, circ, z, tb, gy, 1.f, nullptr);
        }
    } else if (pos_tr >= 0) {
#pragma unroll
        for (int j = 0; j < SP_NP; ++j) {
            const float v = P[g * f.NFP + pos_tr * SP_NP + j];
            p[j] = j < 2 * SP_K ? v * isq : v;
        }
        rqs_setup(p, circ, tb, s);
        float dp[SP_NP];
        gx = rqs_backward(s, p, circ, z, tb, gy, isq, dp);
#pragma unroll
        for (int j = 0; j < SP_NP; ++j) dP[g * f.NFP + pos_tr * SP_NP + j] = dp[j];
    }
    Gin[g * f.D + lane] = gx;
}

// Sampling-direction gradients (x = S^-1(z0; theta) with the noise z0 fixed, S = the log_prob direction): for a cotangent gx of
// x and gl of log q(x), d/d theta = gl d log q / d theta |_x  -  v^T dS / d theta with v = (dS / dx)^-T (gx + gl d log q / dx)
// (implicit function theorem).  The first term is the density tape's; this kernel carries v through ONE coupling layer
// y -> z = (g_u(y_id), g(y_tr; c(y_id))) from the x side to the base side and leaves  -v_z dz / d(parameters)  where the
// density tape keeps its parameter cotangents, so the SAME conditioner backward and the SAME tape GEMMs finish the job:
//   PART 0 (before the conditioner backward): transformed coordinates  v_z = v_y / g'(y),  dP = -v_z dg / dP;  identity
//          coordinates copied (the conditioner backward then ADDS  -A^T v_z,tr  to them, A = dg / dy_id through the net)
//   PART 1 (after it): identity coordinates  v_z = (v_y - A^T v_z,tr) / g_u'(y),  dU = -v_z dg_u / dU
// v = gx + gl * (d log q / dx), in place in v (gx / gl nullable: a cotangent that is not there)
__global__ void k_spline_vjp_seed(float* __restrict__ v, const float* __restrict__ gx, const float* __restrict__ gl, long B, int D) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < B * D; e += (long)gridDim.x * blockDim.x)
        v[e] = (gx ? gx[e] : 0.f) + (gl ? gl[e / D] * v[e] : 0.f);
}

template <int PART>
__global__ __launch_bounds__(256) void k_spline_vsweep(SplineDims f, const float* __restrict__ packed, int layer,
                                                       const float* __restrict__ Zin, const float* __restrict__ P,
                                                       const float* __restrict__ Vin, float* __restrict__ Vout,
                                                       float* __restrict__ dP, long B, float* __restrict__ dU) {
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;
    const float* Lp = packed + (size_t)layer * f.layer_stride;
    const float* meta = Lp + f.o_meta;
    const int n_tr = (int)meta[M_CNT * 64 + 1];
    const float isq = 1.f / sqrtf((float)f.W);
    if (PART == 0)
        for (int j = n_tr * SP_NP + lane; j < f.NFP; j += 64) dP[g * f.NFP + j] = 0.f;
    if (lane >= f.D) return;
    const int pos_id = (int)meta[M_POSID * 64 + lane], pos_tr = (int)meta[M_POSTR * 64 + lane];
    const bool circ = meta[M_CIRC * 64 + lane] != 0.f;
    const float tb = meta[M_TB * 64 + lane];
    const float y = Zin[g * f.D + lane];
    float p[SP_NP], dp[SP_NP];
    Rqs s;
    if (PART == 0) {
        float v = Vin[g * f.D + lane];
        if (pos_tr >= 0) {
#pragma unroll
            for (int j = 0; j < SP_NP; ++j) {
                const float c = P[g * f.NFP + pos_tr * SP_NP + j];
                p[j] = j < 2 * SP_K ? c * isq : c;
            }
            rqs_setup(p, circ, tb, s);
            v = v / rqs_backward<false>(s, p, circ, y, tb, 1.f, isq, nullptr);
            rqs_backward<false>(s, p, circ, y, tb, -v, isq, dp);
#pragma unroll
            for (int j = 0; j < SP_NP; ++j) dP[g * f.NFP + pos_tr * SP_NP + j] = dp[j];
        }
        Vout[g * f.D + lane] = v;
    } else if (pos_id >= 0) {
#pragma unroll
        for (int j = 0; j < SP_NP; ++j) p[j] = Lp[f.o_unc + pos_id * SP_NP + j];
        rqs_setup(p, circ, tb, s);
        const float v = Vout[g * f.D + lane] / rqs_backward<false>(s, p, circ, y, tb, 1.f, 1.f, nullptr);
        rqs_backward<false>(s, p, circ, y, tb, -v, 1.f, dp);
#pragma unroll
        for (int j = 0; j < SP_NP; ++j) dU[g * (SP_MD * SP_NP) + pos_id * SP_NP + j] = dp[j];
        Vout[g * f.D + lane] = v;
    }
}


// ------------------------------------------------------------------------------------------------
// ONE launch for log q (+ d log q / dx): a workgroup carries its 16 chains through all layers - conditioner on the
// matrix cores, its 3K+1 parameters per coordinate kept in LDS, the splines evaluated by the same workgroup (thread =
// (chain, coordinate mod 16)), the state tile never leaving LDS - then, with GRAD, back again (layer inputs, conditioner
// outputs and ReLU decisions of the forward sweep are parked in the workspace, per tile: L2-resident).  Replaces 4 L + 2
// launches of the per-stage kernels above (kept for the training tape and the sampling direction).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sp_identity_from_tile(const SplineDims& f, const float* __restrict__ Lp, const float* ZT,
                                                      float* A0, int AS, const Tid& t) {
    const float* meta = Lp + f.o_meta;
    const int n_id = (int)meta[M_CNT * 64];
    for (int e = t.tid; e < ROWS * AS; e += NTHREADS) {
        const int r = e / AS, i = e % AS;
        float v = 0.f;
        if (i < n_id) {
            v = ZT[r * 64 + (int)meta[M_IDF * 64 + i]];
            if (meta[M_PFON * 64 + i] != 0.f) {
                const int k = (int)meta[M_PFK * 64 + i];
                const float sc = meta[M_PFS * 64 + i];
                v = Lp[f.o_pfw + 2 * k] * sinf(sc * v) + Lp[f.o_pfw + 2 * k + 1] * cosf(sc * v);
            }
        }
        A0[e] = v;
    }
}

// parameters of coordinate j of chain row r in this layer: returns 0 (untouched), 1 (identity: unconditional), 2 (transformed)
__device__ __forceinline__ int sp_coord_params(const SplineDims& f, const float* __restrict__ Lp, const float* meta,
                                               const float* PT, int PS, int r, int j, float isq, float (&p)[SP_NP],
                                               int& pos) {
    const int pos_id = (int)meta[M_POSID * 64 + j], pos_tr = (int)meta[M_POSTR * 64 + j];
    if (pos_id >= 0) {
#pragma unroll
        for (int k = 0; k < SP_NP; ++k) p[k] = Lp[f.o_unc + pos_id * SP_NP + k];
        pos = pos_id;
        return 1;
    }
    if (pos_tr >= 0) {
#pragma unroll
        for (int k = 0; k < SP_NP; ++k) {
            const float v = PT[r * PS + pos_tr * SP_NP + k];
            p[k] = k < 2 * SP_K ? v * isq : v;
        }
        pos = pos_tr;
        return 2;
    }
    pos = -1;
    return 0;
}

template <int NTWM, bool GRAD, bool FAST>
__device__ __forceinline__ void spline_logprob_body(const SplineDims& f, const NetLds& l, const float* __restrict__ packed,
                                                    const float* __restrict__ x, float* __restrict__ log_q,
                                                    float* __restrict__ grad_x, long B, float* __restrict__ Zsave,
                                                    float* __restrict__ Psave, unsigned long long* __restrict__ bitsave, float* lds,
                                                    long long* tlp = nullptr) {
    Tid t;
#define SP_TL(idx) do { if (tlp && blockIdx.x == 0 && threadIdx.x == 0) tlp[idx] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    constexpr int DW = depth_w<NTWM>();
    const long row0 = (long)blockIdx.x * ROWS;
    float* A0 = lds + l.o_A0; float* H0 = lds + l.o_H0; float* T = lds + l.o_T;
    float* X1 = lds + l.o_X1; float* X2 = lds + l.o_X2; float* PT = lds + l.o_DP; float* PART = lds + l.o_PART;
    float* ZT = lds + l.o_ZT; float* GT = lds + l.o_GT;
    const float isq = 1.f / sqrtf((float)f.W);
    const int per = 4 * NTWM * f.KBW * 256;
    const size_t zs = (size_t)B * f.D, ps = (size_t)B * f.NFP;
    // x <- wrap(x - pre-shift of the top layer)   (PeriodicWrap.inverse / the last PeriodicShift.inverse)
    {
        const float* mt = packed + (size_t)(f.L - 1) * f.layer_stride + f.o_meta;
        for (int e = t.tid; e < ROWS * 64; e += NTHREADS) {
            const int r = e >> 6, j = e & 63;
            const long g = row0 + r;
            float v = 0.f;
            if (j < f.D && g < B) {
                v = x[g * f.D + j];
                if (mt[M_PREON * 64 + j] != 0.f) v = sp_wrap(v - mt[M_PRESH * 64 + j], mt[M_TB * 64 + j]);
            }
            ZT[e] = v;
        }
    }
    __syncthreads();
    float ld_acc = 0.f;
    for (int layer = f.L - 1; layer >= 0; --layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        const float* meta = Lp + f.o_meta;
        const bool tl = layer == 1;
        if (tl) SP_TL(0);
        if (GRAD) {                                        // this layer's input state, for the reverse sweep
            for (int e = t.tid; e < ROWS * f.D; e += NTHREADS) {
                const int r = e / f.D, j = e % f.D;
                if (row0 + r < B) Zsave[(size_t)layer * zs + (row0 + r) * f.D + j] = ZT[r * 64 + j];
            }
        }
        sp_identity_from_tile(f, Lp, ZT, A0, l.AS, t);
        __syncthreads();
        if (tl) SP_TL(1);
        sp_net_hidden<NTWM, FAST>(f, l, Lp, lds, t, nullptr, row0, B,
                                  GRAD ? bitsave + ((size_t)layer * gridDim.x + blockIdx.x) * (2 * NWAVE * NTWM * 4) : nullptr);
        if (tl) SP_TL(2);
        for (int c = 0; c < f.NCH; ++c) {
            f32x4 acc[NTWM];
            if constexpr (FAST) sp_gemm_bf16<NTWM, true>(f, X1, l.WS, Lp, 2 + c, Lp + f.o_bf + c * f.Wp, t, acc);
            else sp_gemm<NTWM, DW, true>(X1, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_Wf + (size_t)c * per),
                                         Lp + f.o_bf + c * f.Wp, t, acc);
#pragma unroll
            for (int i = 0; i < NTWM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = c * f.Wp + 16 * (t.wave + 4 * i) + t.n;
                    PT[(4 * t.q + r) * l.PS + col] = acc[i][r];
                }
        }
        __syncthreads();
        if (GRAD) sp_tile_store4(Psave + (size_t)layer * ps, f.NFP, PT, l.PS, f.NFP, row0, B, t);   // for the reverse sweep
        if (tl) SP_TL(3);
        const float* mn = layer > 0 ? packed + (size_t)(layer - 1) * f.layer_stride + f.o_meta : nullptr;
        for (int j = t.c; j < f.D; j += 16) {
            float p[SP_NP];
            int pos;
            const int kind = sp_coord_params(f, Lp, meta, PT, l.PS, t.row, j, isq, p, pos);
            const float tb = meta[M_TB * 64 + j];
            float out = ZT[t.row * 64 + j];
            if (kind) {
                Rqs sp;
                rqs_setup(p, meta[M_CIRC * 64 + j] != 0.f, tb, sp);
                float l1;
                rqs_forward(sp, out, tb, out, l1);
                ld_acc += l1;
            }
            if (mn && mn[M_PREON * 64 + j] != 0.f) out = sp_wrap(out - mn[M_PRESH * 64 + j], tb);   // next stage's shift
            ZT[t.row * 64 + j] = out;
        }
        __syncthreads();
        if (tl) SP_TL(4);
    }
    // base UniformGaussian
    for (int j = t.c; j < f.D; j += 16) {
        const float sc = packed[f.o_base + j];
        const float z = ZT[t.row * 64 + j];
        if (packed[f.o_base + 64 + j] != 0.f) { ld_acc += -logf(sc); if (GRAD) GT[t.row * 64 + j] = 0.f; }
        else {
            ld_acc += -0.5f * 1.8378770664093453f - logf(sc) - 0.5f * ((z / sc) * (z / sc));
            if (GRAD) GT[t.row * 64 + j] = -(z / sc) / sc;
        }
    }
    const float lq = row16_sum(ld_acc);
    if (t.c == 0 && row0 + t.row < B) log_q[row0 + t.row] = lq;
    if (!GRAD) return;
    __syncthreads();
    // ---- reverse sweep: GT = d log q / d(state), layers 0 .. L-1 ---------------------------------------------------
    for (int layer = 0; layer < f.L; ++layer) {
        const float* Lp = packed + (size_t)layer * f.layer_stride;
        const float* meta = Lp + f.o_meta;
        const int n_id = (int)meta[M_CNT * 64], n_tr = (int)meta[M_CNT * 64 + 1];
        const bool tl = layer == 1;
        if (tl) SP_TL(8);
        for (int e = t.tid; e < ROWS * 64; e += NTHREADS) {           // layer input state
            const int r = e >> 6, j = e & 63;
            ZT[e] = (j < f.D && row0 + r < B) ? Zsave[(size_t)layer * zs + (row0 + r) * f.D + j] : 0.f;
        }
        // conditioner output (-> dP in place below) and the ReLU decisions of the forward sweep
        sp_tile_load4(PT, l.PS, Psave + (size_t)layer * ps, f.NFP, f.NFP, row0, B, t);
        unsigned long long m0[NTWM][4], m1[NTWM][4];                       // this thread's ReLU decisions (wave-uniform words)
        {
            const unsigned long long* bw = bitsave + ((size_t)layer * gridDim.x + blockIdx.x) * (2 * NWAVE * NTWM * 4);
#pragma unroll
            for (int i = 0; i < NTWM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m0[i][r] = bw[((0 * NWAVE + t.wave) * NTWM + i) * 4 + r];
                    m1[i][r] = bw[((1 * NWAVE + t.wave) * NTWM + i) * 4 + r];
                }
        }
        __syncthreads();
        if (tl) SP_TL(9);
        for (int j = t.c; j < f.D; j += 16) {
            float p[SP_NP];
            int pos;
            const int kind = sp_coord_params(f, Lp, meta, PT, l.PS, t.row, j, isq, p, pos);
            if (kind) {
                const float tb = meta[M_TB * 64 + j];
                const bool circ = meta[M_CIRC * 64 + j] != 0.f;
                Rqs sp;
                rqs_setup(p, circ, tb, sp);
                const float z = ZT[t.row * 64 + j], gy = GT[t.row * 64 + j];
                if (kind == 2) {
                    float dp[SP_NP];
                    GT[t.row * 64 + j] = rqs_backward(sp, p, circ, z, tb, gy, isq, dp);
#pragma unroll
                    for (int k = 0; k < SP_NP; ++k) PT[t.row * l.PS + pos * SP_NP + k] = dp[k];
                } else {
                    GT[t.row * 64 + j] = rqs_backward(sp, p, circ, z, tb, gy, 1.f, nullptr);
                }
            }
        }
        for (int c = n_tr * SP_NP + t.c; c < l.PS; c += 16) PT[t.row * l.PS + c] = 0.f;    // padding columns of dP
        __syncthreads();
        if (tl) SP_TL(10);
        f32x4 acc[NTWM];
        if constexpr (FAST) {                              // K = NFP: NCH slices of K = Wp of the WfT image
#pragma unroll
            for (int i = 0; i < NTWM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const uint4* Bh = reinterpret_cast<const uint4*>(Lp + f.o_h + (size_t)(2 + f.NCH) * (f.Wp * f.Wp / 2));
            for (int c = 0; c < f.NCH; ++c) gemm_bf16_slice<NTWM>(PT, l.PS, Bh, f.NFP / 32, c * (f.Wp / 32), t, acc);
        } else {
            sp_gemm<NTWM, DW, false>(PT, l.PS, f.NFP / 16, reinterpret_cast<const float4*>(Lp + f.o_WfT), nullptr, t, acc);
        }
#pragma unroll
        for (int i = 0; i < NTWM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) X1[(4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n] = acc[i][r];
        __syncthreads();
        if (tl) SP_TL(11);
        if constexpr (FAST) sp_gemm_bf16<NTWM, false>(f, X1, l.WS, Lp, 2 + 2 * f.NCH, nullptr, t, acc);
        else sp_gemm<NTWM, DW, false>(X1, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_WbT), nullptr, t, acc);
#pragma unroll
        for (int i = 0; i < NTWM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n;
                X2[o] = ((m1[i][r] >> t.lane) & 1ull) ? acc[i][r] : 0.f;
            }
        __syncthreads();
        if constexpr (FAST) sp_gemm_bf16<NTWM, false>(f, X2, l.WS, Lp, 3 + 2 * f.NCH, nullptr, t, acc);
        else sp_gemm<NTWM, DW, false>(X2, l.WS, f.KBW, reinterpret_cast<const float4*>(Lp + f.o_WaT), nullptr, t, acc);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NTWM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (4 * t.q + r) * l.WS + 16 * (t.wave + 4 * i) + t.n;
                T[o] = X1[o] + (((m0[i][r] >> t.lane) & 1ull) ? acc[i][r] : 0.f);
            }
        __syncthreads();
        if (tl) SP_TL(12);
        gemm_ksplit<NTWM>(T, l.WS, reinterpret_cast<const float4*>(Lp + f.o_W0T), 4, PART, l.AS, t);
        __syncthreads();
        if (tl) SP_TL(13);
        for (int e = t.tid; e < ROWS * 64; e += NTHREADS) {
            const int r = e >> 6, i = e & 63;
            if (i < n_id) {
                float d = part_sum(PART, l.AS, r, i);
                const int feat = (int)meta[M_IDF * 64 + i];
                if (meta[M_PFON * 64 + i] != 0.f) {
                    const int k = (int)meta[M_PFK * 64 + i];
                    const float sc = meta[M_PFS * 64 + i], zz = ZT[r * 64 + feat];
                    d = d * (sc * (Lp[f.o_pfw + 2 * k] * cosf(sc * zz) - Lp[f.o_pfw + 2 * k + 1] * sinf(sc * zz)));
                }
                GT[r * 64 + feat] += d;
            }
        }
        __syncthreads();
        if (tl) SP_TL(14);
    }
    for (int e = t.tid; e < ROWS * f.D; e += NTHREADS) {
        const int r = e / f.D, j = e % f.D;
        if (row0 + r < B) grad_x[(row0 + r) * f.D + j] = GT[r * 64 + j];
    }
#undef SP_TL
}

#include "spline_r8.h"

template <int NTWM, bool GRAD>
__global__ __launch_bounds__(NTHREADS) void k_spline_logprob(SplineDims f, NetLds l, const float* __restrict__ packed,
                                                             const float* __restrict__ x, float* __restrict__ log_q,
                                                             float* __restrict__ grad_x, long B,
                                                             float* __restrict__ Zsave, float* __restrict__ Psave,
                                                             unsigned long long* __restrict__ bitsave, long long* tlp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    spline_logprob_body<NTWM, GRAD, false>(f, l, packed, x, log_q, grad_x, B, Zsave, Psave, bitsave, lds, tlp);
}
// fast mode (fabhip_set_fast_mode): the conditioner's Wp x Wp GEMMs on the bf16 matrix cores; gradient evaluations only
template <int NTWM>
__global__ __launch_bounds__(NTHREADS) void k_spline_logprob_fast(SplineDims f, NetLds l, const float* __restrict__ packed,
                                                                  const float* __restrict__ x, float* __restrict__ log_q,
                                                                  float* __restrict__ grad_x, long B,
                                                                  float* __restrict__ Zsave, float* __restrict__ Psave,
                                                                  unsigned long long* __restrict__ bitsave, long long* tlp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    spline_logprob_body<NTWM, true, true>(f, l, packed, x, log_q, grad_x, B, Zsave, Psave, bitsave, lds, tlp);
}

// dev-only stage timeline (FABHIP_TIMELINE=1): s_memtime stamps of workgroup 0 in layer 1 of k_spline_logprob
static long long* g_sp_timeline = nullptr;
static long long* sp_timeline(hipStream_t st) {
    if (!option(FABHIP_OPT_TIMELINE)) return nullptr;
    if (!g_sp_timeline && hipMalloc((void**)&g_sp_timeline, 32 * 8) != hipSuccess) return nullptr;
    hipMemsetAsync(g_sp_timeline, 0, 32 * 8, st);
    return g_sp_timeline;
}

template <int NTWM>
static int launch_logprob(const SplineDims& f, const float* packed, const float* x, float* log_q, float* grad_x, long B,
                          float* Zsave, float* Psave, unsigned long long* bitsave, int fast, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div((int)B, ROWS)), block(NTHREADS);
    const NetLds l = make_net_lds(f, grad_x != nullptr, true);
    const size_t bytes = (size_t)l.total * 4;
    if (grad_x && fast) {
        FAB_TRY(set_max_lds((const void*)k_spline_logprob_fast<NTWM>, bytes));
        hipLaunchKernelGGL((k_spline_logprob_fast<NTWM>), grid, block, bytes, st, f, l, packed, x, log_q, grad_x, B, Zsave,
                           Psave, bitsave, sp_timeline(st));
    } else if (grad_x) {
        FAB_TRY(set_max_lds((const void*)k_spline_logprob<NTWM, true>, bytes));
        hipLaunchKernelGGL((k_spline_logprob<NTWM, true>), grid, block, bytes, st, f, l, packed, x, log_q, grad_x, B, Zsave,
                           Psave, bitsave, sp_timeline(st));
    } else {
        FAB_TRY(set_max_lds((const void*)k_spline_logprob<NTWM, false>, bytes));
        hipLaunchKernelGGL((k_spline_logprob<NTWM, false>), grid, block, bytes, st, f, l, packed, x, log_q, grad_x, B, Zsave,
                           Psave, bitsave, sp_timeline(st));
    }
    return check_launch();
}

// 8-chain tiles (spline_r8.h): hidden width padded to 256, fp32 path.  FABHIP_OPT_TILE_SHAPE 16 / 8 (or 4) forces a shape;
// otherwise 8-chain tiles whenever 16-chain tiles would leave CUs without a workgroup.
// The 4x4x1 stream kernels (spline_r8.h; hidden width padded to 256, fp32 path): 4 / 8 chains per workgroup while that leaves
// no CU with more than one workgroup (the kernel's registers allow no second one: 1024 / 2048 chains on MI355X), 16 above.
// FABHIP_OPT_TILE_SHAPE 4 / 8 / 16 forces the tile; FABHIP_OPT_SPLINE_MFMA = 16 selects the 16x16x4 kernel
// (k_spline_logprob: the only one for other widths and for fast mode).  Returns row blocks (0: not this kernel).
static int r8_row_blocks(const SplineDims& f, long B, int fast, bool grad) {
    if (f.NTWM != 4 || !f.o_r8 || option(FABHIP_OPT_SPLINE_MFMA) == 16) return 0;
    const int sel = option(FABHIP_OPT_TILE_SHAPE);
    int rb = sel == 16 ? 4 : (sel == 8 ? 2 : (sel == 4 ? 1 : (B <= 4L * cu_count() ? 1 : (B <= 8L * cu_count() ? 2 : 4))));
    // fast mode is a PERMISSION to use bf16: up to 8 chains per CU the fp32 stream kernel is the faster one (cfg 3: 0.39 ms
    // against 0.45 for the bf16 16-chain kernel), so fast-mode calls take it too; above, the bf16 kernel (0.68 against 1.03 ms)
    if (fast && rb == 4) return 0;
    // the LDS plan grows with the layer count (ReLU ballots) and the output chunks: deep / wide flows fall back to the smaller
    // tile, then to the 16x16x4 kernel (whose ballots live in the workspace)
    if (rb == 4 && (size_t)make_s8_lds(f, grad, 16).total * 4 > 160 * 1024) rb = 2;
    if (rb == 2 && (size_t)make_s8_lds(f, grad, 8).total * 4 > 160 * 1024) rb = 1;
    if (rb == 1 && (size_t)make_s8_lds(f, grad, 4).total * 4 > 160 * 1024) rb = 0;
    return rb;
}

template <int NCH, int RB, bool TRIM = false>
static int launch_logprob_r8(const SplineDims& f, const float* packed, const float* x, float* log_q, float* grad_x, long B,
                             float* Zsave, float* Psave, hipStream_t st, const SplineLeapDev& lp = SplineLeapDev{}) {
    const dim3 grid((unsigned)ceil_div((int)B, 4 * RB)), block(NTHREADS);
    const S8Lds l = make_s8_lds(f, grad_x != nullptr, 4 * RB);
    const size_t bytes = (size_t)l.total * 4;
    if (grad_x) {
        FAB_TRY(set_max_lds((const void*)k_spline_logprob_r8<NCH, RB, true, TRIM>, bytes));
        hipLaunchKernelGGL((k_spline_logprob_r8<NCH, RB, true, TRIM>), grid, block, bytes, st, f, l, packed, x, log_q, grad_x, B, Zsave,
                           Psave, sp_timeline(st), lp);
    } else {
        FAB_TRY(set_max_lds((const void*)k_spline_logprob_r8<NCH, RB, false, TRIM>, bytes));
        hipLaunchKernelGGL((k_spline_logprob_r8<NCH, RB, false, TRIM>), grid, block, bytes, st, f, l, packed, x, log_q, grad_x, B, Zsave,
                           Psave, sp_timeline(st), SplineLeapDev{});
    }
    return check_launch();
}

template <int RB>
static int launch_logprob_r8_nch(const SplineDims& f, const float* packed, const float* x, float* log_q, float* grad_x, long B,
                                 float* Zsave, float* Psave, hipStream_t st, const SplineLeapDev& lp = SplineLeapDev{}) {
    switch (f.NCH) {
        case 1: return launch_logprob_r8<1, RB>(f, packed, x, log_q, grad_x, B, Zsave, Psave, st, lp);
        case 2: return f.r8_trim ? launch_logprob_r8<2, RB, true>(f, packed, x, log_q, grad_x, B, Zsave, Psave, st, lp)
                                 : launch_logprob_r8<2, RB>(f, packed, x, log_q, grad_x, B, Zsave, Psave, st, lp);
        case 3: return launch_logprob_r8<3, RB>(f, packed, x, log_q, grad_x, B, Zsave, Psave, st, lp);
        case 4: return launch_logprob_r8<4, RB>(f, packed, x, log_q, grad_x, B, Zsave, Psave, st, lp);
        default: return FABHIP_ENOTSUP;
    }
}

template <int NTWM>
static int launch_net(const SplineDims& f, const float* packed, int layer, const float* Z, float* P, const float* dP,
                      float* G, long B, hipStream_t st, const SplineTape& tp, float* act) {
    const dim3 grid((unsigned)ceil_div((int)B, ROWS)), block(NTHREADS);
    const bool bwd = dP != nullptr;
    const NetLds l = make_net_lds(f, bwd);
    const size_t bytes = (size_t)l.total * 4;
    if (bwd) {
        FAB_TRY(set_max_lds((const void*)k_spline_net_bwd<NTWM>, bytes));
        hipLaunchKernelGGL((k_spline_net_bwd<NTWM>), grid, block, bytes, st, f, l, packed, layer, Z, dP, G, B, tp,
                           tp.base ? nullptr : act);
    } else {
        FAB_TRY(set_max_lds((const void*)k_spline_net_fwd<NTWM>, bytes));
        hipLaunchKernelGGL((k_spline_net_fwd<NTWM>), grid, block, bytes, st, f, l, packed, layer, Z, P, B, act);
    }
    return check_launch();
}

static int net(const SplineDims& f, const float* packed, int layer, const float* Z, float* P, const float* dP, float* G,
               long B, hipStream_t st, const SplineTape& tp = SplineTape{}, float* act = nullptr) {
    if (f.NTWM == 1) return launch_net<1>(f, packed, layer, Z, P, dP, G, B, st, tp, act);
    if (f.NTWM == 2) return launch_net<2>(f, packed, layer, Z, P, dP, G, B, st, tp, act);
    if (f.NTWM == 4) return launch_net<4>(f, packed, layer, Z, P, dP, G, B, st, tp, act);
    return FABHIP_ENOTSUP;
}

static inline size_t sp_al(size_t v) { return (v + 255) & ~(size_t)255; }

// launch.h: the outer step's begin / accept / step-size rule inside the leapfrog launches - where the one-launch leapfrog
// applies, the switch is on and the last wave's 2 nblk block sums fit the conditioner-output tile it uses as scratch
size_t spline_fold_scratch_floats(int64_t B) { return (size_t)32 * ((B + 15) / 16) + 64; }
// THE applicability test of the one-launch leapfrog, shared by the query below and by spline_log_prob_leap itself (a precondition
// added to one of them only would turn the caller's fallback into a failure half-way through an outer step): chains per
// workgroup / 8 of the kernel that would run, 0 where it does not apply
static int spline_leap_row_blocks(const fabhip_spline_flow* flow, int64_t B, SplineDims* dims) {
    if (!flow || !flow->packed || B < 1 || !option(FABHIP_OPT_SPLINE_LEAP) || option(FABHIP_OPT_SPLINE_STAGED)) return 0;
    if (check_spline_shape(flow->dim, flow->n_layers, flow->hidden) != FABHIP_OK) return 0;
    *dims = make_spline_dims(flow->dim, flow->n_layers, flow->hidden);
    return r8_row_blocks(*dims, (long)B, resolve_fast(flow->precision), true);
}
bool spline_leap_fold_supported(const fabhip_spline_flow* flow, int64_t B) {
    SplineDims f;
    const int rb = spline_leap_row_blocks(flow, B, &f);
    if (rb == 0 || !option(FABHIP_OPT_ADAPT_FOLD)) return false;
    const S8Lds l = make_s8_lds(f, true, 4 * rb);
    return 2 * ((B + 15) / 16) <= (int64_t)(4 * rb) * l.PS;
}

// launch.h: one leapfrog of the spline family in one launch
int spline_log_prob_leap(const fabhip_spline_flow* flow, const SplineLeap& a, float* log_q, float* grad_x, int64_t B,
                         void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (!flow || !flow->packed || !a.XP || !a.x_out || !a.P || !a.GU || !log_q || !grad_x || !workspace || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_spline_shape(flow->dim, flow->n_layers, flow->hidden));
    FAB_TRY(check_target(&a.tg, flow->dim));
    if (B == 0) return FABHIP_OK;
    SplineDims f;
    const int rb = spline_leap_row_blocks(flow, B, &f);
    if (rb == 0) return FABHIP_ENOTSUP;
    if (workspace_bytes < fabhip_spline_workspace_bytes(flow->dim, flow->n_layers, flow->hidden, B, 1)) return FABHIP_ENOSPC;
    char* ws = (char*)workspace;
    float* Z = (float*)ws; ws += sp_al((size_t)(f.L + 1) * B * f.D * 4);
    float* P = (float*)ws;
    SplineLeapDev d;
    d.XP = a.XP; d.x_out = a.x_out; d.P = a.P; d.GU = a.GU; d.eps_ptr = a.eps_ptr; d.ceps_ptr = a.ceps_ptr; d.mass = a.mass; d.c = a.c;
    d.max_grad = a.max_grad; d.tg = make_target_dev(a.tg); d.prop_lp = a.prop_lp; d.prop_gp = a.prop_gp;
    d.fold = a.fold;
    if (d.fold.flags && !spline_leap_fold_supported(flow, B)) return FABHIP_EINVAL;   // (the caller asks first)
    if (rb == 1) return launch_logprob_r8_nch<1>(f, flow->packed, a.XP, log_q, grad_x, (long)B, Z, P, st, d);
    if (rb == 2) return launch_logprob_r8_nch<2>(f, flow->packed, a.XP, log_q, grad_x, (long)B, Z, P, st, d);
    return launch_logprob_r8_nch<4>(f, flow->packed, a.XP, log_q, grad_x, (long)B, Z, P, st, d);
}

}  // namespace fab

using namespace fab;

extern "C" {

int fabhip_debug_spline_timeline(int64_t* host_out, int32_t n) {
    if (!g_sp_timeline || !host_out || n < 1 || n > 32) return FABHIP_EINVAL;
    return hipMemcpy(host_out, g_sp_timeline, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess ? FABHIP_OK : FABHIP_ELAUNCH;
}

int64_t fabhip_spline_packed_floats(int32_t dim, int32_t n_layers, int32_t hidden) {
    if (check_spline_shape(dim, n_layers, hidden) != FABHIP_OK) return -1;
    return (int64_t)make_spline_dims(dim, n_layers, hidden).total;
}

int fabhip_spline_pack(const fabhip_spline_params* p, float* packed, fabhip_stream_t stream) {
    if (!p || !packed || !p->base_scale || !p->base_circ) return FABHIP_EINVAL;
    FAB_TRY(check_spline_shape(p->dim, p->n_layers, p->hidden));
    const SplineDims f = make_spline_dims(p->dim, p->n_layers, p->hidden);
    hipStream_t st = (hipStream_t)stream;
    for (int l = 0; l < f.L; ++l) {
        if (!p->meta[l] || !p->w0[l] || !p->b0[l] || !p->wa[l] || !p->ba[l] || !p->wb[l] || !p->bb[l] || !p->wf[l] ||
            !p->bf[l] || !p->uw[l] || !p->uh[l] || !p->ud[l])
            return FABHIP_EINVAL;
        const SplineSrc s{p->meta[l], p->w0[l], p->b0[l], p->wa[l], p->ba[l], p->wb[l], p->bb[l], p->wf[l], p->bf[l],
                          p->pfw[l], p->uw[l], p->uh[l], p->ud[l]};
        hipLaunchKernelGGL(k_spline_pack_layer, dim3(ceil_div(f.o_h, 256 * 8)), dim3(256), 0, st, f, s, l, packed);
        hipLaunchKernelGGL(k_spline_pack_bf16, dim3(ceil_div((4 + 2 * f.NCH) * (f.Wp * f.Wp / 2), 256 * 8)), dim3(256), 0, st,
                           f, s, l, packed);
        if (f.o_r8)
            hipLaunchKernelGGL(k_spline_pack_r8, dim3(ceil_div(f.r8_layer, 256 * 8)), dim3(256), 0, st, f, s,
                               l > 0 ? p->meta[l - 1] : (const float*)nullptr, l, packed);
    }
    hipLaunchKernelGGL(k_spline_pack_base, dim3(1), dim3(64), 0, st, f, p->base_scale, p->base_circ, packed);
    return check_launch();
}

size_t fabhip_spline_workspace_bytes(int32_t dim, int32_t n_layers, int32_t hidden, int64_t B, int32_t with_grad) {
    if (check_spline_shape(dim, n_layers, hidden) != FABHIP_OK || B < 0) return 0;
    const SplineDims f = make_spline_dims(dim, n_layers, hidden);
    size_t s = sp_al((size_t)(f.L + 1) * B * f.D * 4);                 // layer input states
    s += sp_al((size_t)(with_grad ? f.L : 1) * B * f.NFP * 4);          // conditioner outputs (kept per layer for the reverse sweep)
    if (with_grad) s += sp_al((size_t)B * f.NFP * 4) + 2 * sp_al((size_t)B * f.D * 4);
    if (with_grad) s += sp_al((size_t)f.L * ceil_div((int)B, ROWS) * (2 * NWAVE * f.NTWM * 4) * 8);   // ReLU decision words
    return s + 256;
}

static int spline_log_prob_impl(const fabhip_spline_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                                float* tape, void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!flow || !flow->packed || !x || !log_q || !workspace || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_spline_shape(flow->dim, flow->n_layers, flow->hidden));
    if (B == 0) return FABHIP_OK;
    if (workspace_bytes < fabhip_spline_workspace_bytes(flow->dim, flow->n_layers, flow->hidden, B, grad_x != nullptr))
        return FABHIP_ENOSPC;
    const SplineDims f = make_spline_dims(flow->dim, flow->n_layers, flow->hidden);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* Z = (float*)ws; ws += sp_al((size_t)(f.L + 1) * B * f.D * 4);       // Z[l+1] = input of layer l, Z[0] = base side
    float* P = (float*)ws; ws += sp_al((size_t)(grad_x ? f.L : 1) * B * f.NFP * 4);
    float* dP = nullptr; float* Ga = nullptr; float* Gb = nullptr;
    if (grad_x) {
        dP = (float*)ws; ws += sp_al((size_t)B * f.NFP * 4);
        Ga = (float*)ws; ws += sp_al((size_t)B * f.D * 4);
        Gb = (float*)ws; ws += sp_al((size_t)B * f.D * 4);
    }
    unsigned long long* bits = grad_x ? (unsigned long long*)ws : nullptr;   // ReLU decisions (one-launch kernel)
    const size_t zs = (size_t)B * f.D, ps = (size_t)B * f.NFP;
    const float* pk = flow->packed;
    if (!tape && !option(FABHIP_OPT_SPLINE_STAGED)) {           // one launch (the staged kernels below: tape, debugging)
        const int rb = r8_row_blocks(f, (long)B, resolve_fast(flow->precision), grad_x != nullptr);
        if (rb == 1) return launch_logprob_r8_nch<1>(f, pk, x, log_q, grad_x, (long)B, Z, P, (hipStream_t)stream);
        if (rb == 2) return launch_logprob_r8_nch<2>(f, pk, x, log_q, grad_x, (long)B, Z, P, (hipStream_t)stream);
        if (rb == 4) return launch_logprob_r8_nch<4>(f, pk, x, log_q, grad_x, (long)B, Z, P, (hipStream_t)stream);
        if (f.NTWM == 1) return launch_logprob<1>(f, pk, x, log_q, grad_x, (long)B, Z, P, bits, resolve_fast(flow->precision), (hipStream_t)stream);
        if (f.NTWM == 2) return launch_logprob<2>(f, pk, x, log_q, grad_x, (long)B, Z, P, bits, resolve_fast(flow->precision), (hipStream_t)stream);
        if (f.NTWM == 4) return launch_logprob<4>(f, pk, x, log_q, grad_x, (long)B, Z, P, bits, resolve_fast(flow->precision), (hipStream_t)stream);
        return FABHIP_ENOTSUP;
    }
    const SplineTape tp = make_spline_tape(f, (long)B, tape);
    const dim3 wgrid((unsigned)((B + 3) / 4)), wblock(256);
    hipLaunchKernelGGL(k_spline_first_stage, dim3((unsigned)((B * f.D + 255) / 256 > 4096 ? 4096 : (B * f.D + 255) / 256)),
                       dim3(256), 0, st, f, pk, x, Z + (size_t)f.L * zs, log_q, (long)B);
    for (int l = f.L - 1; l >= 0; --l) {
        float* Pl = P + (grad_x ? (size_t)l * ps : 0);
        FAB_TRY(net(f, pk, l, Z + (size_t)(l + 1) * zs, Pl, nullptr, nullptr, (long)B, st));
        hipLaunchKernelGGL(k_spline_apply<0>, wgrid, wblock, 0, st, f, pk, l, Z + (size_t)(l + 1) * zs, Pl, Z + (size_t)l * zs,
                           log_q, 1.f, (long)B, (float*)nullptr);
    }
    hipLaunchKernelGGL(k_spline_base, wgrid, wblock, 0, st, f, pk, Z, log_q, Ga, (long)B);
    if (grad_x) {
        float* gin = Ga;                                 // cotangent of Z[0] (the base side): d log p0 / dz
        for (int l = 0; l < f.L; ++l) {
            float* out = (l == f.L - 1) ? grad_x : (gin == Ga ? Gb : Ga);
            float* dPl = tape ? tape + (size_t)l * tp.layer_stride + tp.o_dP : dP;      // kept per layer on the tape
            float* dUl = tape ? tape + (size_t)l * tp.layer_stride + tp.o_dU : nullptr;
            hipLaunchKernelGGL(k_spline_apply_bwd, wgrid, wblock, 0, st, f, pk, l, Z + (size_t)(l + 1) * zs,
                               P + (size_t)l * ps, gin, out, dPl, (long)B, dUl);
            FAB_TRY(net(f, pk, l, Z + (size_t)(l + 1) * zs, nullptr, dPl, out, (long)B, st, tp));
            gin = out;
        }
    }
    return check_launch();
}

int fabhip_spline_log_prob(const fabhip_spline_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                           void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    return spline_log_prob_impl(flow, x, log_q, grad_x, B, nullptr, workspace, workspace_bytes, stream);
}

int fabhip_spline_tape_layout(int32_t dim, int32_t n_layers, int32_t hidden, int64_t B, int64_t out16[16]) {
    if (!out16 || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_spline_shape(dim, n_layers, hidden));
    const SplineDims f = make_spline_dims(dim, n_layers, hidden);
    const SplineTape t = make_spline_tape(f, (long)B, nullptr);
    const int64_t v[16] = {(int64_t)t.layer_stride * n_layers, t.layer_stride, t.o_XI, t.o_A0, t.o_dA0, t.o_R0, t.o_R1, t.o_H1,
                           t.o_dH1, t.o_dT, t.o_dH0, t.o_dP, t.o_dU, f.Wp, f.NFP, SP_MD * SP_NP};
    for (int i = 0; i < 16; ++i) out16[i] = v[i];
    return FABHIP_OK;
}

int fabhip_spline_log_prob_tape(const fabhip_spline_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                                float* tape, int64_t tape_floats, void* workspace, size_t workspace_bytes,
                                fabhip_stream_t stream) {
    if (!flow || !tape || !grad_x) return FABHIP_EINVAL;
    int64_t lay[16];
    FAB_TRY(fabhip_spline_tape_layout(flow->dim, flow->n_layers, flow->hidden, B, lay));
    if (tape_floats < lay[0]) return FABHIP_ENOSPC;
    return spline_log_prob_impl(flow, x, log_q, grad_x, B, tape, workspace, workspace_bytes, stream);
}

int fabhip_spline_sample_vjp_tape(const fabhip_spline_flow* flow, const float* u, const float* eps, const float* gx, const float* gl,
                                  float* v_x, float* v_base, int64_t B, float* tape_density, float* tape_inverse,
                                  int64_t tape_floats, void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!flow || !flow->packed || !u || !eps || !v_x || !tape_density || !tape_inverse || !workspace || B < 0 || (!gx && !gl))
        return FABHIP_EINVAL;
    FAB_TRY(check_spline_shape(flow->dim, flow->n_layers, flow->hidden));
    int64_t lay[16];
    FAB_TRY(fabhip_spline_tape_layout(flow->dim, flow->n_layers, flow->hidden, B, lay));
    if (tape_floats < lay[0]) return FABHIP_ENOSPC;
    if (B == 0) return FABHIP_OK;
    if (workspace_bytes < fabhip_spline_workspace_bytes(flow->dim, flow->n_layers, flow->hidden, B, 1)) return FABHIP_ENOSPC;
    const SplineDims f = make_spline_dims(flow->dim, flow->n_layers, flow->hidden);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* Z = (float*)ws; ws += sp_al((size_t)(f.L + 1) * B * f.D * 4);
    float* P = (float*)ws; ws += sp_al((size_t)f.L * B * f.NFP * 4);
    float* lq = (float*)ws; ws += sp_al((size_t)B * f.NFP * 4);         // (the density sweep's dP scratch: here log q row sums)
    float* Ga = (float*)ws; ws += sp_al((size_t)B * f.D * 4);
    float* Gb = (float*)ws;
    const size_t zs = (size_t)B * f.D, ps = (size_t)B * f.NFP;
    const float* pk = flow->packed;
    const dim3 wgrid((unsigned)((B + 3) / 4)), wblock(256);
    // 1. the sampler again (same kernels, same noise: the same states), every layer's output before its shift and its
    //    conditioner outputs kept: Z[l + 1] / P[l] are what the log_prob direction would recompute from x up to rounding - and
    //    the inverse of a stiff chain of splines amplifies that rounding, so the sweeps below linearise where the sample was made
    hipLaunchKernelGGL(k_spline_base_sample, wgrid, wblock, 0, st, f, pk, u, eps, Z, lq, (long)B);
    const float* zin = Z;
    for (int l = 0; l < f.L; ++l) {
        hipLaunchKernelGGL(k_spline_apply<1>, wgrid, wblock, 0, st, f, pk, l, zin, (const float*)nullptr, v_x, lq, -1.f, (long)B,
                           (float*)nullptr);
        FAB_TRY(net(f, pk, l, v_x, P + (size_t)l * ps, nullptr, nullptr, (long)B, st));
        float* out = (zin == Ga) ? Gb : Ga;
        hipLaunchKernelGGL(k_spline_apply<2>, wgrid, wblock, 0, st, f, pk, l, v_x, P + (size_t)l * ps, out, lq, -1.f, (long)B,
                           Z + (size_t)(l + 1) * zs);
        zin = out;
    }
    // 2. the log_prob direction's reverse sweep on these states (seed 1 per sample): tape_density, d log q / dx -> v_x
    const SplineTape t1 = make_spline_tape(f, (long)B, tape_density);
    hipLaunchKernelGGL(k_spline_base, wgrid, wblock, 0, st, f, pk, Z, lq, Ga, (long)B);
    float* gin = Ga;
    for (int l = 0; l < f.L; ++l) {
        float* out = (l == f.L - 1) ? v_x : (gin == Ga ? Gb : Ga);
        float* tl = tape_density + (size_t)l * t1.layer_stride;
        hipLaunchKernelGGL(k_spline_apply_bwd, wgrid, wblock, 0, st, f, pk, l, Z + (size_t)(l + 1) * zs, P + (size_t)l * ps, gin, out,
                           tl + t1.o_dP, (long)B, tl + t1.o_dU);
        FAB_TRY(net(f, pk, l, Z + (size_t)(l + 1) * zs, nullptr, tl + t1.o_dP, out, (long)B, st, t1));
        gin = out;
    }
    // 3. v at x = gx + gl * d log q / dx (in place)
    hipLaunchKernelGGL(k_spline_vjp_seed, dim3((unsigned)((B * f.D + 255) / 256 > 4096 ? 4096 : (B * f.D + 255) / 256)), dim3(256), 0,
                       st, v_x, gx, gl, (long)B, f.D);
    // 4. v from the x side to the base side (the stage-boundary shifts and wraps have unit derivative): tape_inverse
    const SplineTape t2 = make_spline_tape(f, (long)B, tape_inverse);
    const float* vin = v_x;
    for (int l = f.L - 1; l >= 0; --l) {
        float* out = (l == 0 && v_base) ? v_base : (vin == Ga ? Gb : Ga);
        float* tl = tape_inverse + (size_t)l * t2.layer_stride;
        hipLaunchKernelGGL(k_spline_vsweep<0>, wgrid, wblock, 0, st, f, pk, l, Z + (size_t)(l + 1) * zs, P + (size_t)l * ps, vin,
                           out, tl + t2.o_dP, (long)B, (float*)nullptr);
        FAB_TRY(net(f, pk, l, Z + (size_t)(l + 1) * zs, nullptr, tl + t2.o_dP, out, (long)B, st, t2));
        hipLaunchKernelGGL(k_spline_vsweep<1>, wgrid, wblock, 0, st, f, pk, l, Z + (size_t)(l + 1) * zs, (const float*)nullptr,
                           (const float*)nullptr, out, (float*)nullptr, (long)B, tl + t2.o_dU);
        vin = out;
    }
    return check_launch();
}

int fabhip_spline_sample(const fabhip_spline_flow* flow, const float* u, const float* eps, float* x, float* log_q,
                         int64_t B, void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!flow || !flow->packed || !u || !eps || !x || !log_q || !workspace || B < 0) return FABHIP_EINVAL;
    FAB_TRY(check_spline_shape(flow->dim, flow->n_layers, flow->hidden));
    if (B == 0) return FABHIP_OK;
    if (workspace_bytes < fabhip_spline_workspace_bytes(flow->dim, flow->n_layers, flow->hidden, B, 0)) return FABHIP_ENOSPC;
    const SplineDims f = make_spline_dims(flow->dim, flow->n_layers, flow->hidden);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* Z = (float*)ws; ws += sp_al((size_t)(f.L + 1) * B * f.D * 4);
    float* P = (float*)ws;
    const size_t zs = (size_t)B * f.D;
    float* za = Z; float* zb = Z + zs; float* zc = Z + 2 * zs;
    const float* pk = flow->packed;
    const dim3 wgrid((unsigned)((B + 3) / 4)), wblock(256);
    hipLaunchKernelGGL(k_spline_base_sample, wgrid, wblock, 0, st, f, pk, u, eps, za, log_q, (long)B);
    for (int l = 0; l < f.L; ++l) {
        // identity coordinates through the inverse unconditional spline (zb), conditioner on them, then the rest
        hipLaunchKernelGGL(k_spline_apply<1>, wgrid, wblock, 0, st, f, pk, l, za, (const float*)nullptr, zb, log_q, -1.f, (long)B,
                           (float*)nullptr);
        FAB_TRY(net(f, pk, l, zb, P, nullptr, nullptr, (long)B, st));
        float* out = (l == f.L - 1) ? x : zc;
        hipLaunchKernelGGL(k_spline_apply<2>, wgrid, wblock, 0, st, f, pk, l, zb, P, out, log_q, -1.f, (long)B, (float*)nullptr);
        float* tmp = za; za = zc; zc = tmp;
    }
    return check_launch();
}

}  // extern "C"
