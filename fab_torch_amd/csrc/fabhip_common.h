// fabhip — MI355X (gfx950 / CDNA4) kernels for the fab-torch AIS / flow-density hot path.
// Shared host/device definitions: error codes, flow geometry, packed-weight layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/fabhip.h"

#define FAB_HD __host__ __device__ __forceinline__

namespace fab {

constexpr int ROWS = 16;        // chains per workgroup = M of v_mfma_f32_16x16x4_f32
constexpr int NWAVE = 4;        // one wave per SIMD
constexpr int NTHREADS = 64 * NWAVE;
constexpr int MAX_DIM = 64;     // D <= 64
constexpr int MAX_WIDTH = 512;  // hidden width <= 512 (8 column tiles per wave)

FAB_HD int ceil_div(int a, int b) { return (a + b - 1) / b; }
FAB_HD int pad16(int a) { return ceil_div(a, 16) * 16; }
FAB_HD int pad32(int a) { return ceil_div(a, 32) * 32; }
// hidden width is padded to 64 * (column tiles per wave), tiles per wave in {1, 2, 4, 5, 8}
// fused 4-chain stream (flow_r4f.h): tiles per wave, layer and direction / floats of one layer's bias block
FAB_HD int r4f_tl(int G) { return (4 * G + 4) * G + 1; }
FAB_HD int r4f_tl_fast(int G) { return (2 * G + 4) * G + 1; }   // bf16 W x W tiles: two k-quads per tile
FAB_HD int r4f_bias_stride(int Wp) { return 2 * Wp + 80; }
FAB_HD int ntw_variant(int W) { const int p = ceil_div(W, 64); return p <= 2 ? p : (p <= 4 ? 4 : (p <= 5 ? 5 : 8)); }

// Geometry of one RealNVP flow (experiments/make_flow/make_normflow_model.py:11-30):
// K x [AffineCouplingBlock(MLP[d, W, W, 2(D-d)], exp) + InvertibleAffine(D)], DiagGaussian base.
struct FlowDims {
    int D, d, DO, K, W;         // dim, conditioner width d=ceil(D/2), transformed DO=D-d, layers, hidden
    int Dp, dp, DOp, Wp;        // padded to multiples of 16
    int KBD, NTD;               // D  as k-blocks(16) / column tiles(16)
    int KBd, NTd;               // d
    int KBW, NTW;               // W
    int KBO, NTO;               // 2*DOp (coupling parameter width: [shift | scale])
    // packed offsets (in floats) inside one layer block
    int o_AW, o_AWT, o_AWI, o_W1, o_W2, o_W3, o_W3T, o_W2T, o_W1T, o_b1, o_b2, o_b3, o_logS;
    int o_ac, o_at;             // ActNorm folded into the affine maps: additive terms of the density / sampling direction
    int o_W2h, o_W2Th;          // fast mode: bf16 images of W2 / W2^T (v_mfma_f32_16x16x32_bf16 B-operand tiles), Wp^2/2 floats each
    int o_AWIT;                 // (W'^-1)^T tiles: backward of the sampling direction (fabhip_flow_sample_grad_tape)
    int layer_stride;           // floats per layer block
    int o_base;                 // offset of base block: loc[Dp], log_scale[Dp]
    int o_scratch;              // offset of affine scratch: per layer W[D*D], Winv[D*D]
    int o_r4;                   // 4-chain-tile weight image (flow_r4.h: R4Dims), K layer blocks
    int o_r4s;                  // the same tiles in per-wave consumption order (flow_r4.h: R4Stream; D <= 32, Wp >= 128), else -1
    int o_r8;                   // 8-chain-tile weight image (flow_r8.h; D <= 32, Wp = 256 / 320), else -1
    int o_r8f;                  // 8-chain tiles, fused stages (flow_r8.h: W1' = W'[:, :d] W1^T; needs o_r8 and o_r4fb), else -1
    int o_r4f, o_r4fb;          // 4-chain tiles, fused stages (flow_r4f.h): weight stream / bias blocks (where o_r4s exists), else -1
    int o_r4fh;                 // the same stream with bf16 W x W tiles (fast mode; density direction only), else -1
    int total;                  // total floats
    long long* timeline;        // dev-only: s_memtime stamps of workgroup 0 (nullptr in production)
    int fast;                   // this call runs the fast-mode kernels (resolved from fabhip_flow::precision by the entry point)
};

FAB_HD FlowDims make_flow_dims(int D, int K, int W) {
    FlowDims f;
    f.D = D; f.K = K; f.W = W;
    f.d = (D + 1) / 2;          // int(D/2 + 0.5)
    f.DO = D - f.d;
    // K extents are padded to 32 (even k-block counts: 2-deep weight ring), the hidden width to
    // 64 * tiles-per-wave (4-deep ring, every wave owns the same number of column tiles)
    f.Dp = pad32(D); f.dp = pad32(f.d); f.DOp = pad16(f.DO); f.Wp = 64 * ntw_variant(W);
    f.KBD = f.Dp / 16; f.NTD = pad16(D) / 16;
    f.KBd = f.dp / 16; f.NTd = pad16(f.d) / 16;
    f.KBW = f.Wp / 16; f.NTW = f.Wp / 16;
    f.KBO = 2 * f.DOp / 16; f.NTO = 2 * f.DOp / 16;
    int o = 0;
    auto mat = [&](int kb, int nt) { int r = o; o += kb * nt * 256; return r; };
    f.o_AW = mat(f.KBD, f.NTD);
    f.o_AWT = mat(f.KBD, f.NTD);
    f.o_AWI = mat(f.KBD, f.NTD);
    f.o_W1 = mat(f.KBd, f.NTW);
    f.o_W2 = mat(f.KBW, f.NTW);
    f.o_W3 = mat(f.KBW, f.NTO);
    f.o_W3T = mat(f.KBO, f.NTW);
    f.o_W2T = mat(f.KBW, f.NTW);
    f.o_W1T = mat(f.KBW, f.NTd);
    f.o_b1 = o; o += f.Wp;
    f.o_b2 = o; o += f.Wp;
    f.o_b3 = o; o += 2 * f.DOp;
    f.o_logS = o; o += 16;      // [0] = sum(log_S) (- sum(ActNorm.s)); rest pad (keeps 64-byte alignment)
    f.o_ac = o; o += 64;        // density direction:  z <- x @ W' + ac,  W' = diag(e^-s) W, ac = -(t e^-s) @ W
    f.o_at = o; o += 64;        // sampling direction: z <- x @ W'^-1 + at, W'^-1 = W^-1 diag(e^s), at = t
    f.o_W2h = o; o += f.Wp * f.Wp / 2;
    f.o_W2Th = o; o += f.Wp * f.Wp / 2;
    f.o_AWIT = mat(f.KBD, f.NTD);
    f.layer_stride = o;
    f.o_base = K * f.layer_stride;
    f.o_scratch = f.o_base + 2 * f.Dp;
    f.o_r4 = (f.o_scratch + K * 2 * D * D + 63) & ~63;          // 256-byte aligned: read as float4 tiles
    // per layer: AW, AWT [pad16(D) x 64], W1 [pad16(d) x Wp], W2, W2T [Wp x Wp], W3 [Wp x 2 DOp], W1T [Wp x pad16(d)],
    // W3T [pad16(2 DOp) x Wp]
    f.total = f.o_r4 + K * (2 * pad16(D) * 64 + pad16(f.d) * f.Wp + 2 * f.Wp * f.Wp + f.Wp * 2 * f.DOp + f.Wp * pad16(f.d) +
                            pad16(2 * f.DOp) * f.Wp);
    // stream image: (4 NTW + 4) items per layer and section (density forward, density reverse, sampling), each 4 waves x NTW
    // tiles of 1 KiB; 8 items of padding behind the reverse and behind the sampling section
    f.o_r4s = -1;
    if (D <= 32 && f.Wp >= 128) {
        const int ntw = f.Wp / 64;
        f.o_r4s = f.total;
        f.total += (3 * K * (4 * ntw + 4) + 16) * 4 * ntw * 256;
    }
    // 8-chain-tile image: per layer 4 waves x (forward + reverse = 160 / 238 tiles of 1 KiB for G = Wp / 64 = 4 / 5; flow_r8.h,
    // r4: the narrow matrices as dense tiles)
    f.o_r8 = -1;
    if (D <= 32 && f.DOp == 16 && (f.Wp == 256 || f.Wp == 320)) {
        const int G = f.Wp / 64;
        f.o_r8 = (f.total + 63) & ~63;
        // per wave: K layers x 2 directions x (80 / 120 tiles: 119 padded to a multiple of the ring depth) + 64 tiles of tail
        f.total = f.o_r8 + NWAVE * (2 * K * (G == 5 ? 120 : 80) + 64) * 256;
    }
    // fused-stage 4-chain stream (flow_r4f.h): [K density forward | K density reverse | 1 pad | K + 1 sampling | 1 pad] layer slots
    // of r4f_tl(G) tiles per wave, then the bias blocks (K density, K + 1 sampling)
    f.o_r4f = -1; f.o_r4fb = -1; f.o_r4fh = -1;
    if (f.o_r4s >= 0 && f.DOp == 16) {
        const int G = f.Wp / 64;
        f.o_r4f = (f.total + 63) & ~63;
        f.total = f.o_r4f + (3 * K + 3) * r4f_tl(G) * NWAVE * 256;
        f.o_r4fb = f.total;
        f.total += (2 * K + 1) * r4f_bias_stride(f.Wp);
        // fast mode: [K density forward | K density reverse | 1 pad] layer slots of r4f_tl_fast(G) tiles
        f.o_r4fh = (f.total + 63) & ~63;
        f.total = f.o_r4fh + (2 * K + 1) * r4f_tl_fast(G) * NWAVE * 256;
    }
    // 8-chain tiles with fused stages: per wave K layers x 2 directions x (84 / 126 tiles = 2 / 3 ring depths of 42) + 64 tiles of tail
    f.o_r8f = -1;
    if (f.o_r8 >= 0 && f.o_r4fb >= 0) {
        const int G = f.Wp / 64;
        f.o_r8f = (f.total + 63) & ~63;
        f.total = f.o_r8f + NWAVE * (2 * K * (G == 5 ? 126 : 84) + 64) * 256;
    }
    f.timeline = nullptr;
    f.fast = 0;
    return f;
}

// LDS plan (floats) of one workgroup evaluating the flow on a 16-chain tile.
struct FlowLds {
    int DS, WS, PS, PN;         // leading dims: state, hidden, dparam, k-split partials
    int o_U0, o_U1, o_HA, o_HB, o_PART, o_DP, o_ES, o_V2, o_MASK;
    int total;                  // floats
};

FAB_HD FlowLds make_flow_lds(const FlowDims& f, bool with_grad) {
    FlowLds l;
    l.DS = f.Dp + 4;
    l.WS = f.Wp + 4;
    l.PS = 2 * f.DOp + 4;
    int pn = 2 * f.DOp; if (16 * f.NTd > pn) pn = 16 * f.NTd;
    l.PN = pn + 4;
    int o = 0;
    l.o_U0 = o; o += ROWS * l.DS;
    l.o_U1 = o; o += ROWS * l.DS;
    l.o_HA = o; o += ROWS * l.WS;
    l.o_HB = o; o += ROWS * l.WS;
    l.o_PART = o; o += NWAVE * ROWS * l.PN;
    l.o_DP = o; o += ROWS * l.PS;
    l.o_ES = o; if (with_grad) o += f.K * ROWS * f.DOp;
    l.o_V2 = o; if (with_grad) o += f.K * ROWS * f.DOp;
    l.o_MASK = o; if (with_grad) o += f.K * 2 * NTHREADS;   // one 32-bit ReLU sign word per thread, layer and hidden GEMM
    l.total = (o + 3) & ~3;
    return l;
}

// Tape of the training forward (fabhip_flow_log_prob_tape): what the parameter-gradient GEMMs need, kept in
// HBM as row-major [Bp x width] matrices per layer (Bp = rows padded to the 16-chain tile).  The matrices on
// the input side of a Linear carry a column of ones after their (padded) width, so the bias gradient is one
// more output column of the same GEMM.  All widths are multiples of 16 floats (64-byte rows).
struct TapeDims {
    long Bp;
    int wz, w1, wh, wp, we, wb; // ZA/GZ: 16 NTD | Z1: 16 NTd + 16 | H1/H2: Wp + 16 | DP: 2 DOp | E1/E2: Wp | TB: 2 wz + 16
    long o_ZA, o_GZ, o_Z1, o_H1, o_H2, o_DP, o_E2, o_E1;     // offsets inside one layer block (floats)
    // TB (after the layer blocks): [zn / sc | zn^2 - 1 | ones] of the base distribution, zn = (z - loc) / sc
    long layer_stride, o_TB, total;
};

FAB_HD TapeDims make_tape_dims(const FlowDims& f, long B) {
    TapeDims t;
    t.Bp = (B + ROWS - 1) / ROWS * ROWS;
    t.wz = 16 * f.NTD; t.w1 = 16 * f.NTd + 16; t.wh = f.Wp + 16; t.wp = 2 * f.DOp; t.we = f.Wp;
    t.wb = 2 * t.wz + 16;
    long o = 0;
    t.o_ZA = o; o += t.Bp * t.wz;
    t.o_GZ = o; o += t.Bp * t.wz;
    t.o_Z1 = o; o += t.Bp * t.w1;
    t.o_H1 = o; o += t.Bp * t.wh;
    t.o_H2 = o; o += t.Bp * t.wh;
    t.o_DP = o; o += t.Bp * t.wp;
    t.o_E2 = o; o += t.Bp * t.we;
    t.o_E1 = o; o += t.Bp * t.we;
    t.layer_stride = o;
    t.o_TB = (long)f.K * t.layer_stride;
    t.total = t.o_TB + t.Bp * t.wb;
    return t;
}

}  // namespace fab
