/* fabhip.h — C ABI of libfabhip.so: MI355X (gfx950) kernels for the fab-torch AIS / flow-density
 * hot path.  This is the drop-in boundary: plain device pointers + sizes, no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the field comment says "host";
 *   - all tensors are contiguous row-major float32 unless stated; indices are int64;
 *   - functions only ENQUEUE work on `stream` (a hipStream_t); they never allocate, never
 *     synchronise and never throw; scratch comes from the caller (`*_workspace_bytes`);
 *   - return value: FABHIP_OK or a negative FABHIP_E* code (see fabhip_strerror).
 *
 * Each entry point cites the reference (lollcat/fab-torch) interface it replaces.
 */
#ifndef FABHIP_H
#define FABHIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fabhip_stream_t; /* hipStream_t */

enum {
    FABHIP_OK = 0,
    FABHIP_EINVAL = -1,  /* bad shape / null pointer / misaligned buffer             */
    FABHIP_ENOTSUP = -2, /* dimension beyond compiled limits (dim<=64, width<=512)   */
    FABHIP_ELAUNCH = -3, /* hipGetLastError() != hipSuccess after a launch           */
    FABHIP_ENOSPC = -4   /* workspace too small                                      */
};

#define FABHIP_MAX_LAYERS 64
#define FABHIP_MAX_DIM 64
#define FABHIP_MAX_WIDTH 512

const char* fabhip_strerror(int code);
/* ABI revision of this header: bumped on every change of a struct layout or a function signature.  The host
 * binding compares it (and the struct sizes below) with what it was written against and refuses to run on a
 * mismatch, so that a stale library can never be driven with newer struct layouts. */
#define FABHIP_ABI_VERSION 216
int fabhip_version(void);
/* sizeof() of the argument structs as the library was compiled:
 * {fabhip_flow_params, fabhip_flow, fabhip_target, fabhip_point, fabhip_anneal, fabhip_hmc_args,
 *  fabhip_metropolis_args, fabhip_ais_args}. */
void fabhip_abi_sizes(int64_t out8[8]);

/* ------------------------------------------------------------------------------------------
 * RealNVP flow  (replaces normflows NormalizingFlow.sample / .log_prob as wrapped by
 * fab/wrappers/normflows.py:16-31, architecture of experiments/make_flow/make_normflow_model.py:11-30)
 * ---------------------------------------------------------------------------------------- */

/* Raw parameters in normflows / nn.Linear layout (weight[out][in]); host arrays of device pointers. */
typedef struct {
    int32_t dim, n_layers, width;
    const float* w1[FABHIP_MAX_LAYERS];     /* [width][d]          d = ceil(dim/2)            */
    const float* b1[FABHIP_MAX_LAYERS];     /* [width]                                        */
    const float* w2[FABHIP_MAX_LAYERS];     /* [width][width]                                 */
    const float* b2[FABHIP_MAX_LAYERS];     /* [width]                                        */
    const float* w3[FABHIP_MAX_LAYERS];     /* [2(dim-d)][width]   rows interleaved shift,scale */
    const float* b3[FABHIP_MAX_LAYERS];     /* [2(dim-d)]                                     */
    const float* lu_L[FABHIP_MAX_LAYERS];   /* [dim][dim] InvertibleAffine.L                  */
    const float* lu_U[FABHIP_MAX_LAYERS];   /* [dim][dim] InvertibleAffine.U                  */
    const float* log_S[FABHIP_MAX_LAYERS];  /* [dim]                                          */
    const float* sign_S[FABHIP_MAX_LAYERS]; /* [dim]                                          */
    const float* perm_P[FABHIP_MAX_LAYERS]; /* [dim][dim] permutation matrix                  */
    const float* loc;                       /* [dim] DiagGaussian.loc                         */
    const float* log_scale;                 /* [dim] DiagGaussian.log_scale                   */
    /* normflows ActNorm(dim) after layer k's InvertibleAffine (make_normflow_model.py:27-29, act_norm=True):
     * sampling direction z <- z * exp(s) + t, log_det += sum(s).  NULL (both) = no ActNorm after that layer. */
    const float* an_s[FABHIP_MAX_LAYERS];   /* [dim] ActNorm.s                                */
    const float* an_t[FABHIP_MAX_LAYERS];   /* [dim] ActNorm.t                                */
} fabhip_flow_params;

/* FAST MODE (off by default; NOT the parity path).  When on, the transition kernels (fabhip_hmc_transition,
 * fabhip_ais_run with HMC, fabhip_create_point and fabhip_flow_log_prob when they return gradients) run the two width x width GEMMs of every coupling
 * layer - 87 % of the flow's flops - on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation, weights
 * from a bf16 image every fabhip_flow_pack also writes, activations rounded to bf16 as they are fetched).  log q then
 * differs from the fp32 path at the 1e-3 .. 1e-2 level (the sampler stays a valid HMC-AIS sampler for that slightly
 * different density; fabhip_flow_sample, gradient-free density evaluations, the training path and Metropolis stay fp32).
 * fabhip_spline_log_prob with gradients does the same for the spline conditioner's hidden x hidden GEMMs.
 * Process-wide switch, returns the previous value.  SURVEY section 7's "fp32 parity mode and a fast mode". */
int fabhip_set_fast_mode(int on);
int fabhip_get_fast_mode(void);

/* Developer / test switches (A/B variants of one computation, diagnostics).  A plain table of ints inside the library:
 * initialised ONCE, when the library is loaded, from the environment variables named below, read by the launchers with
 * one load (nothing on a launch path calls getenv), changed at run time only through fabhip_set_option (returns the
 * previous value, or a negative FABHIP_E* code for an unknown key).  No reference counterpart (the reference has no
 * kernel variants); production code never sets them. */
#define FABHIP_OPT_TILE_SHAPE 0          /* FABHIP_TILE: chains per workgroup of the fused RealNVP transitions (4 = flow_r4.h,
                                            8 = flow_r8.h, 16 = flow_device.h tiles; a shape that does not exist for the flow
                                            falls back to 16) and of the 4x4x1 spline density kernel (8 / 16); 0 = by batch
                                            size (default) */
#define FABHIP_OPT_R4_STREAM 1           /* FABHIP_R4_STREAM: 4-chain tiles: 2 = fused stages on their own weight stream (flow_r4f.h: the
                                            D x D map multiplied together with the first / last conditioner Linear, coupling in the
                                            W3 epilogue; default where the image exists), 3 = the same with three items of every W x W
                                            stage prefetched into LDS during the short stages (opt-in: bit-identical to 2 and, measured,
                                            no faster - the copies' issue time lands on the short stages),
                                            1 = the round-3 stream (one stage per matrix), 0 = per-stage request groups.  The 8-chain tiles follow the same switch: >= 2 =
                                            their fused-stage stream (flow_r8.h, default), below = one stage per matrix */
#define FABHIP_OPT_SCAN_VARIANT 2        /* FABHIP_SCAN_VARIANT: fixed-point CDF scan, 3 = LDS-transposed (default), 0-2 = A/B */
#define FABHIP_OPT_SYSTEMATIC_VARIANT 3  /* FABHIP_SYSTEMATIC_VARIANT: 1 = fused systematic sampler (default), 0 = CDF in HBM */
#define FABHIP_OPT_SPLINE_STAGED 4       /* FABHIP_SPLINE_STAGED: 1 = per-layer spline kernels instead of the one-launch kernel */
#define FABHIP_OPT_TIMELINE 5            /* FABHIP_TIMELINE: 1 = workgroup 0 writes s_memtime stage stamps (fabhip_debug_timeline) */
#define FABHIP_OPT_SPLINE_MFMA 6         /* FABHIP_SPLINE_MFMA: spline density kernel for hidden widths padded to 256: 0 = the
                                            4x4x1 stream kernels (8 / 16 chains per workgroup by batch, or by TILE_SHAPE),
                                            16 = the 16x16x4 kernel */
#define FABHIP_OPT_SPLINE_LEAP 7         /* FABHIP_SPLINE_LEAP: fused spline transitions: 1 = one launch per leapfrog (half steps and
                                            the target inside the 4x4x1 spline density kernel, default), 0 = four launches */
#define FABHIP_OPT_FUSED_TAIL 8          /* FABHIP_FUSED_TAIL: AIS calls of <= 2048 chains: 1 = the compaction + ESS / log Z after the chain
                                            initialisation and after the last transition in ONE launch each (default), 0 = the six / five
                                            separate kernels (the same results, bit for bit) */
#define FABHIP_OPT_ADAPT_FOLD 9          /* FABHIP_ADAPT_FOLD: fused AIS calls on 4- / 8-chain tiles: 1 = the step-size rule runs in the LAST
                                            workgroup of every transition kernel (ticket; default), 0 = its own launch per transition.
                                            Spline family (round 5): 1 = an outer HMC step is L launches - begin inside the first
                                            leapfrog launch, accept + rule inside the last; 0 = L + 3 (the same results, bit for bit) */
#define FABHIP_OPT_PGRAD 10              /* FABHIP_PGRAD: parameter gradients from a RealNVP tape: 1 = the stream-K GEMM over all products of all
                                            layers (default; dense 64 x 64 / narrow tiles straight from HBM into the matrix cores, work cut
                                            into equal shares), 0 = the round-1 64 x 64 block kernel (A/B; other summation order) */
#define FABHIP_OPT_TAPE_TILES 11         /* FABHIP_TAPE_TILES: fabhip_flow_log_prob_tape: 0 = 8-chain stream tiles (flow_r8.h) where the flow has
                                            that image (default; inside fabhip_buffer_train_step with the minibatch arithmetic in the kernel's tail), 8 = the same
                                            tiles, the minibatch arithmetic as its own launch (A/B), 16 = always the 16-chain kernel (A/B; other summation order) */
#define FABHIP_OPT_COUNT 12
int fabhip_set_option(int key, int value);
int fabhip_get_option(int key);

/* Number of floats of the MFMA-tiled parameter image for a (dim, n_layers, width) flow. */
int64_t fabhip_flow_packed_floats(int32_t dim, int32_t n_layers, int32_t width);

/* Assemble W = P L U and W^-1 (fp64 triangular inverses) per layer and re-tile every matrix
 * (and its transpose) into v_mfma_f32_16x16x4_f32 B-operand order.  Call after each optimiser step. */
int fabhip_flow_pack(const fabhip_flow_params* params, float* packed, fabhip_stream_t stream);
/* Same without the W^-1 matrices (their float64 triangular inverses are 80 % of the packing time): enough for
 * fabhip_flow_log_prob / _log_prob_tape / the transition kernels between two optimiser steps of a minibatch loop;
 * fabhip_flow_sample and fabhip_ais_run need a full fabhip_flow_pack first. */
int fabhip_flow_pack_density(const fabhip_flow_params* params, float* packed, fabhip_stream_t stream);
/* The image for the TRAINING forward only (round 6): what fabhip_flow_log_prob_tape reads - the assembled affine maps, the 16-chain
 * tiles with the biases and, where the flow has them, the 8-chain stream tiles; ~1/4 of the launches of a density pack.  Used between
 * the optimiser steps of the replay-buffer minibatches (fabhip_buffer_train_step); any other consumer needs fabhip_flow_pack /
 * fabhip_flow_pack_density afterwards. */
int fabhip_flow_pack_train(const fabhip_flow_params* params, float* packed, fabhip_stream_t stream);

/* `precision`: per CALL choice between the fp32 parity kernels and fast mode (below) for the entry points that have both -
 * FABHIP_PRECISION_DEFAULT (0, what a zero-initialised struct gets) follows the process default of fabhip_set_fast_mode,
 * _FP32 / _FAST override it, so two samplers of one process can differ. */
enum { FABHIP_PRECISION_DEFAULT = 0, FABHIP_PRECISION_FP32 = 1, FABHIP_PRECISION_FAST = 2 };
typedef struct {
    int32_t dim, n_layers, width;
    int32_t precision;   /* FABHIP_PRECISION_* */
    const float* packed; /* fabhip_flow_pack output */
} fabhip_flow;

/* x, log_q = flow.sample_and_log_prob given base noise eps[B][dim] ~ N(0,1)
 * (fab/wrappers/normflows.py:16-18 -> NormalizingFlow.sample). */
int fabhip_flow_sample(const fabhip_flow* flow, const float* eps, float* x, float* log_q, int64_t B,
                       fabhip_stream_t stream);

/* log_q[B] = flow.log_prob(x[B][dim]); if grad_x != NULL also d log_q / d x [B][dim]
 * (fab/wrappers/normflows.py:23-24 and the autograd call of fab/sampling_methods/base.py:50-56). */
int fabhip_flow_log_prob(const fabhip_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                         fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Circular / linear-tail rational-quadratic spline COUPLING flow (the alanine-dipeptide flow of
 * experiments/make_flow/make_aldp_model.py:57-71,121-134,146-167: n_layers x normflows
 * CircularCoupledRationalQuadraticSpline(dim, 1 block, hidden, ind_circ, tail_bound, 8 bins, mask) with PeriodicShift /
 * PeriodicWrap layers and a UniformGaussian base; replaces NormalizingFlow.log_prob / .sample behind
 * fab/wrappers/normflows.py:16-31).  dim <= 64, hidden <= 256, 8 bins, one residual block per conditioner.
 *
 * Per layer, raw parameters in normflows / nn.Linear layout (weight[out][in]):
 *   w0 [hidden][n_id], b0 [hidden]            transform_net.initial_layer
 *   wa, wb [hidden][hidden], ba, bb [hidden]  transform_net.blocks.0.linear_layers.{0,1}
 *   wf [25 n_tr][hidden], bf [25 n_tr]        transform_net.final_layer (per transformed coordinate: 8 widths, 8 heights,
 *                                             9 knot derivatives)
 *   pfw [n_pf][2] (may be NULL)               transform_net.preprocessing.weights (periodic features)
 *   uw, uh [n_id][8], ud [n_id][9]            unconditional_transform.unnormalized_{widths,heights,derivatives}
 *   meta [12][64] floats: row 0 identity coordinate indices (n_id valid), 1 transformed coordinate indices (n_tr valid),
 *     2 circular flag per coordinate, 3 tail bound per coordinate, 4 periodic-feature flag per identity position,
 *     5 its scale pi / bound, 6 its row in pfw, 7/8 shift and on-flag applied BEFORE this layer in the log_prob direction
 *     (z <- wrap(z - shift): PeriodicShift.inverse / PeriodicWrap.inverse), 9/10 shift and on-flag applied AFTER this layer
 *     in the sampling direction (PeriodicShift.forward), 11: {n_id, n_tr, n_pf}.
 * base_scale [dim], base_circ [dim] (0/1): UniformGaussian (uniform on [-scale/2, scale/2] for circular coordinates).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dim, n_layers, hidden;
    const float* meta[FABHIP_MAX_LAYERS];
    const float* w0[FABHIP_MAX_LAYERS];
    const float* b0[FABHIP_MAX_LAYERS];
    const float* wa[FABHIP_MAX_LAYERS];
    const float* ba[FABHIP_MAX_LAYERS];
    const float* wb[FABHIP_MAX_LAYERS];
    const float* bb[FABHIP_MAX_LAYERS];
    const float* wf[FABHIP_MAX_LAYERS];
    const float* bf[FABHIP_MAX_LAYERS];
    const float* pfw[FABHIP_MAX_LAYERS];
    const float* uw[FABHIP_MAX_LAYERS];
    const float* uh[FABHIP_MAX_LAYERS];
    const float* ud[FABHIP_MAX_LAYERS];
    const float* base_scale;
    const float* base_circ;
} fabhip_spline_params;

typedef struct {
    int32_t dim, n_layers, hidden;
    int32_t precision;   /* FABHIP_PRECISION_* (the density + gradient kernel has a fast variant) */
    const float* packed; /* fabhip_spline_pack output */
} fabhip_spline_flow;

int64_t fabhip_spline_packed_floats(int32_t dim, int32_t n_layers, int32_t hidden);
/* development aid (FABHIP_TIMELINE=1): s_memtime stamps of one layer of the one-launch density kernel */
int fabhip_debug_spline_timeline(int64_t* host_out, int32_t n);
int fabhip_spline_pack(const fabhip_spline_params* params, float* packed, fabhip_stream_t stream);
size_t fabhip_spline_workspace_bytes(int32_t dim, int32_t n_layers, int32_t hidden, int64_t B, int32_t with_grad);
/* log_q[B] = flow.log_prob(x[B][dim]) and, if grad_x != NULL, d log_q / dx [B][dim] (reverse sweep through all layers). */
int fabhip_spline_log_prob(const fabhip_spline_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                           void* workspace, size_t workspace_bytes, fabhip_stream_t stream);
/* Training tape of the spline flow (the backward of `flow.log_prob(x)` w.r.t. the flow parameters in the forward-KL
 * term of the FAB loss, fab/core.py:114-127, for the normflows spline family).  One call computes log_q, d log_q / dx
 * (required) and writes, per coupling layer l at tape + l * out16[1], row-major [B][width] matrices at offsets
 *   out16[2] XI  [B][64]   raw identity coordinates         out16[3] A0  [B][64]  after the periodic features
 *   out16[4] dA0 [B][64]   cotangent of A0                  out16[5] R0  [B][Wp]  relu(h0)
 *   out16[6] R1  [B][Wp]   relu(t)                          out16[7] H1  [B][Wp]  h1 (input of the final layer)
 *   out16[8] dH1 out16[9] dT out16[10] dH0 [B][Wp]          cotangents of h1, t (ReLU-masked), h0
 *   out16[11] dP [B][NFP]  cotangent of the conditioner output (columns >= 25 n_transform are 0)
 *   out16[12] dU [B][out16[15]] cotangent of the unconditional parameters, [identity position][25]
 * with out16[0] = total floats, out16[13] = Wp, out16[14] = NFP.  All cotangents are for seed 1 per sample: the
 * gradient of sum_b c_b log q(x_b) w.r.t. a Linear's weight is (c * cotangent)^T @ activation - plain GEMMs over the
 * tape, left to the caller's BLAS (rocBLAS; fab_torch_amd/spline_flow.py uses torch.mm). */
int fabhip_spline_tape_layout(int32_t dim, int32_t n_layers, int32_t hidden, int64_t B, int64_t out16[16]);
/* The GEMMs over a tape: for L layers of one shape  C[l][p][q] = sum_b coef[b] Y[l][b][p] X[l][b][q]  (Y: cotangents
 * [B][ldy], X: activations [B][ldx], layer l at Y + l * y_layer_stride / X + l * x_layer_stride; C row-major [L][P][Q])
 * and, if colsum != NULL, the bias gradients colsum[l][p] = sum_b coef[b] Y[l][b][p] - the backward of a Linear
 * (loss.backward() through flow.log_prob, fab/train_with_prioritised_buffer.py:162-177) on the fp32 matrix cores,
 * deterministic summation order.  ldy / ldx multiples of 4 and 16-byte aligned bases take the vector path. */
int fabhip_tape_gemm(const float* Y, int64_t y_layer_stride, int32_t ldy, int32_t P, const float* X, int64_t x_layer_stride,
                     int32_t ldx, int32_t Q, const float* coef, int64_t B, int32_t L, float* C, float* colsum,
                     fabhip_stream_t stream);
int fabhip_spline_log_prob_tape(const fabhip_spline_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                                float* tape, int64_t tape_floats, void* workspace, size_t workspace_bytes,
                                fabhip_stream_t stream);
/* The backward of `x, log_q = flow.sample_and_log_prob()` w.r.t. the parameters and the base noise (the reparameterised baseline
 * losses, fab/core.py:130-152 with the spline flow of experiments/make_flow/make_aldp_model.py), by the implicit function theorem
 * on the log_prob direction S (x = S^-1(z0; theta), z0 = the base sample): for cotangents gx of x and gl of log_q (either nullable),
 *   d/d theta = gl * d log q(x) / d theta |_x  -  v_l^T dS_l / d theta_l,   v = (dS/dx)^-T (gx + gl * d log q / dx).
 * The call re-runs the sampler from (u, eps) keeping every layer's state (the sweeps linearise where the sample was made, not on a
 * trajectory re-derived from the rounded x), runs the log_prob direction's reverse sweep on them (tape_density: seed 1 per sample,
 * as fabhip_spline_log_prob_tape writes it) and carries v from the x side to the base side (tape_inverse, same layout).  The
 * parameter gradients are the tape GEMMs of tape_density with coefficients gl plus those of tape_inverse with coefficients 1.
 * v_x [B][dim]: scratch, on return v at x.  v_base (nullable) [B][dim]: v at the base side = the cotangent of z0.
 * Workspace: fabhip_spline_workspace_bytes(.., with_grad = 1); each tape fabhip_spline_tape_layout's out16[0] floats. */
int fabhip_spline_sample_vjp_tape(const fabhip_spline_flow* flow, const float* u, const float* eps, const float* gx, const float* gl,
                                  float* v_x, float* v_base, int64_t B, float* tape_density, float* tape_inverse,
                                  int64_t tape_floats, void* workspace, size_t workspace_bytes, fabhip_stream_t stream);
/* x, log_q = flow.sample given u[B][dim] ~ U(0,1) (circular coordinates) and eps[B][dim] ~ N(0,1) (the others). */
int fabhip_spline_sample(const fabhip_spline_flow* flow, const float* u, const float* eps, float* x, float* log_q,
                         int64_t B, void* workspace, size_t workspace_bytes, fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Flow training path — the backward of `flow.log_prob(x)` w.r.t. the parameters, i.e. what
 * `loss.backward()` does in fab/train_with_prioritised_buffer.py:162-177 (loss = -mean(w_adjust * log_q_x))
 * and for fab/core.py:112-118 (fab_alpha_div_inner).  Two calls:
 *   fabhip_flow_log_prob_tape : log_q (and optionally d log_q / dx) exactly as fabhip_flow_log_prob, plus a
 *                               tape of per-layer activations / deltas in caller-owned device memory;
 *   fabhip_flow_param_grad    : grads[theta] = sum_b coef[b] * d log_q(x_b) / d theta for every parameter,
 *                               coef = d loss / d log_q (autograd's grad_output), written into one flat image:
 *                               per layer [w1 | b1 | w2 | b2 | w3 | b3 | L | U | log_S] (each in the shape of
 *                               the fabhip_flow_params tensor of that name), then loc, log_scale; when
 *                               params->an_s is set (ActNorm flows) then per layer [an_s | an_t] - `grads` must
 *                               then hold out15[14] floats instead of out15[12].
 * fabhip_flow_grad_layout fills {layer_stride, w1, b1, w2, b2, w3, b3, L, U, log_S (offsets inside a layer
 * block), loc, log_scale (absolute offsets), total, an_base (absolute offset of layer 0's [an_s | an_t] pair, the
 * pairs are 2 dim floats apart), total with ActNorm}, all in floats.
 * ---------------------------------------------------------------------------------------- */
int64_t fabhip_flow_grad_floats(int32_t dim, int32_t n_layers, int32_t width);
int fabhip_flow_grad_layout(int32_t dim, int32_t n_layers, int32_t width, int64_t* out15);
size_t fabhip_flow_tape_bytes(int32_t dim, int32_t n_layers, int32_t width, int64_t B);
/* Layout of the tape (floats): row-major [Bp x width] matrices per layer block, then the base block TB.
 * out18 = {Bp, wz, w1, wh, wp, we, wb (row widths of ZA/GZ, Z1, H1/H2, DP, E1/E2, TB),
 *          o_ZA, o_GZ, o_Z1, o_H1, o_H2, o_DP, o_E2, o_E1 (offsets inside a layer block), layer_stride, o_TB, total}.
 * H1 / H2 hold the hidden activations AFTER the ReLU (their first `width` columns), i.e. the ReLU decisions the
 * gradients were computed with. */
int fabhip_flow_tape_layout(int32_t dim, int32_t n_layers, int32_t width, int64_t B, int64_t* out18);
int fabhip_flow_log_prob_tape(const fabhip_flow* flow, const float* x, float* log_q, float* grad_x, int64_t B,
                              void* tape, size_t tape_bytes, fabhip_stream_t stream);
/* The same with the batch read in place from a larger matrix: batch row g is row rows[g] of x (`self.buffer.x[indices]` of
 * fab/utils/prioritised_replay_buffer.py:98 without materialising the gather). */
int fabhip_flow_log_prob_tape_rows(const fabhip_flow* flow, const float* x, const int64_t* rows, float* log_q, float* grad_x,
                                   int64_t B, void* tape, size_t tape_bytes, fabhip_stream_t stream);
int fabhip_flow_param_grad(const fabhip_flow_params* params, const fabhip_flow* flow, const void* tape,
                           size_t tape_bytes, const float* coef, int64_t B, float* grads, fabhip_stream_t stream);
/* Backward of the SAMPLING direction - `loss.backward()` through `flow.sample_and_log_prob` for the reparameterised
 * baseline losses flow_reverse_kl / flow_alpha_2_div_nis (fab/core.py:130-152).  Given x = the samples fabhip_flow_sample
 * returned, grad_x = d loss / dx [B][dim] and grad_log_q = d loss / d log_q [B], one sweep x -> eps writes the same tape
 * as fabhip_flow_log_prob_tape (same size and layout) with the sampling direction's cotangents, such that
 * fabhip_flow_param_grad(params, flow, tape, coef = ones[B], B, grads) returns d loss / d theta.  grad_eps (nullable)
 * receives d loss / d eps.  `flow->packed` must come from fabhip_flow_pack (with the inverses). */
int fabhip_flow_sample_grad_tape(const fabhip_flow* flow, const float* x, const float* grad_x, const float* grad_log_q,
                                 float* grad_eps, int64_t B, void* tape, size_t tape_bytes, fabhip_stream_t stream);

/* One optimiser step on a flat parameter image (the layout of fabhip_flow_grad_layout): global-norm clipping
 * (torch.nn.utils.clip_grad_norm_(params, max_norm), fab/train_with_prioritised_buffer.py:174; max_norm <= 0 = off)
 * followed by Adam (torch.optim.Adam defaults: no weight decay, no amsgrad).  The total gradient norm is written
 * to grad_norm_out (device, may be NULL); when it is not finite nothing is updated — the reference's "nan grad
 * norm" skip (:175-179) without a host round trip.  m, v: Adam moments, same length.  step_count: device int32,
 * the number of applied steps so far (drives the bias correction); incremented here when the update is applied. */
size_t fabhip_adam_workspace_bytes(int64_t n);
int fabhip_adam_clip_step(float* theta, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                          float beta2, float eps, int32_t* step_count, float max_norm, float* grad_norm_out,
                          void* workspace, size_t workspace_bytes, fabhip_stream_t stream);

/* ONE gradient step on ONE minibatch of the prioritised replay buffer - the body of the minibatch loop of
 * fab/train_with_prioritised_buffer.py:158-185 with the buffer's `adjust` (fab/utils/prioritised_replay_buffer.py:117-131),
 * enqueued without a host synchronisation:
 *   (repack != 0) the training image of the current parameters (fabhip_flow_pack_train);
 *   log_q = flow.log_prob(x) with the tape - x either [B][dim] (rows == NULL) or the buffer's x with the minibatch's row indices;
 *   log_w_adjust = (1 - alpha) (log_q - log_q_old), w = clip(exp(log_w_adjust), max = w_adjust_max_clip) (<= 0: no clip),
 *   loss = -mean(w log_q), coef = -w / B (w is detached in the reference's loss; NaN-poisoned when the loss is not finite, which the
 *   optimiser's finite-norm test turns into the reference's skipped update, :172-181);
 *   (buf_log_w != NULL) buffer.adjust on the minibatch's rows: log_w += log_w_adjust and log_q_old = log_q where both are finite,
 *   log_w = -inf elsewhere (the reference adjusts after the optimiser step with the same values, :184-185; needs unique rows:
 *   sampling without replacement);
 *   grads = sum_b coef_b d log q(x_b) / d theta (fabhip_flow_param_grad), then fabhip_adam_clip_step on (theta, m, v).
 * `params` point INTO theta (the flat layout of fabhip_flow_grad_layout: FlatAdam).  log_q_old: [B] (log_q_old_rows = 0) or the
 * buffer's log_q_old, read at rows[b] (= 1).  stats[8] (device): loss, mean / min / max of exp(log_w_adjust) before the clip,
 * mean(log_q), gradient norm - the reference's logging keys (:188-196).  struct_bytes = sizeof(fabhip_train_step_args). */
typedef struct {
    size_t struct_bytes;
    const fabhip_flow_params* params;
    float* packed;
    int32_t repack, log_q_old_rows;
    const float* x;
    const int64_t* rows;
    const float* log_q_old;
    int64_t B;
    float alpha, w_adjust_max_clip;
    float *buf_log_w, *buf_log_q_old;
    float *log_q, *log_w_adjust, *coef, *grads, *stats;      /* outputs: [B], [B], [B], [n_params], [8] */
    float *theta, *m, *v;
    int64_t n_params;
    float lr, beta1, beta2, eps, max_grad_norm;
    int32_t* step_count;
    void* workspace;                                         /* 256-byte aligned */
    size_t workspace_bytes;
} fabhip_train_step_args;
/* PrioritisedReplayBuffer.add (fab/utils/prioritised_replay_buffer.py:71-85) in one launch: the n new rows go to rows
 * (start + i) mod max_length of the ring (x [max_length][dim], log_w, log_q_old [max_length]). */
int fabhip_buffer_add(const float* x, const float* log_w, const float* log_q_old, int64_t n, int32_t dim, int64_t start,
                      int64_t max_length, float* buf_x, float* buf_log_w, float* buf_log_q_old, fabhip_stream_t stream);
/* PrioritisedReplayBuffer.sample's row selection without replacement (:10-17, :88-97) given the uniform draws: the k rows with
 * the largest log_w + Gumbel(u_gumbel[n]) (fabhip_topk), in a pseudo-random order (`indices[torch.randperm(k)]`): a keyed
 * bijection of [0, k) - four Feistel rounds with cycle walking - whose round keys come from the four uniforms u_order[4]. */
size_t fabhip_buffer_sample_workspace_bytes(int64_t n, int64_t k);
int fabhip_buffer_sample(const float* log_w, const float* u_gumbel, const float* u_order, int64_t n, int64_t k, int64_t* idx_out,
                         void* workspace, size_t workspace_bytes, fabhip_stream_t stream);
size_t fabhip_train_step_workspace_bytes(int32_t dim, int32_t n_layers, int32_t width, int64_t B, int64_t n_params);
int fabhip_buffer_train_step(const fabhip_train_step_args* args, fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Targets (fab/target_distributions/many_well.py:81-90, double_well.py:44-58, gmm.py:57-66)
 * ---------------------------------------------------------------------------------------- */
enum { FABHIP_TARGET_MANYWELL = 1, FABHIP_TARGET_GMM = 2 };

typedef struct {
    int32_t kind, dim;
    float a, b, c;       /* many well: energy a x + b x^2 + c x^4 on even dims, x^2/2 on odd     */
    float log_norm;      /* subtracted from log p (0 unless `normalised`)                        */
    int32_t n_mix;       /* GMM: number of equally weighted components                           */
    const float* locs;   /* GMM: [n_mix][dim]                                                    */
    const float* scales; /* GMM: [n_mix][dim] diagonal of scale_tril                             */
} fabhip_target;

int fabhip_target_log_prob(const fabhip_target* target, const float* x, float* log_p, float* grad_x,
                           int64_t B, fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Point (fab/sampling_methods/base.py:7-47) as a struct of device arrays.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float* x;          /* [B][dim] */
    float* log_q;      /* [B]      */
    float* log_p;      /* [B]      */
    float* grad_log_q; /* [B][dim] or NULL (Metropolis) */
    float* grad_log_p; /* [B][dim] or NULL              */
} fabhip_point;

/* create_point(x, ...) with gradients: fab/sampling_methods/base.py:59-72. point.x is read. */
int fabhip_create_point(const fabhip_flow* flow, const fabhip_target* target, const fabhip_point* point,
                        int32_t with_grad, int64_t B, fabhip_stream_t stream);

/* Coefficients of the annealed density (fab/sampling_methods/base.py:76-118), float32:
 *   log pi_beta = c_q log_q + c_p log_p ;  grad = g_q grad_log_q + g_p grad_log_p
 * (g_p = 2 beta when not p_target: the reference's hard-coded factor, base.py:116). */
/* C callers: ZERO-INITIALISE fabhip_flow / fabhip_spline_flow (`fabhip_flow f = {0};`) before filling them in.  `precision`
 * (ABI 207) sits where padding used to be, so sizeof did not change and fabhip_abi_sizes cannot tell a caller compiled against
 * the older header apart - only FABHIP_ABI_VERSION can; an uninitialised value of 1 / 2 would silently select a precision. */
typedef struct {
    float c_q, c_p, g_q, g_p;
} fabhip_anneal;
void fabhip_anneal_coefs(double beta, double alpha, int32_t p_target, fabhip_anneal* out);

/* ------------------------------------------------------------------------------------------
 * HMC transition  (fab/sampling_methods/transition_operators/hmc.py:105-202)
 * One call = HamiltonianMonteCarlo.transition(point, i, beta): n_outer x (L leapfrogs,
 * Metropolis accept, in-place commit, step-size adaptation) and, if `log_w`, the AIS
 * log-weight increment of ais.py:93-100 with the coefficients `next`.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    fabhip_flow flow;
    fabhip_target target;
    fabhip_point point;     /* in/out, mutated in place like the reference           */
    int64_t B;              /* rows allocated                                        */
    const int32_t* n_valid; /* device scalar: rows in use (NULL -> B)                */
    fabhip_anneal cur;      /* beta_j                                                */
    fabhip_anneal next;     /* beta_{j+1} (used when log_w != NULL)                  */
    float* log_w;           /* [B] in/out or NULL                                    */
    const float* noise_p;   /* [n_outer][B][dim] ~ N(0,1)   (hmc.py:134)             */
    const float* noise_e;   /* [n_outer][B]      ~ Exp(1)   (hmc.py:118)             */
    float* epsilons;        /* -> epsilons[i-1][0..n_outer)  (state buffer, updated) */
    float* common_epsilon;  /* [1] state buffer, updated                             */
    const float* mass;      /* [dim]                                                 */
    int32_t n_outer, L;
    float max_grad, target_p_accept;
    int32_t tune;           /* 1 = adapt step sizes (not eval_mode)                  */
    float* p_accept;        /* [n_outer] out (mean acceptance prob) or NULL          */
    float* avg_distance;    /* [1] out, store_info's distance statistic, or NULL     */
    void* workspace;
    size_t workspace_bytes;
    /* Sharded chains: NULL = adapt from this call's own chains (above).  Otherwise the step sizes, p_accept
     * and avg_distance are left alone and the call publishes its acceptance statistics as ONE slab of
     * fabhip_hmc_partials_floats(B) floats - per 16-chain block the sum of min(1, acceptance) and of store_info's
     * distance, then the number of chains in use: [acc[nblk] | dist[nblk] | n] - which the caller all-gathers over the
     * ranks and hands to fabhip_hmc_adapt_gathered.  Requires n_outer == 1 (every shipped config, setup_run.py:190). */
    float* partials;
} fabhip_hmc_args;

size_t fabhip_hmc_workspace_bytes(int64_t B, int32_t dim, int32_t n_outer);
int fabhip_hmc_transition(const fabhip_hmc_args* args, fabhip_stream_t stream);

/* Step-size adaptation of hmc.py:122-123,162-170 on the mean acceptance of ALL chains of a sharded batch: `gathered` =
 * the partials slabs of n_ranks ranks in rank order (each fabhip_hmc_partials_floats(B_rank) floats, equal B_rank).
 * The sums run over ranks, then blocks, in order - with shards that are multiples of 16 chains exactly the sequence a
 * single device adds for the whole batch, so every rank ends up with the single-device step sizes bit for bit. */
int64_t fabhip_hmc_partials_floats(int64_t B);
int fabhip_hmc_adapt_gathered(const float* gathered, int32_t n_ranks, int64_t B_rank, float* epsilon, float* common_epsilon,
                              float target_p_accept, int32_t tune, float* p_accept, float* avg_distance,
                              fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Metropolis transition (fab/sampling_methods/transition_operators/metropolis.py:51-74)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    fabhip_flow flow;
    fabhip_target target;
    fabhip_point point;      /* grads unused */
    int64_t B;
    const int32_t* n_valid;
    fabhip_anneal cur, next;
    float* log_w;
    const float* noise_x;    /* [n_updates][B][dim] ~ N(0,1)  (metropolis.py:57) */
    const float* noise_u;    /* [n_updates][B]      ~ U(0,1)  (metropolis.py:65) */
    float* noise_scalings;   /* -> noise_scalings[i-1][0..n_updates), updated    */
    int32_t n_updates;
    float target_p_accept;
    int32_t tune;            /* adjust_step_size and not eval_mode               */
    void* workspace;
    size_t workspace_bytes;
} fabhip_metropolis_args;

/* Noise-scaling adaptation of metropolis.py:68-73 on the mean acceptance of ALL chains of a sharded batch.  The scaling
 * of (transition i, update n) is read by that update only and adjusted right after it: nothing later in the same AIS call
 * reads the adjusted value, so a sharded call defers the rule of all M transitions to ONE gather at its end.  `gathered` =
 * the slabs of n_ranks ranks in rank order, each fabhip_metropolis_partials_floats(B_rank, M, n_updates) floats:
 * per transition [n_updates][ceil(B_rank / 16)] sums of min(1, acceptance) per 16-chain block, then the chains in use
 * (fabhip_ais_phase with `partials`).  Sums run over ranks, then blocks, in order: with shards that are multiples of 16
 * chains exactly what one device adds, so every rank ends with the single-device noise_scalings [M][n_updates] bit for bit. */
int64_t fabhip_metropolis_partials_floats(int64_t B, int32_t M, int32_t n_updates);
int fabhip_metropolis_adapt_gathered(const float* gathered, int32_t n_ranks, int64_t B_rank, int32_t M, int32_t n_updates,
                                     float* noise_scalings, float target_p_accept, int32_t tune, fabhip_stream_t stream);

size_t fabhip_metropolis_workspace_bytes(int64_t B, int32_t dim, int32_t n_updates);
int fabhip_metropolis_transition(const fabhip_metropolis_args* args, fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Generic plug-in path: transitions for ANY `Distribution` / `LogProbFunc` plug-in (fab/types_.py:5-27,
 * fab/sampling_methods/ais.py:22-30).  The caller evaluates log q, log p and their gradients with the plug-ins' own
 * code (create_point, fab/sampling_methods/base.py:50-72); everything else of a transition runs here as
 * elementwise kernels.  One workspace of fabhip_generic_workspace_bytes(B, dim) carries the trajectory state
 * (x, p, grad U, -U-K of the current point, acceptance partials) between the calls of one outer step:
 *
 *   HMC outer step n (hmc.py:129-160):
 *     fabhip_hmc_generic_begin(start, cur, ...)        p0 = noise_p * mass, grad U(start), x <- start.x
 *     L x { fabhip_hmc_generic_leap_pre(... x_out)     p -= eps gradU/2 ; x += eps/mass p ; x_out <- x
 *           [caller: log q, log p, gradients at x_out]
 *           fabhip_hmc_generic_leap_post(gq, gp, ...)  gradU <- clamp(-(g_q gq + g_p gp)) ; p -= eps gradU/2 }
 *     fabhip_hmc_generic_accept(prop, cur, ...)        Metropolis test with noise_e, in-place commit into `cur`,
 *                                                      log_w increment (if log_w), step-size adaptation, logging
 *   (`start` = `cur` for n = 0; for n > 0 the reference continues from the previous PROPOSAL, hmc.py:133-142.)
 *
 *   Metropolis update n (metropolis.py:51-74): fabhip_metropolis_generic_propose -> [caller: log q, log p at x_new]
 *     -> fabhip_metropolis_generic_accept (prev_log_prob = fabhip_anneal_log_prob of the point BEFORE the first
 *     update, never refreshed: the reference's stale x_prev_log_prob).
 * ---------------------------------------------------------------------------------------- */
size_t fabhip_generic_workspace_bytes(int64_t B, int32_t dim);
int fabhip_hmc_generic_begin(const fabhip_point* start, const fabhip_point* cur, int64_t B, int32_t dim,
                             fabhip_anneal c, const float* noise_p, const float* mass, float max_grad,
                             void* workspace, size_t workspace_bytes, fabhip_stream_t stream);
int fabhip_hmc_generic_leap_pre(int64_t B, int32_t dim, const float* eps_ptr, const float* ceps_ptr, const float* mass,
                                float* x_out, void* workspace, size_t workspace_bytes, fabhip_stream_t stream);
int fabhip_hmc_generic_leap_post(int64_t B, int32_t dim, const float* grad_log_q, const float* grad_log_p,
                                 fabhip_anneal c, float max_grad, const float* eps_ptr, const float* ceps_ptr,
                                 void* workspace, size_t workspace_bytes, fabhip_stream_t stream);
int fabhip_hmc_generic_accept(const fabhip_point* prop, const fabhip_point* cur, int64_t B, int32_t dim, fabhip_anneal c,
                              fabhip_anneal next, float* log_w, const float* noise_e, const float* mass,
                              float* eps_ptr, float* ceps_ptr, float target_p_accept, int32_t tune, float* p_accept,
                              float* avg_distance, void* workspace, size_t workspace_bytes, fabhip_stream_t stream);
/* out[i] = c_q log_q[i] + c_p log_p[i]  (get_intermediate_log_prob, base.py:76-97) */
int fabhip_anneal_log_prob(const float* log_q, const float* log_p, int64_t n, fabhip_anneal c, float* out,
                           fabhip_stream_t stream);
/* log_w += pi_next(point) - pi_c(point)  (ais.py:93-100) */
int fabhip_log_w_update(const float* log_q, const float* log_p, int64_t n, fabhip_anneal c, fabhip_anneal next,
                        float* log_w, fabhip_stream_t stream);
int fabhip_metropolis_generic_propose(const float* x, const float* noise_x, const float* scale_ptr, int64_t B, int32_t dim,
                                      float* x_new, fabhip_stream_t stream);
int fabhip_metropolis_generic_accept(const float* x_new, const float* new_log_q, const float* new_log_p,
                                     const fabhip_point* cur, const float* prev_log_prob, const float* noise_u,
                                     int64_t B, int32_t dim, fabhip_anneal c, float* scale_ptr, float target_p_accept,
                                     int32_t tune, void* workspace, size_t workspace_bytes, fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole AIS call  (fab/sampling_methods/ais.py:53-105): sample the flow, create the point,
 * initial log-weights, NaN/inf filtering (stable compaction, ais.py:190-213), M transitions
 * with log-weight accumulation, final filtering, ESS / log Z statistics — all enqueued by
 * ONE host call, no host synchronisation inside.
 * ---------------------------------------------------------------------------------------- */
enum { FABHIP_TRANSITION_HMC = 1, FABHIP_TRANSITION_METROPOLIS = 2 };

typedef struct {
    fabhip_flow flow;
    fabhip_target target;
    int64_t B;                /* requested batch size                                        */
    int32_t M;                /* n_intermediate_distributions                                */
    const double* betas;      /* host, [M+2] (B_space, ais.py:108-129)                       */
    double alpha;
    int32_t p_target;
    int32_t transition;       /* FABHIP_TRANSITION_*                                         */
    const float* eps0;        /* [B][dim] base noise                                         */
    const float* noise_a;     /* HMC: [M][n_inner][B][dim] momenta ; Metropolis: proposals   */
    const float* noise_b;     /* HMC: [M][n_inner][B] Exp(1)       ; Metropolis: U(0,1)      */
    float* step_state;        /* HMC: epsilons[M][n_outer] ; Metropolis: noise_scalings[M][n_updates] */
    float* common_epsilon;    /* HMC only                                                    */
    const float* mass;        /* HMC only                                                    */
    int32_t n_inner;          /* n_outer (HMC) / n_updates (Metropolis)                      */
    int32_t L;
    float max_grad, target_p_accept;
    int32_t tune;
    fabhip_point point;       /* out [B] rows; first n_valid[1] rows are the result          */
    float* log_w;             /* out [B]                                                     */
    int32_t* n_valid;         /* out device int32[2]: rows after "chain init" / "chain end"  */
    float* stats;             /* out device float[16]: [0] ess_base [1] - [2] rows after init
                                 [3] ess_ais [4] log_Z [5] rows at chain end, [6..15] reserved (the init phase
                                 writes zeros): the caller need not initialise any of it                 */
    /* HMC logging slots (hmc.py:173-183, store_info), device, each may be NULL: mean acceptance probability of
     * every outer loop [n_inner] and store_info's mean distance [1], for the first (i = 1) and the last (i = M)
     * intermediate distribution.  With M = 1 only the "first" pair is written, like the reference. */
    float* p_accept_first;
    float* p_accept_last;
    float* avg_distance_first;
    float* avg_distance_last;
    /* Evaluation outputs (ais.py:152-166, generate_eval_data), each may be NULL: the chains' starting points after
     * the "chain init" filtering, base_x [B][dim], and their importance weights w.r.t. the target,
     * base_log_w [B] = log p(x0) - log q(x0) with log q as returned by the flow's sampling pass; the first
     * n_valid[0] rows are the result. */
    float* base_x;
    float* base_log_w;
    void* workspace;
    size_t workspace_bytes;
} fabhip_ais_args;

size_t fabhip_ais_workspace_bytes(int64_t B, int32_t dim, int32_t n_inner);
int fabhip_ais_run(const fabhip_ais_args* args, fabhip_stream_t stream);

/* The same call in pieces, for chains sharded over ranks with exact step-size adaptation: `phases` selects
 * FABHIP_AIS_INIT (chain initialisation, "chain init" filter, base ESS), the transitions j_begin .. j_end (1-based,
 * j_begin > j_end: none) and FABHIP_AIS_FINISH ("chain end" filter, ESS / log Z).  point / log_w / n_valid / stats are
 * in/out across the calls of one AIS run, eps0 is read by INIT only.  With `partials` != NULL (HMC, n_inner == 1,
 * j_begin == j_end) the transition leaves the step sizes alone and publishes its acceptance slab instead
 * (fabhip_hmc_args.partials); fabhip_ais_run(args) == fabhip_ais_phase(args, INIT | FINISH, 1, M, NULL).
 * Metropolis with `partials` != NULL (any j range, one call may hold the whole run): transition j leaves its noise scalings
 * alone and writes its block sums + chain count at partials + (j - 1) (n_inner nblk + 1) - the layout of
 * fabhip_metropolis_partials_floats(B, M, n_inner) - for fabhip_metropolis_adapt_gathered after the call. */
enum { FABHIP_AIS_INIT = 1, FABHIP_AIS_FINISH = 2,
       FABHIP_AIS_CONTINUE = 4 };   /* this call continues a run whose INIT phase was enqueued with the SAME workspace on the same
                                      * stream and nothing else touched the workspace since (lets the later phases keep using the
                                      * state INIT left there: the ticket of the in-kernel step-size rule).  noise_a / noise_b are
                                      * read by the transitions only: a call without transitions may pass NULL (a host can draw the
                                      * transition noise while the device runs the chain initialisation). */
int fabhip_ais_phase(const fabhip_ais_args* args, int32_t phases, int32_t j_begin, int32_t j_end, float* partials,
                     fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same with the RQ-spline flow (fabhip_spline_*) as base distribution
 * ---------------------------------------------------------------------------------------- */
/* One HMC transition (hmc.py:129-160: n_outer x (momentum refresh, L leapfrogs, Metropolis test, in-place commit,
 * step-size adaptation) + the AIS log-weight increment when log_w != NULL) with the SPLINE flow as base distribution and a
 * native target, enqueued by one host call: per leapfrog the generic element-wise kernels, the one-launch spline density +
 * gradient kernel and the target kernel.  Same argument meaning as fabhip_hmc_args; `n_valid` (device scalar, may be
 * NULL) = rows in use.  workspace: fabhip_spline_hmc_workspace_bytes. */
typedef struct {
    fabhip_spline_flow flow;
    fabhip_target target;
    fabhip_point point;     /* in/out */
    int64_t B;
    const int32_t* n_valid;
    fabhip_anneal cur, next;
    float* log_w;
    const float* noise_p;   /* [n_outer][B][dim] */
    const float* noise_e;   /* [n_outer][B]      */
    float* epsilons;        /* -> epsilons[i-1][0..n_outer) */
    float* common_epsilon;
    const float* mass;
    int32_t n_outer, L;
    float max_grad, target_p_accept;
    int32_t tune;
    float* p_accept;        /* [n_outer] or NULL */
    float* avg_distance;    /* [1] or NULL       */
    void* workspace;
    size_t workspace_bytes;
} fabhip_spline_hmc_args;
size_t fabhip_spline_hmc_workspace_bytes(int32_t dim, int32_t n_layers, int32_t hidden, int64_t B);
int fabhip_spline_hmc_transition(const fabhip_spline_hmc_args* args, fabhip_stream_t stream);

/* AnnealedImportanceSampler.sample_and_log_weights (ais.py:53-105) with the spline flow as base distribution, a native
 * target and HMC transitions - the spline family's fabhip_ais_run: flow sample, point creation (log q re-evaluated
 * through log_prob, base.py:65-68), initial log-weights, "chain init" filter, base ESS, M transitions, "chain end"
 * filter, ESS / log Z, enqueued by ONE host call without host synchronisation.  u0 / eps0 [B][dim]: the uniforms /
 * normals of the flow's base sample (fabhip_spline_sample); everything else as in fabhip_ais_args. */
typedef struct {
    fabhip_spline_flow flow;
    fabhip_target target;
    int64_t B;
    int32_t M;
    const double* betas;
    double alpha;
    int32_t p_target;
    const float* u0;
    const float* eps0;
    const float* noise_p;   /* [M][n_outer][B][dim] */
    const float* noise_e;   /* [M][n_outer][B]      */
    float* epsilons;        /* [M][n_outer]         */
    float* common_epsilon;
    const float* mass;
    int32_t n_outer, L;
    float max_grad, target_p_accept;
    int32_t tune;
    fabhip_point point;     /* out */
    float* log_w;
    int32_t* n_valid;       /* out device int32[2] */
    float* stats;           /* out device float[16], layout of fabhip_ais_args.stats */
    float* p_accept_first;
    float* p_accept_last;
    float* avg_distance_first;
    float* avg_distance_last;
    float* base_x;          /* optional, see fabhip_ais_args */
    float* base_log_w;
    void* workspace;
    size_t workspace_bytes;
} fabhip_spline_ais_args;
size_t fabhip_spline_ais_workspace_bytes(int32_t dim, int32_t n_layers, int32_t hidden, int64_t B);
int fabhip_spline_ais_run(const fabhip_spline_ais_args* args, fabhip_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * ESS / log Z  (fab/utils/numerical.py:18-23, fab/sampling_methods/ais.py:80-86)
 * out[0] = normalised ESS of log_w[0..n), out[1] = logsumexp(log_w) - log(n_norm), out[2] = n used.
 * n is read from n_ptr (device) when n_ptr != NULL.
 * ---------------------------------------------------------------------------------------- */
size_t fabhip_ess_workspace_bytes(int64_t n);
int fabhip_ess_logz(const float* log_w, int64_t n, const int32_t* n_ptr, double n_norm, float* out,
                    void* workspace, size_t workspace_bytes, fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Resampling  (fab/sampling_methods/base.py:121-124 -> torch.multinomial)
 * ---------------------------------------------------------------------------------------- */
/* Bit-exact restatement of torch's CPU multinomial-with-replacement given probs[n] (float32) and
 * the float64 uniforms it consumed: sequential fp32 cumsum, /sum, last bucket = 1, lower bound. */
size_t fabhip_multinomial_torch_workspace_bytes(int64_t n);
int fabhip_multinomial_torch(const float* probs, int64_t n, const double* u, int64_t n_samples,
                             int64_t* idx, void* workspace, size_t workspace_bytes,
                             fabhip_stream_t stream);

/* Scalable fixed-point CDF resamplers from log-weights (any n): p_i = exp_spec(w_i - max w) (a specified fp32
 * exponential, oracle/numerical.py), q_i = floor(p_i 2^36), C = inclusive integer prefix sum.
 *   multinomial: idx_k = first j with C_j > floor(u_k * C_total)                (decoupled look-back scan + search)
 *   systematic : idx_k = first j with C_j > floor((k + u0) * (C_total / n_samples))
 *                fused: wave-tile sums -> tile prefix -> every tile emits the strata that fall inside its CDF range;
 *                the CDF never touches HBM (12n + 8 n_samples bytes of traffic).                              */
size_t fabhip_resample_workspace_bytes(int64_t n);
/* The fixed-point inclusive CDF C[0..n) (uint64) of log-weights, built in the workspace by ONE decoupled-look-back scan
 * pass (4n bytes read, 8n written) after a max pass (4n read); *cdf_out (host pointer-to-device-pointer, may be
 * NULL) receives its device address inside the workspace.  reuse_max != 0 skips the max pass and uses the maximum a
 * previous call on the same log_w left in this workspace (measurement of the scan pass alone). */
int fabhip_fixed_cdf(const float* log_w, int64_t n, int32_t reuse_max, const uint64_t** cdf_out, void* workspace,
                     size_t workspace_bytes, fabhip_stream_t stream);
int fabhip_resample_multinomial(const float* log_w, int64_t n, const double* u, int64_t n_samples,
                                int64_t* idx, void* workspace, size_t workspace_bytes,
                                fabhip_stream_t stream);
int fabhip_resample_systematic(const float* log_w, int64_t n, double u0, int64_t n_samples, int64_t* idx,
                               void* workspace, size_t workspace_bytes, fabhip_stream_t stream);
/* dst[k][:] = src[idx[k]][:]  (Point.__getitem__ / tensor indexing used by resample) */
int fabhip_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t n_out, int64_t row_len,
                       fabhip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Top-k selection — the `torch.topk(gumbel + logits, n)` of the prioritised buffer's sampling without
 * replacement (fab/utils/prioritised_replay_buffer.py:10-17, called from :100-110 with n = n_batches * batch).
 * sorted = 0: idx_out = the indices of the k largest keys in ascending index order (ties at the threshold: lowest
 * indices first) — all the buffer needs, it permutes them randomly afterwards; any k <= n.
 * sorted = 1: descending key order, ties by ascending index (torch.topk with sorted=True up to its unspecified tie
 * order), k <= 16384 (one LDS-resident sort, else FABHIP_ENOTSUP); key_out (optional) = those keys.
 * NaN ranks above +inf as in torch; n < 2^32.
 * ---------------------------------------------------------------------------------------- */
size_t fabhip_topk_workspace_bytes(int64_t n, int64_t k);
int fabhip_topk(const float* keys, int64_t n, int64_t k, int32_t sorted, int64_t* idx_out, float* key_out,
                void* workspace, size_t workspace_bytes, fabhip_stream_t stream);

/* Diagnostics (development only): with FABHIP_TIMELINE=1 in the environment fabhip_flow_log_prob records
 * s_memtime stamps at the stage boundaries of one forward and one backward layer of workgroup 0;
 * this call copies the first n (<= 64) stamps to the host (synchronises). */
int fabhip_debug_timeline(int64_t* host_out, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* FABHIP_H */
