// Shared by the training-path translation units (train_kernels.hip: tape forward of the 16-chain tiles, LU chain rule, Adam;
// train_step.hip: the stream-K parameter-gradient GEMM, the 8-chain-tile tape forward, the replay-buffer minibatch step).
#pragma once
#include "fabhip_common.h"
#include "launch.h"

namespace fab {

// ---- flat gradient image: per layer [w1 | b1 | w2 | b2 | w3 | b3 | L | U | log_S], then loc, log_scale; flows with
// ActNorm layers: then per layer [an_s | an_t] from `an_base` (= total without ActNorm) on
struct GradLayout {
    long layer_stride, w1, b1, w2, b2, w3, b3, L, U, logS, loc, log_scale, total, an_base, total_an;
};

FAB_HD GradLayout make_grad_layout(const FlowDims& f) {
    GradLayout g;
    long o = 0;
    g.w1 = o; o += (long)f.W * f.d;
    g.b1 = o; o += f.W;
    g.w2 = o; o += (long)f.W * f.W;
    g.b2 = o; o += f.W;
    g.w3 = o; o += (long)2 * f.DO * f.W;
    g.b3 = o; o += 2 * f.DO;
    g.L = o; o += (long)f.D * f.D;
    g.U = o; o += (long)f.D * f.D;
    g.logS = o; o += f.D;
    g.layer_stride = o;
    g.loc = (long)f.K * o;
    g.log_scale = g.loc + f.D;
    g.total = g.log_scale + f.D;
    g.an_base = g.total;
    g.total_an = g.total + (long)f.K * 2 * f.D;
    return g;
}

constexpr int LBATCH_T = 16;       // layers per launch of the LU chain-rule kernel (pointer tables travel as kernel arguments)
struct AffineSrc {
    const float *L, *U, *logS, *signS, *P, *an_s, *an_t;
};
struct AffineSrcTab {
    const float *L[LBATCH_T], *U[LBATCH_T], *logS[LBATCH_T], *signS[LBATCH_T], *P[LBATCH_T], *an_s[LBATCH_T], *an_t[LBATCH_T];
};

// floats behind the tape proper: per-layer affine dW scratch + sum(coef)
static inline size_t tape_floats(const FlowDims& f, const TapeDims& td) {
    return (size_t)td.total + (size_t)f.K * td.wz * td.wz + 16;
}
static inline size_t tape_ga_offset(const TapeDims& td) { return (size_t)td.total; }

// all weight / bias gradients + the affine dW scratch (ga_ws) + the base distribution's gradients from a tape (train_step.hip)
int launch_param_grad_tiles(const FlowDims& f, const TapeDims& td, const GradLayout& gl, const float* tape, const float* coef,
                            long B, float* grads, float* ga_ws, hipStream_t st);
// The arithmetic between `log_q_x = flow.log_prob(x)` and `loss.backward()` of one replay-buffer minibatch, and the buffer's
// `adjust`, in the TAIL of the 8-chain tape kernel (each workgroup has the log q of its 8 rows): per row log_w_adjust, the weight
// w (coef = -w / B), the buffer update; per wave (4 rows) the partial sums of the loss and of the logging statistics.  The optimiser
// step's first launch adds the partials in order and decides the skip of a non-finite loss.  coef == nullptr: no tail.
struct MbTail {
    const float* log_q_old;            // the buffer's log_q_old (read at rows[g]) or, rows_old == 0, the minibatch's own [B]
    int rows_old;
    float one_minus_alpha, w_clip, neg_inv_B;
    float *coef, *log_w_adjust;        // [B] out
    float *buf_log_w, *buf_log_q_old;  // adjusted in place at rows[g] (nullptr: no adjust)
    float* partials;                   // [2 * workgroups][8]: sum w lq | sum exp(adj) | sum lq | min | max | saw NaN
};
constexpr int MB_PART = 8;

// the 8-chain-tile tape forward (train_step.hip); FABHIP_ENOTSUP where the 8-chain image does not exist
int launch_log_prob_tape_r8(const FlowDims& f, const TapeDims& td, const float* packed, const float* x, const int64_t* rows,
                            float* log_q, float* grad, float* tape, long B, hipStream_t st, const MbTail* mb = nullptr);
// fabhip_adam_clip_step with the minibatch's loss partials (train_kernels.hip): the norm launch's block 0 adds them, writes
// stats[0 .. 4] (loss, mean / min / max of exp(log_w_adjust), mean log q) and the update is skipped when the loss is not finite
int adam_clip_step_impl(float* theta, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        int32_t* step_count, float max_norm, float* grad_norm_out, void* workspace, size_t workspace_bytes,
                        const float* mb_partials, int n_partials, float* stats, long B, hipStream_t st);
// LU chain rule of the InvertibleAffine layers (+ ActNorm) from ga_ws (train_kernels.hip)
int launch_affine_grads(const FlowDims& f, const TapeDims& td, const GradLayout& gl, const fabhip_flow_params* params,
                        const float* ga_ws, float* grads, const float* tape, const float* coef, long B, hipStream_t st);

}  // namespace fab
