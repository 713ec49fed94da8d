// Generic plug-in path of the transition operators: when the base distribution / target are
// not fabhip-native (any `Distribution` / `LogProbFunc`, fab/types_.py:5-27), the densities and their gradients are
// evaluated by the CALLER (the plug-in's own code + autograd, fab/sampling_methods/base.py:50-72) and everything
// else of a transition runs here as elementwise HIP kernels on [B][D] row-major state:
//   HMC        hmc.py:129-160   begin (momentum, grad U, -U - K of the current point), leapfrog halves, accept +
//                               in-place commit + AIS log-weight increment + acceptance partials (then k_hmc_adapt's
//                               twin below for the step sizes)
//   Metropolis metropolis.py:51-74  propose, accept + commit (stale x_prev_log_prob kept), step adaptation
// HBM-bound trivial kernels: 16 lanes per chain row (coalesced 64-byte segments), row sums by a fixed xor tree.
#include "fabhip_common.h"
#include "launch.h"

#pragma clang fp contract(off)   // like the eager reference: a*b+c stays two roundings

namespace fab {

__device__ __forceinline__ float g_row16_sum(float v) {
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

__device__ __forceinline__ float g_clamp_nan0(float g, float mg) {      // hmc.py:194-199
    return (g != g) ? 0.f : fminf(fmaxf(g, -mg), mg);
}

// ---- HMC ---------------------------------------------------------------------------------------------------------
// begin of an outer step: p0 = noise * mass (hmc.py:134), grad U of the start point, x working copy,
// logp_cur = -U(current) - K(p0)
__global__ __launch_bounds__(256) void k_gen_hmc_begin(long B, int D, const float* __restrict__ start_x,
                                                       const float* __restrict__ start_gq,
                                                       const float* __restrict__ start_gp,
                                                       const float* __restrict__ cur_lq, const float* __restrict__ cur_lp,
                                                       const float* __restrict__ noise_p, const float* __restrict__ mass,
                                                       fabhip_anneal c, float max_grad, float* __restrict__ XP,
                                                       float* __restrict__ P, float* __restrict__ GU,
                                                       float* __restrict__ logp_cur, const int* __restrict__ n_valid) {
    const long B_all = B;
    if (n_valid) B = *n_valid < B ? *n_valid : B;           // (fused AIS calls: rows in use after the "chain init" filter)
    const long g = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int cc = threadIdx.x & 15;
    float k0 = 0.f;
    if (g >= B && g < B_all) {                              // rows of dropped chains: the leapfrog / density / target kernels run
        for (int j = cc; j < D; j += 16) {                  // over all B_all rows - give them a defined state
            XP[g * D + j] = 0.f; P[g * D + j] = 0.f; GU[g * D + j] = 0.f;
        }
    }
    if (g < B) {
        for (int j = cc; j < D; j += 16) {
            const float m = mass[j];
            const float p = noise_p[g * D + j] * m;
            const float gr = -(c.g_q * start_gq[g * D + j] + c.g_p * start_gp[g * D + j]);
            XP[g * D + j] = start_x[g * D + j];
            P[g * D + j] = p;
            GU[g * D + j] = g_clamp_nan0(gr, max_grad);
            k0 += p * p / m;
        }
    }
    k0 = g_row16_sum(k0) / 2.f;
    if (g < B && cc == 0) logp_cur[g] = (c.c_q * cur_lq[g] + c.c_p * cur_lp[g]) - k0;
}

// first half of a leapfrog (hmc.py:140-142): p -= eps gradU / 2 ; x += eps / mass * p
__global__ void k_gen_leap_pre(long n, int D, float* __restrict__ XP, float* __restrict__ P, const float* __restrict__ GU,
                               const float* __restrict__ eps_ptr, const float* __restrict__ ceps_ptr,
                               const float* __restrict__ mass) {
    const float eps = *eps_ptr + *ceps_ptr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float m = mass[i % D];
        const float p = P[i] - eps * GU[i] / 2.f;
        P[i] = p;
        XP[i] = XP[i] + eps / m * p;
    }
}

// second half (hmc.py:145-147) with the gradients of the re-evaluated point: gradU = clamp(...) ; p -= eps gradU / 2
__global__ void k_gen_leap_post(long n, const float* __restrict__ gq, const float* __restrict__ gp, fabhip_anneal c,
                                float max_grad, float* __restrict__ P, float* __restrict__ GU,
                                const float* __restrict__ eps_ptr, const float* __restrict__ ceps_ptr) {
    const float eps = *eps_ptr + *ceps_ptr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gu = g_clamp_nan0(-(c.g_q * gq[i] + c.g_p * gp[i]), max_grad);
        GU[i] = gu;
        P[i] = P[i] - eps * gu / 2.f;
    }
}

struct GenAccK {
    long B;
    int D;
    const float *XP, *P, *prop_lq, *prop_lp, *prop_gq, *prop_gp;   // the proposal
    float *cur_x, *cur_lq, *cur_lp, *cur_gq, *cur_gp;              // committed in place (hmc.py:154)
    const float *logp_cur, *noise_e, *mass;
    fabhip_anneal c, nx;
    float* log_w;                                                   // nullptr: no AIS increment in this outer step
    float *part_acc, *part_dist;                                    // [gridDim.x]
    const int* n_valid;                                             // device scalar: rows in use (nullptr: B)
};

__global__ __launch_bounds__(256) void k_gen_hmc_accept(GenAccK a) {
    __shared__ float rowb[32];
    const int r = threadIdx.x >> 4, cc = threadIdx.x & 15;
    const long g = (long)blockIdx.x * 16 + r;
    const int D = a.D;
    const long nv = a.n_valid ? (*a.n_valid < a.B ? (long)*a.n_valid : a.B) : a.B;
    const bool active = g < nv;
    float k1 = 0.f, dist2 = 0.f;
    if (active) {
        for (int j = cc; j < D; j += 16) {
            const float p = a.P[g * D + j];
            k1 += p * p / a.mass[j];
            const float dx = a.cur_x[g * D + j] - a.XP[g * D + j];
            dist2 += dx * dx;
        }
    }
    k1 = g_row16_sum(k1) / 2.f;
    dist2 = g_row16_sum(dist2);
    float contrib = 0.f, dist = 0.f;
    if (active) {
        const float lq = a.prop_lq[g], lp = a.prop_lp[g];
        const float lq_c = a.cur_lq[g], lp_c = a.cur_lp[g];
        const float delta = ((a.c.c_q * lq + a.c.c_p * lp) - k1) - a.logp_cur[g];
        const bool valid = isfinite(delta);
        const float dd = valid ? delta : -INFINITY;
        const bool accept = valid && (dd > -a.noise_e[g]);            // hmc.py:105-124
        contrib = expf(fminf(dd, 0.f));
        dist = accept ? 0.f : sqrtf(dist2);                          // store_info sees the committed point
        if (accept) {
            for (int j = cc; j < D; j += 16) {
                a.cur_x[g * D + j] = a.XP[g * D + j];
                a.cur_gq[g * D + j] = a.prop_gq[g * D + j];
                a.cur_gp[g * D + j] = a.prop_gp[g * D + j];
            }
        }
        if (cc == 0) {
            if (accept) { a.cur_lq[g] = lq; a.cur_lp[g] = lp; }
            if (a.log_w) {                                            // ais.py:93-100
                const float lqf = accept ? lq : lq_c, lpf = accept ? lp : lp_c;
                a.log_w[g] = a.log_w[g] + ((a.nx.c_q * lqf + a.nx.c_p * lpf) - (a.c.c_q * lqf + a.c.c_p * lpf));
            }
        }
    }
    if (cc == 0) { rowb[r] = contrib; rowb[16 + r] = dist; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f, d = 0.f;
        for (int i = 0; i < 16; ++i) { s += rowb[i]; d += rowb[16 + i]; }
        a.part_acc[blockIdx.x] = s;
        a.part_dist[blockIdx.x] = d;
    }
}

// step-size adaptation (hmc.py:122-123,162-170), same fixed-order reduction as the fused path's k_hmc_adapt
__global__ void k_gen_hmc_adapt(const float* __restrict__ part_acc, const float* __restrict__ part_dist, int nblk, long B,
                                float* eps_ptr, float* ceps_ptr, float target_p_accept, int tune, float* p_accept_out,
                                float* dist_out, const int* __restrict__ n_valid) {
    if (n_valid) B = *n_valid < B ? *n_valid : B;
    if (threadIdx.x != 0 || blockIdx.x != 0 || B <= 0) return;
    float s = 0.f, d = 0.f;
    for (int i = 0; i < nblk; ++i) { s += part_acc[i]; d += part_dist[i]; }
    const float log_mean = logf(s) - logf((float)B);
    if (p_accept_out) *p_accept_out = expf(log_mean);
    if (dist_out) *dist_out = d / (float)B;
    if (tune) {
        if (log_mean > logf(target_p_accept)) { *eps_ptr = *eps_ptr * 1.05f; *ceps_ptr = *ceps_ptr * 1.02f; }
        else { *eps_ptr = *eps_ptr / 1.05f; *ceps_ptr = *ceps_ptr / 1.02f; }
    }
}

// ---- Metropolis --------------------------------------------------------------------------------------------------
// annealed log-density c_q log_q + c_p log_p (base.py:76-97): x_prev_log_prob and the AIS log-weight terms
__global__ void k_gen_anneal(long n, const float* __restrict__ lq, const float* __restrict__ lp, fabhip_anneal c,
                             float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = c.c_q * lq[i] + c.c_p * lp[i];
}

// log_w += pi_{beta_next}(point) - pi_{beta}(point)   (ais.py:93-100)
__global__ void k_gen_logw_update(long n, const float* __restrict__ lq, const float* __restrict__ lp, fabhip_anneal c,
                                  fabhip_anneal nx, float* __restrict__ log_w) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        log_w[i] = log_w[i] + ((nx.c_q * lq[i] + nx.c_p * lp[i]) - (c.c_q * lq[i] + c.c_p * lp[i]));
}

__global__ void k_gen_met_propose(long n, const float* __restrict__ x, const float* __restrict__ noise,
                                  const float* __restrict__ scale_ptr, float* __restrict__ xn) {
    const float sc = *scale_ptr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        xn[i] = x[i] + noise[i] * sc;                                    // metropolis.py:57
}

struct GenMetK {
    long B;
    int D;
    const float *XN, *new_lq, *new_lp, *prev_lp, *noise_u;
    float *cur_x, *cur_lq, *cur_lp;
    fabhip_anneal c;
    float* part_acc;
};

__global__ __launch_bounds__(256) void k_gen_met_accept(GenMetK a) {
    __shared__ float rowb[16];
    const int r = threadIdx.x >> 4, cc = threadIdx.x & 15;
    const long g = (long)blockIdx.x * 16 + r;
    float contrib = 0.f;
    if (g < a.B) {
        const float lq = a.new_lq[g], lp = a.new_lp[g];
        float acc = expf((a.c.c_q * lq + a.c.c_p * lp) - a.prev_lp[g]);   // prev_lp is never refreshed (:53)
        if (!isfinite(acc)) acc = 0.f;                                      // nan_to_num(nan=0, posinf=0, neginf=0)
        const bool accept = acc > a.noise_u[g];
        if (accept) {
            for (int j = cc; j < a.D; j += 16) a.cur_x[g * a.D + j] = a.XN[g * a.D + j];
            if (cc == 0) { a.cur_lq[g] = lq; a.cur_lp[g] = lp; }
        }
        contrib = fminf(acc, 1.f);
    }
    if (cc == 0) rowb[r] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += rowb[i];
        a.part_acc[blockIdx.x] = s;
    }
}

__global__ void k_gen_met_adapt(const float* __restrict__ part_acc, int nblk, long B, float* scale_ptr,
                                float target_p_accept) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || B <= 0) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += part_acc[i];
    const float p_accept = s / (float)B;
    *scale_ptr = (p_accept > target_p_accept) ? *scale_ptr * 1.05f : *scale_ptr / 1.05f;
}

static inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
static inline int row_blocks(long B) { return (int)((B + 15) / 16); }

static void split_ws(void* ws, long B, int D, float*& XP, float*& P, float*& GU, float*& row, float*& pa, float*& pd) {
    float* w = (float*)ws;
    XP = w; w += B * D;
    P = w; w += B * D;
    GU = w; w += B * D;
    row = w; w += B;
    pa = w; w += row_blocks(B);
    pd = w;
}

// begin / accept with the row count read on the device (launch.h): the fused spline AIS call (ais_kernels.hip) keeps
// fixed-size buffers after the "chain init" filter, like fabhip_ais_run
void gen_hmc_state(void* workspace, long B, int dim, float** XP, float** P, float** GU) {
    float *row, *pa, *pd;
    split_ws(workspace, B, dim, *XP, *P, *GU, row, pa, pd);
}

int gen_hmc_begin(const fabhip_point* start, const fabhip_point* cur, long B, int dim, fabhip_anneal c, const float* noise_p,
                  const float* mass, float max_grad, void* workspace, const int* n_valid, hipStream_t st) {
    float *XP, *P, *GU, *row, *pa, *pd;
    split_ws(workspace, B, dim, XP, P, GU, row, pa, pd);
    hipLaunchKernelGGL(k_gen_hmc_begin, dim3(row_blocks(B)), dim3(256), 0, st, B, dim, start->x, start->grad_log_q,
                       start->grad_log_p, cur->log_q, cur->log_p, noise_p, mass, c, max_grad, XP, P, GU, row, n_valid);
    return check_launch();
}

int gen_hmc_accept(const fabhip_point* prop, const fabhip_point* cur, long B, int dim, fabhip_anneal c, fabhip_anneal next,
                   float* log_w, const float* noise_e, const float* mass, float* eps_ptr, float* ceps_ptr,
                   float target_p_accept, int tune, float* p_accept, float* avg_distance, void* workspace,
                   const int* n_valid, hipStream_t st) {
    float *XP, *P, *GU, *row, *pa, *pd;
    split_ws(workspace, B, dim, XP, P, GU, row, pa, pd);
    GenAccK a;
    a.B = B; a.D = dim; a.XP = XP; a.P = P;
    a.prop_lq = prop->log_q; a.prop_lp = prop->log_p; a.prop_gq = prop->grad_log_q; a.prop_gp = prop->grad_log_p;
    a.cur_x = cur->x; a.cur_lq = cur->log_q; a.cur_lp = cur->log_p; a.cur_gq = cur->grad_log_q; a.cur_gp = cur->grad_log_p;
    a.logp_cur = row; a.noise_e = noise_e; a.mass = mass; a.c = c; a.nx = next; a.log_w = log_w;
    a.part_acc = pa; a.part_dist = pd; a.n_valid = n_valid;
    const int nblk = row_blocks(B);
    hipLaunchKernelGGL(k_gen_hmc_accept, dim3(nblk), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_gen_hmc_adapt, dim3(1), dim3(64), 0, st, pa, pd, nblk, B, eps_ptr, ceps_ptr, target_p_accept, tune,
                       p_accept, avg_distance, n_valid);
    return check_launch();
}

}  // namespace fab

using namespace fab;

extern "C" {

size_t fabhip_generic_workspace_bytes(int64_t B, int32_t dim) {
    const size_t nblk = (size_t)row_blocks(B);
    //  XP, P, GU [B][dim] ; logp_cur / prev_lp [B] ; two partial arrays
    return ((size_t)3 * B * dim + B + 2 * nblk) * sizeof(float) + 1024;
}

int fabhip_hmc_generic_begin(const fabhip_point* start, const fabhip_point* cur, int64_t B, int32_t dim,
                             fabhip_anneal c, const float* noise_p, const float* mass, float max_grad,
                             void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!start || !cur || !start->x || !start->grad_log_q || !start->grad_log_p || !cur->log_q || !cur->log_p ||
        !noise_p || !mass || !workspace || B < 0 || dim < 1)
        return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_generic_workspace_bytes(B, dim)) return FABHIP_ENOSPC;
    if (B == 0) return FABHIP_OK;
    return gen_hmc_begin(start, cur, (long)B, dim, c, noise_p, mass, max_grad, workspace, nullptr, (hipStream_t)stream);
}

int fabhip_hmc_generic_leap_pre(int64_t B, int32_t dim, const float* eps_ptr, const float* ceps_ptr, const float* mass,
                                float* x_out, void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!eps_ptr || !ceps_ptr || !mass || !x_out || !workspace || B < 0 || dim < 1) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_generic_workspace_bytes(B, dim)) return FABHIP_ENOSPC;
    if (B == 0) return FABHIP_OK;
    float *XP, *P, *GU, *row, *pa, *pd;
    split_ws(workspace, (long)B, dim, XP, P, GU, row, pa, pd);
    const long n = (long)B * dim;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gen_leap_pre, dim3(ew_grid(n)), dim3(256), 0, st, n, (int)dim, XP, P, GU, eps_ptr, ceps_ptr, mass);
    if (hipMemcpyAsync(x_out, XP, (size_t)n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return FABHIP_ELAUNCH;
    return check_launch();
}

int fabhip_hmc_generic_leap_post(int64_t B, int32_t dim, const float* grad_log_q, const float* grad_log_p,
                                 fabhip_anneal c, float max_grad, const float* eps_ptr, const float* ceps_ptr,
                                 void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!grad_log_q || !grad_log_p || !eps_ptr || !ceps_ptr || !workspace || B < 0 || dim < 1) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_generic_workspace_bytes(B, dim)) return FABHIP_ENOSPC;
    if (B == 0) return FABHIP_OK;
    float *XP, *P, *GU, *row, *pa, *pd;
    split_ws(workspace, (long)B, dim, XP, P, GU, row, pa, pd);
    const long n = (long)B * dim;
    hipLaunchKernelGGL(k_gen_leap_post, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, n, grad_log_q, grad_log_p, c,
                       max_grad, P, GU, eps_ptr, ceps_ptr);
    return check_launch();
}

int fabhip_hmc_generic_accept(const fabhip_point* prop, const fabhip_point* cur, int64_t B, int32_t dim, fabhip_anneal c,
                              fabhip_anneal next, float* log_w, const float* noise_e, const float* mass,
                              float* eps_ptr, float* ceps_ptr, float target_p_accept, int32_t tune, float* p_accept,
                              float* avg_distance, void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!prop || !cur || !prop->log_q || !prop->log_p || !prop->grad_log_q || !prop->grad_log_p || !cur->x ||
        !cur->log_q || !cur->log_p || !cur->grad_log_q || !cur->grad_log_p || !noise_e || !mass || !eps_ptr ||
        !ceps_ptr || !workspace || B < 0 || dim < 1)
        return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_generic_workspace_bytes(B, dim)) return FABHIP_ENOSPC;
    if (B == 0) return FABHIP_OK;
    return gen_hmc_accept(prop, cur, (long)B, dim, c, next, log_w, noise_e, mass, eps_ptr, ceps_ptr, target_p_accept, (int)tune,
                          p_accept, avg_distance, workspace, nullptr, (hipStream_t)stream);
}

int fabhip_anneal_log_prob(const float* log_q, const float* log_p, int64_t n, fabhip_anneal c, float* out,
                           fabhip_stream_t stream) {
    if (!log_q || !log_p || !out || n < 0) return FABHIP_EINVAL;
    if (n == 0) return FABHIP_OK;
    hipLaunchKernelGGL(k_gen_anneal, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (long)n, log_q, log_p, c, out);
    return check_launch();
}

int fabhip_log_w_update(const float* log_q, const float* log_p, int64_t n, fabhip_anneal c, fabhip_anneal next,
                        float* log_w, fabhip_stream_t stream) {
    if (!log_q || !log_p || !log_w || n < 0) return FABHIP_EINVAL;
    if (n == 0) return FABHIP_OK;
    hipLaunchKernelGGL(k_gen_logw_update, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (long)n, log_q, log_p, c,
                       next, log_w);
    return check_launch();
}

int fabhip_metropolis_generic_propose(const float* x, const float* noise_x, const float* scale_ptr, int64_t B, int32_t dim,
                                      float* x_new, fabhip_stream_t stream) {
    if (!x || !noise_x || !scale_ptr || !x_new || B < 0 || dim < 1) return FABHIP_EINVAL;
    if (B == 0) return FABHIP_OK;
    const long n = (long)B * dim;
    hipLaunchKernelGGL(k_gen_met_propose, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, n, x, noise_x, scale_ptr,
                       x_new);
    return check_launch();
}

int fabhip_metropolis_generic_accept(const float* x_new, const float* new_log_q, const float* new_log_p,
                                     const fabhip_point* cur, const float* prev_log_prob, const float* noise_u,
                                     int64_t B, int32_t dim, fabhip_anneal c, float* scale_ptr, float target_p_accept,
                                     int32_t tune, void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!x_new || !new_log_q || !new_log_p || !cur || !cur->x || !cur->log_q || !cur->log_p || !prev_log_prob ||
        !noise_u || !scale_ptr || !workspace || B < 0 || dim < 1)
        return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_generic_workspace_bytes(B, dim)) return FABHIP_ENOSPC;
    if (B == 0) return FABHIP_OK;
    GenMetK a;
    a.B = B; a.D = dim; a.XN = x_new; a.new_lq = new_log_q; a.new_lp = new_log_p; a.prev_lp = prev_log_prob;
    a.noise_u = noise_u; a.cur_x = cur->x; a.cur_lq = cur->log_q; a.cur_lp = cur->log_p; a.c = c;
    a.part_acc = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = row_blocks(B);
    hipLaunchKernelGGL(k_gen_met_accept, dim3(nblk), dim3(256), 0, st, a);
    if (tune)
        hipLaunchKernelGGL(k_gen_met_adapt, dim3(1), dim3(64), 0, st, a.part_acc, nblk, (long)B, scale_ptr, target_p_accept);
    return check_launch();
}

}  // extern "C"
