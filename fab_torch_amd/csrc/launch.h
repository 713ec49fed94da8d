// Host-side launch helpers (no allocation, no synchronisation).
#pragma once
#include <mutex>
#include <vector>
#include "fabhip_common.h"

namespace fab {

#define FAB_TRY(expr)                     \
    do {                                  \
        const int _e = (expr);            \
        if (_e != FABHIP_OK) return _e;   \
    } while (0)

// process-wide fast-mode switch (fabhip_set_fast_mode, flow_kernels.hip)
int fast_mode();
// fast mode of ONE call: the flow descriptor's FABHIP_PRECISION_* over the process default
static inline int resolve_fast(int precision) {
    return precision == FABHIP_PRECISION_FAST ? 1 : (precision == FABHIP_PRECISION_FP32 ? 0 : fast_mode());
}
static inline FlowDims flow_dims_of(const fabhip_flow& fl) {
    FlowDims f = make_flow_dims(fl.dim, fl.n_layers, fl.width);
    f.fast = resolve_fast(fl.precision);
    return f;
}
// developer switches (fabhip_set_option; flow_kernels.hip): one int load, initialised from the environment at load time
int option(int key);
// generic HMC pieces with the row count on the device (generic_kernels.hip; used by the fused spline AIS call)
int gen_hmc_begin(const fabhip_point* start, const fabhip_point* cur, long B, int dim, fabhip_anneal c, const float* noise_p,
                  const float* mass, float max_grad, void* workspace, const int* n_valid, hipStream_t st);
int gen_hmc_accept(const fabhip_point* prop, const fabhip_point* cur, long B, int dim, fabhip_anneal c, fabhip_anneal next,
                   float* log_w, const float* noise_e, const float* mass, float* eps_ptr, float* ceps_ptr,
                   float target_p_accept, int tune, float* p_accept, float* avg_distance, void* workspace,
                   const int* n_valid, hipStream_t st);

// the transition state inside a generic-HMC workspace (generic_kernels.hip)
void gen_hmc_state(void* workspace, long B, int dim, float** XP, float** P, float** GU);
// One leapfrog of the spline family in ONE launch (spline_kernels.hip; r4): first half step, spline density + gradient, target
// + gradient and second half step inside k_spline_logprob_r8.  FABHIP_ENOTSUP where that kernel does not apply (the caller
// then runs the four launches of the generic pieces - the same arithmetic, bit for bit).
// Round 5: the rest of the outer step inside the leapfrog launches (spline_r8.h) - flags & 1: first leapfrog of the outer step
// (k_gen_hmc_begin's work at the launch's top), & 2: the last one (k_gen_hmc_accept's and k_gen_hmc_adapt's at its end).
struct SplineFold {
    int flags;
    const float *start_x, *start_gq, *start_gp, *noise_p;      // first
    float* logp_cur;                                           // [B] written by the first launch, read by the last
    float *cur_x, *cur_lq, *cur_lp, *cur_gq, *cur_gp;          // committed in place by the last launch (hmc.py:154)
    const float* noise_e;
    fabhip_anneal nx;
    float* log_w;                                              // nullptr: no AIS increment in this outer step
    const int* n_valid;                                        // device scalar: rows in use (nullptr: B)
    float *row_acc, *row_dist;                                 // [16 nblk] per-chain min(1, acceptance prob) / store_info distance
    int* ticket;                                               // zero before the launch; reset by the wave that draws the last one
    float *eps_w, *ceps_w;
    float target_p_accept;
    int tune;
    float *p_accept_out, *dist_out;
    int nblk;
};
// whether spline_log_prob_leap applies to this shape AND can carry the fold (FABHIP_OPT_ADAPT_FOLD, LDS for the block sums)
bool spline_leap_fold_supported(const fabhip_spline_flow* flow, int64_t B);
// floats of the fold's scratch behind a generic-HMC workspace: row_acc, row_dist [16 nblk] + the ticket word
size_t spline_fold_scratch_floats(int64_t B);
struct SplineLeap {
    SplineFold fold;                   // flags == 0: a plain leapfrog
    float *XP, *P, *GU;                // [B][D] of the transition workspace: position, momentum, clamped grad U
    float* x_out;                      // the proposal's x (a copy of XP, as fabhip_hmc_generic_leap_pre leaves it)
    const float *eps_ptr, *ceps_ptr, *mass;
    fabhip_anneal c;
    float max_grad;
    fabhip_target tg;
    float *prop_lp, *prop_gp;          // target log-density and gradient of the proposal
};
int spline_log_prob_leap(const fabhip_spline_flow* flow, const SplineLeap& lp, float* log_q, float* grad_x, int64_t B,
                         void* workspace, size_t workspace_bytes, hipStream_t st);

// The tail of a chain phase in one launch for small batches (reduce_resample.hip: k_tail_small): compaction of the rows with
// finite log_p / log_q, optionally diff = log_p - log_q over all B rows, ESS / log Z over the survivors (of diff, or of log_w).
// FABHIP_ENOTSUP above 2048 rows (the caller then runs the separate kernels: the same results, bit for bit).
struct TailArgs {
    float *x, *lq, *lp, *gq, *gp;      // the point's fields (gq / gp null: a Metropolis run)
    float* log_w;
    float* extra;                      // optional per-row scalar carried along by the compaction
    const int* n_in;                   // rows in (device; null: B)
    int* n_out;                        // rows out (device)
    long B;
    int D;
    float* diff;                       // null: ESS over log_w
    double n_norm;
    float* stats_out;                  // [3]: ESS, log Z, n
    int* zero_word;                    // optional: one word the kernel zeroes (the ticket of the in-kernel step-size rule)
    float* zero_f;                     // optional: n_zero_f floats the kernel zeroes (the unused tail of the caller's stats[16]:
    int n_zero_f;                      //   the binding allocates stats uninitialised - one fill launch less per call)
};
int tail_small(const TailArgs& a, int* dest, hipStream_t st);

// 4-chain tiles (flow_r4.h) pay when 16-chain tiles cannot fill the chip: up to 288 workgroups of 4 chains (B <= 1152); off in fast
// mode (no bf16 variant).  FABHIP_OPT_TILE_SHAPE = 16 / 4 forces the choice (tests exercise both).  D <= 32 and hidden width <= 320
// only (the other instantiations spill registers and are not compiled).  Used by the transitions, the chain initialisation and
// the flow sample alike.
bool r4f_lds_fits(const FlowDims& f);                  // (flow_kernels.hip: the LDS-fit test of use_r4_fused, flow_r4f.h)
static inline bool use_r4_tiles(const FlowDims& f, long B) {
    if (f.D > 32 || f.NTW / 4 > 5) return false;
    // fast mode: only where the fused-stage stream has its bf16 image AND the fused kernel's LDS plan fits (flow_r4f.h; its bias
    // blocks and ReLU masks grow with the layer count) - else the 16-chain fast kernels, never silently the fp32 stream
    if (f.fast && !(f.o_r4fh >= 0 && f.NTW / 4 >= 2 && option(FABHIP_OPT_R4_STREAM) >= 2 && r4f_lds_fits(f))) return false;
    const int shape = option(FABHIP_OPT_TILE_SHAPE);
    if (shape == 16 || shape == 8) return false;
    if (shape == 4) return true;
    return B <= 1152;
}

static inline int check_launch() { return hipGetLastError() == hipSuccess ? FABHIP_OK : FABHIP_ELAUNCH; }

// Allow > 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU).
// (the attribute sticks to the function on its device: the driver call is made once per (kernel, device, larger size) and
// remembered - it sat on the host path of every launch, and the first launch of a call is one the GPU waits for)
static inline int set_max_lds(const void* fn, size_t bytes) {
    if (bytes > 160 * 1024) return FABHIP_ENOTSUP;
    if (bytes > 48 * 1024) {
        struct Seen { const void* fn; int dev; size_t bytes; };
        static std::mutex mu;
        static std::vector<Seen> seen;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return FABHIP_ELAUNCH; }
        std::lock_guard<std::mutex> lk(mu);
        Seen* hit = nullptr;
        for (Seen& e : seen)
            if (e.fn == fn && e.dev == dev) { hit = &e; break; }
        if (hit && hit->bytes >= bytes) return FABHIP_OK;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return FABHIP_ELAUNCH;
        }
        if (hit) hit->bytes = bytes; else seen.push_back(Seen{fn, dev, bytes});
    }
    return FABHIP_OK;
}

// compute units of the current device (cached: the tile-shape choices compare the batch with chains-per-workgroup x CUs)
static inline int cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
        (void)hipGetLastError();
        return v > 0 ? v : 256;
    }();
    return n;
}

static inline int check_flow_shape(int dim, int n_layers, int width) {
    if (dim < 2 || n_layers < 1 || width < 1) return FABHIP_EINVAL;
    if (dim > FABHIP_MAX_DIM || n_layers > FABHIP_MAX_LAYERS || width > FABHIP_MAX_WIDTH) return FABHIP_ENOTSUP;
    return FABHIP_OK;
}

// column tiles per wave: ceil(NTW / 4) rounded up to a compiled variant {1, 2, 4, 5, 8}
#define FAB_DISPATCH_NTW(f, fn, ...)                      \
    do {                                                  \
        const int _per = (f).NTW / 4;       \
        if (_per <= 1) return fn<1>(__VA_ARGS__);         \
        if (_per <= 2) return fn<2>(__VA_ARGS__);         \
        if (_per <= 4) return fn<4>(__VA_ARGS__);         \
        if (_per <= 5) return fn<5>(__VA_ARGS__);         \
        if (_per <= 8) return fn<8>(__VA_ARGS__);         \
        return FABHIP_ENOTSUP;                            \
    } while (0)

// same, but falls through on success (for call sites that continue afterwards)
#define FAB_DISPATCH_NTW_NORET(f, fn, ...)                \
    do {                                                  \
        const int _per = (f).NTW / 4;       \
        int _rc;                                          \
        if (_per <= 1) _rc = fn<1>(__VA_ARGS__);          \
        else if (_per <= 2) _rc = fn<2>(__VA_ARGS__);     \
        else if (_per <= 4) _rc = fn<4>(__VA_ARGS__);     \
        else if (_per <= 5) _rc = fn<5>(__VA_ARGS__);     \
        else if (_per <= 8) _rc = fn<8>(__VA_ARGS__);     \
        else _rc = FABHIP_ENOTSUP;                        \
        if (_rc != FABHIP_OK) return _rc;                 \
    } while (0)

}  // namespace fab
