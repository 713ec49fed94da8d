// ESS / log Z reductions, multinomial + systematic resampling, row gather, target log-prob kernel.
//
// The resample path is HBM-bound integer/byte work: coalesced 16-byte loads, wave ballots and LDS
// partials, a single-pass decoupled-look-back prefix scan over fixed-point weights (integer sums are
// associative => bit-exact for any scan order), two-level binary search (tile prefixes, then inside
// one 4096-entry tile) and a vectorised row gather.
#include "flow_device.h"
#include "target_device.h"
#include "launch.h"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace fab {

// ------------------------------------------------------------------------------------------------
// ESS / log Z (fab/utils/numerical.py:18-23; ais.py:80-86), float64 accumulation
// ------------------------------------------------------------------------------------------------
struct Msum {   // running (max, sum exp(x-max), sum exp(2(x-max)))
    double m, s1, s2;
};
__device__ __forceinline__ Msum msum_id() { return Msum{-INFINITY, 0.0, 0.0}; }
__device__ __forceinline__ Msum msum_merge(const Msum& a, const Msum& b) {
    if (a.m != a.m || b.m != b.m) return Msum{NAN, NAN, NAN};
    const double m = fmax(a.m, b.m);
    if (m == -INFINITY) return Msum{m, 0.0, 0.0};
    if (m == INFINITY) return Msum{m, NAN, NAN};      // softmax of +inf is NaN, like torch
    const double ea = exp(a.m - m), eb = exp(b.m - m);
    return Msum{m, a.s1 * ea + b.s1 * eb, a.s2 * ea * ea + b.s2 * eb * eb};
}
__device__ __forceinline__ Msum msum_push(const Msum& a, double x) {
    if (x != x || a.m != a.m || x == INFINITY) return Msum{NAN, NAN, NAN};   // softmax(+inf) is NaN in torch
    if (x == -INFINITY) return a;                       // weight 0
    if (x <= a.m) {
        const double e = (double)expf((float)(x - a.m));   // fp32 exp (1e-7 rel), fp64 accumulation
        return Msum{a.m, a.s1 + e, a.s2 + e * e};
    }
    const double r = (double)expf((float)(a.m - x));    // a.m == -inf -> 0
    return Msum{x, a.s1 * r + 1.0, a.s2 * r * r + 1.0};
}

// tree over the first `bdim` threads of the workgroup (every thread of the workgroup must call it: barriers)
__device__ __forceinline__ Msum msum_block_reduce(Msum v, Msum* sh, int bdim) {
    const int tid = threadIdx.x;
    if (tid < bdim) sh[tid] = v;
    __syncthreads();
    for (int s = bdim >> 1; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = msum_merge(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    return sh[0];
}
__device__ __forceinline__ Msum msum_block_reduce(Msum v, Msum* sh) { return msum_block_reduce(v, sh, (int)blockDim.x); }

constexpr int ESS_THREADS = 256;
constexpr int ESS_MAX_BLOCKS = 1024;

// chunk of values -> (max, sum exp(x - max), sum exp(2 (x - max))) with ONE rescale per chunk: the max is taken in fp32
// first, the exponentials are fp32 (1e-7 relative) and summed in fp32 over the <= 16 values of the chunk, the running
// totals are float64.  Same special values as torch's softmax: NaN or +inf anywhere -> NaN, -inf -> weight 0.
template <int NV>
__device__ __forceinline__ Msum msum_push_chunk(const Msum& a, const float (&x)[NV]) {
    float m = -INFINITY;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < NV; ++i) { bad |= (x[i] != x[i]) | (x[i] == INFINITY); m = fmaxf(m, x[i]); }
    if (bad || a.m != a.m) return Msum{NAN, NAN, NAN};
    if (m == -INFINITY) return a;                       // every weight of the chunk is zero
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float e = expf(x[i] - m); s1 += e; s2 += e * e; }
    const double md = (double)m;
    if (md <= a.m) {
        const double r = (double)expf((float)(md - a.m));
        return Msum{a.m, a.s1 + r * (double)s1, a.s2 + r * r * (double)s2};
    }
    const double r = (double)expf((float)(a.m - md));   // a.m == -inf -> 0
    return Msum{md, a.s1 * r + (double)s1, a.s2 * r * r + (double)s2};
}

// the work of block `vb` of `nb` of the partial pass (ltid = thread in the block): shared by k_ess_partial and the one-launch
// tail kernel, which runs the blocks one after the other - the same additions in the same order, bit for bit
// (`bdim`: threads per block of the partial pass; threads beyond it - the tail kernel runs 1024 - only keep the barriers company)
__device__ __forceinline__ Msum ess_partial_body(const float* __restrict__ lw, long n, int vb, int nb, Msum* sh, int bdim) {
    Msum v = msum_id();
    const long stride = (long)nb * bdim;
    const long tid = (long)vb * bdim + threadIdx.x;
    long done = 0;
    if ((int)threadIdx.x >= bdim) return msum_block_reduce(v, sh, bdim);
    if ((((size_t)lw) & 15) == 0) {                     // 16 values per thread per step: four coalesced 16-byte loads
        const long n16 = n / (16 * stride) * (16 * stride);
        const float4* p4 = reinterpret_cast<const float4*>(lw);
        for (long base = 0; base < n16; base += 16 * stride) {
            float x[16];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float4 q = p4[(base >> 2) + h * stride + tid];
                x[4 * h] = q.x; x[4 * h + 1] = q.y; x[4 * h + 2] = q.z; x[4 * h + 3] = q.w;
            }
            v = msum_push_chunk<16>(v, x);
        }
        done = n16;
    }
    for (long i = done + tid; i < n; i += stride) v = msum_push(v, (double)lw[i]);
    return msum_block_reduce(v, sh, bdim);
}
__device__ __forceinline__ Msum ess_partial_body(const float* __restrict__ lw, long n, int vb, int nb, Msum* sh) {
    return ess_partial_body(lw, n, vb, nb, sh, (int)blockDim.x);
}
__device__ __forceinline__ void ess_final_body(const Msum* __restrict__ part, int nblk, long n, double n_norm,
                                               float* __restrict__ out, Msum* sh, int bdim) {
    Msum v = msum_id();
    if ((int)threadIdx.x < bdim)
        for (int i = threadIdx.x; i < nblk; i += bdim) v = msum_merge(v, part[i]);
    v = msum_block_reduce(v, sh, bdim);
    if (threadIdx.x == 0) {
        const double ess = (v.s1 * v.s1 / v.s2) / (double)n;       // 1 / sum(softmax^2) / n
        const double logz = v.m + log(v.s1) - log(n_norm);         // logsumexp - log(n_norm)
        out[0] = (float)ess;
        out[1] = (float)logz;
        out[2] = (float)n;
    }
}

__global__ __launch_bounds__(ESS_THREADS) void k_ess_partial(const float* __restrict__ lw, long n_cap, const int* n_ptr,
                                                             Msum* __restrict__ part) {
    __shared__ Msum sh[ESS_THREADS];
    const long n = n_ptr ? (long)*n_ptr : n_cap;
    const Msum v = ess_partial_body(lw, n, blockIdx.x, gridDim.x, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = v;
}

__global__ __launch_bounds__(ESS_THREADS) void k_ess_final(const Msum* __restrict__ part, int nblk, long n_cap,
                                                           const int* n_ptr, double n_norm, float* __restrict__ out) {
    __shared__ Msum sh[ESS_THREADS];
    const long n = n_ptr ? (long)*n_ptr : n_cap;
    ess_final_body(part, nblk, n, n_norm, out, sh, (int)blockDim.x);
}

// ------------------------------------------------------------------------------------------------
// The tail of a chain phase for SMALL batches in ONE launch (launch.h: TailArgs; <= 2048 rows): stable compaction of the
// rows with finite log_p and log_q (ais.py:190-213), optionally log_p - log_q, then ESS / log Z over the survivors - what
// k_valid_scan, k_compact_scatter, k_compact_copyback, k_sub, k_ess_partial and k_ess_final do in six launches (~5 us each on an
// otherwise idle GPU: the work of 1024 rows is a few hundred nanoseconds).  One workgroup: the rows move IN PLACE, in row order,
// through an LDS chunk (destination <= source, so a chunk's writes only touch rows already read); nothing moves when no row was
// dropped.  The ESS runs the partial pass's blocks one after the other (ess_partial_body): bit-identical to the two-kernel form.
// ------------------------------------------------------------------------------------------------
constexpr int TAIL_THREADS = 1024;                     // scan / move / difference on 1024 threads, the ESS on the first ESS_THREADS
constexpr int TAIL_CHUNK = 32;                         // rows staged in LDS per step of the in-place move
constexpr int TAIL_MAX_BLOCKS = 2;                     // ESS blocks of 1024 values: batches of <= 2048 rows (tools/time_tail.py: one
                                                       // workgroup is level with the separate kernels at 2048 rows and behind from 4096)

__global__ __launch_bounds__(TAIL_THREADS) void k_tail_small(TailArgs a, int* __restrict__ dest, int nb) {
    extern __shared__ __attribute__((aligned(16))) float stage[];          // [TAIL_CHUNK][3 D + 4]
    __shared__ Msum sh[ESS_THREADS];
    __shared__ Msum part[TAIL_MAX_BLOCKS];
    __shared__ int wsum[TAIL_THREADS / 64];
    __shared__ int running;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NWV = TAIL_THREADS / 64;
    const long n_in = a.n_in ? (long)*a.n_in : a.B;
    if (tid == 0) { running = 0; if (a.zero_word) *a.zero_word = 0; }
    if (a.zero_f && tid < a.n_zero_f) a.zero_f[tid] = 0.f;
    __syncthreads();
    for (long base = 0; base < n_in; base += TAIL_THREADS) {               // k_valid_scan's ranks (integers: any order of work agrees)
        const long r = base + tid;
        const bool v = r < n_in && isfinite(a.lq[r]) && isfinite(a.lp[r]);
        const unsigned long long bal = __ballot(v);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int i = 0; i < NWV; ++i) { if (i < w) woff += wsum[i]; tot += wsum[i]; }
        if (r < n_in) dest[r] = v ? running + woff + pre : -1;
        __syncthreads();
        if (tid == 0) running += tot;
        __syncthreads();
    }
    const long n_out = running;
    if (tid == 0) *a.n_out = (int)n_out;
    const int D = a.D, RW = 3 * D + 4;
    const bool hg = a.gq != nullptr;
    if (n_out != n_in) {
        for (long r0 = 0; r0 < n_in; r0 += TAIL_CHUNK) {
            const int nr = (int)(n_in - r0 < TAIL_CHUNK ? n_in - r0 : TAIL_CHUNK);
            for (int e = tid; e < nr * D; e += TAIL_THREADS) {
                const int i = e / D, j = e % D;
                const long r = r0 + i;
                float* o = stage + i * RW;
                o[j] = a.x[r * D + j];
                if (hg) { o[D + j] = a.gq[r * D + j]; o[2 * D + j] = a.gp[r * D + j]; }
            }
            if (tid < nr) {
                const long r = r0 + tid;
                float* o = stage + tid * RW;
                o[3 * D] = a.lq[r]; o[3 * D + 1] = a.lp[r]; o[3 * D + 2] = a.log_w[r];
                if (a.extra) o[3 * D + 3] = a.extra[r];
            }
            __syncthreads();
            for (int e = tid; e < nr * D; e += TAIL_THREADS) {
                const int i = e / D, j = e % D;
                const long d = dest[r0 + i];
                if (d < 0 || d == r0 + i) continue;
                const float* o = stage + i * RW;
                a.x[d * D + j] = o[j];
                if (hg) { a.gq[d * D + j] = o[D + j]; a.gp[d * D + j] = o[2 * D + j]; }
            }
            if (tid < nr) {
                const long d = dest[r0 + tid];
                if (d >= 0 && d != r0 + tid) {
                    const float* o = stage + tid * RW;
                    a.lq[d] = o[3 * D]; a.lp[d] = o[3 * D + 1]; a.log_w[d] = o[3 * D + 2];
                    if (a.extra) a.extra[d] = o[3 * D + 3];
                }
            }
            __syncthreads();
        }
    }
    const float* lw = a.log_w;
    if (a.diff) {
        for (long i = tid; i < a.B; i += TAIL_THREADS) a.diff[i] = a.lp[i] - a.lq[i];
        lw = a.diff;
    }
    __syncthreads();
    for (int vb = 0; vb < nb; ++vb) {
        const Msum v = ess_partial_body(lw, n_out, vb, nb, sh, ESS_THREADS);
        if (tid == 0) part[vb] = v;
        __syncthreads();
    }
    if (nb == 1) {
        // k_ess_final with one partial merges it with 255 identities: msum_merge(v, identity) == v bit for bit (exp(0) = 1,
        // exp(-inf) = 0; -inf and NaN maxima propagate the same way), so the tree is skipped
        if (tid == 0) {
            const Msum v = msum_merge(msum_id(), part[0]);
            const double ess = (v.s1 * v.s1 / v.s2) / (double)n_out;
            const double logz = v.m + log(v.s1) - log(a.n_norm);
            a.stats_out[0] = (float)ess; a.stats_out[1] = (float)logz; a.stats_out[2] = (float)n_out;
        }
    } else {
        ess_final_body(part, nb, n_out, a.n_norm, a.stats_out, sh, ESS_THREADS);
    }
}

// ------------------------------------------------------------------------------------------------
// torch-compatible multinomial (sequential fp32 CDF, exactly the CPU kernel's arithmetic)
// ------------------------------------------------------------------------------------------------
// c[j] = fl32(c[j-1] + p[j]) is inherently serial (the rounding of every partial sum feeds the next one), and it
// is what torch's CPU multinomial computes.  One lane runs the add chain out of LDS (float4 at a time, ~6 cycles per
// element instead of a global-memory round trip per few elements); the second wave of the workgroup streams the
// next chunk in and the previous chunk's results out meanwhile (double buffer, one barrier per 2048 elements).
constexpr int SEQ_CH = 2048;

__global__ __launch_bounds__(128) void k_seq_cdf(const float* __restrict__ p, long n, float* __restrict__ c) {
    __shared__ __attribute__((aligned(16))) float buf[2][SEQ_CH];      // inputs  (double buffer)
    __shared__ __attribute__((aligned(16))) float res[2][SEQ_CH];      // results (separate: lets the reads run ahead)
    __shared__ float carry_sh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long nch = (n + SEQ_CH - 1) / SEQ_CH;
    auto load_chunk = [&](long ch, float* dst) {
        const long base = ch * SEQ_CH;
        for (int i = lane; i < SEQ_CH; i += 64) dst[i] = (base + i < n) ? p[base + i] : 0.f;   // + 0.f is exact
    };
    auto store_chunk = [&](long ch, const float* src) {
        const long base = ch * SEQ_CH;
        for (int i = lane; i < SEQ_CH; i += 64)
            if (base + i < n) c[base + i] = src[i];
    };
    if (wave == 1 && nch > 0) load_chunk(0, buf[0]);
    if (tid == 0) carry_sh = 0.f;
    __syncthreads();
    for (long ch = 0; ch < nch; ++ch) {
        if (wave == 0) {
            if (lane == 0) {
                float s = carry_sh;
                const float4* __restrict__ vin = reinterpret_cast<const float4*>(buf[ch & 1]);
                float4* __restrict__ vout = reinterpret_cast<float4*>(res[ch & 1]);
#pragma unroll 8
                for (int j = 0; j < SEQ_CH / 4; ++j) {
                    float4 x = vin[j];
                    s += x.x; x.x = s;
                    s += x.y; x.y = s;
                    s += x.z; x.z = s;
                    s += x.w; x.w = s;
                    vout[j] = x;
                }
                carry_sh = s;
            }
        } else {
            if (ch > 0) store_chunk(ch - 1, res[(ch & 1) ^ 1]);
            if (ch + 1 < nch) load_chunk(ch + 1, buf[(ch & 1) ^ 1]);
        }
        __syncthreads();
    }
    if (wave == 1 && nch > 0) store_chunk(nch - 1, res[(nch - 1) & 1]);
    if (tid == 0) c[n] = carry_sh;   // total kept in the extra slot
}

__global__ void k_cdf_normalise(float* __restrict__ c, long n) {
    const float s = c[n];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = c[i] / s;
        if (i == n - 1) v = 1.f;
        c[i] = v;
    }
}

__global__ void k_search_f32cdf(const float* __restrict__ c, long n, const double* __restrict__ u, long ns,
                                long long* __restrict__ idx) {
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < ns; k += (long)gridDim.x * blockDim.x) {
        const double uk = u[k];
        long lo = 0, hi = n;
        while (hi - lo > 0) {
            const long mid = lo + (hi - lo) / 2;
            if ((double)c[mid] < uk) lo = mid + 1; else hi = mid;
        }
        idx[k] = lo;
    }
}

// ------------------------------------------------------------------------------------------------
// scalable fixed-point CDF:  q_i = floor(fl32(exp(fl64(w_i) - fl64(max))) * 2^36)
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;                         // per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;   // 4096 weights per tile
constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_INC = 2ull << 62, VAL_MASK = (1ull << 62) - 1ull;

struct ScanWs {          // workspace layout (all 256-byte aligned)
    float* max_part;             // [1024]
    float* max_val;              // [1]
    unsigned long long* desc;    // [ntiles] tile descriptors  (flag | value)
    unsigned int* ticket;        // [1] dynamic tile id
    unsigned long long* tile_inc;// [ntiles] inclusive prefix at the end of each tile
    unsigned long long* cdf;     // [n]
    unsigned long long* cdf16;   // [ceil(n / 16)] cdf16[g] = cdf[min(16 g + 15, n - 1)] (LDS scan only)
    unsigned long long* cdf256;  // [ceil(n / 256)] the same, every 256th value
    int variant;                 // store-pattern variant (A/B experiments)
    int want_sub;                // write cdf16 / cdf256 (only the multinomial sampler reads them)
};

__global__ __launch_bounds__(256) void k_max_partial(const float* __restrict__ lw, long n, float* __restrict__ part) {
    __shared__ float sh[256];
    float m = -INFINITY;
    const long n4 = (((size_t)lw & 15) == 0) ? (n >> 2) : 0;          // 16-byte vector body
    const float4* p4 = reinterpret_cast<const float4*>(lw);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = p4[i];
        if (isfinite(v.x)) m = fmaxf(m, v.x);
        if (isfinite(v.y)) m = fmaxf(m, v.y);
        if (isfinite(v.z)) m = fmaxf(m, v.z);
        if (isfinite(v.w)) m = fmaxf(m, v.w);
    }
    for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = lw[i];
        if (isfinite(v)) m = fmaxf(m, v);
    }
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void k_max_final(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    __shared__ float sh[256];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < nblk; i += 256) m = fmaxf(m, part[i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (sh[0] == -INFINITY) ? 0.f : sh[0];
}

// Specified fp32 exponential of the resampler (oracle/numerical.py:exp_spec restates it op for op):
//   r = x * log2(e); k = rint(r); f = r - k; p = 2^k * P6(f)   with every multiply and add individually rounded
// (no FMA contraction in this file), x <= 0.  Deterministic on any IEEE machine => bit-exact parity of the
// resulting fixed-point CDF; |p/exp(x) - 1| <= 2e-6 (3e-7 near the maximum).
__device__ __forceinline__ float exp_spec(float x) {
    const float r = x * 1.44269504088896341f;
    const float k = rintf(r);
    const float f = r - k;
    float p = 1.5403530393381609e-4f;
    p = p * f + 1.3333558146428443e-3f;
    p = p * f + 9.6181291076284772e-3f;
    p = p * f + 5.5504108664821580e-2f;
    p = p * f + 2.4022650695910071e-1f;
    p = p * f + 6.9314718055994531e-1f;
    p = p * f + 1.0f;
    return (k < -60.f) ? 0.f : ldexpf(p, (int)k);     // below 2^-60 the fixed-point weight is 0 anyway
}

__device__ __forceinline__ unsigned long long fixed_weight(float w, float mx) {
    // floor(p * 2^36): the scaling is exact, and so is the split into two 32-bit halves (v < 2^37 carries 24 significant
    // bits: hi = trunc(v 2^-32), v - hi 2^32 is a sub-range of those bits) - two v_cvt_u32_f32 instead of the generic
    // float -> uint64 sequence.  Non-finite weights count 0.
    const float v = isfinite(w) ? exp_spec(w - mx) * 68719476736.0f : 0.f;
    const unsigned hi = (unsigned)(v * 2.3283064365386963e-10f);
    const unsigned lo = (unsigned)(v - (float)hi * 4294967296.0f);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int off) {
    const unsigned lo = __shfl_up((unsigned)v, off), hi = __shfl_up((unsigned)(v >> 32), off);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    const unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// Single-pass inclusive scan with decoupled look-back.  One workgroup scans SCAN_BLOCK = 16384 weights
// (one dynamic ticket + one look-back per 16384 items: a single atomic word saturates at ~88 tickets/us);
// wave w owns search tile 4*blk + w (4096 consecutive weights) slab-wise — slab v = 256 consecutive
// weights, lane l owns 4 of them — so every global load/store instruction of a wave is a fully coalesced
// 1-2 KiB access.  Descriptors are 8-byte {flag,value} granules written/read with relaxed agent-scope
// atomics: the data IS the flag.
constexpr int SCAN_NW = 8;                              // waves per scan workgroup
constexpr int SCAN_WAVE_ITEMS = 2048;                   // weights per wave (half a search tile)
constexpr int SCAN_SLABS = SCAN_WAVE_ITEMS / 256;       // 8 slabs per wave
constexpr int SCAN_BLOCK = SCAN_WAVE_ITEMS * SCAN_NW;   // 16384 weights per workgroup / ticket

__global__ __launch_bounds__(64 * SCAN_NW) void k_scan_fixed(const float* __restrict__ lw, long n,
                                                             const float* __restrict__ max_val, ScanWs ws) {
    __shared__ unsigned long long wave_tot[SCAN_NW];
    __shared__ unsigned long long blk_prefix;
    __shared__ unsigned int blk_id_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) blk_id_sh = atomicAdd(ws.ticket, 1u);
    __syncthreads();
    const unsigned int blk = blk_id_sh;
    const long wbase = (long)blk * SCAN_BLOCK + (long)wave * SCAN_WAVE_ITEMS;
    const float mx = max_val[0];
    unsigned long long q[SCAN_SLABS][4];
    const bool vec = (wbase + SCAN_WAVE_ITEMS <= n) && (((size_t)lw & 15) == 0);
    if (vec) {
#pragma unroll
        for (int h = 0; h < SCAN_SLABS; h += 8) {          // 8 loads in flight per lane
            float4 x[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) x[v] = *reinterpret_cast<const float4*>(lw + wbase + (h + v) * 256 + lane * 4);
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                q[h + v][0] = fixed_weight(x[v].x, mx); q[h + v][1] = fixed_weight(x[v].y, mx);
                q[h + v][2] = fixed_weight(x[v].z, mx); q[h + v][3] = fixed_weight(x[v].w, mx);
            }
        }
    } else {
#pragma unroll
        for (int v = 0; v < SCAN_SLABS; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long i = wbase + v * 256 + lane * 4 + e;
                q[v][e] = (i < n) ? fixed_weight(lw[i], mx) : 0ull;
            }
    }
    // lane-local inclusive sums per slab, wave scan of the lane totals, running slab offsets
    unsigned long long excl[SCAN_SLABS];
    unsigned long long slab_off = 0ull;
#pragma unroll
    for (int v = 0; v < SCAN_SLABS; ++v) {
        q[v][1] += q[v][0]; q[v][2] += q[v][1]; q[v][3] += q[v][2];
        unsigned long long incl = q[v][3];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = shfl_up_u64(incl, off);
            if (lane >= off) incl += o;
        }
        excl[v] = slab_off + (incl - q[v][3]);
        slab_off += shfl_u64(incl, 63);
    }
    if (lane == 0) wave_tot[wave] = slab_off;      // wave (= search tile) total, identical in every lane
    __syncthreads();
    unsigned long long wave_off = 0ull, blk_tot = 0ull;
#pragma unroll
    for (int w = 0; w < SCAN_NW; ++w) { if (w < wave) wave_off += wave_tot[w]; blk_tot += wave_tot[w]; }
    // publish the aggregate, look back for the exclusive prefix (wave 0)
    if (wave == 0) {
        if (blk == 0) {
            if (lane == 0) {
                __hip_atomic_store(ws.desc + 0, FLAG_INC | blk_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                blk_prefix = 0ull;
            }
        } else {
            if (lane == 0)
                __hip_atomic_store(ws.desc + blk, FLAG_AGG | blk_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long ex = 0ull;
            long look = (long)blk - 1;
            while (true) {
                const long j = look - lane;
                unsigned long long d = 0ull;
                if (j >= 0) {
                    do {
                        d = __hip_atomic_load(ws.desc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((d >> 62) == 0ull);
                } else {
                    d = FLAG_INC;          // virtual block before the first: inclusive prefix 0
                }
                const unsigned long long inc_mask = __ballot((d >> 62) == 2ull);
                const int first_inc = __ffsll((long long)inc_mask) - 1;
                unsigned long long contrib = (first_inc < 0 || lane <= first_inc) ? (d & VAL_MASK) : 0ull;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) contrib += shfl_u64(contrib, lane ^ off);
                ex += contrib;
                if (first_inc >= 0) break;
                look -= 64;
            }
            if (lane == 0) {
                __hip_atomic_store(ws.desc + blk, FLAG_INC | (ex + blk_tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                blk_prefix = ex;
            }
        }
    }
    __syncthreads();
    const unsigned long long off0 = blk_prefix + wave_off;
    // inclusive prefix at the end of every 4096-weight search tile (= two waves)
    if (lane == 63 && (wave & 1) && wbase - SCAN_WAVE_ITEMS < n)
        ws.tile_inc[((long)blk * SCAN_BLOCK + (long)(wave - 1) * SCAN_WAVE_ITEMS) / SCAN_TILE] = off0 + slab_off;
    if (vec && ws.variant == 1) {
        // fully contiguous 1-KiB store instructions: lane j stores items (2j, 2j+1) of each half slab, which
        // live in lane j/2 (first half) / 32 + j/2 (second half) -> one lane exchange per value
        const int srcA = lane >> 1, srcB = 32 + (lane >> 1);
        const bool odd = lane & 1;
#pragma unroll
        for (int v = 0; v < SCAN_SLABS; ++v) {
            const unsigned long long b = off0 + excl[v];
            const unsigned long long c0 = b + q[v][0], c1 = b + q[v][1], c2 = b + q[v][2], c3 = b + q[v][3];
            const unsigned long long lo = odd ? c2 : c0, hi = odd ? c3 : c1;   // what a consumer of MY data wants
            // consumer lane j reads from lane src; the value it needs depends on ITS parity, so send both pairs
            const unsigned long long a0 = shfl_u64(c0, srcA), a1 = shfl_u64(c1, srcA), a2 = shfl_u64(c2, srcA), a3 = shfl_u64(c3, srcA);
            const unsigned long long b0 = shfl_u64(c0, srcB), b1 = shfl_u64(c1, srcB), b2 = shfl_u64(c2, srcB), b3 = shfl_u64(c3, srcB);
            (void)lo; (void)hi;
            ulonglong2* oA = reinterpret_cast<ulonglong2*>(ws.cdf + wbase + v * 256 + lane * 2);
            ulonglong2* oB = reinterpret_cast<ulonglong2*>(ws.cdf + wbase + v * 256 + 128 + lane * 2);
            *oA = odd ? make_ulonglong2(a2, a3) : make_ulonglong2(a0, a1);
            *oB = odd ? make_ulonglong2(b2, b3) : make_ulonglong2(b0, b1);
        }
    } else if (vec && ws.variant == 2) {
        // same pattern, streaming (non-temporal) stores: the CDF is consumed by a later kernel, not by this one
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const int srcA = lane >> 1, srcB = 32 + (lane >> 1);
        const bool odd = lane & 1;
#pragma unroll
        for (int v = 0; v < SCAN_SLABS; ++v) {
            const unsigned long long b = off0 + excl[v];
            const unsigned long long c0 = b + q[v][0], c1 = b + q[v][1], c2 = b + q[v][2], c3 = b + q[v][3];
            const unsigned long long a0 = shfl_u64(c0, srcA), a1 = shfl_u64(c1, srcA), a2 = shfl_u64(c2, srcA), a3 = shfl_u64(c3, srcA);
            const unsigned long long b0 = shfl_u64(c0, srcB), b1 = shfl_u64(c1, srcB), b2 = shfl_u64(c2, srcB), b3 = shfl_u64(c3, srcB);
            u64x2* oA = reinterpret_cast<u64x2*>(ws.cdf + wbase + v * 256 + lane * 2);
            u64x2* oB = reinterpret_cast<u64x2*>(ws.cdf + wbase + v * 256 + 128 + lane * 2);
            const u64x2 vA = odd ? (u64x2){a2, a3} : (u64x2){a0, a1};
            const u64x2 vB = odd ? (u64x2){b2, b3} : (u64x2){b0, b1};
            __builtin_nontemporal_store(vA, oA);
            __builtin_nontemporal_store(vB, oB);
        }
    } else if (vec) {
#pragma unroll
        for (int v = 0; v < SCAN_SLABS; ++v) {
            ulonglong2* o2 = reinterpret_cast<ulonglong2*>(ws.cdf + wbase + v * 256 + lane * 4);
            const unsigned long long b = off0 + excl[v];
            o2[0] = make_ulonglong2(b + q[v][0], b + q[v][1]);
            o2[1] = make_ulonglong2(b + q[v][2], b + q[v][3]);
        }
    } else {
#pragma unroll
        for (int v = 0; v < SCAN_SLABS; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long i = wbase + v * 256 + lane * 4 + e;
                if (i < n) ws.cdf[i] = off0 + excl[v] + q[v][e];
            }
    }
}

// Same scan, LDS-transposed: the wave's 2048 weights are loaded with fully coalesced 1-KiB instructions, turned
// in LDS so that lane l owns the 32 CONSECUTIVE weights [32 l, 32 l + 32), summed serially in registers (one
// 6-step wave scan per 2048 weights instead of one per 256), and the 2048 CDF values are turned back through
// LDS into fully coalesced non-temporal 1-KiB stores.  Integer sums: bit-identical to k_scan_fixed.
// LDS per wave: 64 lanes x 18 u64 (the CDF goes back in two halves of 16 values per lane; lane stride 144 B);
// the float staging of the input (lane stride 36 floats = 144 B) aliases the same region.  73 KB per
// workgroup -> two workgroups per CU.
constexpr int SCAN_LSTRIDE = 18;                        // u64 per lane row (16 + 2 pad)
constexpr int SCAN_FSTRIDE = 36;                        // floats per lane row of the input staging

__global__ __launch_bounds__(64 * SCAN_NW) void k_scan_fixed_lds(const float* __restrict__ lw, long n,
                                                                 const float* __restrict__ max_val, ScanWs ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long scan_lds[];
    __shared__ unsigned long long wave_tot[SCAN_NW];
    __shared__ unsigned long long blk_prefix;
    __shared__ unsigned int blk_id_sh;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) blk_id_sh = atomicAdd(ws.ticket, 1u);
    __syncthreads();
    const unsigned int blk = blk_id_sh;
    const long wbase = (long)blk * SCAN_BLOCK + (long)wave * SCAN_WAVE_ITEMS;
    const float mx = max_val[0];
    unsigned long long* wl = scan_lds + (size_t)wave * 64 * SCAN_LSTRIDE;
    float* wf = reinterpret_cast<float*>(wl);
    const bool vec = (wbase + SCAN_WAVE_ITEMS <= n) && (((size_t)lw & 15) == 0);
    unsigned long long run[32];                          // inclusive sums of this lane's 32 consecutive weights
    if (vec) {
        float4 x[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) x[h] = *reinterpret_cast<const float4*>(lw + wbase + h * 256 + lane * 4);
#pragma unroll
        for (int h = 0; h < 8; ++h)                      // item 256 h + 4 lane -> row (8 h + lane / 8), column 4 (lane % 8)
            *reinterpret_cast<float4*>(wf + (8 * h + (lane >> 3)) * SCAN_FSTRIDE + 4 * (lane & 7)) = x[h];
    } else {
        for (int i = lane; i < SCAN_WAVE_ITEMS; i += 64) {
            const long g = wbase + i;
            wf[(i >> 5) * SCAN_FSTRIDE + (i & 31)] = (g < n) ? lw[g] : -INFINITY;
        }
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long acc = 0ull;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 y = *reinterpret_cast<const float4*>(wf + lane * SCAN_FSTRIDE + 4 * k);
        acc += fixed_weight(y.x, mx); run[4 * k + 0] = acc;
        acc += fixed_weight(y.y, mx); run[4 * k + 1] = acc;
        acc += fixed_weight(y.z, mx); run[4 * k + 2] = acc;
        acc += fixed_weight(y.w, mx); run[4 * k + 3] = acc;
    }
    unsigned long long incl = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = shfl_up_u64(incl, off);
        if (lane >= off) incl += o;
    }
    const unsigned long long lane_excl = incl - acc;
    const unsigned long long wtot = shfl_u64(incl, 63);
    if (lane == 0) wave_tot[wave] = wtot;
    __syncthreads();
    unsigned long long wave_off = 0ull, blk_tot = 0ull;
#pragma unroll
    for (int w = 0; w < SCAN_NW; ++w) { if (w < wave) wave_off += wave_tot[w]; blk_tot += wave_tot[w]; }
    if (wave == 0) {
        if (blk == 0) {
            if (lane == 0) {
                __hip_atomic_store(ws.desc + 0, FLAG_INC | blk_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                blk_prefix = 0ull;
            }
        } else {
            if (lane == 0)
                __hip_atomic_store(ws.desc + blk, FLAG_AGG | blk_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long ex = 0ull;
            long look = (long)blk - 1;
            while (true) {
                const long j = look - lane;
                unsigned long long d = 0ull;
                if (j >= 0) {
                    do {
                        d = __hip_atomic_load(ws.desc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((d >> 62) == 0ull);
                } else {
                    d = FLAG_INC;
                }
                const unsigned long long inc_mask = __ballot((d >> 62) == 2ull);
                const int first_inc = __ffsll((long long)inc_mask) - 1;
                unsigned long long contrib = (first_inc < 0 || lane <= first_inc) ? (d & VAL_MASK) : 0ull;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) contrib += shfl_u64(contrib, lane ^ off);
                ex += contrib;
                if (first_inc >= 0) break;
                look -= 64;
            }
            if (lane == 0) {
                __hip_atomic_store(ws.desc + blk, FLAG_INC | (ex + blk_tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                blk_prefix = ex;
            }
        }
    }
    __syncthreads();
    const unsigned long long off0 = blk_prefix + wave_off;
    if (lane == 63 && (wave & 1) && wbase - SCAN_WAVE_ITEMS < n)
        ws.tile_inc[((long)blk * SCAN_BLOCK + (long)(wave - 1) * SCAN_WAVE_ITEMS) / SCAN_TILE] = off0 + wtot;
    const unsigned long long base = off0 + lane_excl;
    if (ws.want_sub) {                                   // sub-sampled CDF: the last value of each 16-group (padding weighs 0)
        const long g0 = (wbase + 32 * lane) >> 4;
        if ((g0 << 4) < n) ws.cdf16[g0] = base + run[15];
        if (((g0 + 1) << 4) < n) ws.cdf16[g0 + 1] = base + run[31];
        if ((lane & 7) == 7 && wbase + 32 * (lane - 7) < n) ws.cdf256[(wbase + 32 * lane) >> 8] = base + run[31];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                        // values 16 h .. 16 h + 15 of every lane
#pragma unroll
        for (int k = 0; k < 8; ++k)
            *reinterpret_cast<u64x2*>(wl + lane * SCAN_LSTRIDE + 2 * k) =
                (u64x2){base + run[16 * h + 2 * k], base + run[16 * h + 2 * k + 1]};
        __builtin_amdgcn_wave_barrier();
        if (vec) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {                // 8 lanes write one full 128-byte line of row 8 s + lane / 8
                const int row = 8 * s + (lane >> 3), col = 2 * (lane & 7);
                const u64x2 c = *reinterpret_cast<const u64x2*>(wl + row * SCAN_LSTRIDE + col);
                __builtin_nontemporal_store(c, reinterpret_cast<u64x2*>(ws.cdf + wbase + 32 * row + 16 * h + col));
            }
        } else {
            for (int i = lane; i < SCAN_WAVE_ITEMS / 2; i += 64) {
                const long g = wbase + 32 * (i >> 4) + 16 * h + (i & 15);
                if (g < n) ws.cdf[g] = wl[(i >> 4) * SCAN_LSTRIDE + (i & 15)];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// first j with C[j] > t : search the tile prefixes, then inside the tile
__device__ __forceinline__ long search_cdf(const ScanWs& ws, long n, long ntiles, unsigned long long t) {
    long lo = 0, hi = ntiles;
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (ws.tile_inc[mid] > t) hi = mid; else lo = mid + 1; }
    if (lo >= ntiles) return n - 1;
    long a = lo * SCAN_TILE, b = a + SCAN_TILE;
    if (b > n) b = n;
    while (a < b) { const long mid = (a + b) >> 1; if (ws.cdf[mid] > t) b = mid; else a = mid + 1; }
    return a < n ? a : n - 1;
}

// Multinomial sampler: the thresholds are unordered, so every draw walks the CDF on its own.  A bisection of a 4096-entry
// tile of the full CDF touches 9 distinct cache lines of an 8N-byte array per draw, and - with thousands of searches in
// flight per CU - even the last steps inside one line miss the 32 KB L1 again (measured: splitting the tile search over
// sub-sampled tables cut the time only from 9.6 to 5.1 ms at N = 2^26, more searches in flight per thread made it slower).
// So below the tile the search is 16-ary over three tables the LDS scan writes - cdf256 (every 256th value, N/32 bytes,
// L2-resident), cdf16 (N/2 bytes, MALL), the CDF itself - and each 16-entry node is ONE 128-byte line fetched ONCE by a
// group of 8 lanes (16 bytes each, coalesced); the position inside the node is a ballot + popcount.  A wave serves its
// 64 draws in 8 rounds (group g, round r: the draw of lane 8 g + r), all 8 rounds of a level in flight together.
// The tile prefixes are one more 16-ary level; above it the last prefix of every `cs`-th 16-tile node (<= 4096 entries,
// cs = 1 up to N = 2^28) is bisected in LDS.
constexpr int MN_COARSE = 4096;

__device__ __forceinline__ long shfl_i64(long v, int src) { return (long)shfl_u64((unsigned long long)v, src); }

// one 16-ary level for the wave's 8 rounds: p[r] (node start, multiple of 16) += entries <= t among the node's 16
__device__ __forceinline__ void node16_level(const unsigned long long* __restrict__ table, long len, long (&p)[8],
                                             const unsigned long long (&t)[8], int sub, int gbase) {
    ulonglong2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = *reinterpret_cast<const ulonglong2*>(table + p[r] + 2 * sub);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const long i0 = p[r] + 2 * sub;
        const unsigned long long bx = __ballot(i0 < len && v[r].x <= t[r]);
        const unsigned long long by = __ballot(i0 + 1 < len && v[r].y <= t[r]);
        p[r] += __popc((unsigned)((bx >> gbase) & 0xffull)) + __popc((unsigned)((by >> gbase) & 0xffull));
    }
}

__global__ __launch_bounds__(256) void k_sample_multinomial(ScanWs ws, long n, long ntiles, const double* __restrict__ u,
                                                            long ns, long long* __restrict__ idx) {
    __shared__ unsigned long long coarse[MN_COARSE];
    const unsigned long long total = ws.tile_inc[ntiles - 1];
    if (ws.variant != 3) {                           // only the LDS scan writes the sub-sampled tables
        for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < ns; k += (long)gridDim.x * blockDim.x) {
            unsigned long long t = 0ull;
            if (total > 0ull) {
                t = (unsigned long long)floor(u[k] * (double)total);
                if (t > total - 1ull) t = total - 1ull;
            }
            idx[k] = search_cdf(ws, n, ntiles, t);
        }
        return;
    }
    // LDS: the inclusive prefix at the end of every `cs`-th 16-tile node (cs = 1 up to N = 2^28)
    const long nnodes = (ntiles + 15) >> 4;
    const long cs = (nnodes + MN_COARSE - 1) / MN_COARSE, nc = (nnodes + cs - 1) / cs;
    for (long i = threadIdx.x; i < nc; i += blockDim.x) {
        const long j = 16 * (i + 1) * cs - 1;
        coarse[i] = ws.tile_inc[j < ntiles ? j : ntiles - 1];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = lane & 7, gbase = lane & ~7;
    const long n256 = (n + 255) >> 8, n16 = (n + 15) >> 4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long w0 = (long)blockIdx.x * blockDim.x + (threadIdx.x & ~63); w0 < ns; w0 += stride) {   // wave-uniform trip count
        const long k = w0 + lane;
        unsigned long long t = 0ull;
        if (k < ns && total > 0ull) {
            t = (unsigned long long)floor(u[k] * (double)total);
            if (t > total - 1ull) t = total - 1ull;
        }
        // this lane's 16-tile node: first node whose last inclusive prefix exceeds t
        long lo = 0, hi = nc;
        while (lo < hi) { const long mid = (lo + hi) >> 1; if (coarse[mid] > t) hi = mid; else lo = mid + 1; }
        lo *= cs; hi = lo + cs;
        if (hi > nnodes) hi = nnodes;
        while (lo < hi) {
            const long mid = (lo + hi) >> 1, j = 16 * mid + 15;
            if (ws.tile_inc[j < ntiles ? j : ntiles - 1] > t) hi = mid; else lo = mid + 1;
        }
        if (lo > nnodes - 1) lo = nnodes - 1;        // (only when total == 0)
        long p[8];
        unsigned long long tr[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            tr[r] = shfl_u64(t, gbase + r);
            p[r] = shfl_i64(lo, gbase + r) << 4;
        }
        node16_level(ws.tile_inc, ntiles, p, tr, sub, gbase);
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r] = (p[r] < ntiles ? p[r] : ntiles - 1) * (SCAN_TILE / 256);
        node16_level(ws.cdf256, n256, p, tr, sub, gbase);
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r] = (p[r] < n256 ? p[r] : n256 - 1) << 4;
        node16_level(ws.cdf16, n16, p, tr, sub, gbase);
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r] = (p[r] < n16 ? p[r] : n16 - 1) << 4;
        node16_level(ws.cdf, n, p, tr, sub, gbase);
        long res = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) res = sub == r ? p[r] : res;
        if (k < ns) idx[k] = (total > 0ull && res < n) ? res : n - 1;
    }
}

// stratum map of the systematic resampler (see the fused path below and oracle/numerical.py:systematic_strata)
struct Strata {            // t_k = (k * S + U) >> F
    unsigned long long total, S, U;
    int F;
};

__device__ __forceinline__ Strata make_strata(unsigned long long total, double u0, long ns) {
    Strata m;
    m.total = total;
    m.F = total ? __clzll((long long)total) - 2 : 0;           // 62 - bit_length(total)
    m.S = total ? (total << m.F) / (unsigned long long)ns : 0ull;
    unsigned long long U = (unsigned long long)floor(u0 * (double)m.S);
    m.U = (m.S && U > m.S - 1ull) ? m.S - 1ull : U;
    return m;
}
__device__ __forceinline__ Strata load_strata(const unsigned long long* p) {
    Strata m;
    m.total = p[0]; m.S = p[1]; m.U = p[2]; m.F = (int)p[3];
    return m;
}

// Systematic thresholds are sorted, so a chunk of 4096 consecutive samples lands in a short run of CDF
// tiles: stage each tile (32 KiB) in LDS and binary-search there -> indices are written coalesced and the
// work is balanced in SAMPLES whatever the weight distribution (a tile-driven variant was 1.75x slower on
// heavy-tailed weights).  Chunks that span many (near-empty) tiles fall back to the global two-level search;
// the total number of staged tiles is bounded by ntiles + nchunks.
constexpr int SYS_CHUNK = 4096, SYS_PER_THREAD = SYS_CHUNK / 256, SYS_MAX_SPAN = 8;

__global__ __launch_bounds__(256) void k_sample_systematic(ScanWs ws, long n, long ntiles,
                                                           const unsigned long long* __restrict__ strata_ptr, long ns,
                                                           long long* __restrict__ idx) {
    __shared__ unsigned long long cdf_sh[SCAN_TILE];
    __shared__ long span_sh[2];
    const int tid = threadIdx.x;
    const Strata sm = load_strata(strata_ptr);
    const unsigned long long total = sm.total;
    const long k0 = (long)blockIdx.x * SYS_CHUNK;
    unsigned long long t[SYS_PER_THREAD];
#pragma unroll
    for (int m = 0; m < SYS_PER_THREAD; ++m) {
        const long k = k0 + tid + 256 * m;
        t[m] = (total > 0ull && k < ns) ? ((unsigned long long)k * sm.S + sm.U) >> sm.F : 0ull;
    }
    if (tid == 0) {                 // first / last tile touched by this chunk (thresholds are monotone in k)
        const long kl = (k0 + SYS_CHUNK <= ns ? k0 + SYS_CHUNK : ns) - 1;
        const unsigned long long tl = total > 0ull ? ((unsigned long long)kl * sm.S + sm.U) >> sm.F : 0ull;
        long lo = 0, hi = ntiles;
        while (lo < hi) { const long mid = (lo + hi) >> 1; if (ws.tile_inc[mid] > t[0]) hi = mid; else lo = mid + 1; }
        span_sh[0] = lo < ntiles ? lo : ntiles - 1;
        lo = 0; hi = ntiles;
        while (lo < hi) { const long mid = (lo + hi) >> 1; if (ws.tile_inc[mid] > tl) hi = mid; else lo = mid + 1; }
        span_sh[1] = lo < ntiles ? lo : ntiles - 1;
    }
    __syncthreads();
    const long ta = span_sh[0], tb = span_sh[1];
    if (tb - ta + 1 > SYS_MAX_SPAN) {
#pragma unroll
        for (int m = 0; m < SYS_PER_THREAD; ++m) {
            const long k = k0 + tid + 256 * m;
            if (k < ns) idx[k] = search_cdf(ws, n, ntiles, t[m]);
        }
        return;
    }
    unsigned done = 0u;
    for (long tile = ta; tile <= tb; ++tile) {
        const long base = tile * SCAN_TILE;
        const long len = (base + SCAN_TILE <= n) ? SCAN_TILE : n - base;
        for (int i = tid * 2; i < SCAN_TILE; i += 512) {
            if (i + 1 < len) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(ws.cdf + base + i);
                cdf_sh[i] = v.x; cdf_sh[i + 1] = v.y;
            } else {
                cdf_sh[i] = (i < len) ? ws.cdf[base + i] : ~0ull;
                cdf_sh[i + 1] = ~0ull;
            }
        }
        __syncthreads();
        const unsigned long long c_end = ws.tile_inc[tile];
#pragma unroll
        for (int m = 0; m < SYS_PER_THREAD; ++m) {
            const long k = k0 + tid + 256 * m;
            if (k < ns && !((done >> m) & 1u) && (t[m] < c_end || tile == tb)) {
                int a = 0, b = (int)len;
                while (a < b) { const int mid = (a + b) >> 1; if (cdf_sh[mid] > t[m]) b = mid; else a = mid + 1; }
                long r = base + a;
                idx[k] = r < n ? r : n - 1;
                done |= 1u << m;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Systematic resampling.  Stratum map (oracle/numerical.py:systematic_strata restates it): pure 64-bit integers
//     F = 62 - bit_length(total),  S = floor(total 2^F / ns),  U = min(floor(u0 S), S - 1),   t_k = (k S + U) >> F
// i.e. ns equal strata of width S / 2^F offset by u0 strata.  Its inverse, "the first stratum whose threshold
// reaches C", is K(C) = ceil(((C << F) - U) / S), evaluated as a float64 estimate + ONE exact integer correction step.
//
// Fused path (no CDF in HBM, no per-stratum search):
//   pass T  k_emit_wave_sums : fixed-point weight sum of every 1024-weight wave tile + of every 4096-weight block (EM_NW waves)
//           k_emit_prefix    : exclusive prefix over the block sums (one workgroup), total, (F, S, U)      (reads 4N)
//   pass E  k_emit_systematic: every wave re-derives the CDF of its 1024 weights in registers (coalesced loads, LDS
//           transpose so that a lane owns 16 consecutive weights, serial sums + one 6-step wave scan).  Weight j owns the
//           CONTIGUOUS strata [K(C_{j-1}), K(C_j)): the resampled index array is an EXPANSION - weight j repeated
//           K(C_j) - K(C_{j-1}) times.  Every weight with a non-empty run drops ONE marker (its id) at the first
//           position of its run in a 1536-entry LDS window; an inclusive max-scan (ids grow with position) fills the
//           runs - no data-dependent loop, no divergence, a heavy weight costs nothing extra - and the window is
//           flushed with coalesced 32-byte-per-lane stores.  Runs >= 32768 strata (all the mass on a few particles)
//           are published and filled by a grid-wide kernel instead of one wave.                 (reads 4N, writes 8 ns)
// Integer prefix sums and an exact inverse => bit-identical to the scan + search path and to the oracle.
// HBM traffic 12N + 8 ns bytes instead of 4N + 12N + 16 ns.
// ------------------------------------------------------------------------------------------------
constexpr int EM_NW = 4;                                // waves per workgroup (r4: 4 x 7 KiB of LDS -> 5 workgroups = 20 waves per CU at <= 96
                                                        // VGPRs, was 8 waves x 9 KiB -> 16 per CU: 402 -> 382 us end to end at 2^26)
constexpr int EM_WAVE_ITEMS = 1024;                     // weights per wave: 16 consecutive per lane
constexpr int EM_BLOCK = EM_WAVE_ITEMS * EM_NW;
constexpr int EM_FSTRIDE = 20;                          // floats per lane row of the input staging (16 + 4 pad)
constexpr int EM_WIN = 1536;                            // int32 marker window (entries) per wave: EM_ROW = 24 per lane row (a wave owns ~1024
                                                        // strata when ns = N; 1024 entries: two passes for half the waves, 402 us)
constexpr int EM_WSTRIDE = 28;                          // dwords per lane row of the window (24 + 4 pad: 28 i mod 64 distinct multiples of 4
                                                        // over 16 lanes: conflict-free b128)
constexpr int EM_WAVE_LDS = 64 * EM_WSTRIDE * 4;        // bytes per wave (>= 64 * EM_FSTRIDE * 4)
constexpr int EM_ROW = EM_WIN / 64;                     // window entries per lane row
__device__ __forceinline__ int em_waddr(int e) { const int r = (int)((unsigned)e / (unsigned)EM_ROW); return r * EM_WSTRIDE + (e - r * EM_ROW); }
constexpr int EM_GIANT = 32768;                         // runs at least this long are filled by a grid-wide kernel

__global__ __launch_bounds__(64 * EM_NW) void k_emit_wave_sums(const float* __restrict__ lw, long n,
                                                               const float* __restrict__ max_val,
                                                               unsigned long long* __restrict__ wave_sum,
                                                               unsigned long long* __restrict__ block_sum) {
    __shared__ unsigned long long wsum[EM_NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wt = (long)blockIdx.x * EM_NW + wave;
    const long wbase = wt * EM_WAVE_ITEMS;
    const float mx = max_val[0];
    unsigned long long acc = 0ull;
    if (wbase + EM_WAVE_ITEMS <= n && (((size_t)lw & 15) == 0)) {
        float4 x[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) x[h] = *reinterpret_cast<const float4*>(lw + wbase + h * 256 + lane * 4);
#pragma unroll
        for (int h = 0; h < 4; ++h)
            acc += fixed_weight(x[h].x, mx) + fixed_weight(x[h].y, mx) + fixed_weight(x[h].z, mx) + fixed_weight(x[h].w, mx);
    } else {
        for (int i = lane; i < EM_WAVE_ITEMS; i += 64)
            if (wbase + i < n) acc += fixed_weight(lw[wbase + i], mx);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += shfl_u64(acc, lane ^ off);
    if (lane == 0) { wsum[wave] = acc; if (wbase < n) wave_sum[wt] = acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = 0ull;
        for (int w = 0; w < EM_NW; ++w) b += wsum[w];
        block_sum[blockIdx.x] = b;
    }
}

// exclusive prefix over the block sums (nb = ceil(N / EM_BLOCK)): one 1024-thread workgroup, a contiguous chunk per thread;
// also the stratum map {total, S, U, F} and the reset of the giant-run counter
__global__ __launch_bounds__(1024) void k_emit_prefix(const unsigned long long* __restrict__ block_sum, long nb,
                                                      unsigned long long* __restrict__ block_excl, double u0, long ns,
                                                      unsigned long long* __restrict__ strata_out,
                                                      unsigned long long* __restrict__ giant_count) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    if (tid == 0) *giant_count = 0ull;
    const long chunk = (nb + 1023) / 1024;
    const long a = (long)tid * chunk, b = a + chunk < nb ? a + chunk : nb;
    unsigned long long s = 0ull;
    for (long i = a; i < b; ++i) s += block_sum[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {               // Hillis-Steele inclusive scan of the chunk sums
        const unsigned long long v = tid >= off ? part[tid - off] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = tid ? part[tid - 1] : 0ull;
    for (long i = a; i < b; ++i) { block_excl[i] = run; run += block_sum[i]; }
    if (tid == 1023) {
        const Strata m = make_strata(part[1023], u0, ns);
        strata_out[0] = m.total; strata_out[1] = m.S; strata_out[2] = m.U; strata_out[3] = (unsigned long long)m.F;
    }
}

// stratum map for the scan + search path (total = last tile prefix)
__global__ void k_strata_params(const unsigned long long* __restrict__ total_ptr, double u0, long ns,
                                unsigned long long* __restrict__ strata_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Strata m = make_strata(*total_ptr, u0, ns);
    strata_out[0] = m.total; strata_out[1] = m.S; strata_out[2] = m.U; strata_out[3] = (unsigned long long)m.F;
}

// K(C) - Kref for a CDF value C = c0 + d (d < 2^47), as int32: float64 estimate of ceil(((C << F) - U) / S) with
// num0 = (double)((c0 << F) - U) (signed) supplied by the caller, then one exact integer correction step
// (estimate error << 1 stratum: the conversions lose < 2^-26 of a stratum, the reciprocal 2^-27).
__device__ __forceinline__ int stratum_rel(const Strata& m, unsigned long long c0, unsigned long long d, double num0,
                                           double pow2F, double invS, long Kref, long ns) {
    const unsigned long long C = c0 + d;
    if (C >= m.total) return (int)(ns - Kref);
    const double y = ((double)d * pow2F + num0) * invS;
    long k = (long)(int)ceil(y);
    if (k < 0) k = 0;
    if (k > ns) k = ns;
    const unsigned long long B = C << m.F;
    const unsigned long long v = (unsigned long long)k * m.S + m.U;          // (t_k << F) + fractional bits
    if (v < B) { if (k < ns) ++k; }                                           // t_k < C: the next stratum is the first
    else if (k > 0 && v - m.S >= B) --k;                                      // t_{k-1} >= C already
    return (int)(k - Kref);
}

// The same for the items of ONE wave whose strata span T < 2^20: relative to the wave's first stratum K0 the quotient is
// small, so an fp32 estimate of ceil(X / S), X = Bw + (d << F) with the wave constant Bw = (c_start << F) - U - K0 S in
// (-S, 0], is off by less than one stratum and ONE exact 64-bit check decides (branch-free: 22 instead of ~45
// instructions + 3 divergent branches per item).  Same value as stratum_rel(..., Kref = K0) by construction.
__device__ __forceinline__ int stratum_rel_small(const Strata& m, long long Bw, float invS_f, unsigned long long C_abs,
                                                 unsigned long long d, int ns_rel) {
    const long long X = Bw + (long long)(d << m.F);
    const float xf = (float)(int)(X >> 32) * 4294967296.0f + (float)(unsigned)X;
    int k = (int)ceilf(xf * invS_f);
    k = k < 0 ? 0 : (k > ns_rel ? ns_rel : k);
    const long long R = X - (long long)((unsigned long long)(unsigned)k * m.S);        // X - k S, exact
    k += (R > 0 && k < ns_rel) ? 1 : 0;
    k -= (R + (long long)m.S <= 0 && k > 0) ? 1 : 0;
    return C_abs >= m.total ? ns_rel : k;
}

__global__ __launch_bounds__(64 * EM_NW) void k_emit_systematic(const float* __restrict__ lw, long n,
                                                                const float* __restrict__ max_val,
                                                                const unsigned long long* __restrict__ wave_sum,
                                                                const unsigned long long* __restrict__ block_excl,
                                                                const unsigned long long* __restrict__ strata_ptr,
                                                                long ns, long long* __restrict__ idx,
                                                                unsigned long long* __restrict__ giant_count,
                                                                long long* __restrict__ giant_desc, long giant_cap) {
    __shared__ __attribute__((aligned(16))) unsigned char em_lds[EM_NW * EM_WAVE_LDS];
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wt = (long)blockIdx.x * EM_NW + wave;
    const long wbase = wt * EM_WAVE_ITEMS;
    if (wbase >= n) return;
    const Strata m = load_strata(strata_ptr);
    if (m.total == 0ull) {                                 // all weights zero: every stratum maps to the last index
        if (wt == 0)
            for (long k = lane; k < ns; k += 64) idx[k] = n - 1;
        return;
    }
    float* wf = reinterpret_cast<float*>(em_lds + (size_t)wave * EM_WAVE_LDS);
    int* win = reinterpret_cast<int*>(wf);
    const float mx = max_val[0];
    unsigned long long c_start = block_excl[blockIdx.x];
    for (int w = 0; w < wave; ++w) c_start += wave_sum[(long)blockIdx.x * EM_NW + w];
    // ---- CDF of this wave's 1024 weights: lane owns items [16 lane, 16 lane + 16) ----
    if (wbase + EM_WAVE_ITEMS <= n && (((size_t)lw & 15) == 0)) {
        float4 x[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) x[h] = *reinterpret_cast<const float4*>(lw + wbase + h * 256 + lane * 4);
#pragma unroll
        for (int h = 0; h < 4; ++h)                      // item 256 h + 4 lane -> row 16 h + lane / 4, column 4 (lane % 4)
            *reinterpret_cast<float4*>(wf + (16 * h + (lane >> 2)) * EM_FSTRIDE + 4 * (lane & 3)) = x[h];
    } else {
        for (int i = lane; i < EM_WAVE_ITEMS; i += 64)
            wf[(i >> 4) * EM_FSTRIDE + (i & 15)] = (wbase + i < n) ? lw[wbase + i] : -INFINITY;
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long run[16];
    unsigned long long acc = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 y = *reinterpret_cast<const float4*>(wf + lane * EM_FSTRIDE + 4 * k);
        acc += fixed_weight(y.x, mx); run[4 * k + 0] = acc;
        acc += fixed_weight(y.y, mx); run[4 * k + 1] = acc;
        acc += fixed_weight(y.z, mx); run[4 * k + 2] = acc;
        acc += fixed_weight(y.w, mx); run[4 * k + 3] = acc;
    }
    unsigned long long incl = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = shfl_up_u64(incl, off);
        if (lane >= off) incl += o;
    }
    const unsigned long long wtot = shfl_u64(incl, 63);
    if (wtot == 0ull) return;                              // no stratum lands in a zero-weight tile
    const unsigned long long lane_off = incl - acc;        // CDF offset of this lane's first item inside the wave
    // ---- stratum ranges relative to K0 = K(c_start): item j owns [ke[j-1], ke[j]) (ke[-1] = previous lane's ke[15]) ----
    const double pow2F = (double)(1ull << m.F), invS = 1.0 / (double)m.S;
    const unsigned long long B0 = c_start << m.F;
    const double num0 = B0 >= m.U ? (double)(B0 - m.U) : -(double)(m.U - B0);
    const long K0 = (long)stratum_rel(m, c_start, 0ull, num0, pow2F, invS, 0, ns) ;   // wave-uniform (Kref = 0: < 2^31)
    const int T = stratum_rel(m, c_start, wtot, num0, pow2F, invS, K0, ns);            // strata owned by this wave (uniform)
    if (T <= 0) return;
    int ke[16];
    if (T < (1 << 20)) {                                   // wave-uniform: the usual case
        const long long Bw = (long long)(B0 - m.U - (unsigned long long)K0 * m.S);
        const float invS_f = 1.0f / (float)m.S;
        const int ns_rel = (int)(ns - K0);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            ke[j] = stratum_rel_small(m, Bw, invS_f, c_start + lane_off + run[j], lane_off + run[j], ns_rel);
    } else {                                               // giant runs: float64 estimate per item
#pragma unroll
        for (int j = 0; j < 16; ++j)
            ke[j] = (j > 0 && run[j] == run[j - 1]) ? ke[j - 1]
                                                    : stratum_rel(m, c_start, lane_off + run[j], num0, pow2F, invS, K0, ns);
    }
    int kprev = __shfl_up(ke[15], 1);
    if (lane == 0) kprev = 0;
    __builtin_amdgcn_wave_barrier();                       // every lane is done reading the float staging
    const long item0 = wbase;
    // giant runs: publish, the grid-wide fill kernel writes them; this wave skips windows entirely inside one
    unsigned gmask = 0u;                                   // bit j: item j of this lane is a PUBLISHED giant run
    if (T >= EM_GIANT) {                                   // wave-uniform
        int sg = kprev;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (ke[j] - sg >= EM_GIANT) {
                const unsigned long long slot = atomicAdd(giant_count, 1ull);
                if ((long)slot < giant_cap) {              // (list full: the run stays with this wave)
                    long id = item0 + 16 * lane + j;
                    giant_desc[3 * slot] = K0 + sg; giant_desc[3 * slot + 1] = K0 + ke[j];
                    giant_desc[3 * slot + 2] = id < n ? id : n - 1;
                    gmask |= 1u << j;
                }
            }
            sg = ke[j];
        }
    }
    const bool wave_has_giant = __ballot(gmask != 0u) != 0ull;
    for (int w0 = 0; w0 < T; w0 += EM_WIN) {
        const int wlen = T - w0 < EM_WIN ? T - w0 : EM_WIN;
        const int whi = w0 + wlen;
        if (wave_has_giant) {                              // a window entirely inside ONE published giant run: skip
            bool cover = false;
            int sg = kprev;
#pragma unroll
            for (int j = 0; j < 16; ++j) { if (((gmask >> j) & 1u) && sg <= w0 && ke[j] >= whi) cover = true; sg = ke[j]; }
            if (__ballot(cover) != 0ull) continue;
        }
        // (a) clear the window, (b) every non-empty run drops its id + 1 at max(run start, window start)
#pragma unroll
        for (int i = 0; i < 64 * EM_WSTRIDE / 256; ++i)
            *reinterpret_cast<int4*>(win + 4 * lane + 256 * i) = make_int4(0, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
        {
            int sg = kprev;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int e = ke[j];
                const int pos = sg > w0 ? sg : w0;
                if (e > pos && pos < whi) win[em_waddr(pos - w0)] = 16 * lane + j + 1;
                sg = e;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // (c) inclusive max-scan: lane owns entries [32 lane, 32 lane + 32)
        int v[EM_ROW];
#pragma unroll
        for (int i = 0; i < EM_ROW / 4; ++i) {
            const int4 q = *reinterpret_cast<const int4*>(win + EM_WSTRIDE * lane + 4 * i);
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        }
#pragma unroll
        for (int i = 1; i < EM_ROW; ++i) v[i] = v[i] > v[i - 1] ? v[i] : v[i - 1];
        int mx_in = v[EM_ROW - 1];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(mx_in, off);
            if (lane >= off) mx_in = mx_in > o ? mx_in : o;
        }
        int carry = __shfl_up(mx_in, 1);
        if (lane == 0) carry = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < EM_ROW / 4; ++i) {
            int4 q;
            q.x = (v[4 * i] > carry ? v[4 * i] : carry) - 1;
            q.y = (v[4 * i + 1] > carry ? v[4 * i + 1] : carry) - 1;
            q.z = (v[4 * i + 2] > carry ? v[4 * i + 2] : carry) - 1;
            q.w = (v[4 * i + 3] > carry ? v[4 * i + 3] : carry) - 1;
            *reinterpret_cast<int4*>(win + EM_WSTRIDE * lane + 4 * i) = q;
        }
        __builtin_amdgcn_wave_barrier();
        // (d) flush: 4 entries per lane per step -> two 16-byte stores
        const long obase = K0 + w0;
        for (int p = 4 * lane; p < wlen; p += 256) {
            const int4 q = *reinterpret_cast<const int4*>(win + em_waddr(p));
            long r0 = item0 + q.x, r1 = item0 + q.y, r2 = item0 + q.z, r3 = item0 + q.w;
            if (r3 >= n) { r0 = r0 < n ? r0 : n - 1; r1 = r1 < n ? r1 : n - 1; r2 = r2 < n ? r2 : n - 1; r3 = n - 1; }
            if (p + 4 <= wlen) {
                i64x2* o = reinterpret_cast<i64x2*>(idx + obase + p);
                o[0] = (i64x2){r0, r1};
                o[1] = (i64x2){r2, r3};
            } else {
                idx[obase + p] = r0;
                if (p + 1 < wlen) idx[obase + p + 1] = r1;
                if (p + 2 < wlen) idx[obase + p + 2] = r2;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// constant fill of the published giant runs: workgroup b takes 16384-entry chunks (run r, chunk c) round-robin
__global__ __launch_bounds__(256) void k_emit_fill_runs(const unsigned long long* __restrict__ giant_count,
                                                        const long long* __restrict__ giant_desc, long giant_cap,
                                                        long long* __restrict__ idx) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    long cnt = (long)*giant_count;
    if (cnt > giant_cap) cnt = giant_cap;
    constexpr long CH = 16384;
    unsigned work = 0;
    for (long r = 0; r < cnt; ++r) {
        const long a = giant_desc[3 * r], b = giant_desc[3 * r + 1], id = giant_desc[3 * r + 2];
        const long nch = (b - a + CH - 1) / CH;
        for (long c = 0; c < nch; ++c, ++work) {
            if (work % gridDim.x != blockIdx.x) continue;
            const long lo = a + c * CH, hi = lo + CH < b ? lo + CH : b;
            long p = lo + 2 * threadIdx.x;
            if ((lo & 1) && threadIdx.x == 0) idx[lo] = id;           // align the 16-byte stores
            p += (lo & 1);
            for (; p + 1 < hi; p += 512) *reinterpret_cast<i64x2*>(idx + p) = (i64x2){id, id};
            if (p < hi) idx[p] = id;
        }
    }
}

__global__ void k_gather_rows(const float* __restrict__ src, const long long* __restrict__ idx, float* __restrict__ dst,
                              long n_out, long row_len) {
    if ((row_len & 3) == 0 && ((size_t)src & 15) == 0 && ((size_t)dst & 15) == 0) {
        const long r4 = row_len >> 2;
        const long total = n_out * r4;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
            const long k = e / r4, j = e % r4;
            reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[idx[k] * r4 + j];
        }
    } else {
        const long total = n_out * row_len;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
            const long k = e / row_len, j = e % row_len;
            dst[e] = src[idx[k] * row_len + j];
        }
    }
}

// standalone target log-prob (+grad) kernel
template <bool GRAD>
__global__ __launch_bounds__(NTHREADS) void k_target(TargetDev tg, const float* __restrict__ x, float* __restrict__ lp,
                                                     float* __restrict__ grad, long B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Tid t;
    const int D = tg.dim;
    float* X = lds;
    float* G = lds + ROWS * D;
    const long row0 = (long)blockIdx.x * ROWS;
    for (int e = t.tid; e < ROWS * D; e += NTHREADS) {
        const long g = row0 + e / D;
        X[e] = g < B ? x[g * D + e % D] : 0.f;
    }
    __syncthreads();
    const float v = target_tile<GRAD>(tg, X, D, G, D, t);
    const long g = row0 + t.row;
    if (g < B) {
        if (t.c == 0) lp[g] = v;
        if (GRAD) for (int j = t.c; j < D; j += 16) grad[g * D + j] = G[t.row * D + j];
    }
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int grid_for(long n, int threads, int cap) {
    long b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

static inline long scan_tiles(long n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

static ScanWs carve_scan_ws(void* workspace, long n) {
    char* p = (char*)workspace;
    ScanWs ws;
    ws.max_part = (float*)p; p += al256(1024 * 4);
    ws.max_val = (float*)p; p += 256;
    ws.desc = (unsigned long long*)p; p += al256((size_t)scan_tiles(n) * 8);
    ws.ticket = (unsigned int*)p; p += 256;
    ws.tile_inc = (unsigned long long*)p; p += al256((size_t)scan_tiles(n) * 8);
    ws.cdf = (unsigned long long*)p; p += al256((size_t)n * 8);
    ws.cdf16 = (unsigned long long*)p; p += al256(((size_t)n / 16 + 1) * 8);
    ws.cdf256 = (unsigned long long*)p;
    ws.want_sub = 0;
    ws.variant = option(FABHIP_OPT_SCAN_VARIANT);   // 3 = LDS-transposed scan; 0-2 = register/shuffle variants (A/B)
    return ws;
}

static int build_fixed_cdf(const float* log_w, long n, const ScanWs& ws, hipStream_t st, bool reuse_max = false) {
    const int mb = grid_for(n, 256 * 16, 1024);
    if (!reuse_max) {
        hipLaunchKernelGGL(k_max_partial, dim3(mb), dim3(256), 0, st, log_w, n, ws.max_part);
        hipLaunchKernelGGL(k_max_final, dim3(1), dim3(256), 0, st, ws.max_part, mb, ws.max_val);
    }
    // zero descriptors + ticket (one contiguous region: desc .. ticket)
    const size_t zbytes = (size_t)((char*)ws.ticket - (char*)ws.desc) + 256;
    if (hipMemsetAsync(ws.desc, 0, zbytes, st) != hipSuccess) return FABHIP_ELAUNCH;
    const dim3 grid((unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK)), block(64 * SCAN_NW);
    if (ws.variant == 3) {
        const size_t lds = (size_t)SCAN_NW * 64 * SCAN_LSTRIDE * sizeof(unsigned long long);
        if (hipFuncSetAttribute((const void*)k_scan_fixed_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            (void)hipGetLastError();
            return FABHIP_ELAUNCH;
        }
        hipLaunchKernelGGL(k_scan_fixed_lds, grid, block, lds, st, log_w, n, ws.max_val, ws);
    } else {
        hipLaunchKernelGGL(k_scan_fixed, grid, block, 0, st, log_w, n, ws.max_val, ws);
    }
    return check_launch();
}

int tail_small(const TailArgs& a, int* dest, hipStream_t st) {
    if (a.B > (long)TAIL_MAX_BLOCKS * ESS_THREADS * 4 || a.B < 1) return FABHIP_ENOTSUP;
    const int nb = grid_for(a.B, ESS_THREADS * 4, ESS_MAX_BLOCKS);          // fabhip_ess_logz's grid for the same row capacity
    const size_t bytes = (size_t)TAIL_CHUNK * (3 * a.D + 4) * 4;
    if (bytes > 48 * 1024) return FABHIP_ENOTSUP;
    hipLaunchKernelGGL(k_tail_small, dim3(1), dim3(TAIL_THREADS), bytes, st, a, dest, nb);
    return check_launch();
}

}  // namespace fab

using namespace fab;

extern "C" {

const char* fabhip_strerror(int code) {
    switch (code) {
        case FABHIP_OK: return "ok";
        case FABHIP_EINVAL: return "invalid argument (shape, null pointer or alignment)";
        case FABHIP_ENOTSUP: return "not supported (dimension beyond compiled limits)";
        case FABHIP_ELAUNCH: return "kernel launch failed (hipGetLastError)";
        case FABHIP_ENOSPC: return "workspace too small";
        default: return "unknown fabhip error";
    }
}

int fabhip_version(void) { return FABHIP_ABI_VERSION; }

void fabhip_abi_sizes(int64_t out8[8]) {
    if (!out8) return;
    out8[0] = (int64_t)sizeof(fabhip_flow_params); out8[1] = (int64_t)sizeof(fabhip_flow);
    out8[2] = (int64_t)sizeof(fabhip_target);      out8[3] = (int64_t)sizeof(fabhip_point);
    out8[4] = (int64_t)sizeof(fabhip_anneal);      out8[5] = (int64_t)sizeof(fabhip_hmc_args);
    out8[6] = (int64_t)sizeof(fabhip_metropolis_args); out8[7] = (int64_t)sizeof(fabhip_ais_args);
}

int fabhip_target_log_prob(const fabhip_target* target, const float* x, float* log_p, float* grad_x, int64_t B,
                           fabhip_stream_t stream) {
    if (!target || !x || !log_p || B < 0) return FABHIP_EINVAL;
    if (target->dim < 1 || target->dim > FABHIP_MAX_DIM) return FABHIP_ENOTSUP;
    FAB_TRY(check_target(target, target->dim));
    if (B == 0) return FABHIP_OK;
    const TargetDev tg = make_target_dev(*target);
    const size_t bytes = (size_t)2 * ROWS * tg.dim * 4;
    const dim3 grid((unsigned)((B + ROWS - 1) / ROWS)), block(NTHREADS);
    if (grad_x) hipLaunchKernelGGL((k_target<true>), grid, block, bytes, (hipStream_t)stream, tg, x, log_p, grad_x, (long)B);
    else hipLaunchKernelGGL((k_target<false>), grid, block, bytes, (hipStream_t)stream, tg, x, log_p, grad_x, (long)B);
    return check_launch();
}

size_t fabhip_ess_workspace_bytes(int64_t n) { (void)n; return al256(sizeof(Msum) * ESS_MAX_BLOCKS) + 256; }

int fabhip_ess_logz(const float* log_w, int64_t n, const int32_t* n_ptr, double n_norm, float* out, void* workspace,
                    size_t workspace_bytes, fabhip_stream_t stream) {
    if (!log_w || !out || !workspace || n < 0) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_ess_workspace_bytes(n)) return FABHIP_ENOSPC;
    Msum* part = (Msum*)workspace;
    const int nblk = grid_for(n, ESS_THREADS * 4, ESS_MAX_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ess_partial, dim3(nblk), dim3(ESS_THREADS), 0, st, log_w, (long)n, n_ptr, part);
    hipLaunchKernelGGL(k_ess_final, dim3(1), dim3(ESS_THREADS), 0, st, part, nblk, (long)n, n_ptr, n_norm, out);
    return check_launch();
}

size_t fabhip_multinomial_torch_workspace_bytes(int64_t n) { return al256((size_t)(n + 1) * 4) + 256; }

int fabhip_multinomial_torch(const float* probs, int64_t n, const double* u, int64_t n_samples, int64_t* idx,
                             void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!probs || !u || !idx || !workspace || n < 1 || n_samples < 0) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_multinomial_torch_workspace_bytes(n)) return FABHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    float* c = (float*)workspace;
    hipLaunchKernelGGL(k_seq_cdf, dim3(1), dim3(128), 0, st, probs, (long)n, c);
    hipLaunchKernelGGL(k_cdf_normalise, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, c, (long)n);
    if (n_samples > 0)
        hipLaunchKernelGGL(k_search_f32cdf, dim3(grid_for(n_samples, 256, 4096)), dim3(256), 0, st, c, (long)n, u,
                           (long)n_samples, (long long*)idx);
    return check_launch();
}

size_t fabhip_resample_workspace_bytes(int64_t n) {
    const size_t tiles = (size_t)scan_tiles(n);
    return al256(1024 * 4) + 256 + al256(tiles * 8) + 256 + al256(tiles * 8) + al256((size_t)n * 8) +
           al256(((size_t)n / 16 + 1) * 8) + al256(((size_t)n / 256 + 1) * 8) + 256;
}

int fabhip_fixed_cdf(const float* log_w, int64_t n, int32_t reuse_max, const uint64_t** cdf_out, void* workspace,
                     size_t workspace_bytes, fabhip_stream_t stream) {
    if (!log_w || !workspace || n < 1) return FABHIP_EINVAL;
    if (((size_t)workspace & 255) != 0) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_resample_workspace_bytes(n)) return FABHIP_ENOSPC;
    const ScanWs ws = carve_scan_ws(workspace, n);
    if (cdf_out) *cdf_out = (const uint64_t*)ws.cdf;
    return build_fixed_cdf(log_w, n, ws, (hipStream_t)stream, reuse_max != 0);
}

int fabhip_resample_multinomial(const float* log_w, int64_t n, const double* u, int64_t n_samples, int64_t* idx,
                                void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!log_w || !u || !idx || !workspace || n < 1 || n_samples < 0) return FABHIP_EINVAL;
    if (((size_t)workspace & 255) != 0) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_resample_workspace_bytes(n)) return FABHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    ScanWs ws = carve_scan_ws(workspace, n);
    ws.want_sub = 1;
    FAB_TRY(build_fixed_cdf(log_w, n, ws, st));
    if (n_samples > 0)
        hipLaunchKernelGGL(k_sample_multinomial, dim3(grid_for(n_samples, 256, 4096)), dim3(256), 0, st, ws, (long)n,
                           scan_tiles(n), u, (long)n_samples, (long long*)idx);
    return check_launch();
}

int fabhip_resample_systematic(const float* log_w, int64_t n, double u0, int64_t n_samples, int64_t* idx,
                               void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!log_w || !idx || !workspace || n < 1 || n_samples < 0 || !(u0 >= 0.0 && u0 < 1.0)) return FABHIP_EINVAL;
    if (((size_t)workspace & 255) != 0) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_resample_workspace_bytes(n)) return FABHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    const ScanWs ws = carve_scan_ws(workspace, n);
    if (n_samples == 0) return FABHIP_OK;
    const long nwt = ((long)n + EM_WAVE_ITEMS - 1) / EM_WAVE_ITEMS;
    const long nbk = ((long)n + EM_BLOCK - 1) / EM_BLOCK;
    // FABHIP_OPT_SYSTEMATIC_VARIANT 0 = scan + CDF in HBM + search (A/B reference)
    if (option(FABHIP_OPT_SYSTEMATIC_VARIANT) == 0 || n_samples >= (1ll << 31) - 2) {
        FAB_TRY(build_fixed_cdf(log_w, n, ws, st));
        unsigned long long* strata = ws.desc;                    // the descriptors are dead once the scan is done
        hipLaunchKernelGGL(k_strata_params, dim3(1), dim3(1), 0, st, ws.tile_inc + (scan_tiles(n) - 1), u0,
                           (long)n_samples, strata);
        hipLaunchKernelGGL(k_sample_systematic, dim3((unsigned)((n_samples + SYS_CHUNK - 1) / SYS_CHUNK)), dim3(256), 0,
                           st, ws, (long)n, scan_tiles(n), strata, (long)n_samples, (long long*)idx);
        return check_launch();
    }
    // fused path: max -> wave-tile sums -> block prefix + stratum map -> emit (no CDF in HBM); scratch lives in the
    // (8 n)-byte `cdf` region of the workspace
    const int mb = grid_for(n, 256 * 16, 1024);
    hipLaunchKernelGGL(k_max_partial, dim3(mb), dim3(256), 0, st, log_w, (long)n, ws.max_part);
    hipLaunchKernelGGL(k_max_final, dim3(1), dim3(256), 0, st, ws.max_part, mb, ws.max_val);
    unsigned long long* wave_sum = ws.cdf;
    unsigned long long* block_sum = ws.cdf + nwt;
    unsigned long long* block_excl = block_sum + nbk;
    unsigned long long* strata = block_excl + nbk;               // {total, S, U, F}
    unsigned long long* giant_count = strata + 4;
    long long* giant_desc = (long long*)(giant_count + 1);
    // capacity: what is left of the CDF region; at most n_samples / EM_GIANT runs can be giant
    long giant_cap = ((long)n - (nwt + 2 * nbk + 5)) / 3;
    if (giant_cap < 0) giant_cap = 0;
    if (giant_cap > n_samples / EM_GIANT + 1) giant_cap = n_samples / EM_GIANT + 1;
    const dim3 grid((unsigned)nbk), block(64 * EM_NW);
    hipLaunchKernelGGL(k_emit_wave_sums, grid, block, 0, st, log_w, (long)n, ws.max_val, wave_sum, block_sum);
    hipLaunchKernelGGL(k_emit_prefix, dim3(1), dim3(1024), 0, st, block_sum, nbk, block_excl, u0, (long)n_samples, strata,
                       giant_count);
    hipLaunchKernelGGL(k_emit_systematic, grid, block, 0, st, log_w, (long)n, ws.max_val, wave_sum, block_excl, strata,
                       (long)n_samples, (long long*)idx, giant_count, giant_desc, giant_cap);
    if (giant_cap > 0 && n_samples >= EM_GIANT)
        hipLaunchKernelGGL(k_emit_fill_runs, dim3(256), dim3(256), 0, st, giant_count, giant_desc, giant_cap,
                           (long long*)idx);
    return check_launch();
}

int fabhip_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t n_out, int64_t row_len,
                       fabhip_stream_t stream) {
    if (!src || !idx || !dst || n_out < 0 || row_len < 1) return FABHIP_EINVAL;
    if (n_out == 0) return FABHIP_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n_out * row_len, 256 * 4, 8192)), dim3(256), 0, (hipStream_t)stream,
                       src, (const long long*)idx, dst, (long)n_out, (long)row_len);
    return check_launch();
}

}  // extern "C"
