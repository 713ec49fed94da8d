// Top-k selection over fp32 keys — the device half of the prioritised buffer's sampling without replacement
// (fab/utils/prioritised_replay_buffer.py:10-17: `torch.topk(gumbel + logits, n)` over <= 512 000 stored
// log-weights, n = minibatches x batch = 16 384).  SURVEY.md section 8f, rank 1.
//
//   1. three histogram passes of an MSB-first radix select (11 + 11 + 10 bits of the order-preserving unsigned
//      image of the key) find the k-th largest key T, how many keys are larger, and how many of the keys equal to
//      T are needed (`need_eq`, taken in ascending index order);
//   2. an index-ordered compaction (per-block counts -> one-block scan -> scatter) writes the k selected
//      (key, index) pairs as 64-bit composites  ord(key) << 32 | ~index  (unique, so the order is total);
//   3. unsorted mode (what the buffer needs: it permutes the selection randomly anyway) stops here, the indices
//      come out in ascending index order; sorted mode adds one workgroup that sorts the k <= 16384 composites in
//      LDS (bitonic, 128 KB): descending key order, ties by ascending index - torch.topk(sorted=True) up to its
//      tie rule.
// Integer histograms and ranks only: the result is exact and independent of scheduling.
#include "launch.h"

namespace fab {

constexpr int TK_BLOCK = 256, TK_ITEMS = 16, TK_TILE = TK_BLOCK * TK_ITEMS;      // 4096 keys per workgroup
constexpr int TK_BINS = 2048;
constexpr int TK_MAXK = 16384;

struct TopkState {            // device-resident scalars
    unsigned prefix;          // decided high bits (right-aligned)
    unsigned bits_done;       // how many high bits are decided
    long long k_rem;          // how many keys still to take among those matching the prefix
    long long count_gt;       // keys strictly above the final threshold
};

struct TopkWs {
    TopkState* st;
    unsigned* hist;           // [3][TK_BINS]
    long long* blk;           // [nblk][2] (gt, eq) counts, then exclusive prefixes in place
    unsigned long long* sel;  // [TK_MAXK] selected composites
};

__device__ __forceinline__ unsigned ord32(float f) {             // larger float <-> larger unsigned (NaN above +inf)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void k_topk_init(TopkState* st, long long k) {
    st->prefix = 0u; st->bits_done = 0u; st->k_rem = k; st->count_gt = 0;
}

// histogram of the next `nbits` bits among the keys whose decided high bits equal the prefix
__global__ __launch_bounds__(TK_BLOCK) void k_topk_hist(const float* __restrict__ keys, long n, const TopkState* st,
                                                        int nbits, unsigned* __restrict__ hist) {
    __shared__ unsigned h[TK_BINS];
    for (int i = threadIdx.x; i < TK_BINS; i += TK_BLOCK) h[i] = 0u;
    __syncthreads();
    const unsigned done = st->bits_done, prefix = st->prefix;
    const int shift = 32 - (int)done - nbits;
    const long base = (long)blockIdx.x * TK_TILE;
#pragma unroll 4
    for (int j = 0; j < TK_ITEMS; ++j) {
        const long i = base + (long)j * TK_BLOCK + threadIdx.x;
        if (i < n) {
            const unsigned u = ord32(keys[i]);
            const bool match = done == 0u || (u >> (32 - done)) == prefix;
            if (match) atomicAdd(&h[(u >> shift) & ((1u << nbits) - 1u)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << nbits); i += TK_BLOCK)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// one workgroup: the digit that holds the k_rem-th largest matching key
__global__ __launch_bounds__(TK_BLOCK) void k_topk_pick(TopkState* st, const unsigned* __restrict__ hist, int nbits) {
    __shared__ long long part[TK_BLOCK];
    __shared__ int digit_sh;
    __shared__ long long above_sh;
    const int nb = 1 << nbits, per = nb / TK_BLOCK;             // bins per thread (8 or 4), thread 0 = highest bins
    const int tid = threadIdx.x;
    const int hi = nb - 1 - tid * per;                          // this thread's highest bin
    long long s = 0;
    for (int j = 0; j < per; ++j) s += hist[hi - j];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        const long long k = st->k_rem;
        long long above = 0;
        int t = 0;
        while (t < TK_BLOCK - 1 && above + part[t] < k) { above += part[t]; ++t; }
        const int h0 = nb - 1 - t * per;
        int d = h0;
        for (int j = 0; j < per; ++j) {
            d = h0 - j;
            if (above + hist[d] >= k || j == per - 1) break;
            above += hist[d];
        }
        digit_sh = d;
        above_sh = above;
    }
    __syncthreads();
    if (tid == 0) {
        st->prefix = (st->prefix << nbits) | (unsigned)digit_sh;
        st->bits_done += (unsigned)nbits;
        st->k_rem -= above_sh;
        st->count_gt += above_sh;
    }
}

__device__ __forceinline__ void block_excl_scan2(long long& a, long long& b, long long* sh) {   // exclusive, 256 threads
    const int tid = threadIdx.x;
    sh[tid] = a; sh[TK_BLOCK + tid] = b;
    __syncthreads();
    for (int off = 1; off < TK_BLOCK; off <<= 1) {
        const long long xa = tid >= off ? sh[tid - off] : 0, xb = tid >= off ? sh[TK_BLOCK + tid - off] : 0;
        __syncthreads();
        sh[tid] += xa; sh[TK_BLOCK + tid] += xb;
        __syncthreads();
    }
    a = sh[tid] - a; b = sh[TK_BLOCK + tid] - b;
    __syncthreads();
}

// per workgroup: how many keys above / equal to the threshold (thread t owns TK_ITEMS consecutive keys)
__global__ __launch_bounds__(TK_BLOCK) void k_topk_count(const float* __restrict__ keys, long n, const TopkState* st,
                                                         long long* __restrict__ blk) {
    __shared__ long long sh[2 * TK_BLOCK];
    const unsigned T = st->prefix;
    const long base = (long)blockIdx.x * TK_TILE + (long)threadIdx.x * TK_ITEMS;
    long long gt = 0, eq = 0;
    for (int j = 0; j < TK_ITEMS; ++j) {
        const long i = base + j;
        if (i < n) { const unsigned u = ord32(keys[i]); gt += u > T; eq += u == T; }
    }
    sh[threadIdx.x] = gt; sh[TK_BLOCK + threadIdx.x] = eq;
    __syncthreads();
    for (int s = TK_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sh[threadIdx.x] += sh[threadIdx.x + s]; sh[TK_BLOCK + threadIdx.x] += sh[TK_BLOCK + threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { blk[2 * blockIdx.x] = sh[0]; blk[2 * blockIdx.x + 1] = sh[TK_BLOCK]; }
}

// one workgroup: exclusive prefix over the per-block counts, in place
__global__ __launch_bounds__(TK_BLOCK) void k_topk_scan_blocks(long long* __restrict__ blk, long nblk) {
    __shared__ long long sh[2 * TK_BLOCK];
    __shared__ long long carry[2];
    if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; }
    __syncthreads();
    for (long b0 = 0; b0 < nblk; b0 += TK_BLOCK) {
        const long b = b0 + threadIdx.x;
        long long a = b < nblk ? blk[2 * b] : 0, e = b < nblk ? blk[2 * b + 1] : 0;
        const long long a0 = a, e0 = e;
        block_excl_scan2(a, e, sh);
        if (b < nblk) { blk[2 * b] = carry[0] + a; blk[2 * b + 1] = carry[1] + e; }
        __syncthreads();
        if (threadIdx.x == TK_BLOCK - 1) { carry[0] += a + a0; carry[1] += e + e0; }
        __syncthreads();
    }
}

// scatter the selected composites: slot = rank among the larger keys, or count_gt + rank among the equal keys
__global__ __launch_bounds__(TK_BLOCK) void k_topk_scatter(const float* __restrict__ keys, long n, const TopkState* st,
                                                           const long long* __restrict__ blk,
                                                           unsigned long long* __restrict__ sel,
                                                           long long* __restrict__ idx_direct) {
    __shared__ long long sh[2 * TK_BLOCK];
    const unsigned T = st->prefix;
    const long long need_eq = st->k_rem, count_gt = st->count_gt;
    const long base = (long)blockIdx.x * TK_TILE + (long)threadIdx.x * TK_ITEMS;
    long long gt = 0, eq = 0;
    for (int j = 0; j < TK_ITEMS; ++j) {
        const long i = base + j;
        if (i < n) { const unsigned u = ord32(keys[i]); gt += u > T; eq += u == T; }
    }
    block_excl_scan2(gt, eq, sh);
    gt += blk[2 * blockIdx.x];
    eq += blk[2 * blockIdx.x + 1];
    for (int j = 0; j < TK_ITEMS; ++j) {
        const long i = base + j;
        if (i >= n) break;
        const unsigned u = ord32(keys[i]);
        const unsigned long long comp = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        if (idx_direct) {                       // unsorted mode: ascending index = slot order of (gt + taken eq) merged
            if (u > T || (u == T && eq < need_eq)) idx_direct[gt + (eq < need_eq ? eq : need_eq)] = (long long)i;
            gt += u > T; eq += u == T;
        } else {
            if (u > T) sel[gt++] = comp;
            else if (u == T) { if (eq < need_eq) sel[count_gt + eq] = comp; ++eq; }
        }
    }
}

// one workgroup: bitonic sort (descending) of k <= 16384 composites in LDS, then the indices
__global__ __launch_bounds__(1024) void k_topk_sort(const unsigned long long* __restrict__ sel, long k,
                                                    long long* __restrict__ idx_out, float* __restrict__ key_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s[];
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += 1024) s[i] = i < k ? sel[i] : 0ull;      // 0 sorts last
    __syncthreads();
    for (int len = 2; len <= np2; len <<= 1) {
        for (int stride = len >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < np2 / 2; i += 1024) {
                const int lo = 2 * i - (i & (stride - 1));              // element with the stride bit clear
                const int hi = lo + stride;
                const bool desc = (lo & len) == 0;
                const unsigned long long a = s[lo], b = s[hi];
                if (desc ? a < b : a > b) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += 1024) {
        const unsigned long long c = s[i];
        idx_out[i] = (long long)(0xFFFFFFFFu - (unsigned)(c & 0xFFFFFFFFull));
        if (key_out) {
            const unsigned u = (unsigned)(c >> 32);
            key_out[i] = __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
        }
    }
}

// Gumbel-max keys of the buffer's sampling without replacement (fab/utils/prioritised_replay_buffer.py:10-13):
// key = logit - log(-log(u)), u ~ U(0, 1) clamped away from 0 (torch.finfo(float32).tiny) as the host expression does
__global__ __launch_bounds__(256) void k_gumbel_keys(const float* __restrict__ logits, const float* __restrict__ u, long n,
                                                     float* __restrict__ keys) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float uu = fmaxf(u[i], 1.17549435e-38f);
    keys[i] = -logf(-logf(uu)) + logits[i];
}

// A pseudo-random order of the k selected rows - what `indices[torch.randperm(n)]` of :14-16 draws - without a sort: a keyed
// bijection of [0, k), four Feistel rounds on the next power of two with cycle walking back into range (format-preserving
// permutation), the round keys from four uniform draws of torch's generator.  O(1) per row, any k; sorting 16 384 random keys in
// one workgroup's LDS (tried first) took 178 us.
__device__ __forceinline__ unsigned tk_mix(unsigned x) {               // (murmur3 finaliser)
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void k_random_order(const long long* __restrict__ idx_in, const float* __restrict__ u4, long k,
                                                      long long* __restrict__ idx_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= k) return;
    int bits = 2;
    while ((1l << bits) < k) ++bits;
    const int lb = bits >> 1, rb = bits - lb;                            // left / right half widths
    const unsigned lm = (1u << lb) - 1u, rm = (1u << rb) - 1u;
    unsigned key[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) key[j] = tk_mix(__float_as_uint(u4[j]) * 0x9E3779B9u + (unsigned)j);
    unsigned long v = (unsigned long)i;
    do {                                                                 // cycle walking: expected < 2 trips (2^bits < 2 k... 4 k)
        unsigned l = (unsigned)(v >> rb) & lm, r = (unsigned)v & rm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j & 1) r ^= tk_mix(l ^ key[j]) & rm;                    // alternate the halves: each round is an involution
            else l ^= tk_mix(r ^ key[j]) & lm;
        }
        v = ((unsigned long)l << rb) | r;
    } while ((long)v >= k);
    idx_out[i] = idx_in[v];
}

// PrioritisedReplayBuffer.add (fab/utils/prioritised_replay_buffer.py:71-85): rows (start + i) mod max_length of the ring
__global__ __launch_bounds__(256) void k_buffer_add(const float* __restrict__ x, const float* __restrict__ log_w,
                                                    const float* __restrict__ log_q, long n, int dim, long start, long max_length,
                                                    float* __restrict__ bx, float* __restrict__ blw, float* __restrict__ blq) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * dim) return;
    const long i = e / dim;
    const int j = (int)(e - i * dim);
    const long row = (start + i) % max_length;
    bx[row * dim + j] = x[e];
    if (j == 0) { blw[row] = log_w[i]; blq[row] = log_q[i]; }
}

static inline size_t tk_al(size_t x) { return (x + 255) & ~(size_t)255; }
static inline long tk_blocks(long n) { return (n + TK_TILE - 1) / TK_TILE; }

}  // namespace fab

using namespace fab;

extern "C" {

size_t fabhip_topk_workspace_bytes(int64_t n, int64_t k) {
    if (n < 1 || k < 1) return 0;
    return 256 + tk_al(3 * TK_BINS * 4) + tk_al((size_t)tk_blocks((long)n) * 16) + tk_al((size_t)TK_MAXK * 8);
}

int fabhip_topk(const float* keys, int64_t n, int64_t k, int32_t sorted, int64_t* idx_out, float* key_out,
                void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!keys || !idx_out || !workspace || n < 1 || k < 1 || k > n) return FABHIP_EINVAL;
    if (!sorted && key_out) return FABHIP_EINVAL;
    if ((sorted && k > TK_MAXK) || n > 0xFFFFFFFFll) return FABHIP_ENOTSUP;
    if (workspace_bytes < fabhip_topk_workspace_bytes(n, k)) return FABHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    TopkWs ws;
    ws.st = (TopkState*)p; p += 256;
    ws.hist = (unsigned*)p; p += tk_al(3 * TK_BINS * 4);
    ws.blk = (long long*)p; p += tk_al((size_t)tk_blocks((long)n) * 16);
    ws.sel = (unsigned long long*)p;
    const long nblk = tk_blocks((long)n);
    if (hipMemsetAsync(ws.hist, 0, 3 * TK_BINS * 4, st) != hipSuccess) return FABHIP_ELAUNCH;
    hipLaunchKernelGGL(k_topk_init, dim3(1), dim3(1), 0, st, ws.st, (long long)k);
    const int bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(k_topk_hist, dim3((unsigned)nblk), dim3(TK_BLOCK), 0, st, keys, (long)n, ws.st, bits[pass],
                           ws.hist + pass * TK_BINS);
        hipLaunchKernelGGL(k_topk_pick, dim3(1), dim3(TK_BLOCK), 0, st, ws.st, ws.hist + pass * TK_BINS, bits[pass]);
    }
    hipLaunchKernelGGL(k_topk_count, dim3((unsigned)nblk), dim3(TK_BLOCK), 0, st, keys, (long)n, ws.st, ws.blk);
    hipLaunchKernelGGL(k_topk_scan_blocks, dim3(1), dim3(TK_BLOCK), 0, st, ws.blk, nblk);
    hipLaunchKernelGGL(k_topk_scatter, dim3((unsigned)nblk), dim3(TK_BLOCK), 0, st, keys, (long)n, ws.st, ws.blk, ws.sel,
                       sorted ? (long long*)nullptr : (long long*)idx_out);
    if (!sorted) return check_launch();
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    const size_t lds = (size_t)np2 * 8;
    FAB_TRY(set_max_lds((const void*)k_topk_sort, lds));
    hipLaunchKernelGGL(k_topk_sort, dim3(1), dim3(1024), lds, st, ws.sel, (long)k, (long long*)idx_out, key_out);
    return check_launch();
}

int fabhip_buffer_add(const float* x, const float* log_w, const float* log_q_old, int64_t n, int32_t dim, int64_t start,
                      int64_t max_length, float* buf_x, float* buf_log_w, float* buf_log_q_old, fabhip_stream_t stream) {
    if (!x || !log_w || !log_q_old || !buf_x || !buf_log_w || !buf_log_q_old || n < 0 || dim < 1 || max_length < 1 || start < 0 ||
        n > max_length)
        return FABHIP_EINVAL;
    if (n == 0) return FABHIP_OK;
    const long tot = (long)n * dim;
    hipLaunchKernelGGL(k_buffer_add, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, log_w, log_q_old,
                       (long)n, (int)dim, (long)start, (long)max_length, buf_x, buf_log_w, buf_log_q_old);
    return check_launch();
}

size_t fabhip_buffer_sample_workspace_bytes(int64_t n, int64_t k) {
    const size_t t = fabhip_topk_workspace_bytes(n, k);
    return t ? t + tk_al((size_t)n * 4) + tk_al((size_t)k * 8) : 0;      // + the keys, the selection in index order
}

int fabhip_buffer_sample(const float* log_w, const float* u_gumbel, const float* u_order, int64_t n, int64_t k, int64_t* idx_out,
                         void* workspace, size_t workspace_bytes, fabhip_stream_t stream) {
    if (!log_w || !u_gumbel || !u_order || !idx_out || !workspace || n < 1 || k < 1 || k > n) return FABHIP_EINVAL;
    if (workspace_bytes < fabhip_buffer_sample_workspace_bytes(n, k)) return FABHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    float* keys = (float*)p; p += tk_al((size_t)n * 4);
    long long* sel = (long long*)p; p += tk_al((size_t)k * 8);
    hipLaunchKernelGGL(k_gumbel_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, log_w, u_gumbel, (long)n, keys);
    FAB_TRY(fabhip_topk(keys, n, k, 0, (int64_t*)sel, nullptr, p, workspace_bytes - (size_t)(p - (char*)workspace), stream));
    hipLaunchKernelGGL(k_random_order, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, (const long long*)sel, u_order, (long)k,
                       (long long*)idx_out);
    return check_launch();
}

}  // extern "C"
