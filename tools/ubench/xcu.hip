// Round 4, DESIGN section 10: what would it cost to split ONE tile of chains over several CUs of the same XCD?
//
// The 4-chain headline kernel is bound by the L2 -> CU weight stream: every CU pulls the whole 47.9 MB image per transition
// for its 4 chains.  The only lever that changes that bound is reuse: G CUs that share a tile of 4 G chains, each multiplying
// 1 / G of the output columns, stream 1 / G of the weights each - at the price of an all-to-all of the activations between
// the G CUs at every GEMM stage (~10 stages per layer pair, ~2500 per transition).  CDNA4 has no workgroup clusters: the
// exchange goes through the L2 of the XCD the G workgroups share (workgroup i runs on XCD i mod 8), data written with plain
// stores (the L2 is the coherence point inside an XCD), flags polled with L2 read-modify-writes or carried as tags in the data,
// data read with sc1 loads (measured here: sc0 loads and buffer_inv sc0 + plain loads are served by the CU's own vector L1).
//
// This benchmark prices exactly that exchange: 256 workgroups (one per CU: the LDS request keeps a second one out), groups
// of G workgroups {xcd + 8 (G g + k)}, per stage every member stores a slice (BYTES per member), waits for its stores,
// raises its flag, polls the flags of its group, reads the G slices back.  Optional WORK cycles of s_sleep-free ALU filler
// between stages stand for the member's share of the stage's MFMAs.  Prints cycles per stage (s_memtime) and the wall time.
// Every poll loop is bounded: a group that cannot see its peers (not co-resident / not on one XCD) reports it instead of
// hanging the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NT = 256;                       // threads per workgroup
constexpr int MAXG = 8;
constexpr unsigned POLL_LIMIT = 1u << 17;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// POLL 0: buffer_inv sc0 (drop the CU's vector-L1 lines) + plain load per poll; POLL 1: an atomic OR of 0 that returns the word
// (read-modify-writes execute in the L2).  (A load with sc0 alone is NOT enough: outside threadgroup-split mode the L1 may serve
// "workgroup scope" - the first version of this benchmark polled its own L1 copy until the limit.)
__device__ __forceinline__ void inv_l1() { asm volatile("buffer_inv sc0" ::: "memory"); }
template <int POLL>
__device__ __forceinline__ unsigned poll_word(unsigned* p) {
    unsigned v = 0;
    if constexpr (POLL == 2) return v;
    if constexpr (POLL == 0) {
        asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else {
        const unsigned zero = 0u;
        asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
    }
    return v;
}
// DATA 0: one buffer_inv sc0 after the flags, then plain loads; DATA 1: loads with sc1 (device scope), no invalidate;
// DATA 2: plain loads without any invalidate (expected to return stale lines from the second stage pair on: the control)
template <int DATA>
__device__ __forceinline__ void issue_ld4(f32x4& v, const f32x4* p) {
    if constexpr (DATA == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct Out {
    long long cycles;
    unsigned xcc, fail, pad0, pad1;
};

// F4 = float4 per member and stage (F4 / NT per thread), G = members per group
template <int G, int F4, int POLL, int DATA>
__global__ __launch_bounds__(NT) void k_xcu(f32x4* __restrict__ data, unsigned* __restrict__ flags, int n_stages, int work,
                                            float* __restrict__ sink, Out* __restrict__ out) {
    extern __shared__ float lds[];
    __shared__ int abort_flag;
    constexpr int PER = F4 / NT;
    const int tid = threadIdx.x, wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3, group = slot / G, member = slot % G;
    const int gid = xcd * (32 / G) + group;                                  // group id, 0 .. 256 / G - 1
    f32x4* gdata = data + (size_t)gid * 2 * G * F4;                         // [parity][member][F4]
    unsigned* gflags = flags + (size_t)gid * MAXG * 32;                      // one 128-byte line per member
    if (tid == 0) abort_flag = 0;
    lds[tid] = (float)tid;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float filler = (float)tid * 1e-3f;
    unsigned fail = 0, stale = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 1; it <= n_stages; ++it) {
        f32x4* mine = gdata + ((size_t)(it & 1) * G + member) * F4;
#pragma unroll
        for (int p = 0; p < PER; ++p) mine[p * NT + tid] = (f32x4){(float)it, filler, acc.x, (float)member};
        f32x4 v[G * PER];
        if constexpr (POLL == 2) {
            // tagged data, no flags: every float4 carries its stage in .x; a thread re-reads its own G x PER elements (sc1)
            // until all of them carry this stage - no store wait, no flag, no workgroup barrier inside the exchange
            unsigned n = 0;
            bool ok;
            do {
#pragma unroll
                for (int k = 0; k < G; ++k)
#pragma unroll
                    for (int p = 0; p < PER; ++p)
                        issue_ld4<1>(v[k * PER + p], gdata + ((size_t)(it & 1) * G + k) * F4 + p * NT + tid);
                wait_all();
                ok = true;
#pragma unroll
                for (int q = 0; q < G * PER; ++q) { asm volatile("" : "+v"(v[q])); ok = ok && v[q].x == (float)it; }
            } while (!ok && ++n <= POLL_LIMIT);
            if (!ok) abort_flag = 1;
            __syncthreads();                                                 // (the stage's own barrier in a real kernel)
            if (abort_flag) { fail = (unsigned)it; break; }
        } else {
        wait_all();                                                          // this thread's stores are in the L2
        __syncthreads();
        if (tid == 0) gflags[member * 32] = (unsigned)it;
        if (tid < G) {                                                       // lane k watches member k
            unsigned n = 0;
            while (poll_word<POLL>(gflags + tid * 32) < (unsigned)it) {
                if (++n > POLL_LIMIT) { abort_flag = 1; break; }
            }
        }
        __syncthreads();
        if (abort_flag) { fail = (unsigned)it; break; }
        if constexpr (DATA == 0) inv_l1();                                   // the slices were last read one stage pair ago
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
            for (int p = 0; p < PER; ++p) issue_ld4<DATA>(v[k * PER + p], gdata + ((size_t)(it & 1) * G + k) * F4 + p * NT + tid);
        wait_all();
        }
#pragma unroll
        for (int q = 0; q < G * PER; ++q) {
            asm volatile("" : "+v"(v[q]));
            acc += v[q];
            stale += v[q].x != (float)it;                                    // every slice carries the stage it was written in
        }
        for (int w = 0; w < work; ++w) filler = filler * 1.0000001f + 1e-7f;          // the member's own arithmetic
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        out[wg].cycles = t1 - t0;
        out[wg].xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));               // HW_REG_XCC_ID[3:0]
        out[wg].fail = fail;
    }
    if (stale) atomicAdd(&out[wg].pad0, stale);
    sink[(size_t)wg * NT + tid] = acc.x + acc.y + acc.z + acc.w + filler;
}

template <int G, int F4, int POLL = 1, int DATA = 0>
void run(const char* name, f32x4* data, unsigned* flags, float* sink, Out* out, int n_stages, int work) {
    auto kern = k_xcu<G, F4, POLL, DATA>;
    const size_t lds = 96 * 1024;                                            // one workgroup per CU
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(flags, 0, 256 * MAXG * 32 * 4);
        (void)hipMemset(out, 0, 256 * sizeof(Out));
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(NT), lds, 0, data, flags, n_stages, work, sink, out);
        (void)hipEventRecord(e1, 0);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%s failed: %s\n", name, hipGetErrorString(e)); return; }
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<Out> h(256);
    (void)hipMemcpy(h.data(), out, 256 * sizeof(Out), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    int failed = 0, split = 0;
    unsigned long long stale = 0;
    for (int wg = 0; wg < 256; ++wg) {
        mean += (double)h[wg].cycles;
        mx = h[wg].cycles > mx ? (double)h[wg].cycles : mx;
        failed += h[wg].fail != 0;
        stale += h[wg].pad0;
        const int first = (wg & 7) + 8 * ((wg >> 3) / G * G);                // first member of this workgroup's group
        split += h[wg].xcc != h[first].xcc;
    }
    mean /= 256;
    printf("%-16s poll=%d data=%d G=%d %5d B/member work=%4d: %7.0f cycles per stage (max WG %7.0f), %6.2f us per stage by events; "
           "%llu stale float4 read, %d WGs timed out, %d WGs not on their group's XCD (xcc of WG 0..7: %u %u %u %u %u %u %u %u)\n",
           name, POLL, DATA, G, F4 * 16, work, mean / n_stages, mx / n_stages, 1e3 * ms / n_stages, stale, failed, split, h[0].xcc, h[1].xcc,
           h[2].xcc, h[3].xcc, h[4].xcc, h[5].xcc, h[6].xcc, h[7].xcc);
    fflush(stdout);
}

int main() {
    f32x4* data; unsigned* flags; float* sink; Out* out;
    (void)hipMalloc((void**)&data, (size_t)256 * 2 * MAXG * 1024 * 16);
    (void)hipMalloc((void**)&flags, 256 * MAXG * 32 * 4);
    (void)hipMalloc((void**)&sink, 256 * NT * 4);
    (void)hipMalloc((void**)&out, 256 * sizeof(Out));
    (void)hipMemset(data, 0, (size_t)256 * 2 * MAXG * 1024 * 16);
    const int n = 2000;
    // the poll that does not work, once (kept as the record of what buffer_inv sc0 + a plain load sees: its own L1 line)
    run<4, 256, 0, 0>("exchange only", data, flags, sink, out, 20, 0);
    // exchange alone (the latency of one all-to-all through the XCD's L2): atomic poll, three ways to read the slices
    run<2, 256, 1, 0>("exchange only", data, flags, sink, out, n, 0);
    run<4, 256, 1, 0>("exchange only", data, flags, sink, out, n, 0);
    run<8, 256, 1, 0>("exchange only", data, flags, sink, out, n, 0);
    run<4, 256, 1, 1>("exchange only", data, flags, sink, out, n, 0);
    run<4, 256, 1, 2>("exchange only", data, flags, sink, out, n, 0);
    run<4, 512, 1, 0>("exchange only", data, flags, sink, out, n, 0);
    run<4, 1024, 1, 0>("exchange only", data, flags, sink, out, n, 0);
    run<4, 1024, 1, 1>("exchange only", data, flags, sink, out, n, 0);
    // tagged data instead of flags (poll=2)
    run<2, 256, 2, 1>("tagged data", data, flags, sink, out, n, 0);
    run<4, 256, 2, 1>("tagged data", data, flags, sink, out, n, 0);
    run<8, 256, 2, 1>("tagged data", data, flags, sink, out, n, 0);
    run<4, 512, 2, 1>("tagged data", data, flags, sink, out, n, 0);
    run<4, 1024, 2, 1>("tagged data", data, flags, sink, out, n, 0);
    // with the member's own work between exchanges (a dependent ALU chain of `work` steps on every wave)
    run<4, 256, 1, 0>("exchange + work", data, flags, sink, out, n, 250);
    run<4, 256, 1, 0>("exchange + work", data, flags, sink, out, n, 1000);
    run<4, 512, 1, 1>("exchange + work", data, flags, sink, out, n, 1000);
    return 0;
}
