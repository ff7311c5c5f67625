// pc_middle.hip -- the glue of the middle scan (Pipeline.phase_c) as four small kernels.
//
// Between the scans of a step, porechop_amd/pipeline.py decides -- per read -- what nanopore_read.py:56-62,210-243 decide:
// the trimmed interval the middle scan looks at, which alignments are hits (full identity >= --middle_threshold), which
// adapter of a masked read hits first.  As torch expressions that is ~150 elementwise launches of a few microseconds each per
// step, issued by a host that cannot launch them faster than they run: the GPU idles a third of a 17 ms step between them
// (tools/r6_gaps.py).  One launch each here; the torch formulation stays for aligners without these entry points (tests).
// HBM-bound integer / double work, one thread per element, block-reduced statistics with one atomic per block and value.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pc_kernels.h"

namespace pck {

namespace {

__device__ __forceinline__ double identity6(int matches, int len)
{
    const double x = (100.0 * (double)matches) / (double)len;      // exactly pc_reduce.hip's identity(): %f printed, parsed back
    return rint(x * 1e6) / 1e6;
}

// sum / max / min of a value over the block -> one atomic each from thread 0 (256 threads)
template <class T, class Op> __device__ __forceinline__ T block_reduce(T v, T *scratch, Op op)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = op(v, __shfl_xor(v, s));
    if (lane == 0) scratch[wv] = v;
    __syncthreads();
    T r = scratch[0];
    for (int k = 1; k < 4; ++k) r = op(r, scratch[k]);
    __syncthreads();
    return r;
}

}  // namespace

// ---- seq[start_trim : len - end_trim] with Python's slice semantics (nanopore_read.py:56-62; pipeline.trimmed_interval) --------
// stats: [reads with a non-empty interval, longest, kStatBig - shortest non-empty (0: none), sum of lengths]  (zero-initialised: a
// memset on the stream, no upload -- a pageable upload would drain the queue)
constexpr long long kStatBig = 1ll << 40;
__global__ __launch_bounds__(256) void trim_windows_kernel(const int64_t *off, const int32_t *len, const int32_t *st, const int32_t *et, int64_t n,
                                                           int64_t *toff, int32_t *tlen, long long *stats)
{
    __shared__ long long scratch[4];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    long long t = 0;
    if (r < n) {
        const long long ln = len[r], s0 = st[r], e0 = et[r];
        long long s_pos = s0 < ln ? s0 : ln;
        long long e_pos = ln - e0;
        if (e_pos < 0) { e_pos = ln + e_pos; if (e_pos < 0) e_pos = 0; }
        if (s0 == 0 && e0 == 0) { s_pos = 0; e_pos = ln; }
        t = e_pos - s_pos;
        if (t < 0) t = 0;
        toff[r] = off[r] + s_pos;
        tlen[r] = (int32_t)t;
    }
    const long long live = block_reduce<long long>(t > 0 ? 1 : 0, scratch, [](long long a, long long b) { return a + b; });
    const long long mx = block_reduce<long long>(t, scratch, [](long long a, long long b) { return a > b ? a : b; });
    const long long mn = block_reduce<long long>(t > 0 ? kStatBig - t : 0, scratch, [](long long a, long long b) { return a > b ? a : b; });
    const long long sm = block_reduce<long long>(t, scratch, [](long long a, long long b) { return a + b; });
    if (threadIdx.x == 0) {
        if (live) atomicAdd((unsigned long long *)stats + 0, (unsigned long long)live);
        atomicMax(stats + 1, mx);
        atomicMax(stats + 2, mn);
        if (sm) atomicAdd((unsigned long long *)stats + 3, (unsigned long long)sm);
    }
}

// ---- full-adapter identity of whole-read records and whether they are hits (nanopore_read.py:218-226) ------------------
__global__ __launch_bounds__(256) void middle_hits_kernel(const int32_t *rec, int64_t n, double threshold, double *full, uint8_t *hit)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 a = ((const int4 *)(rec + i * TRACE_OUT_INTS))[0];
    const int4 b = ((const int4 *)(rec + i * TRACE_OUT_INTS))[1];
    double f = 0.0;
    if (a.x != -1) { f = identity6(b.y, b.w); if (f != f) f = 0.0; }      // (an all-zero record -- "not computed" -- is 0 / 0: not a hit)
    full[i] = f;
    hit[i] = (a.x != -1 && f >= threshold) ? 1 : 0;
}

// ---- which windows survive the prefilter for each adapter SET: cand[g][w] = mask[w] & gmask[g] != 0, and how many ----------
__global__ __launch_bounds__(256) void group_survivors_kernel(const int32_t *mask, int64_t n, int words, const int32_t *gmask, int ngroups,
                                                              uint8_t *cand, unsigned long long *counts)
{
    __shared__ long long scratch[4];
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int g = 0; g < ngroups; ++g) {
        int any = 0;
        if (w < n) {
            for (int k = 0; k < words; ++k) any |= mask[w * words + k] & gmask[g * words + k];
            cand[(int64_t)g * n + w] = any ? 1 : 0;
        }
        const long long c = block_reduce<long long>(any ? 1 : 0, scratch, [](long long a, long long b) { return a + b; });
        if (threadIdx.x == 0 && c) atomicAdd(counts + g, (unsigned long long)c);
    }
}

// ---- one round of mask-and-realign, the consuming part (nanopore_read.py:210-243 for every active read at once) -------------
// Active read k = dirty read d = act[k], standing at adapter cur[d]: the first adapter a >= cur[d] whose record of d is a hit.
// stats: [alignments consumed, reads that hit, kStatBig - smallest hit adapter (0: none), bases to mask]  (zero-initialised)
__global__ __launch_bounds__(256) void round_consume_kernel(const double *full_all, const int32_t *rec_all, const int64_t *cur, const int64_t *act,
                                                            int64_t nact, int A, int64_t Dn, double threshold, uint8_t *anyh, int32_t *a_hit,
                                                            int64_t *cnt, long long *stats)
{
    __shared__ long long scratch[4];
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    long long used = 0, hitc = 0, amin = 0, masked = 0;
    if (k < nact) {
        const int64_t d = act[k];
        const int c = (int)cur[d];
        int found = -1;
        for (int a = c; a < A; ++a) {
            if (full_all[(int64_t)a * Dn + d] >= threshold && rec_all[((int64_t)a * Dn + d) * TRACE_OUT_INTS] != -1) { found = a; break; }
        }
        anyh[k] = found >= 0 ? 1 : 0;
        a_hit[k] = found >= 0 ? found : 0;
        long long m = 0;
        if (found >= 0) {
            const int32_t *r = rec_all + ((int64_t)found * Dn + d) * TRACE_OUT_INTS;
            m = (long long)r[1] + 1 - (long long)r[0];
            if (m < 0) m = 0;
            used = found - c + 1; hitc = 1; amin = kStatBig - found;
        } else {
            used = A - c;
        }
        cnt[k] = m;
        masked = m;
    }
    const long long u = block_reduce<long long>(used, scratch, [](long long a, long long b) { return a + b; });
    const long long h = block_reduce<long long>(hitc, scratch, [](long long a, long long b) { return a + b; });
    const long long mn = block_reduce<long long>(amin, scratch, [](long long a, long long b) { return a > b ? a : b; });
    const long long ms = block_reduce<long long>(masked, scratch, [](long long a, long long b) { return a + b; });
    if (threadIdx.x == 0) {
        if (u) atomicAdd((unsigned long long *)stats + 0, (unsigned long long)u);
        if (h) atomicAdd((unsigned long long *)stats + 1, (unsigned long long)h);
        atomicMax(stats + 2, mn);
        if (ms) atomicAdd((unsigned long long *)stats + 3, (unsigned long long)ms);
    }
}

int launch_trim_windows(const int64_t *off, const int32_t *len, const int32_t *st, const int32_t *et, int64_t n, int64_t *toff, int32_t *tlen,
                        int64_t *stats, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(trim_windows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, off, len, st, et, n, toff, tlen,
                       (long long *)stats);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_middle_hits(const int32_t *rec, int64_t n, double threshold, double *full, uint8_t *hit, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(middle_hits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rec, n, threshold, full, hit);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_group_survivors(const int32_t *mask, int64_t n, int words, const int32_t *gmask, int ngroups, uint8_t *cand, int64_t *counts, void *stream)
{
    if (n <= 0 || ngroups <= 0) return 0;
    hipLaunchKernelGGL(group_survivors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, n, words, gmask, ngroups, cand,
                       (unsigned long long *)counts);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_round_consume(const double *full_all, const int32_t *rec_all, const int64_t *cur, const int64_t *act, int64_t nact, int A, int64_t Dn,
                         double threshold, uint8_t *anyh, int32_t *a_hit, int64_t *cnt, int64_t *stats, void *stream)
{
    if (nact <= 0) return 0;
    hipLaunchKernelGGL(round_consume_kernel, dim3((unsigned)((nact + 255) / 256)), dim3(256), 0, (hipStream_t)stream, full_all, rec_all, cur, act, nact, A,
                       Dn, threshold, anyh, a_hit, cnt, (long long *)stats);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace pck
