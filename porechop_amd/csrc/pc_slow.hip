// pc_slow.hip -- the plain-int32 alignment kernel: one LANE per (window, adapter) pair, any scoring scheme, adapters up
// to pcs::MAX_ADAPTER bases.  It exists so that the drop-in boundary is TOTAL: the reference takes any four integers
// and any adapter (porechop/porechop.py:145,196-202, porechop/src/adapter_align.cpp:11-31), the packed 16-bit kernels
// of pc_kernels.hip only the schemes pcb::scores_supported proves exact.  The arithmetic is pcs::align_pair
// (pc_slow.h), the very function tests/host/test_slow.cpp checks against the oracle on unrestricted schemes.
//
// Layout: the column state (M, H per adapter row) sits in LDS as [row][lane] when the adapter has at most 128 rows
// (64 KB per wave), otherwise in HBM as [row][pair]; the trace, one nibble-holding byte per cell, in HBM as
// [column][row][pair] -- every store of a wave is 64 consecutive bytes, the traceback's loads are per-lane gathers.
// Bound by memory latency, not throughput: this is the refuge path, ~100x slower per cell than the packed kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pc_kernels.h"
#include "pc_slow.h"

namespace pck {

namespace {

__device__ __forceinline__ int code_of(uint8_t c)
{
    // seqan/basic/alphabet_residue_tabs.h:113-140
    if (c == 'A' || c == 'a') return 0;
    if (c == 'C' || c == 'c') return 1;
    if (c == 'G' || c == 'g') return 2;
    if (c == 'T' || c == 't' || c == 'U' || c == 'u') return 3;
    return 4;
}

struct LdsMem {
    int *st; int m, lane; uint8_t *tr; int64_t P, p;
    __device__ __forceinline__ int &M(int i) { return st[(i - 1) * 64 + lane]; }
    __device__ __forceinline__ int &H(int i) { return st[(m + i - 1) * 64 + lane]; }
    __device__ __forceinline__ uint8_t &T(int j, int i) { return tr[((int64_t)(j - 1) * m + (i - 1)) * P + p]; }
};

struct HbmMem {
    int *st; int m; uint8_t *tr; int64_t P, p;
    __device__ __forceinline__ int &M(int i) { return st[(int64_t)(i - 1) * P + p]; }
    __device__ __forceinline__ int &H(int i) { return st[(int64_t)(m + i - 1) * P + p]; }
    __device__ __forceinline__ uint8_t &T(int j, int i) { return tr[((int64_t)(j - 1) * m + (i - 1)) * P + p]; }
};

template <bool LDS>
__global__ __launch_bounds__(64) void slow_kernel(SlowArgs a)
{
    extern __shared__ int dyn[];
    const int lane = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * 64 + lane;
    if (p >= a.count) return;
    const int64_t w = a.first_window + p;
    const int n = a.win_len[w];
    int32_t *o = a.out + (a.out_base + p) * TRACE_OUT_INTS;
    if (n > a.max_len) {                      // the host sized the trace from max_len
        atomicAdd(a.err, 1u);
        return;
    }
    const uint8_t *rd = a.arena + a.win_off[w];
    const uint8_t *ad = a.adapter;            // Dna5 codes already
    pcw::Digest d;
    int err;
    auto rdf = [&](int k) { return code_of(rd[k]); };
    auto adf = [&](int k) { return (int)ad[k]; };
    if constexpr (LDS)
        err = pcs::align_pair(n, a.m, rdf, adf, a.match, a.mismatch, a.gap_open, a.gap_extend,
                              LdsMem{dyn, a.m, lane, a.trace, a.P, p}, d);
    else
        err = pcs::align_pair(n, a.m, rdf, adf, a.match, a.mismatch, a.gap_open, a.gap_extend,
                              HbmMem{a.state, a.m, a.trace, a.P, p}, d);
    if (err) atomicAdd(a.err, 1u);
    ((int4 *)o)[0] = make_int4(d.read_start, d.read_end, d.adapter_start, d.adapter_end);
    ((int4 *)o)[1] = make_int4(d.score, d.matches, d.aligned_len, d.full_len);
}

}  // namespace

int launch_slow(const SlowArgs &a, void *stream)
{
    if (a.count <= 0) return 0;
    const unsigned grid = (unsigned)((a.count + 63) / 64);
    if (a.state == nullptr) {
        const size_t lds = (size_t)a.m * 2 * 64 * sizeof(int);
        (void)hipFuncSetAttribute((const void *)slow_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(slow_kernel<true>, dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(slow_kernel<false>, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace pck
