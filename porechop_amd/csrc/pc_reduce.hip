// pc_reduce.hip -- per-read reduction of the end-window records on the device.
//
// What porechop/nanopore_read.py does per read with the alignments of phase B, for a whole batch:
//   find_start_trim / find_end_trim   nanopore_read.py:166-208   (trim amounts)
//   determine_barcode                 nanopore_read.py:399-466   (barcode call; without the Albacore rule)
// Input: the records pc_scan_device wrote for J jobs over the same n reads (job j's records are
// contiguous, read r of job j at rec[job_off[j] + r]), each job being one adapter sequence
// against the start (side 0) or the end (side 1) window of every read.  HBM-bound integer work:
// one thread per read, adjacent threads read adjacent 32-byte records (coalesced per job).
//
// Identities are the doubles Python sees: (100.0 * matches) / length printed with %f and parsed
// back, i.e. rounded half-to-even at 6 decimals (porechop/src/alignment.cpp:113-121,
// nanopore_read.py:476-491); a failed alignment (field 0 == -1) scores 0.0, and so does a score-only record
// (field 0 == -2) that the exact pruning of phase B proved irrelevant and left untraced (pc_select.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pc_kernels.h"

namespace pck {

namespace {

struct Rec { int32_t rs, re, as, ae, score, matches, aligned_len, full_len; };

__device__ __forceinline__ Rec load_rec(const int32_t *base, int64_t idx)
{
    const int4 a = ((const int4 *)(base + idx * TRACE_OUT_INTS))[0];
    const int4 b = ((const int4 *)(base + idx * TRACE_OUT_INTS))[1];
    return {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}

__device__ __forceinline__ double identity(int matches, int len)
{
    const double x = (100.0 * (double)matches) / (double)len;      // 0/0 -> NaN: compares false, like Python's nan
    return rint(x * 1e6) / 1e6;
}

__device__ __forceinline__ double full_identity(const Rec &r)
{
    return r.rs < 0 ? 0.0 : identity(r.matches, r.full_len);
}

}  // namespace

__global__ __launch_bounds__(256) void reduce_kernel(ReduceArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    // ---- trims: every job, in job order (the order is irrelevant to a max) -----------------
    int start_trim = 0, end_trim = 0;
    // (a wave's 64 reads share one mask word per job: a single broadcast load decides for all of them)
    auto untraced = [&](int j) -> bool {
        return a.traced_mask && !((a.traced_mask[(int64_t)j * a.mask_words + (r >> 6)] >> (r & 63)) & 1ull);
    };
    for (int j = 0; j < a.njobs; ++j) {
        if (untraced(j)) continue;
        const Rec rec = load_rec(a.records, a.job_off[j] + r);
        if (rec.rs < 0) continue;                      // -1: no alignment; -2: a score record left untraced (pc_select.hip)
        const double partial = identity(rec.matches, rec.aligned_len);
        const int rs = rec.rs, re = rec.re + 1;
        if (!(partial > a.end_threshold) || re - rs < a.min_trim_size) continue;
        if (a.job_side[j] == 0) {
            if (re != a.end_size) { const int t = re + a.extra_end_trim; start_trim = t > start_trim ? t : start_trim; }
        } else {
            if (rs != 0) { const int t = (a.end_size - rs) + a.extra_end_trim; end_trim = t > end_trim ? t : end_trim; }
        }
    }
    a.start_trim[r] = start_trim;
    a.end_trim[r] = end_trim;
    if (a.nbins <= 0 || !a.call) return;

    // ---- barcode call ------------------------------------------------------------------------
    // bin k has a start entry (job bin_start[k]) and an end entry (job bin_end[k]); a missing entry
    // scores 0.0.  Python sorts (name, score) lists with a stable descending sort: among equal
    // scores the entry inserted first wins -- start entries before end entries, bins in order.
    auto score_of = [&](const int32_t *jobs, int k) -> double {
        const int j = jobs[k];
        return (j < 0 || untraced(j)) ? 0.0 : full_identity(load_rec(a.records, a.job_off[j] + r));
    };
    int call = -1;
    if (a.require_two) {
        // best and second best (the next entry of the sorted list) of each side
        auto best_two = [&](const int32_t *jobs, int &bi, double &bv, double &second) {
            bi = 0; bv = score_of(jobs, 0); second = -1.0;
            for (int k = 1; k < a.nbins; ++k) {
                const double v = score_of(jobs, k);
                if (v > bv) { second = bv; bv = v; bi = k; }
                else if (v > second) second = v;
            }
            if (a.nbins < 2) second = 0.0;
        };
        int si, ei; double sv, s2, ev, e2;
        best_two(a.bin_start, si, sv, s2);
        best_two(a.bin_end, ei, ev, e2);
        if (sv >= a.barcode_threshold && ev >= a.barcode_threshold && sv >= s2 + a.barcode_diff && ev >= e2 + a.barcode_diff && si == ei)
            call = si;
    } else {
        int bk = 0; double bv = score_of(a.bin_start, 0);
        for (int k = 1; k < a.nbins; ++k) { const double v = score_of(a.bin_start, k); if (v > bv) { bv = v; bk = k; } }
        for (int k = 0; k < a.nbins; ++k) { const double v = score_of(a.bin_end, k); if (v > bv) { bv = v; bk = k; } }
        // second best: the best score among the OTHER bins (a bin keeps the better of its two entries)
        double second = 0.0;
        for (int k = 0; k < a.nbins; ++k) {
            if (k == bk) continue;
            const double s = score_of(a.bin_start, k), e = score_of(a.bin_end, k);
            const double v = s > e ? s : e;
            second = v > second ? v : second;
        }
        if (bv >= a.barcode_threshold && bv >= second + a.barcode_diff) call = bk;
    }
    a.call[r] = call;
}

int launch_reduce(const ReduceArgs &a, void *stream)
{
    if (a.n <= 0) return 0;
    const unsigned grid = (unsigned)((a.n + 255) / 256);
    hipLaunchKernelGGL(reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// Tile table from runs: one thread per tile finds its run (runs are sorted by tile0) and writes the
// tile's descriptors.  A scan of 1 M reads against 98 adapter pairs is 1.5 M tiles (86 MB) described by 98
// runs: built here in microseconds instead of on the host and across PCIe.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_tiles_kernel(const TileRun *runs, int nruns, Tile *tiles, int64_t ntiles)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    int lo = 0, hi = nruns - 1;
    while (lo < hi) {                         // last run with tile0 <= t
        const int mid = (lo + hi + 1) >> 1;
        if (runs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
    }
    const TileRun r = runs[lo];
    const int64_t k = t - r.tile0;
    Tile o;
    if (r.dual) {
        const int64_t s = 64 * k;
        o.win_lo = r.win0 + s; o.win_hi = r.win0 + s;
        o.out_lo = r.out0 + s; o.out_hi = r.out0 + r.n + s;
        const int64_t c = r.n - s < 64 ? r.n - s : 64;
        o.count_lo = o.count_hi = (int32_t)(c > 0 ? c : 0);
    } else {
        const int64_t s = 128 * k;
        o.win_lo = r.win0 + s; o.win_hi = r.win0 + s + 64;
        o.out_lo = r.out0 + s; o.out_hi = r.out0 + s + 64;
        const int64_t cl = r.n - s < 64 ? r.n - s : 64, ch = r.n - s - 64 < 64 ? r.n - s - 64 : 64;
        o.count_lo = (int32_t)(cl > 0 ? cl : 0); o.count_hi = (int32_t)(ch > 0 ? ch : 0);
    }
    o.adapter_lo = r.adapter_lo; o.adapter_hi = r.adapter_hi; o.rows = r.rows; o.pad_ = 0;
    tiles[t] = o;
}

int launch_expand_tiles(const TileRun *d_runs, int nruns, Tile *d_tiles, int64_t ntiles, void *stream)
{
    if (ntiles <= 0 || nruns <= 0) return 0;
    const int64_t blocks = (ntiles + 255) / 256;
    hipLaunchKernelGGL(expand_tiles_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_runs, nruns, d_tiles, ntiles);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// Packed private copies of windows (the maskable copies of reads with middle hits): one workgroup per
// window copies its bytes and pads up to the next copy's start.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_windows_kernel(const uint8_t *arena, const int64_t *src_off, const int32_t *len,
                                                           uint8_t *dst, const int64_t *dst_off, int pad)
{
    const int64_t i = blockIdx.x;
    const uint8_t *s = arena + src_off[i];
    uint8_t *d = dst + dst_off[i];
    const int64_t n = len[i], total = dst_off[i + 1] - dst_off[i];
    for (int64_t b = threadIdx.x; b < total; b += blockDim.x) d[b] = b < n ? s[b] : (uint8_t)pad;
}

int launch_copy_windows(const uint8_t *arena, const int64_t *src_off, const int32_t *len, int64_t n, uint8_t *dst,
                        const int64_t *dst_off, int pad, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(copy_windows_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, arena, src_off, len, dst, dst_off, pad);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- pc_unpack_device: 2 bits per base (+ the positions of the non-ACGT bases) -> one byte per base ------------------
// HBM-bound: a thread turns one dword of the plane (16 bases) into one 16-byte store; 0.25 B read + 1 B written per base.
__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t *packed, int64_t nbases, uint8_t *arena, int pad)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // group of 16 bases
    const int64_t base = g * 16;
    if (base >= nbases + pad) return;
    uint32_t out[4];
    if (base + 16 <= nbases) {
        const uint32_t w = packed[g];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t code = (w >> (2 * (4 * q + k))) & 3u;
                v |= ((0x54474341u >> (8 * code)) & 0xFFu) << (8 * k);      // "ACGT"
            }
            out[q] = v;
        }
        *(uint4 *)(arena + base) = uint4{out[0], out[1], out[2], out[3]};
        return;
    }
    // the tail: the plane's last (possibly partial) dword is read a byte at a time, then `pad` bytes of 'N'
    const uint8_t *pb = (const uint8_t *)packed;
    for (int k = 0; k < 16 && base + k < nbases + pad; ++k) {
        const int64_t i = base + k;
        arena[i] = i < nbases ? (uint8_t)((0x54474341u >> (8 * ((pb[i >> 2] >> (2 * (i & 3))) & 3u))) & 0xFFu) : (uint8_t)'N';
    }
}

__global__ __launch_bounds__(256) void unpack_exceptions_kernel(const int64_t *exc_pos, int64_t nexc, int64_t nbases, uint8_t *arena)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nexc) return;
    const int64_t p = exc_pos[i];
    if (p >= 0 && p < nbases) arena[p] = (uint8_t)'N';
}

int launch_unpack(const void *packed, int64_t nbases, const int64_t *exc_pos, int64_t nexc, void *arena, int pad, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int64_t groups = (nbases + pad + 15) / 16;
    if (groups > 0)
        hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, (const uint32_t *)packed, nbases,
                           (uint8_t *)arena, pad);
    if (nexc > 0)
        hipLaunchKernelGGL(unpack_exceptions_kernel, dim3((unsigned)((nexc + 255) / 256)), dim3(256), 0, st, exc_pos, nexc, nbases,
                           (uint8_t *)arena);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace pck
