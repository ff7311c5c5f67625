// pc_select.hip -- exact pruning of phase B on the device: which end-window alignments have to be traced.
//
// Phase B aligns every adapter of the matching sets (with a barcode panel: ~200 sequences) against both end windows
// of every read and keeps, per read, the MAXIMUM trim over the alignments that qualify (nanopore_read.py:166-208)
// and the barcode call from the best full identities (nanopore_read.py:399-466).  Two or three of those ~200
// alignments decide the result.  The score-only pass (5 instead of 13.25 packed ops per two cells, no trace slab)
// gives every alignment's end cell -- I adapter bases and Jc window columns consumed -- and its score S: the
// reference's own end cell, the one a traced scan starts its traceback from.  From (I, Jc, S) follow UPPER BOUNDS on
// what the alignment can contribute (derivations: porechop_amd/pipeline.py, "Exact pruning of phase B"):
//   full identity <= 100 min(I, Jc, m) / m
//   start window:  trim <= Jc + 1 + extra; none if Jc + 1 < min_trim_size or the path ends in the last column of a full window
//   end window:    none if Jc - 1 < ceil((S + g I) / (match + g)), else trim <= end_size - max(1, Jc - bmax) + extra,
//                  bmax = I + (match I - S) / g,  g = min(|open|, |extend|)
//   either side:   a trim needs S > (tau match - (1 - tau) P) min(I, Jc)    (P = the dearest non-matching column)
// select_kernel, round 1: per read and side, the two best-SCORING pairs that can trim at all and the two best-scoring
// barcode pairs.  They are traced, scattered over their score records (scatter_kernel), and the exact reduction
// (pc_reduce.hip, which reads a score record as "no alignment") gives the trims so far and the best barcode identity
// per side.  Round 2: the pairs whose bound still exceeds the trim so far, and the barcode pairs that could reach
// max(best, threshold) - diff.  Every pair left untraced is proven unable to change the maximum or the call.
//
// HBM-bound integer work: one thread per read, adjacent threads read adjacent 32-byte records (coalesced per job);
// the selection leaves the kernel as one bit per (job, read) -- a ballot per job -- and gather_kernel turns the bits
// of each job into the window list of a traced scan.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pc_kernels.h"

namespace pck {

namespace {

struct Bound { int32_t ub; double ub_full; int32_t S; };

__device__ __forceinline__ int64_t floor_div(int64_t a, int64_t b)      // b > 0
{
    int64_t q = a / b;
    if ((a % b) != 0 && a < 0) --q;
    return q;
}

__device__ __forceinline__ Bound bound_of(const SelectArgs &a, const int32_t *rec, int side, int m, int nwin)
{
    const int4 v = ((const int4 *)rec)[0];
    const int flag = v.x, Jc = v.y, I = v.z, S = rec[4];
    Bound b;
    b.S = S;
    if (flag != -2) {                                   // anything that is not a plain score record: trace it
        b.ub = 1 << 20; b.ub_full = 100.0;
        return b;
    }
    const int mn = I < Jc ? I : Jc;
    b.ub_full = 100.0 * (double)(mn < m ? mn : m) / (double)m;
    int ub;
    if (side == 0) {
        const bool ok = (Jc + 1 >= a.min_trim_size) && !((Jc == nwin) && (nwin == a.end_size));
        ub = ok ? Jc + 1 + a.extra_end_trim : 0;
    } else {
        const int g = (-a.gap_open < -a.gap_extend) ? -a.gap_open : -a.gap_extend;
        if (g > 0 && a.match + g > 0) {
            const int64_t bmin = floor_div((int64_t)S + (int64_t)g * I + (a.match + g - 1), a.match + g);
            const int64_t over = (int64_t)a.match * I - S;
            const int64_t bmax = I + (over > 0 ? over : 0) / g;
            const bool ok = (Jc - 1 >= bmin) && (bmax + 1 >= a.min_trim_size);
            const int64_t back = Jc - bmax > 1 ? Jc - bmax : 1;
            ub = ok ? (int)(a.end_size - back + a.extra_end_trim) : 0;
        } else {
            ub = a.end_size - 1 + a.extra_end_trim;
        }
    }
    if (a.ident_c > 0.0) {
        const int64_t need = (int64_t)floor(a.ident_c * (double)(mn > 1 ? mn : 1));
        if (!((int64_t)S > need)) ub = 0;
    }
    b.ub = ub;
    return b;
}

struct Top2 { int32_t s0, j0, s1, j1; };
__device__ __forceinline__ void top2_init(Top2 &t) { t.s0 = t.s1 = -1; t.j0 = t.j1 = -1; }
__device__ __forceinline__ void top2_add(Top2 &t, int s, int j)
{
    if (s > t.s0) { t.s1 = t.s0; t.j1 = t.j0; t.s0 = s; t.j0 = j; }
    else if (s > t.s1) { t.s1 = s; t.j1 = j; }
}

}  // namespace

__global__ __launch_bounds__(256) void select_kernel(SelectArgs a)
{
    extern __shared__ unsigned s_cnt[];                 // [njobs] pairs selected by this block (lds_counts)
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = r < a.n;
    const int64_t word = r >> 6;                        // 64 consecutive reads = one wave = one mask word per job
    const int lane = threadIdx.x & 63;
    if (a.lds_counts) {
        for (int j = threadIdx.x; j < a.njobs; j += 256) s_cnt[j] = 0;
        __syncthreads();
    }
    const int n_start = live ? a.start_len[r] : 0, n_end = live ? a.end_len[r] : 0;
    auto emit = [&](int j, bool pick) {
        const unsigned long long bits = __ballot(pick);
        if (lane == 0 && word < a.words) {              // (the last block's waves beyond the last read own no word)
            a.mask_out[(int64_t)j * a.words + word] = bits;
            if (bits) {
                const unsigned c = (unsigned)__popcll(bits);
                if (a.lds_counts) atomicAdd(&s_cnt[j], c);
                else atomicAdd(a.counts + j, (unsigned long long)c);
            }
        }
    };
    if (a.round == 1) {
        Top2 trim[2], call[2];
        top2_init(trim[0]); top2_init(trim[1]); top2_init(call[0]); top2_init(call[1]);
        const bool calls_on = a.call_level < 1e8;
        if (live) {
            for (int j = 0; j < a.njobs; ++j) {
                const int side = a.job_side[j];
                const Bound b = bound_of(a, a.records + (a.job_off[j] + r) * TRACE_OUT_INTS, side, a.job_len[j], side ? n_end : n_start);
                if (a.ub_trim_out) a.ub_trim_out[(int64_t)j * a.n + r] = b.ub;
                if (a.ub_full_out) a.ub_full_out[(int64_t)j * a.n + r] = b.ub_full;
                if (b.S < 0) continue;
                if (b.ub > 0) top2_add(trim[side], b.S, j);
                if (calls_on && a.job_call[j]) top2_add(call[side], b.S, j);
            }
        }
        for (int j = 0; j < a.njobs; ++j) {
            const bool pick = live && (j == trim[0].j0 || j == trim[0].j1 || j == trim[1].j0 || j == trim[1].j1 ||
                                       j == call[0].j0 || j == call[0].j1 || j == call[1].j0 || j == call[1].j1);
            emit(j, pick);
        }
    } else {
        const int so_far[2] = {live ? a.start_trim[r] : 0, live ? a.end_trim[r] : 0};
        // A barcode pair left untraced counts as identity 0.  That changes no call as long as its identity is below
        // max(best traced on its side, --barcode_threshold) - --barcode_diff: it can then neither become the best nor
        // come within the difference of it.  An identity of t needs the geometric bound >= t AND S >= m (t match - (1 - t) P).
        const bool calls_on = a.call_level < 1e8;
        double lvl[2] = {0.0, 0.0};
        if (calls_on && live) {
            for (int s = 0; s < 2; ++s) {
                double best = a.best_full[(int64_t)s * a.n + r];
                const double floor_level = a.call_level + a.call_level_diff;
                if (best < floor_level) best = floor_level;
                lvl[s] = best - a.call_level_diff - 1e-6;
            }
        }
        for (int j = 0; j < a.njobs; ++j) {
            const unsigned long long prev = word < a.words ? a.mask_prev[(int64_t)j * a.words + word] : 0ull;   // wave-uniform
            bool pick = false;
            if (live && !((prev >> lane) & 1ull)) {
                const int side = a.job_side[j], m = a.job_len[j];
                const Bound b = bound_of(a, a.records + (a.job_off[j] + r) * TRACE_OUT_INTS, side, m, side ? n_end : n_start);
                pick = b.ub > so_far[side];
                if (calls_on && a.job_call[j]) {
                    const double smin = floor((double)m * ((lvl[side] / 100.0) * (double)(a.match + a.pen_max) - (double)a.pen_max) - 1e-9);
                    pick = pick || (b.ub_full >= lvl[side] && (double)b.S >= smin);
                }
            }
            emit(j, pick);
        }
    }
    if (a.lds_counts) {
        __syncthreads();
        for (int j = threadIdx.x; j < a.njobs; j += 256)
            if (s_cnt[j]) atomicAdd(a.counts + j, (unsigned long long)s_cnt[j]);
    }
}

int launch_select(const SelectArgs &a, void *stream)
{
    if (a.n <= 0 || a.njobs <= 0) return 0;
    SelectArgs b = a;
    b.lds_counts = a.njobs <= 8192 ? 1 : 0;
    const unsigned grid = (unsigned)((a.n + 255) / 256);
    hipLaunchKernelGGL(select_kernel, dim3(grid), dim3(256), b.lds_counts ? (size_t)a.njobs * 4 : 0, (hipStream_t)stream, b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// One block per (256 mask words, job): the selected reads of the job become windows [first[j], first[j] + count[j]) of
// the traced scan, in no particular order (blocks claim their share of the job's range from a cursor).
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a)
{
    __shared__ unsigned s_wave[4];
    __shared__ unsigned long long s_base;
    const int j = blockIdx.y;
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long bits = w < a.words ? a.mask[(int64_t)j * a.words + w] : 0ull;
    const unsigned c = (unsigned)__popcll(bits);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned incl = c;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const unsigned o = __shfl_up(incl, s); if (lane >= s) incl += o; }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
    for (int k = 0; k < 4; ++k) { if (k < wv) before += s_wave[k]; total += s_wave[k]; }
    if (threadIdx.x == 0) s_base = total ? atomicAdd(a.cursor + j, (unsigned long long)total) : 0ull;
    __syncthreads();
    if (!c) return;
    int64_t idx = a.first[j] + (int64_t)s_base + before + (incl - c);
    const int side = a.job_side[j];
    const int64_t *off = side ? a.end_off : a.start_off;
    const int32_t *len = side ? a.end_len : a.start_len;
    unsigned long long rest = bits;
    while (rest) {
        const int b = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        const int64_t r = 64 * w + b;
        a.win_off[idx] = off[r];
        a.win_len[idx] = len[r];
        a.dest[idx] = a.job_off[j] + r;
        a.pair_job[idx] = j;
        a.pair_read[idx] = r;
        ++idx;
    }
}

int launch_gather(const GatherArgs &a, void *stream)
{
    if (a.words <= 0 || a.njobs <= 0) return 0;
    const unsigned gx = (unsigned)((a.words + 255) / 256);
    hipLaunchKernelGGL(gather_kernel, dim3(gx, (unsigned)a.njobs), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// out[k] = records[index[k]]: the score records of the selected pairs, in the order of the traced scan's windows
// (PC_MODE_TRACE_AT reads their end cells from its output buffer)
__global__ __launch_bounds__(256) void gather_records_kernel(const int32_t *records, const int64_t *index, int64_t count, int32_t *out)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const int4 *s = (const int4 *)(records + index[k] * TRACE_OUT_INTS);
    int4 *d = (int4 *)(out + k * TRACE_OUT_INTS);
    const int4 lo = s[0], hi = s[1];
    d[0] = lo; d[1] = hi;
}

int launch_gather_records(const int32_t *records, const int64_t *index, int64_t count, int32_t *out, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, records, index, count, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Traced records over the score records they replace; the best full identity of the barcode pairs per (side, read)
// is kept as the bit pattern of a non-negative double (ordered like the value).
__global__ __launch_bounds__(256) void scatter_kernel(ScatterArgs a)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.count) return;
    const int4 lo = ((const int4 *)(a.traced + k * TRACE_OUT_INTS))[0];
    const int4 hi = ((const int4 *)(a.traced + k * TRACE_OUT_INTS))[1];
    int4 *d = (int4 *)(a.records + a.dest[k] * TRACE_OUT_INTS);
    d[0] = lo; d[1] = hi;
    const int j = a.pair_job[k];
    if (a.best_full && a.job_call[j] && lo.x != -1 && hi.w > 0) {
        const double x = (100.0 * (double)hi.y) / (double)hi.w;            // matches / full length, as pc_reduce.hip
        const double full = rint(x * 1e6) / 1e6;
        if (full > 0.0)
            atomicMax((unsigned long long *)a.best_full + (int64_t)a.job_side[j] * a.n + a.pair_read[k],
                      (unsigned long long)__double_as_longlong(full));
    }
}

int launch_scatter(const ScatterArgs &a, void *stream)
{
    if (a.count <= 0) return 0;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace pck
