// pc_prefilter.hip -- exact bit-parallel prefilter for the whole-read ("middle") adapter scan.
//
// Porechop aligns every adapter of the matching sets against every whole read and then keeps only the
// alignments whose full-adapter identity reaches --middle_threshold (porechop/nanopore_read.py:224-241);
// nothing else about the other alignments is ever used.  Its README names the cure for the cost of that
// search (README.md:355-357: SeqAn "is very flexible, but not as fast as some alternatives, such as Edlib").
// An alignment with full-adapter identity >= t has at most k = floor(m (1-t)/t) non-matching columns inside
// the adapter's span (pc_prefilter_max_edits, include/porechop_amd.h), i.e. the adapter lies within k edits
// (unit-cost substitutions / insertions / deletions, adapter global, read local -- overhanging adapter bases
// are deletions) of some substring of the read.  This kernel decides exactly that for every (read, adapter)
// pair with Myers' bit-vector algorithm (G. Myers, J. ACM 46(3), 1999; search variant: horizontal delta 0
// at row 0): pairs it rejects are PROVEN not to be hits, and only the survivors run the DP.
//
// Mapping: one LANE per (read chunk, adapter piece) -- the vertical-delta vectors Pv / Mv of a <= 32-base
// piece are one 32-bit VGPR each, every lane busy, no cross-lane traffic.  A lane carries P pieces
// (P independent dependency chains: the VALU always has work) and streams its own read bytes, 16 at a time.
// A piece occupies the TOP bits of its word (row i of an m-base piece at bit 32-m+i), so the horizontal
// delta of the last row is the carry of `Ph + Ph` / `Mh + Mh`, which is also the shift the recurrence needs:
// the score costs one add-with-carry / subtract-with-borrow per column.  The bits below the piece are rows of
// a wildcard prefix with vertical deltas 0 (Eq = 1, Pv = Mv = 0 there): they keep horizontal deltas 0
// flowing into the piece's first row and never generate a carry.
// Per column and piece: or, and, add, 3 x bitop3, and, 2 x add-with-carry-out, addc, subb, and = 12 VALU, + half a
// v_min3_u32 (two columns' scores per op), + 1 shared address op per column -- against 2.5 per adapter ROW (60-82 per column for 24-33 bases) in the
// specialised score kernel.
// Equality is the reference's: Dna5 codes, everything that is not ACGTU is N, N == N (the per-byte Eq table
// is built by the host from the same code table).
// Adapters above 32 bases are cut into p = ceil(m/32) pieces; by pigeonhole one of them has <= floor(k/p)
// edits, so the pair survives iff some piece does (a superset for those; exact for m <= 32).
// Reads are cut into column chunks (warm-up = piece length + k columns before the chunk: an occurrence
// with <= k edits ending in the chunk starts inside the warm-up), so a launch fills the chip whatever the
// read lengths.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pc_kernels.h"

namespace pck {

namespace {

// read bytes are consumed as delivered: a window starts at any byte (the hardware handles unaligned vector loads)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const int o = __shfl_xor(v, s, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const int o = __shfl_xor(v, s, 64); v = o > v ? o : v; }
    return v;
}

// One column of Myers' recurrence for one piece (see the header comment): 12 VALU instructions, written out
// because hipcc does not find the carry idiom or the three-input boolean forms by itself (18 per step from
// plain C++).  v_bitop3_b32's table is f(a = 0xF0, b = 0xCC, c = 0xAA):  (a ^ b) | c = 0xBE,  a | ~(b | c) = 0xF1.
// `ph + ph` is the shift the recurrence needs AND leaves the last row's horizontal +1 in the carry (`mh + mh`: the
// -1), which v_addc / v_subbrev fold into the score.  sc_out is a fresh register so that two consecutive columns'
// scores can share one v_min3_u32.
__device__ __forceinline__ uint32_t myers_step(uint32_t eq, uint32_t &pv, uint32_t &mv, uint32_t sc)
{
    uint32_t xv, t, xh, sc_out;
    asm("v_or_b32 %[xv], %[eq], %[mv]\n\t"
        "v_and_b32 %[t], %[eq], %[pv]\n\t"
        "v_add_u32 %[t], %[t], %[pv]\n\t"
        "v_bitop3_b32 %[xh], %[t], %[pv], %[eq] bitop3:0xbe\n\t"      // xh = ((eq & pv) + pv) ^ pv | eq
        "v_bitop3_b32 %[t], %[mv], %[xh], %[pv] bitop3:0xf1\n\t"      // ph = mv | ~(xh | pv)
        "v_and_b32 %[xh], %[pv], %[xh]\n\t"                           // mh = pv & xh
        "v_add_co_u32 %[t], vcc, %[t], %[t]\n\t"                      // ph << 1, carry = the last row's +1
        "v_addc_co_u32 %[sco], vcc, 0, %[sc], vcc\n\t"
        "v_add_co_u32 %[xh], vcc, %[xh], %[xh]\n\t"                   // mh << 1, carry = the last row's -1
        "v_subbrev_co_u32 %[sco], vcc, 0, %[sco], vcc\n\t"
        "v_bitop3_b32 %[pv], %[xh], %[xv], %[t] bitop3:0xf1\n\t"      // pv = mh' | ~(xv | ph')
        "v_and_b32 %[mv], %[t], %[xv]"                                  // mv = ph' & xv
        : [xv] "=&v"(xv), [t] "=&v"(t), [xh] "=&v"(xh), [sco] "=&v"(sc_out), [pv] "+v"(pv), [mv] "+v"(mv)
        : [eq] "v"(eq), [sc] "v"(sc)
        : "vcc");
    return sc_out;
}

// the P Eq words of one byte value: one ds_read_b128 per four pieces
template <int P>
__device__ __forceinline__ void load_row(const uint32_t *tab, uint32_t byte, uint32_t (&e)[P])
{
    if constexpr (P >= 4) {
#pragma unroll
        for (int i = 0; i < P; i += 4) {
            const uint4 v = *(const uint4 *)(tab + byte * P + i);
            e[i] = v.x; e[i + 1] = v.y; e[i + 2] = v.z; e[i + 3] = v.w;
        }
    } else if constexpr (P == 2) {
        const uint2 v = *(const uint2 *)(tab + byte * P);
        e[0] = v.x; e[1] = v.y;
    } else {
        e[0] = tab[byte];
    }
}

template <int P>
__global__ __launch_bounds__(256) void prefilter_kernel(PrefilterArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[256 * P];
    const int group = blockIdx.y;
    {   // the group's Eq table: [byte value][piece]
        const uint4 *src = (const uint4 *)(a.tables + (size_t)group * 256 * P);
        uint4 *dst = (uint4 *)tab;
        for (int i = threadIdx.x; i < 256 * P / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int32_t *meta = a.piece_meta + (size_t)group * P * 4;      // m, k, mask word, mask bit

    // unit = (window, chunk), chunk-major: a wave holds 64 consecutive windows at one chunk index
    const int64_t wblocks = (a.nwindows + 255) / 256;
    const int chunk = (int)(blockIdx.x / wblocks);
    const int64_t w = (int64_t)(blockIdx.x % wblocks) * 256 + threadIdx.x;
    int n = 0;
    const uint8_t *p = a.arena;
    if (w < a.nwindows) {
        const int len = a.win_len[w];
        const int c0 = chunk * a.chunk_len;
        if (c0 < len) {
            const int start = c0 > a.warm ? c0 - a.warm : 0;
            const int end = (c0 + a.chunk_len < len) ? c0 + a.chunk_len : len;
            n = end - start;
            p = a.arena + a.win_off[w] + start;
        }
    }
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max(n));
    if (nmax == 0) return;
    // shortest chunk among the lanes that have one (lanes past the end of their read sit the whole launch out)
    const int nmin = __builtin_amdgcn_readfirstlane(wave_min(n > 0 ? n : 0x7FFFFFFF));
    if (n <= 0) return;

    uint32_t pv[P], mv[P], sc[P], mn[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int m = meta[i * 4 + 0];                // 0: unused slot of the group (Eq table all ones, score 0)
        pv[i] = m >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> m);
        if (m == 0) pv[i] = 0;
        mv[i] = 0;
        sc[i] = (uint32_t)m;
        mn[i] = m == 0 ? 0x7FFFFFFFu : (uint32_t)m;
    }

    // ---- 16-column blocks every active lane of the wave has: no masking, next block's bytes in flight ----------
    const int full = nmin & ~15;
    u32x4 cur = {0, 0, 0, 0};
    if (full > 0) cur = *(const u32x4_unaligned *)p;
    for (int j0 = 0; j0 < full; j0 += 16) {
        u32x4 nxt = cur;
        if (j0 + 16 < full) nxt = *(const u32x4_unaligned *)(p + j0 + 16);
        const uint32_t wd[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int b = 0; b < 4; b += 2) {
                uint32_t e0[P], e1[P];
                load_row<P>(tab, (wd[q] >> (8 * b)) & 0xFFu, e0);
                load_row<P>(tab, (wd[q] >> (8 * b + 8)) & 0xFFu, e1);
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const uint32_t s1 = myers_step(e0[i], pv[i], mv[i], sc[i]);
                    const uint32_t s2 = myers_step(e1[i], pv[i], mv[i], s1);
                    const uint32_t lo = s1 < s2 ? s1 : s2;
                    mn[i] = lo < mn[i] ? lo : mn[i];
                    sc[i] = s2;
                }
            }
        }
        cur = nxt;
    }
    // ---- the rest (chunks of unequal length in one wave, the last < 16 columns): lanes sit out column by column.
    // A lane's last block may read up to 15 bytes past its window: the arena is readable 16 bytes past its end.
    for (int j0 = full; j0 < nmax; j0 += 16) {
        if (j0 < n) {
            const u32x4 blk = *(const u32x4_unaligned *)(p + j0);
            const uint32_t wd[4] = {blk.x, blk.y, blk.z, blk.w};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (j0 + c < n) {
                    uint32_t e[P];
                    load_row<P>(tab, (wd[c >> 2] >> (8 * (c & 3))) & 0xFFu, e);
#pragma unroll
                    for (int i = 0; i < P; ++i) {
                        sc[i] = myers_step(e[i], pv[i], mv[i], sc[i]);
                        mn[i] = sc[i] < mn[i] ? sc[i] : mn[i];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int m = meta[i * 4 + 0], k = meta[i * 4 + 1];
        if (m > 0 && (int)mn[i] <= k) atomicOr(a.mask + w * a.words + meta[i * 4 + 2], (uint32_t)meta[i * 4 + 3]);
    }
}

}  // namespace

int launch_prefilter(const PrefilterArgs &a, int pieces_per_lane, int ngroups, void *stream)
{
    if (a.nwindows <= 0 || ngroups <= 0) return 0;
    const int64_t wblocks = (a.nwindows + 255) / 256;
    const int64_t gx = wblocks * a.chunks;
    if (gx > 0x7FFFFFFFll || ngroups > 65535) return -1;
    const dim3 grid((unsigned)gx, (unsigned)ngroups), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (pieces_per_lane) {
        case 1: hipLaunchKernelGGL(prefilter_kernel<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(prefilter_kernel<2>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(prefilter_kernel<4>, grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL(prefilter_kernel<8>, grid, block, 0, s, a); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace pck
