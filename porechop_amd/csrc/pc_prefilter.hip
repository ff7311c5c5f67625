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
// Adapters above 32 bases: with k <= 8 the first 32 bases stand for the adapter with the same bound (every substring
// of a string within k edits is within k edits); beyond that the adapter is cut into p = ceil(m/32) pieces, of which
// one has <= floor(k/p) edits by pigeonhole.  Either way the pair survives iff some piece does (a superset for those
// adapters; exact for m <= 32).
// Reads are cut into column chunks (warm-up = piece length + k columns before the chunk: an occurrence
// with <= k edits ending in the chunk starts inside the warm-up), so a launch fills the chip whatever the
// read lengths.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pc_kernels.h"

namespace pck {

namespace {

// read bytes are consumed as delivered: a window starts at any byte; the kernels fetch the aligned 16-byte blocks around it
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));

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
    // Columns are fetched as ALIGNED 16-byte blocks (an unaligned vector load is split by the memory pipeline): a lane
    // starts `skip` bytes before its chunk, at the 16-byte boundary below it, and sits those columns out.  Below,
    // columns are counted from that boundary: the chunk is [skip, n).
    int n = 0, skip = 0;
    const uint8_t *p = a.arena;
    if (w < a.nwindows) {
        const int len = a.win_len[w];
        if (chunk == 0 && len > a.max_len) atomicAdd(a.err, 1u);      // columns beyond chunks * chunk_len would go unscanned
        const int c0 = chunk * a.chunk_len;
        if (c0 < len) {
            const int start = c0 > a.warm ? c0 - a.warm : 0;
            const int end = (c0 + a.chunk_len < len) ? c0 + a.chunk_len : len;
            p = a.arena + a.win_off[w] + start;
            skip = (int)((uintptr_t)p & 15u);
            p -= skip;
            n = end - start + skip;
        }
    }
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max(n));
    if (nmax == 0) return;
    // shortest chunk among the lanes that have one (lanes past the end of their read sit the whole launch out)
    const int nmin = __builtin_amdgcn_readfirstlane(wave_min(n > 0 ? n : 0x7FFFFFFF));
    const int any_skip = __builtin_amdgcn_readfirstlane(wave_max(skip));
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

    // ---- first block, when some lane has columns to sit out in it: column by column ------------------------------
    auto masked_block = [&](int j0) {
        if (j0 < n) {
            const u32x4 blk = *(const u32x4 *)(p + j0);
            const uint32_t wd[4] = {blk.x, blk.y, blk.z, blk.w};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (j0 + c >= skip && j0 + c < n) {
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
    };
    const int first = any_skip ? 16 : 0;
    if (any_skip) masked_block(0);
    // ---- 16-column blocks every active lane of the wave has: no masking, next block's bytes in flight ----------
    const int full = nmin & ~15;
    u32x4 cur = {0, 0, 0, 0};
    if (full > first) cur = *(const u32x4 *)(p + first);
    for (int j0 = first; j0 < full; j0 += 16) {
        u32x4 nxt = cur;
        if (j0 + 16 < full) nxt = *(const u32x4 *)(p + j0 + 16);
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
    for (int j0 = full > first ? full : first; j0 < nmax; j0 += 16) masked_block(j0);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int m = meta[i * 4 + 0], k = meta[i * 4 + 1];
        if (m > 0 && (int)mn[i] <= k) atomicOr(a.mask + w * a.words + meta[i * 4 + 2], (uint32_t)meta[i * 4 + 3]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Seed stage: the same decision, most of it without any edit-distance arithmetic.
//
// Pigeonhole (the partition lemma of approximate matching): cut a piece of len bases that may take k edits into
// k + 1 disjoint parts; an occurrence with <= k edits leaves at least one part untouched, so that part's first q
// bases occur EXACTLY in the read, where the alignment puts them.  seed_scan_kernel streams every read once, keeps
// the last 16 bases as 2-bit codes in a register and tests the last q against a bitmap of all seeds of that length
// (one bitmap per seed length present, at most three, all in LDS): 2 + 4 VALU operations per read base and seed
// length, whatever the number of adapters -- the kernel is bound by HBM, not by the VALU.  Positions that hit (a few per thousand)
// are appended to a candidate list; seed_verify_kernel then runs Myers' recurrence for the candidate's piece over
// the len + 2k read columns the occurrence would have to lie in.  A pair is marked iff some candidate verifies:
// exactly the pairs within k edits (the verification is the same exact test on a window that contains every
// occurrence through that seed), so the mask is identical to the exhaustive kernel's.
// Seeds hold only A/C/G/T (a piece with another letter in a seed is left to the exhaustive kernel); read bytes that
// are not bases are scanned as 'A', which can only add candidates -- the verifier reads the real bytes.
// ---------------------------------------------------------------------------------------------------------
// Bitmap slots in LDS (words): the seed lengths are handed over longest first, so slot c always fits 4^q[c] bits
constexpr int kBmOff0 = 0, kBmOff1 = (1 << 16) / 32, kBmOff2 = kBmOff1 + (1 << 14) / 32, kBmWords = kBmOff2 + (1 << 12) / 32;

constexpr int kWaveBuf = 512;                        // finds a wave can keep in LDS; it flushes them from half that on

__device__ __forceinline__ void flush_wave(const SeedScanArgs &a, int *wcnt, uint2 (*wbuf)[kWaveBuf], int wv, int lane)
{
    int n = wcnt[wv];
    n = n < kWaveBuf ? n : kWaveBuf;
    n = __builtin_amdgcn_readfirstlane(n);
    if (n <= 0) return;
    unsigned long long base = 0;
    if (lane == __builtin_amdgcn_readfirstlane(lane)) base = atomicAdd(a.count, (unsigned long long)n);
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)bhi << 32) | blo;
    for (int i = lane; i < n; i += 64)
        if (base + i < (unsigned long long)a.cap) ((uint2 *)a.cand)[base + i] = wbuf[wv][i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == __builtin_amdgcn_readfirstlane(lane)) wcnt[wv] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int NQ>
__global__ __launch_bounds__(256) void seed_scan_kernel(SeedScanArgs a)
{
    __shared__ uint32_t bm[kBmWords];
    __shared__ int wcnt[4];
    __shared__ uint2 wbuf[4][kWaveBuf];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 4) wcnt[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < kBmWords; i += 256) bm[i] = a.bitmaps[i];
    __syncthreads();
    const int64_t wblocks = (a.nwindows + 255) / 256;
    const int chunk = (int)(blockIdx.x / wblocks);
    const int64_t w = (int64_t)(blockIdx.x % wblocks) * 256 + threadIdx.x;
    int n = 0, start = 0, c0 = 0;
    const uint8_t *p = a.arena;
    if (w < a.nwindows) {
        const int len = a.win_len[w];
        if (chunk == 0 && len > a.max_len) atomicAdd(a.err, 1u);
        c0 = chunk * a.chunk_len;
        if (c0 < len) {
            start = c0 > a.warm ? c0 - a.warm : 0;
            const int end = (c0 + a.chunk_len < len) ? c0 + a.chunk_len : len;
            p = a.arena + a.win_off[w] + start;
            // Whole aligned 128-byte LINES: start at the line boundary below the chunk.  The extra bytes are scanned like
            // any others (they can only add finds, and a find is kept only where its q-gram lies inside window and chunk).
            const int skip = (int)((uintptr_t)p & 127u);
            p -= skip; start -= skip;
            n = end - start;
        }
    }
    // (lanes without a chunk stay: the wave's last flush below is wave-wide)
    // A base's 2-bit code is bits 1-2 of its ASCII byte (A/a 0, C/c 1, T/t/U/u 2, G/g 3: seed_code below): one v_bfe_u32
    // straight from the fetched dword, no table.  Any other byte gets whatever code its bits spell: that can only ADD
    // candidates (the verifier looks at the real bytes and drops them); a true seed occurrence consists of bases and is
    // always found.  Likewise the register starts as "AAAAAAAA" and the warm-up columns before a chunk may re-find the
    // previous chunk's last seeds.
    const int qb[3] = {2 * a.q[0], 2 * a.q[1], 2 * a.q[2]};
    constexpr int boff[3] = {kBmOff0, kBmOff1, kBmOff2};
    uint32_t x = 0;
    // A lane takes its stream a whole cache line (128 columns) at a time, the next line in flight meanwhile.  Sixteen
    // bytes per visit -- eight visits per line, microseconds apart, by two thousand lanes per CU -- made every visit
    // after the first an L2 miss served by the Infinity Cache: 8x the traffic behind the L2 and 0.85 TB/s of useful
    // bytes; with one visit per line the L2 sees each line once.
    const int nlines = (n + 127) >> 7;               // 0 for a lane without a chunk
    const int nl_wave = __builtin_amdgcn_readfirstlane(wave_max(nlines));
    u32x4 r[8], nx[8];
    // (only the 16-byte blocks that hold columns of the chunk are fetched: never more than 15 bytes past a window's end)
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i] = u32x4{0, 0, 0, 0}; if (16 * i < n) r[i] = *(const u32x4 *)(p + 16 * i); }
    for (int L = 0; L < nl_wave; ++L) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { nx[i] = r[i]; if (128 * (L + 1) + 16 * i < n) nx[i] = *(const u32x4 *)(p + 128 * (L + 1) + 16 * i); }
        uint32_t hb[NQ][8];                          // per seed length and 16-column block: bit 15 - t = a seed ends at column t
        uint32_t any = 0;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            const uint32_t wd[4] = {r[blk].x, r[blk].y, r[blk].z, r[blk].w};
#pragma unroll
            for (int c = 0; c < NQ; ++c) hb[c][blk] = 0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                x = (x << 2) | __builtin_amdgcn_ubfe(wd[t >> 2], 8 * (t & 3) + 1, 2);
#pragma unroll
                for (int c = 0; c < NQ; ++c) {
                    const uint32_t word = bm[boff[c] + __builtin_amdgcn_ubfe(x, 5, qb[c] - 5)];
                    hb[c][blk] = (hb[c][blk] << 1) | __builtin_amdgcn_ubfe(word, x & 31u, 1);
                }
            }
#pragma unroll
            for (int c = 0; c < NQ; ++c) any |= hb[c][blk];
        }
        if (any && L < nlines) {                     // a few lines per hundred per lane
            const int left = n - 128 * L;            // columns of this line that belong to the chunk
#pragma unroll 1
            for (int cb = 0; cb < NQ * 8; ++cb) {
                const int c = cb >> 3, blk = cb & 7;
                uint32_t h = 0;
#pragma unroll
                for (int cc = 0; cc < NQ; ++cc)
#pragma unroll
                    for (int bb = 0; bb < 8; ++bb) h = (cc == c && bb == blk) ? hb[cc][bb] : h;
                while (h) {
                    const int t = 15 - (31 - __builtin_clz(h));
                    h &= ~(1u << (15 - t));
                    const int col = 16 * blk + t;
                    const int j = start + 128 * L + col;           // window coordinate of the seed's last base
                    if (col < left && j >= c0 && j >= a.q[c] - 1) {  // (warm-up columns belong to the previous chunk)
                        const uint2 cd = make_uint2((uint32_t)w, (uint32_t)j | ((uint32_t)c << 28));
                        const int slot = atomicAdd(&wcnt[wv], 1);  // the wave's own buffer in LDS
                        if (slot < kWaveBuf) {
                            wbuf[wv][slot] = cd;
                        } else {                                   // (more than the buffer holds: straight to the list)
                            const unsigned long long g = atomicAdd(a.count, 1ull);
                            if (g < (unsigned long long)a.cap) ((uint2 *)a.cand)[g] = cd;
                        }
                    }
                }
            }
        }
        // The finds of a wave are handed to the global list 64 and more at a time: ONE atomic on the list's counter per
        // flush.  (One atomic per find -- seven million of them on one address -- took 50 ms; the scan itself takes 2.)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (wcnt[wv] >= kWaveBuf / 2) flush_wave(a, wcnt, wbuf, wv, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = nx[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    flush_wave(a, wcnt, wbuf, wv, lane);
}

// One lane per candidate: every piece that has the seed (class c, q-gram value) at some offset is verified over the read
// columns an occurrence through that seed can span.
__global__ __launch_bounds__(256) void seed_verify_kernel(SeedVerifyArgs a)
{
    // (the launch may cover the whole candidate list's capacity -- pc_prefilter_defer_count: blocks beyond the count leave first)
    if ((unsigned long long)blockIdx.x * 256 >= *a.count) return;
    extern __shared__ uint32_t eq_tab[];             // [npieces][8]: Eq word of codes 0..4 (5..7 unused)
    for (int i = threadIdx.x; i < a.npieces * 8; i += 256) eq_tab[i] = a.piece_eq[i];
    __shared__ uint8_t code_of[256];
    for (int i = threadIdx.x; i < 256; i += 256) {
        uint8_t v = 4;
        switch (i) {
            case 'A': case 'a': v = 0; break;
            case 'C': case 'c': v = 1; break;
            case 'G': case 'g': v = 2; break;
            case 'T': case 't': case 'U': case 'u': v = 3; break;
            default: break;
        }
        code_of[i] = v;
    }
    __syncthreads();
    const unsigned long long total = *a.count < (unsigned long long)a.cap ? *a.count : (unsigned long long)a.cap;
    const unsigned long long ci = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (ci >= total) return;
    const uint2 cd = ((const uint2 *)a.cand)[ci];
    const int64_t w = (int64_t)cd.x;
    const int j = (int)(cd.y & 0x0FFFFFFFu);          // window column of the seed's last base
    const int cls = (int)(cd.y >> 28);
    const int q = a.q[cls];
    const int n = a.win_len[w];
    const uint8_t *rd = a.arena + a.win_off[w];
    uint32_t *mrow = a.mask + w * a.words;
    // the q-gram from the read's real bytes; a byte that is not a base means the scan's find was not a seed
    uint32_t idx = 0;
    for (int col = j - q + 1; col <= j; ++col) {
        const uint32_t byte = rd[col];
        if (code_of[byte] > 3u) return;
        idx = (idx << 2) | ((byte >> 1) & 3u);       // the scan's code (seed_code)
    }
    const uint32_t e0 = a.first[a.first_off[cls] + idx], e1 = a.first[a.first_off[cls] + idx + 1];
    for (uint32_t e = e0; e < e1; ++e) {
        const int4 en = ((const int4 *)a.entries)[e];                 // piece, offset of the seed in the piece, -, -
        const int pi = en.x, off = en.y;
        const int4 pm = ((const int4 *)a.piece_meta)[pi];             // len, k, mask word, mask bit
        const int len = pm.x, k = pm.y;
        if (mrow[pm.z] & (uint32_t)pm.w) continue;    // this pair is already marked
        // the seed's first base is at column j - q + 1 and at piece offset off: the occurrence starts within k columns
        // of  j - q + 1 - off  and ends within k columns of that + len
        int lo = j - q + 1 - off - k, hi = j - q + 1 - off + len + k;
        lo = lo < 0 ? 0 : lo; hi = hi > n ? n : hi;
        uint32_t pv = len >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> len), mv = 0, sc = (uint32_t)len, mn = (uint32_t)len;
        const uint32_t *eqp = eq_tab + pi * 8;
        // sixteen columns per fetch (four dwords at any alignment; up to 15 bytes past the window are readable): one
        // memory round trip per sixteen columns instead of one per column
        for (int base = lo; base < hi; base += 16) {
            uint32_t wd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wd[i] = *(const u32_unaligned *)(rd + base + 4 * i);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (base + t < hi) {
                    sc = myers_step(eqp[code_of[(wd[t >> 2] >> (8 * (t & 3))) & 0xFFu]], pv, mv, sc);
                    mn = sc < mn ? sc : mn;
                }
            }
        }
        if ((int)mn <= k) atomicOr(mrow + pm.z, (uint32_t)pm.w);
    }
}

// ---------------------------------------------------------------------------------------------------------
// The seed stage over reads held at 2 bits per base (north_star: "2-bit-packed read windows"): a quarter of the bytes of
// the one HBM-bound kernel of the path, and no byte -> code step at all -- sixteen bases ARE a dword.  A lane takes its
// stream a 128-byte line (512 bases) at a time; the q-gram that ends at base t of a dword is one funnel shift of
// (previous dword, this dword) by a compile-time amount, its low 5 bits pick the bit and the rest the word of the bitmap:
// shift, bfe, LDS probe, bfe, shift-or = 4 VALU operations and one probe per base and seed length.
// (Tried and dropped: a copy of the bitmap per LDS bank, 64 KB, so that no probe ever conflicts -- 2.88 instead of 2.21 ms
// for 8 Gbase: the probes' latency is hidden by resident waves, and 64 KB of LDS leaves two per SIMD instead of five.)
// ---------------------------------------------------------------------------------------------------------
// Q0 >= Q1 >= Q2 are the seed lengths present (0 = none): one bitmap each, in the slots the host lays them out in
// (kBmOff0 / 1 / 2, longest first).  A batch of few adapters has ONE length (pc_api.cpp); a barcode panel keeps its lengths apart.
template <int Q0, int Q1, int Q2>
__global__ __launch_bounds__(256) void seed_scan_packed_kernel(SeedScanArgs a)
{
    constexpr int NQ = 1 + (Q1 > 0) + (Q2 > 0);
    constexpr int QS[3] = {Q0, Q1 > 0 ? Q1 : 6, Q2 > 0 ? Q2 : 6};
    constexpr int BOFF[3] = {kBmOff0, kBmOff1, kBmOff2};
    constexpr int BMW = NQ == 1 ? (1 << (2 * Q0)) / 32 : kBmWords;
    __shared__ uint32_t bm[BMW];
    __shared__ int wcnt[4];
    __shared__ uint2 wbuf[4][kWaveBuf];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 4) wcnt[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < BMW; i += 256) bm[i] = a.bitmaps[i];
    __syncthreads();
    const uint32_t *plane = (const uint32_t *)a.arena;
    const int64_t wblocks = (a.nwindows + 255) / 256;
    const int chunk = (int)(blockIdx.x / wblocks);
    const int64_t w = (int64_t)(blockIdx.x % wblocks) * 256 + threadIdx.x;
    int n = 0, start = 0, c0 = 0;
    const uint32_t *p = plane;
    if (w < a.nwindows) {
        const int len = a.win_len[w];
        if (chunk == 0 && len > a.max_len) atomicAdd(a.err, 1u);
        c0 = chunk * a.chunk_len;
        if (c0 < len) {
            start = c0 > a.warm ? c0 - a.warm : 0;
            const int end = (c0 + a.chunk_len < len) ? c0 + a.chunk_len : len;
            // whole aligned 128-byte lines of the plane: start at the line boundary (512 bases) below the chunk; the bases
            // before the window are scanned like any others (a find is kept only where its q-gram lies inside window and chunk)
            const int64_t b = a.win_off[w] + start;
            const int skip = (int)(b & 511);
            p = plane + ((b - skip) >> 4);
            start -= skip;
            n = end - start;
        }
    }
    const int nlines = (n + 511) >> 9;               // 0 for a lane without a chunk
    const int nl_wave = __builtin_amdgcn_readfirstlane(wave_max(nlines));
    u32x4 r[8], nx[8];
    // (only the 16-byte blocks that hold bases of the chunk are fetched: never more than 63 bases past a window's end)
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i] = u32x4{0, 0, 0, 0}; if (64 * i < n) r[i] = *(const u32x4 *)(p + 4 * i); }
    uint32_t prev = 0;                               // the dword before the one being scanned ("AAAA..." before the first)
    for (int L = 0; L < nl_wave; ++L) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { nx[i] = r[i]; if (512 * (L + 1) + 64 * i < n) nx[i] = *(const u32x4 *)(p + 32 * (L + 1) + 4 * i); }
#pragma unroll
        for (int half = 0; half < 4; ++half) {       // a quarter line: 8 dwords = 128 bases
            uint32_t hb[NQ][8];                      // per seed length and dword: bit 15 - t = a seed ends at its base t
            uint32_t any = 0;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const u32x4 v = r[2 * half + (d >> 2)];
                const uint32_t cur = (d & 3) == 0 ? v.x : (d & 3) == 1 ? v.y : (d & 3) == 2 ? v.z : v.w;
                uint32_t h[NQ];
#pragma unroll
                for (int c = 0; c < NQ; ++c) h[c] = 0;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
#pragma unroll
                    for (int c = 0; c < NQ; ++c) {
                        // the QS[c] bases ending at base t of `cur`, first base in the lowest bits: bits [st, st + 2 q) of prev:cur
                        const int st = 32 + 2 * (t + 1) - 2 * QS[c];
                        const uint32_t g = st < 32 ? __builtin_amdgcn_alignbit(cur, prev, st) : (cur >> (st - 32));
                        const uint32_t word = bm[(NQ == 1 ? 0 : BOFF[c]) + __builtin_amdgcn_ubfe(g, 5, 2 * QS[c] - 5)];
                        h[c] = (h[c] << 1) | __builtin_amdgcn_ubfe(word, g, 1);   // (the offset operand is taken modulo 32)
                    }
                }
#pragma unroll
                for (int c = 0; c < NQ; ++c) { hb[c][d] = h[c]; any |= h[c]; }
                prev = cur;
            }
            if (any && L < nlines) {
                const int base0 = 512 * L + 128 * half;  // stream position of this quarter line's first base
#pragma unroll 1
                for (int cd_ = 0; cd_ < NQ * 8; ++cd_) {
                    const int c = cd_ >> 3, d = cd_ & 7;
                    uint32_t h = 0;
#pragma unroll
                    for (int cc = 0; cc < NQ; ++cc)
#pragma unroll
                        for (int dd = 0; dd < 8; ++dd) h = (cc == c && dd == d) ? hb[cc][dd] : h;
                    const int qc = c == 0 ? QS[0] : c == 1 ? QS[1] : QS[2];
                    while (h) {
                        const int t = 15 - (31 - __builtin_clz(h));
                        h &= ~(1u << (15 - t));
                        const int pos = base0 + 16 * d + t;          // stream position of the seed's last base
                        const int j = start + pos;                     // window coordinate
                        if (pos < n && j >= c0 && j >= qc - 1) {       // (warm-up columns belong to the previous chunk)
                            const uint2 cd = make_uint2((uint32_t)w, (uint32_t)j | ((uint32_t)c << 28));
                            const int slot = atomicAdd(&wcnt[wv], 1);
                            if (slot < kWaveBuf) {
                                wbuf[wv][slot] = cd;
                            } else {
                                const unsigned long long g2 = atomicAdd(a.count, 1ull);
                                if (g2 < (unsigned long long)a.cap) ((uint2 *)a.cand)[g2] = cd;
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (wcnt[wv] >= kWaveBuf / 2) flush_wave(a, wcnt, wbuf, wv, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = nx[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    flush_wave(a, wcnt, wbuf, wv, lane);
}

// One lane per candidate, bases from the plane: see seed_verify_kernel.
__global__ __launch_bounds__(256) void seed_verify_packed_kernel(SeedVerifyArgs a)
{
    if ((unsigned long long)blockIdx.x * 256 >= *a.count) return;      // (see seed_verify_kernel)
    extern __shared__ uint32_t eq_tab[];             // [npieces][8]: Eq word of codes 0..3 (the plane holds nothing else)
    for (int i = threadIdx.x; i < a.npieces * 8; i += 256) eq_tab[i] = a.piece_eq[i];
    __syncthreads();
    const unsigned long long total = *a.count < (unsigned long long)a.cap ? *a.count : (unsigned long long)a.cap;
    const unsigned long long ci = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (ci >= total) return;
    const uint2 cd = ((const uint2 *)a.cand)[ci];
    const int64_t w = (int64_t)cd.x;
    const int j = (int)(cd.y & 0x0FFFFFFFu);          // window column of the seed's last base
    const int cls = (int)(cd.y >> 28);
    const int q = a.q[cls];
    const int n = a.win_len[w];
    const uint32_t *plane = (const uint32_t *)a.arena;
    const int64_t b0 = a.win_off[w];                  // base index of the window's column 0
    auto code_at = [&](int col) -> uint32_t { const int64_t b = b0 + col; return (plane[b >> 4] >> (2 * (int)(b & 15))) & 3u; };
    uint32_t *mrow = a.mask + w * a.words;
    uint32_t idx = 0;                                 // little-endian q-gram: first base in the lowest bits
    for (int t = 0; t < q; ++t) idx |= code_at(j - q + 1 + t) << (2 * t);
    const uint32_t e0 = a.first[a.first_off[cls] + idx], e1 = a.first[a.first_off[cls] + idx + 1];
    for (uint32_t e = e0; e < e1; ++e) {
        const int4 en = ((const int4 *)a.entries)[e];
        const int pi = en.x, off = en.y;
        const int4 pm = ((const int4 *)a.piece_meta)[pi];
        const int len = pm.x, k = pm.y;
        if (mrow[pm.z] & (uint32_t)pm.w) continue;
        int lo = j - q + 1 - off - k, hi = j - q + 1 - off + len + k;
        lo = lo < 0 ? 0 : lo; hi = hi > n ? n : hi;
        uint32_t pv = len >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> len), mv = 0, sc = (uint32_t)len, mn = (uint32_t)len;
        const uint32_t *eqp = eq_tab + pi * 8;
        // a dword of the plane (16 bases) per fetch
        int col = lo;
        while (col < hi) {
            const int64_t b = b0 + col;
            const uint32_t wd = plane[b >> 4];
            const int t0 = (int)(b & 15);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (t >= t0 && col + (t - t0) < hi) {
                    sc = myers_step(eqp[(wd >> (2 * t)) & 3u], pv, mv, sc);
                    mn = sc < mn ? sc : mn;
                }
            }
            col += 16 - t0;
        }
        if ((int)mn <= k) atomicOr(mrow + pm.z, (uint32_t)pm.w);
    }
}

// ---- windows of the plane as bytes: one workgroup per window ---------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_windows_kernel(const uint32_t *plane, const int64_t *src_off, const int32_t *len,
                                                             uint8_t *dst, const int64_t *dst_off, int pad)
{
    const int64_t i = blockIdx.x;
    const int64_t b0 = src_off[i];
    uint8_t *d = dst + dst_off[i];
    const int64_t n = len[i], total = dst_off[i + 1] - dst_off[i];
    for (int64_t c = threadIdx.x; c < total; c += blockDim.x) {
        uint8_t v = (uint8_t)pad;
        if (c < n) { const int64_t b = b0 + c; v = (uint8_t)((0x54474341u >> (8 * ((plane[b >> 4] >> (2 * (int)(b & 15))) & 3u))) & 0xFFu); }
        d[c] = v;
    }
}

// one thread per exception: the window that holds it (windows ascending by src_off, not overlapping), if any
__global__ __launch_bounds__(256) void unpack_windows_exceptions_kernel(const int64_t *exc_pos, int64_t nexc, const int64_t *src_off,
                                                                        const int32_t *len, int64_t n, uint8_t *dst, const int64_t *dst_off)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nexc) return;
    const int64_t pos = exc_pos[e];
    int64_t lo = 0, hi = n - 1;
    if (n <= 0 || pos < src_off[0]) return;
    while (lo < hi) {                                 // last window with src_off <= pos
        const int64_t mid = (lo + hi + 1) >> 1;
        if (src_off[mid] <= pos) lo = mid; else hi = mid - 1;
    }
    if (pos - src_off[lo] < (int64_t)len[lo]) dst[dst_off[lo] + (pos - src_off[lo])] = (uint8_t)'N';
}

}  // namespace

int launch_seed_scan_packed(const SeedScanArgs &a, void *stream)
{
    if (a.nwindows <= 0) return 0;
    const int64_t wblocks = (a.nwindows + 255) / 256;
    const int64_t gx = wblocks * a.chunks;
    if (gx > 0x7FFFFFFFll) return -1;
    hipStream_t s = (hipStream_t)stream;
    // the lengths present, longest first (8 >= q[0] > q[1] > q[2] >= 6)
    const int key = a.q[0] * 100 + (a.nq > 1 ? a.q[1] * 10 : 0) + (a.nq > 2 ? a.q[2] : 0);
#define PC_SSP(K, A, B, C) case K: hipLaunchKernelGGL((seed_scan_packed_kernel<A, B, C>), dim3((unsigned)gx), dim3(256), 0, s, a); break;
    switch (key) {
        PC_SSP(600, 6, 0, 0) PC_SSP(700, 7, 0, 0) PC_SSP(800, 8, 0, 0)
        PC_SSP(870, 8, 7, 0) PC_SSP(860, 8, 6, 0) PC_SSP(760, 7, 6, 0) PC_SSP(876, 8, 7, 6)
        default: return -1;
    }
#undef PC_SSP
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_seed_verify_packed(const SeedVerifyArgs &a, int64_t ncand, void *stream)
{
    if (ncand <= 0) return 0;
    const int64_t gx = (ncand + 255) / 256;
    if (gx > 0x7FFFFFFFll) return -1;
    hipLaunchKernelGGL(seed_verify_packed_kernel, dim3((unsigned)gx), dim3(256), (size_t)a.npieces * 32, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_unpack_windows(const void *plane, const int64_t *exc_pos, int64_t nexc, const int64_t *src_off, const int32_t *len,
                          int64_t n, uint8_t *dst, const int64_t *dst_off, int pad, void *stream)
{
    if (n <= 0) return 0;
    if (n > 0x7FFFFFFFll) return -1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(unpack_windows_kernel, dim3((unsigned)n), dim3(256), 0, s, (const uint32_t *)plane, src_off, len, dst, dst_off, pad);
    if (nexc > 0)
        hipLaunchKernelGGL(unpack_windows_exceptions_kernel, dim3((unsigned)((nexc + 255) / 256)), dim3(256), 0, s, exc_pos, nexc, src_off,
                           len, n, dst, dst_off);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_seed_scan(const SeedScanArgs &a, void *stream)
{
    if (a.nwindows <= 0) return 0;
    const int64_t wblocks = (a.nwindows + 255) / 256;
    const int64_t gx = wblocks * a.chunks;
    if (gx > 0x7FFFFFFFll) return -1;
    hipStream_t s = (hipStream_t)stream;
    switch (a.nq) {
        case 1: hipLaunchKernelGGL(seed_scan_kernel<1>, dim3((unsigned)gx), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(seed_scan_kernel<2>, dim3((unsigned)gx), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(seed_scan_kernel<3>, dim3((unsigned)gx), dim3(256), 0, s, a); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_seed_verify(const SeedVerifyArgs &a, int64_t ncand, void *stream)
{
    if (ncand <= 0) return 0;
    const int64_t gx = (ncand + 255) / 256;
    if (gx > 0x7FFFFFFFll) return -1;
    hipLaunchKernelGGL(seed_verify_kernel, dim3((unsigned)gx), dim3(256), (size_t)a.npieces * 32, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

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
