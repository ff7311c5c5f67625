// pc_kernels.hip -- the adapter-alignment hot path as hand-written HIP for gfx950 (CDNA4).
//
// What is computed is exactly the reference's path:
//   Gotoh affine-gap DP with all four end gaps free, SeqAn's tie-breaking
//   (seqan/align/dp_formula_affine.h:456-495), its max search over last row / last column
//   (dp_scout.h:165-179), its traceback (dp_traceback_impl.h:376-552) and Porechop's
//   ScoredAlignment digest (porechop/src/alignment.cpp:6-111).
// How it is computed is MI355X-first and shares nothing with the reference's structure:
//
//   * inter-pair SIMD: one LANE owns two (window, adapter) pairs, packed in the low/high int16
//     halves of every VGPR, so a wavefront advances 128 independent alignments with zero
//     cross-lane traffic, no ramp-up/down and every lane busy (adapters are 22..50 rows, so an
//     anti-diagonal-per-wave mapping would idle most of a 64-wide wave);
//   * the whole DP column (M+open and H per adapter row) lives in VGPRs, rows fully unrolled;
//     the adapter is uniform per half-wave, so its bases are SGPR (or LDS-broadcast) operands;
//   * all cell arithmetic is v_pk_*_i16 (2 cells per VALU op): 11 ops per cell pair for the
//     score recurrence, +12 for the 4 trace bits in the tracing variant;
//   * the 4-bit trace is written to a per-wave slab with fully coalesced 256-B stores and read
//     back only along the path; whole-read scans never store trace for the whole read: a
//     score-only pass finds the end cell, a bounded window is re-run with trace (pc_bounds.h);
//   * read bytes are consumed as delivered (1 B/base ASCII); byte -> Dna5 code is a 512-B LDS
//     table lookup per column.
//
// No MFMA: this is integer max-plus DP, not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pc_bounds.h"
#include "pc_kernels.h"
#include "pc_walk.h"

namespace pck {

typedef uint32_t u32;

// ---- packed int16 primitives.  Plain ext-vector C++ compiles 1:1 to v_pk_*_i16/u16 on gfx950.
// The trace-bit idiom min_u16(x - y, 1) must use an OPAQUE 1 (a kernel argument): with a literal
// hipcc rewrites it into v_cmp/v_cndmask/v_perm chains that cost 2.5x more.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 SV(u32 x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ u16x2 UV(u32 x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ u32 WV(s16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 WV(u16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return WV(SV(a) + SV(b)); }
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b) { return WV(SV(a) - SV(b)); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return WV(__builtin_elementwise_max(SV(a), SV(b))); }
__device__ __forceinline__ u32 pk_minu(u32 a, u32 b) { return WV(__builtin_elementwise_min(UV(a), UV(b))); }
__device__ __forceinline__ u32 pk_madu(u32 a, u32 k, u32 c) { return WV(UV(a) * UV(k) + UV(c)); }

__device__ __forceinline__ u32 pack2(int v) { return ((u32)v & 0xFFFFu) | ((u32)v << 16); }
__device__ __forceinline__ int lo16(u32 x) { return (int)(short)(x & 0xFFFFu); }
__device__ __forceinline__ int hi16(u32 x) { return (int)(short)(x >> 16); }

__device__ __forceinline__ int dna5_code(int c) {
    // seqan/basic/alphabet_residue_tabs.h:113-140
    c &= 0xFF;
    if (c == 'A' || c == 'a') return 0;
    if (c == 'C' || c == 'c') return 1;
    if (c == 'G' || c == 'g') return 2;
    if (c == 'T' || c == 't' || c == 'U' || c == 'u') return 3;
    return 4;
}

typedef u32 u32_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ u32 load_u32_unaligned(const uint8_t *p) {
    // gfx9 global loads are byte-addressable at any alignment (SH_MEM_CONFIG unaligned mode)
    return *(const u32_unaligned *)p;
}


// Scout state of one packed half
struct Best { int score, I, J, tie; };

// ---------------------------------------------------------------------------------------------
// One DP column for both halves of every lane, in DRIFTING COORDINATES: a value X of register row
// rho = r+1, jj columns after the window start / the last renormalisation, is held as
// X + (rho + jj) * eps with eps = -gap_extend (T one step ahead: + (rho + jj + 1) * eps).  A gap
// extension moves one row or one column and costs -eps, so it needs no add:
//   H' = max(H, T_left)      V' = max(V_up, T_up)      d = (T_diag + (match - open + eps)) - z
//   M = max(d, H', V')       T' = M + (open + eps)
// 9 packed ops per two cells instead of 11 (plus 12 for the trace bits, which are differences of
// values in the same coordinates and so unchanged).  Row 0 (M = 0) becomes the per-column values
// topT = T(0,j) and topD = T(0,j-1) + (match - open + eps), advanced by eps per column by the caller.
//   T[r] = M[r][j-1] + open      U[r] = H[r][j-1]          (packed lo/hi int16, drifted)
//   h2   = spaced Dna5 codes of the two reads' bases at this column
//   per-row adapter constants: spaced code vc, and the min-constant dm (D=match-mismatch for a
//   real adapter row, `match` for a padding row above the adapter: padding rows then
//   reproduce row 0 exactly -- M stays 0, V re-opens -- see DESIGN.md "top padding")
//   PAD=false: both adapters fill all R rows, dm is the uniform D and vc lives in SGPRs.
//   PAD=true : (vc, dm) come from an LDS broadcast read per row.
// The row loop is software-pipelined by hand: the ops of row r+2 that do not depend on the
// vertical chain are issued between the chain ops of row r, so dependent v_pk ops are never
// adjacent (gfx950 needs a wait state between them) and one wave alone keeps the VALU busy.
// ---------------------------------------------------------------------------------------------
struct KConst { u32 A2, AO2, E2, O2, NEG2, D2, ONE2, TWO2, SIXTEEN2;
                u32 AOE2, OE2, EPS2; };      // drifting coordinates: match-open+eps, open+eps, eps

// where the per-row adapter constants live
enum ConstMode { CONST_EXACT = 0,   // no padding rows: code in SGPRs, min-constant is the uniform D
                 CONST_REGS = 1,    // padded, R <= 40: code in VGPRs (uniform value), min-constant in SGPRs
                 CONST_LDS = 2 };   // padded, larger R: (code, min-constant) by LDS broadcast read per row
template <int R, bool PAD> struct Cfg {
    static constexpr int kMode = !PAD ? CONST_EXACT : (R <= 40 ? CONST_REGS : CONST_LDS);
    static constexpr int kNS = (kMode == CONST_LDS) ? 1 : R;                        // SGPR array extent
    static constexpr int kNV = (kMode == CONST_REGS) ? R : 1;                       // VGPR array extent
};

template <int R, bool PAD, bool TRACE>
__device__ __forceinline__ void column_step(u32 (&T)[R], u32 (&U)[R], u32 h2,
                                            const u32 (&cs)[Cfg<R, PAD>::kNS], const u32 (&cv)[Cfg<R, PAD>::kNV],
                                            const uint2 *lds_const, const KConst &k, const u32 topD, const u32 topT,
                                            u32 (&trw)[(R + 3) / 4], u32 &last_tie01)
{
    constexpr int MODE = Cfg<R, PAD>::kMode;
    constexpr int K = 2;                    // pipeline depth (rows ahead)
    u32 dd[R], Hh[R], dh[R], b0s[TRACE ? R : 1];   // only a window of K+1 entries is ever live
    u32 dq = topD;                          // row 0's diagonal term (M = 0 there)
    auto ind = [&](int r) {
        u32 vc, dm;
        if constexpr (MODE == CONST_LDS) {
            asm volatile("" : "+s"(lds_const));     // pin this LDS-broadcast read to its row
            const uint2 c = lds_const[r]; vc = c.x; dm = c.y;
        } else if constexpr (MODE == CONST_REGS) { vc = cv[r]; dm = cs[r]; }
        else { vc = cs[r]; dm = k.D2; }
        const u32 y = pk_sub(h2, vc);
        const u32 z = pk_minu(y, dm);
        const u32 Hs = pk_max(U[r], T[r]);
        const u32 d = pk_sub(dq, z);
        dq = pk_add(T[r], k.AOE2);          // diagonal term of row r+1, from the OLD T[r]
        dd[r] = d; Hh[r] = Hs;
        dh[r] = pk_max(d, Hs);              // off the vertical chain: M = max(max(d,H), V)
        if constexpr (TRACE) b0s[r] = pk_minu(pk_sub(Hs, U[r]), k.ONE2);   // HOPEN
    };
#pragma clang loop unroll(full)
    for (int r = 0; r < K && r < R; ++r) ind(r);
    u32 Tup = topT, Vprev = k.NEG2, acc = 0;
#pragma clang loop unroll(full)
    for (int r = 0; r < R; ++r) {
        if (r + K < R) {
            // data-dependence fence: row r+K's independent ops may not start before the vertical
            // chain has reached row r (keeps live ranges to a K+1 row window at every level of
            // the compiler, which a sched_barrier alone does not)
            asm volatile("" : "+v"(U[r + K]), "+v"(T[r + K]), "+v"(Vprev));
            ind(r + K);
        }
        // the vertical chain: 3 dependent ops per row (Vs, M, T')
        const u32 Vs = pk_max(Vprev, Tup);
        const u32 Mn = pk_max(dh[r], Vs);
        const u32 Tn = pk_add(Mn, k.OE2);
        if constexpr (TRACE) {
            const u32 g = pk_max(Hh[r], Vs);
            const u32 b1 = pk_minu(pk_sub(Vs, Vprev), k.ONE2);   // VOPEN
            const u32 b2 = pk_minu(pk_sub(g, Vs), k.ONE2);       // FROMH
            const u32 b3 = pk_minu(pk_sub(Mn, dd[r]), k.ONE2);   // NOTDIAG
            u32 nib = pk_madu(b3, k.TWO2, b2);
            nib = pk_madu(nib, k.TWO2, b1);
            nib = pk_madu(nib, k.TWO2, b0s[r]);
            acc = pk_madu(acc, k.SIXTEEN2, nib);
            if ((r & 3) == 3 || r == R - 1) { trw[r >> 2] = acc; acc = 0; }
            if (r == R - 1) last_tie01 = pk_minu(dd[r] ^ g, k.ONE2);   // 0 where d == max(H,V)
        } else {
            if (r == R - 1) last_tie01 = pk_minu(dd[r] ^ pk_max(Hh[r], Vs), k.ONE2);
        }
        T[r] = Tn; U[r] = Hh[r];
        Tup = Tn; Vprev = Vs;
        if constexpr (TRACE) {
            // same kind of fence for the trace bits: they must retire inside this row's window
            // instead of being sunk below the whole column (which keeps 6 values per row alive)
            if ((r & 3) == 3 || r == R - 1) asm volatile("" : "+v"(trw[r >> 2]), "+v"(Tup));
            else asm volatile("" : "+v"(acc), "+v"(Tup));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Traceback + digest of a lane's two pairs from the trace slab, SIDE BY SIDE: each walk is a chain
// of dependent loads (trace word -> next cell), so the two are advanced in one loop and their loads
// are in flight together (pc_walk.h: Walk::consume takes one cell's nibble).  Shared by the int16
// and the fp16 traced kernels (same slab layout: [column][word][lane], 16 bits per 4 rows per half,
// first row of a group in the highest nibble).
// ---------------------------------------------------------------------------------------------
// COUNT (the range-checking builds): the matches are also counted from the bases on the walk's diagonal steps and must
// equal the count pc_walk.h derives from the score -- with match - mismatch = 1 a wrong end-cell score would otherwise turn
// into a wrong identity without tripping finish()'s divisibility check.
// GROUPED (the fp16 kernel's slab): a column's words lie in groups of four, [column][group][lane][word of the group] -- a lane's
// four words are 16 contiguous bytes, written by ONE buffer_store_dwordx4 (a last group of 1..3 words by a narrower store);
// the column pitch is NW x 64 dwords either way.
template <bool COUNT = false, bool GROUPED = false>
__device__ __forceinline__ void traceback_pairs(const ScanArgs &a, const u32 *slab, const int rows, const int NW, const int lane,
                                                const Best b_lo, const Best b_hi, const int pad_lo, const int pad_hi,
                                                const bool have_lo, const bool have_hi, const int n_lo, const int n_hi,
                                                const int c0_lo, const int c0_hi, const int m_lo, const int m_hi,
                                                const int64_t p_lo, const int64_t p_hi, const int notrace_upto,
                                                const uint8_t *w_lo = nullptr, const uint8_t *w_hi = nullptr,
                                                const int ad_lo = 0, const int ad_hi = 0)
{
    // end-aligned windows: a negative col0 is the lead-in before the read's column 0
    const int cmin_lo = c0_lo < 0 ? -c0_lo : 0, cmin_hi = c0_hi < 0 ? -c0_hi : 0;
    // trace nibble of cell (col, adapter row) of half hf; the bases themselves are never needed
    // (pc_walk.h derives the match count from the score)
    auto fetch = [&](int hf, int pad, int col, int row) -> int {
        const int r = pad + row - 1;
        const int wq = r >> 2;
        const int rows_in_group = (rows - 4 * wq) < 4 ? (rows - 4 * wq) : 4;
        const int pos = rows_in_group - 1 - (r & 3);
        int64_t at;
        if constexpr (GROUPED) {
            const int g = wq >> 2, kg = (NW - 4 * g) < 4 ? (NW - 4 * g) : 4;
            at = (int64_t)(col - 1) * NW * 64 + g * 256 + lane * kg + (wq & 3);
        } else {
            at = ((int64_t)(col - 1) * NW + wq) * 64 + lane;
        }
        const u32 dw = slab[at];
        return (int)((dw >> (16 * hf + 4 * pos)) & 0xFu);
    };
    pcw::Walk wk_lo, wk_hi;
    const int nt_lo = a.n_total ? (have_lo ? a.n_total[p_lo] : 0) : n_lo;
    const int nt_hi = a.n_total ? (have_hi ? a.n_total[p_hi] : 0) : n_hi;
    // _correctTraceValue needs the end cell's nibble before the walk starts
    auto tie_fix_of = [&](int hf, bool have, const Best &b, int pad) -> int {
        if (!have || !(b.J > 0 && b.I > 0) || a.linear) return 0;
        const int nb = fetch(hf, pad, b.J, b.I);
        return ((nb & pcw::NIB_NOTDIAG) || b.tie) ? ((nb & pcw::NIB_FROMH) ? 2 : 1) : 0;
    };
    const int tiefix_lo = tie_fix_of(0, have_lo, b_lo, pad_lo);
    const int tiefix_hi = tie_fix_of(1, have_hi, b_hi, pad_hi);
    wk_lo.start(b_lo.I, b_lo.J, m_lo, c0_lo, nt_lo, b_lo.score, tiefix_lo, cmin_lo);
    wk_hi.start(b_hi.I, b_hi.J, m_hi, c0_hi, nt_hi, b_hi.score, tiefix_hi, cmin_hi);
    if (!have_lo) wk_lo.done = 1;
    if (!have_hi) wk_hi.done = 1;
    int left_trace = 0;
    int seen_lo = 0, seen_hi = 0;                      // COUNT: matches counted from the bases
    // (a path has at most rows + columns steps: the cap turns a corrupted trace into a reported error, not a hang)
    int steps_left = 2 * (rows + (n_lo > n_hi ? n_lo : n_hi)) + 8;
    while (!(wk_lo.done & wk_hi.done)) {
        if (--steps_left < 0) { left_trace = 1; break; }
        // a walk that leaves the traced columns of a pass-2 window is stopped and flagged (never
        // expected: the bound of pc_bounds.h).  Both walks fetch every round -- a finished one from
        // a clamped, valid cell -- and step under a predicate: no divergent branches in the loop.
        int go_lo = wk_lo.done ^ 1, go_hi = wk_hi.done ^ 1;
        const int out_lo = go_lo & (wk_lo.col <= notrace_upto ? 1 : 0), out_hi = go_hi & (wk_hi.col <= notrace_upto ? 1 : 0);
        left_trace |= out_lo | out_hi;
        wk_lo.done |= out_lo; wk_hi.done |= out_hi;
        go_lo &= out_lo ^ 1; go_hi &= out_hi ^ 1;
        const int nb_lo = fetch(0, pad_lo, wk_lo.col > 1 ? wk_lo.col : 1, wk_lo.row > 1 ? wk_lo.row : 1);
        const int nb_hi = fetch(1, pad_hi, wk_hi.col > 1 ? wk_hi.col : 1, wk_hi.row > 1 ? wk_hi.row : 1);
        (void)cmin_lo; (void)cmin_hi;
        if constexpr (COUNT) {
            const int d_lo = wk_lo.ndiag, d_hi = wk_hi.ndiag;
            const int cl = wk_lo.col, rl = wk_lo.row, ch = wk_hi.col, rh = wk_hi.row;
            wk_lo.step(nb_lo, go_lo);
            wk_hi.step(nb_hi, go_hi);
            if (wk_lo.ndiag != d_lo) seen_lo += dna5_code(w_lo[cl - 1]) == (int)a.ad_codes[ad_lo * 128 + rl - 1] ? 1 : 0;
            if (wk_hi.ndiag != d_hi) seen_hi += dna5_code(w_hi[ch - 1]) == (int)a.ad_codes[ad_hi * 128 + rh - 1] ? 1 : 0;
        } else {
            wk_lo.step(nb_lo, go_lo);
            wk_hi.step(nb_hi, go_hi);
        }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const bool have = hf ? have_hi : have_lo;
        if (!have) continue;
        pcw::Walk &wk = hf ? wk_hi : wk_lo;
        const Best b = hf ? b_hi : b_lo;
        const int64_t p = hf ? p_hi : p_lo;
        pcw::Digest dg;
        int err = wk.finish(dg, a.match, a.mismatch, a.gap_open, a.init_extend);
        if (a.force_score && a.force_score[p] != b.score) err = 1;
        if (left_trace) err = 1;
        if constexpr (COUNT) { if (dg.matches != (hf ? seen_hi : seen_lo)) err = 1; }
        if (err) atomicAdd(a.err, 1u);
        int4 o0 = {dg.read_start, dg.read_end, dg.adapter_start, dg.adapter_end};
        int4 o1 = {dg.score, dg.matches, dg.aligned_len, dg.full_len};
        int4 *op = (int4 *)(a.out + p * TRACE_OUT_INTS);
        op[0] = o0; op[1] = o1;
    }
}

// ---------------------------------------------------------------------------------------------
// The scan kernel.  TRACE=true : full alignment (trace slab, traceback, digest) -> 8 ints/pair
//                   TRACE=false: score-only forward pass -> (score, I, J) per pair
// R > 0 : DP column in VGPRs, R rows fully unrolled (the fast path; PAD as in column_step).
// R == 0: "generic" variant for adapters longer than the register variants: the column lives in
//         LDS ([row][lane], conflict-free), rows = tile.rows at run time, rolled row loop.
// One wavefront per block; blocks stride over tiles.
// ---------------------------------------------------------------------------------------------
template <bool B> struct BoolTag { static constexpr bool value = B; };

// Register budget: 3 resident waves per SIMD (<= 168 VGPRs) up to 34 rows, 2 (<= 256) up to 72 --
// the traced variants of 33..34 and 68..72 rows land a few registers above that without the hint.
template <int R, bool PAD, bool TRACE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((R > 0 && R <= 34) ? 3 : (R > 0 && R <= 72) ? 2 : 1))) void scan_kernel(ScanArgs a)
{
    constexpr bool GEN = (R == 0);
    constexpr int RS = GEN ? 1 : R;          // static array extent
    __shared__ uint16_t lut[256];
    __shared__ uint2 lds_const_s[RS];        // per-row (spaced code, min-constant), packed lo|hi
    extern __shared__ __attribute__((aligned(16))) uint2 dyn_lds[];   // GEN: consts + column state
    uint2 *lds_const = GEN ? dyn_lds : lds_const_s;
    uint2 *lds_col = dyn_lds + a.gen_max_rows;                        // GEN only: the DP column [row][lane]
    // previous column kept for the (rare) last-column scan of the register variants: global scratch,
    // [lane][row] so that the few lanes that finish at a given column write contiguous bytes
    uint2 *fin = GEN ? nullptr : (uint2 *)a.fin_scratch + (int64_t)blockIdx.x * RS * 64;

    const int lane = threadIdx.x;
    const int D = a.match - a.mismatch;
    for (int c = lane; c < 256; c += 64) lut[c] = (uint16_t)(dna5_code(c) * D);

    KConst k;
    k.A2 = pack2(a.match); k.AO2 = pack2(a.match - a.gap_open); k.E2 = pack2(a.gap_extend);
    k.O2 = pack2(a.gap_open); k.NEG2 = pack2(pcb::NEG16); k.D2 = pack2(D);
    // register variants run in drifting coordinates (column_step); the LDS-state generic variant --
    // linear-gap schemes, schemes whose gap extension is too large to drift -- does not
    const int eps = GEN ? 0 : -a.gap_extend;
    k.AOE2 = pack2(a.match - a.gap_open + eps); k.OE2 = pack2(a.gap_open + eps); k.EPS2 = pack2(eps);
    k.ONE2 = a.one2; k.TWO2 = a.two2; k.SIXTEEN2 = a.sixteen2;   // opaque on purpose
    const u32 NEG2 = k.NEG2;
    u32 *slab = TRACE ? a.slab + (int64_t)blockIdx.x * a.slab_stride : nullptr;

    // score-only whole-read passes may be cut into `chunks` column chunks per tile (virtual
    // tiles) when real tiles are too few to fill the chip: chunk c tracks the last-row cells of
    // columns (c*L, (c+1)*L] after a warm-up of SPAN columns that makes its values exact
    // (pc_bounds.h); the planner kernel then merges the per-chunk maxima in visiting order.
    const int nchunks = (!TRACE && a.chunks > 1) ? a.chunks : 1;
    // units beyond the grid come from a counter in launch order (see pc_spec_score, pc_jit_source.h)
    auto next_unit = [&](int vt) -> int {
        if (TRACE || !a.work_counter) return vt + (int)gridDim.x;
        u32 v = 0;
        if (lane == 0) v = atomicAdd(a.work_counter, 1u);
        return (int)gridDim.x + (int)__builtin_amdgcn_readfirstlane(v);
    };
    const int total_units = (!TRACE && a.unit_prefix) ? a.unit_prefix[a.ntiles] : a.ntiles * nchunks;
    for (int vt = blockIdx.x; vt < total_units; vt = next_unit(vt)) {
        int t, chunk;
        if (!TRACE && a.unit_prefix) {                               // the last tile whose first unit is <= vt (wave-uniform)
            int lo = 0, hi = a.ntiles - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.unit_prefix[mid] <= vt) lo = mid; else hi = mid - 1; }
            t = lo; chunk = vt - a.unit_prefix[lo];
        } else {
            t = vt / nchunks; chunk = vt - t * nchunks;
        }
        const Tile tile = a.tiles[t];
        const int rows = GEN ? tile.rows : R;
        const int NW = (rows + 3) >> 2;      // trace dwords per column per lane
        const int m_lo = a.ad_len[tile.adapter_lo];
        const int m_hi = a.ad_len[tile.adapter_hi];
        const u32 *codes_lo = a.ad_codes + (int64_t)tile.adapter_lo * pcb::MAX_ADAPTER;
        const u32 *codes_hi = a.ad_codes + (int64_t)tile.adapter_hi * pcb::MAX_ADAPTER;
        const int pad_lo = rows - m_lo, pad_hi = rows - m_hi;     // top padding rows of each half
        if (pad_lo < 0 || pad_hi < 0 || (GEN && rows > a.gen_max_rows)) {   // host bug
            if (lane == 0) atomicAdd(a.err, 1u);
            continue;
        }

        // ---- per-row adapter constants -> LDS (-> SGPRs for the exact variants) -----------
        __syncthreads();
        for (int r = lane; r < rows; r += 64) {
            const int il = r - pad_lo, ih = r - pad_hi;
            const u32 rl = codes_lo[il >= 0 ? il : 0], rh = codes_hi[ih >= 0 ? ih : 0];
            const u32 cl = il >= 0 ? rl * (u32)D : 6u * (u32)D;
            const u32 ch = ih >= 0 ? rh * (u32)D : 6u * (u32)D;
            const u32 dl = il >= 0 ? (u32)D : (u32)a.match;
            const u32 dh = ih >= 0 ? (u32)D : (u32)a.match;
            lds_const[r] = make_uint2(cl | (ch << 16), dl | (dh << 16));
        }
        __syncthreads();
        constexpr int MODE = GEN ? (int)CONST_LDS : Cfg<RS, PAD>::kMode;
        u32 cs[GEN ? 1 : Cfg<RS, PAD>::kNS], cv[GEN ? 1 : Cfg<RS, PAD>::kNV];
        cs[0] = 0; cv[0] = 0;
        if constexpr (MODE == CONST_EXACT) {
            if (pad_lo != 0 || pad_hi != 0) { if (lane == 0) atomicAdd(a.err, 1u); continue; }   // host bug
#pragma clang loop unroll(full)
            for (int r = 0; r < RS; ++r) cs[r] = __builtin_amdgcn_readfirstlane(lds_const[r].x);
        } else if constexpr (MODE == CONST_REGS) {
#pragma clang loop unroll(full)
            for (int r = 0; r < RS; ++r) {
                const uint2 c = lds_const[r];
                cv[r] = c.x; cs[r] = __builtin_amdgcn_readfirstlane(c.y);
            }
        }

        // ---- this lane's two pairs -----------------------------------------------------
        const int64_t p_lo = tile.out_lo + lane, p_hi = tile.out_hi + lane;            // output slots
        const int64_t wi_lo = a.win_by_out ? p_lo : tile.win_lo + lane;                  // window slots
        const int64_t wi_hi = a.win_by_out ? p_hi : tile.win_hi + lane;
        const bool have_lo = lane < tile.count_lo, have_hi = lane < tile.count_hi;
        const bool one_stream = !a.win_by_out && tile.win_lo == tile.win_hi;             // same windows, two adapters
        const uint8_t *w_lo = a.arena + (have_lo ? a.win_off[wi_lo] : 0);
        const uint8_t *w_hi = a.arena + (have_hi ? a.win_off[wi_hi] : 0);
        int n_lo = have_lo ? a.win_len[wi_lo] : 0;
        int n_hi = have_hi ? a.win_len[wi_hi] : 0;
        int c0_lo = (have_lo && a.col0) ? a.col0[p_lo] : 0;
        int c0_hi = (have_hi && a.col0) ? a.col0[p_hi] : 0;
        // chunked pass: [tf+1 .. n] are the tracked local columns, tail = this chunk ends the read
        int tf_lo = 0, tf_hi = 0;
        bool tail_lo = true, tail_hi = true;
        if (nchunks > 1) {
            const int L = a.chunk_len, start = chunk * L;
            auto cut = [&](int nfull, int span, const uint8_t *&w, int &n, int &c0, int &tf, bool &tail) {
                if (start >= nfull) { n = 0; c0 = 0; tf = 0; tail = false; return; }   // empty chunk
                c0 = start - span > 0 ? start - span : 0;
                const int end = start + L < nfull ? start + L : nfull;
                w += c0; n = end - c0; tf = start - c0; tail = (end == nfull);
            };
            // a longer warm-up is still exact: halves sharing one read stream must share it
            int sp_lo = a.ad_span[tile.adapter_lo], sp_hi = a.ad_span[tile.adapter_hi];
            if (one_stream) sp_lo = sp_hi = (sp_lo > sp_hi ? sp_lo : sp_hi);
            cut(n_lo, sp_lo, w_lo, n_lo, c0_lo, tf_lo, tail_lo);
            cut(n_hi, sp_hi, w_hi, n_hi, c0_hi, tf_hi, tail_hi);
        }
        const int fr_lo = (have_lo && a.force_row) ? a.force_row[p_lo] : -1;   // adapter row or -1
        const int fr_hi = (have_hi && a.force_row) ? a.force_row[p_hi] : -1;

        // ---- column 0 state --------------------------------------------------------------
        // scout start: M=0 everywhere (free leading gaps).  interior start (window not at the
        // read's column 0): every state is a real lower-bound path, row-0 start + vertical gap:
        // M = open + (i-1)*ext for adapter row i>=1, i.e. T = M + open.
        u32 T[RS], U[RS];
        auto init_T = [&](int r) -> u32 {
            const int vl = (c0_lo > 0 && r >= pad_lo) ? 2 * a.gap_open + (r - pad_lo) * a.init_extend : a.gap_open;
            const int vh = (c0_hi > 0 && r >= pad_hi) ? 2 * a.gap_open + (r - pad_hi) * a.init_extend : a.gap_open;
            return ((u32)vl & 0xFFFFu) | ((u32)vh << 16);
        };
        if constexpr (GEN) {
#pragma unroll 1
            for (int r = 0; r < rows; ++r) lds_col[r * 64 + lane] = make_uint2(init_T(r), NEG2);
            T[0] = 0; U[0] = 0;
        } else {
#pragma clang loop unroll(full)
            for (int r = 0; r < R; ++r) { T[r] = pk_add(init_T(r), pack2((r + 2) * eps)); U[r] = NEG2; }
        }
        // row 0 in drifting coordinates: T(0,j) = open + (jj+1)*eps; unshift = what to take off the
        // bottom row's T to get the true M: open + (R + jj + 1)*eps, jj = columns since (re)start
        u32 topT = pack2(a.gap_open + eps);            // T(0,0)
        int jj = 0;
        Best b_lo = {0, m_lo, 0, 0}, b_hi = {0, m_hi, 0, 0};
        if (chunk > 0) { b_lo.score = -32768; b_lo.J = -1; b_hi.score = -32768; b_hi.J = -1; }   // (m,0) belongs to chunk 0

        int nmax = n_lo > n_hi ? n_lo : n_hi;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { const int o = __shfl_xor(nmax, s); nmax = o > nmax ? o : nmax; }
        if (TRACE && nmax > a.slab_cols) {   // host sized the slab from a wrong bound: refuse
            if (lane == 0) atomicAdd(a.err, 1u);
            nmax = 0;
        }

        // last-column scan of one row (tracked cells visited top to bottom, strict '>',
        // dp_scout.h:165-179; a forced end cell just records that row)
        auto scan_row = [&](int r, int j, u32 Tn, u32 t01, bool fin_lo, bool fin_hi) {
            const int il = r - pad_lo + 1, ih = r - pad_hi + 1;
            const int off = a.gap_open + (GEN ? 0 : (r + 1 + jj + 1) * eps);      // T -> true M of this row
            const int cl = lo16(Tn) - off, ch = hi16(Tn) - off;
            if (fin_lo && il >= 1 && (fr_lo >= 0 ? (il == fr_lo) : (cl > b_lo.score))) {
                b_lo.score = cl; b_lo.I = il; b_lo.J = j; b_lo.tie = !(t01 & 0xFFFFu);
            }
            if (fin_hi && ih >= 1 && (fr_hi >= 0 ? (ih == fr_hi) : (ch > b_hi.score))) {
                b_hi.score = ch; b_hi.I = ih; b_hi.J = j; b_hi.tie = !(t01 >> 16);
            }
        };

        // Forced-end windows (pass 2 of a whole-read scan) only need trace for the last W+2 columns:
        // the traced path cannot reach further left (pc_bounds.h), the columns before it are only
        // the SPAN warm-up that makes the values exact.  Those run the score-only column.
        int notrace_upto = 0;
        if (TRACE && a.force_row && a.ad_window) {
            const int w_lo = a.ad_window[tile.adapter_lo] - a.ad_span[tile.adapter_lo] - 1;
            const int w_hi = a.ad_window[tile.adapter_hi] - a.ad_span[tile.adapter_hi] - 1;
            int t0 = 1 << 30;
            if (have_lo) t0 = n_lo - w_lo - 2;
            if (have_hi && n_hi - w_hi - 2 < t0) t0 = n_hi - w_hi - 2;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) { const int o = __shfl_xor(t0, s); t0 = o < t0 ? o : t0; }
            notrace_upto = __builtin_amdgcn_readfirstlane((t0 > 0 && t0 < nmax) ? t0 : 0);   // (a tile of empty windows has no t0)   // wave-uniform: a scalar branch below
        }

        u32 cur_lo = 0, cur_hi = 0;
        // one column; TR = trace bits are formed and stored (always false in score-only kernels)
        auto column = [&](const int j, auto trace_tag) {
            constexpr bool TR = decltype(trace_tag)::value;
            if (((j - 1) & 3) == 0) {
                // next 4 bases of each stream; finished streams re-read their last dword
                const int kl = (j - 1 < n_lo) ? j - 1 : (n_lo > 0 ? ((n_lo - 1) & ~3) : 0);
                cur_lo = load_u32_unaligned(w_lo + kl);
                if (!one_stream) {
                    const int kh = (j - 1 < n_hi) ? j - 1 : (n_hi > 0 ? ((n_hi - 1) & ~3) : 0);
                    cur_hi = load_u32_unaligned(w_hi + kh);
                }
            }
            u32 h2;
            if (one_stream) {
                const u32 c = lut[cur_lo & 0xFF];
                h2 = c | (c << 16);
            } else {
                h2 = (u32)lut[cur_lo & 0xFF] | ((u32)lut[cur_hi & 0xFF] << 16);
                cur_hi >>= 8;
            }
            cur_lo >>= 8;

            u32 tie01 = 0, Tlast = 0;
            const bool fin_lo = (j == n_lo) && tail_lo, fin_hi = (j == n_hi) && tail_hi;
            const bool any_fin = __any(fin_lo || fin_hi);
            u32 *trace_dst = TR ? slab + ((int64_t)(j - 1) * NW) * 64 + lane : nullptr;

            if constexpr (GEN) {
                // ---- rolled column over the LDS-resident state --------------------------
                u32 dq = k.A2, Tup = k.O2, Vprev = k.NEG2, acc = 0;
#pragma unroll 2
                for (int r = 0; r < rows; ++r) {
                    const uint2 old = lds_col[r * 64 + lane];      // (T, U) of column j-1
                    const uint2 c = lds_const[r];
                    const u32 z = pk_minu(pk_sub(h2, c.x), c.y);
                    const u32 d = pk_sub(dq, z);
                    const u32 Hx = pk_add(old.y, k.E2);
                    const u32 Hs = pk_max(Hx, old.x);
                    const u32 Vx = pk_add(Vprev, k.E2);
                    const u32 Vs = pk_max(Vx, Tup);
                    const u32 g = pk_max(Hs, Vs);
                    const u32 Mn = pk_max(d, g);
                    const u32 Tn = pk_add(Mn, k.O2);
                    const u32 t01 = pk_minu(d ^ g, k.ONE2);
                    if constexpr (TR) {
                        const u32 b0 = pk_minu(pk_sub(Hs, Hx), k.ONE2);
                        const u32 b1 = pk_minu(pk_sub(Vs, Vx), k.ONE2);
                        const u32 b2 = pk_minu(pk_sub(g, Vs), k.ONE2);
                        const u32 b3 = pk_minu(pk_sub(Mn, d), k.ONE2);
                        u32 nib = pk_madu(b3, k.TWO2, b2);
                        nib = pk_madu(nib, k.TWO2, b1);
                        nib = pk_madu(nib, k.TWO2, b0);
                        acc = pk_madu(acc, k.SIXTEEN2, nib);
                        if ((r & 3) == 3 || r == rows - 1) { trace_dst[(r >> 2) * 64] = acc; acc = 0; }
                    }
                    if (any_fin) scan_row(r, j, Tn, t01, fin_lo, fin_hi);
                    lds_col[r * 64 + lane] = make_uint2(Tn, Hs);
                    dq = pk_add(old.x, k.AO2); Tup = Tn; Vprev = Vs;
                    tie01 = t01; Tlast = Tn;
                }
            } else {
                if (jj >= a.kren) {
                    // shift the state back down before int16 runs out (every value moves by the same amount)
                    const u32 DK = pack2(jj * eps);
#pragma clang loop unroll(full)
                    for (int r = 0; r < R; ++r) { T[r] = pk_sub(T[r], DK); U[r] = pk_sub(U[r], DK); }
                    topT = pk_sub(topT, DK);
                    jj = 0;
                }
                if (any_fin) {
                    // some pair reaches its last column: keep the previous column for the scan below
                    if (fin_lo || fin_hi) {
#pragma clang loop unroll(full)
                        for (int r = 0; r < R; ++r) fin[r * 64 + lane] = make_uint2(T[r], U[r]);
                    }
                }
                u32 trw[(RS + 3) / 4];
                const u32 topD = pk_add(topT, k.AOE2);             // T(0,j-1) + (match - open + eps)
                topT = pk_add(topT, k.EPS2);                        // T(0,j)
                ++jj;
                const u32 topT_col = topT;
                column_step<RS, PAD, TR>(T, U, h2, cs, cv, lds_const, k, topD, topT_col, trw, tie01);
                if constexpr (TR) {
#pragma unroll
                    for (int w = 0; w < (RS + 3) / 4; ++w) trace_dst[w * 64] = trw[w];
                }
                Tlast = T[RS - 1];
                if (any_fin) {
                    // Rare, so it is a rolled re-run of the column from the saved state (same
                    // packed arithmetic) that also yields the d == max(H,V) flag of every row.
                    u32 dq = topD, Tup = topT_col, Vprev = k.NEG2;
#pragma unroll 1
                    for (int r = 0; r < R; ++r) {
                        const uint2 old = (fin_lo || fin_hi) ? fin[r * 64 + lane] : make_uint2(0u, 0u);
                        const uint2 c = lds_const[r];
                        const u32 z = pk_minu(pk_sub(h2, c.x), c.y);
                        const u32 d = pk_sub(dq, z);
                        const u32 Hs = pk_max(old.y, old.x);
                        const u32 Vs = pk_max(Vprev, Tup);
                        const u32 g = pk_max(Hs, Vs);
                        const u32 Tn = pk_add(pk_max(d, g), k.OE2);
                        const u32 t01 = pk_minu(d ^ g, k.ONE2);
                        dq = pk_add(old.x, k.AOE2); Tup = Tn; Vprev = Vs;
                        scan_row(r, j, Tn, t01, fin_lo, fin_hi);
                    }
                }
            }
            // last adapter row of columns 1..n-1 (bottom register row for both halves)
            {
                const int off = a.gap_open + (GEN ? 0 : (rows + jj + 1) * eps);
                const int cl = lo16(Tlast) - off, ch = hi16(Tlast) - off;
                const bool tr_lo = j > tf_lo && (tail_lo ? j < n_lo : j <= n_lo);
                const bool tr_hi = j > tf_hi && (tail_hi ? j < n_hi : j <= n_hi);
                if (tr_lo && fr_lo < 0 && cl > b_lo.score) { b_lo.score = cl; b_lo.I = m_lo; b_lo.J = j; b_lo.tie = !(tie01 & 0xFFFFu); }
                if (tr_hi && fr_hi < 0 && ch > b_hi.score) { b_hi.score = ch; b_hi.I = m_hi; b_hi.J = j; b_hi.tie = !(tie01 >> 16); }
            }
        };
        // two loops rather than a branch per column: each gets its own register allocation
        int j = 1;
        if constexpr (TRACE) {
            for (; j <= notrace_upto; ++j) column(j, BoolTag<false>{});
        }
        for (; j <= nmax; ++j) column(j, BoolTag<TRACE>{});

        // ---- results -----------------------------------------------------------------
        if constexpr (!TRACE) {
            // J is reported in whole-window columns; chunk results go to [pair][chunk]
            // (chunks that start beyond their window's end are neither written nor read: plan_kernel merges the real ones only)
            const bool live_lo = have_lo && (chunk == 0 || n_lo > 0), live_hi = have_hi && (chunk == 0 || n_hi > 0);
            if (live_lo) { int4 o = {b_lo.score, b_lo.I, b_lo.J + c0_lo, 0}; *(int4 *)(a.out + (p_lo * nchunks + chunk) * SCORE_OUT_INTS) = o; }
            if (live_hi) { int4 o = {b_hi.score, b_hi.I, b_hi.J + c0_hi, 0}; *(int4 *)(a.out + (p_hi * nchunks + chunk) * SCORE_OUT_INTS) = o; }
        } else {
            traceback_pairs(a, slab, rows, NW, lane, b_lo, b_hi, pad_lo, pad_hi,
                            have_lo, have_hi, n_lo, n_hi, c0_lo, c0_hi, m_lo, m_hi, p_lo, p_hi, notrace_upto);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The traced scan in PACKED FP16 (the fast path of the end windows and of pass-2 windows).
//
// Same recurrence, same drifting coordinates, same trace slab and traceback as scan_kernel<R,*,true>,
// but 13.25 instead of 21 packed VALU ops per two cells:
//   * the DP values are the reference's integers held as fp16 numbers X + (rho + jj [+1])*eps - C
//     (pc_bounds.h f16_plan: every value ever formed is an integer of magnitude <= 2040, which
//     fp16 holds and adds exactly), so  M = v_pk_maximum3_f16(d, H, V)  is one op;
//   * the diagonal term is ONE op,  d = T_diag + S,  with the substitution terms S of this tile's
//     adapter pair fetched from an LDS table [25 letter pairs of the two streams][R rows] by one
//     ds_read_b128 per four rows (scan_kernel builds it from run-time adapter codes in 4 ops);
//   * each trace bit is ONE op,  b = v_pk_add_f16(x, -y) clamp  = min(max(x - y, 0), 1)  exactly
//     1.0 or 0.0 for integers, and joins the byte of its row pair by ONE fma,  acc = 2*acc + b:
//     started at 4.0, two rows (8 bits) later acc = 1024 + byte, whose fp16 bit pattern is
//     0x6400 | byte -- the integer appears in the mantissa without a conversion; v_perm_b32 joins
//     two such bytes per half into the slab's 16 bits per four rows.
// Rows are issued as one hand-interleaved asm statement each (independent half of row r+2, the
// serial V -> M -> T chain of row r, the trace bits of row r, the byte accumulation of row r-1):
// no two dependent packed ops are adjacent.
// ---------------------------------------------------------------------------------------------
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h16x2 HV(u32 x) { return __builtin_bit_cast(h16x2, x); }
__device__ __forceinline__ u32 WH(h16x2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 hk_add(u32 a, u32 b) { return WH(HV(a) + HV(b)); }
__device__ __forceinline__ u32 hk_sub(u32 a, u32 b) { return WH(HV(a) - HV(b)); }
__device__ __forceinline__ u32 hk_max(u32 a, u32 b) { return WH(__builtin_elementwise_max(HV(a), HV(b))); }
// IEEE-754-2019 maximum: lowers to v_pk_maximum3_f16 (two nested ones fuse into ONE three-input op) and, unlike
// maxnum, needs no canonicalisation of operands the compiler cannot prove quiet (values here are never NaN)
__device__ __forceinline__ u32 hk_maximum(u32 a, u32 b) { return WH(__builtin_elementwise_maximum(HV(a), HV(b))); }
__device__ __forceinline__ u32 hk_min(u32 a, u32 b) { return WH(__builtin_elementwise_min(HV(a), HV(b))); }
__device__ __forceinline__ u32 hpack2x(int l, int h) { const h16x2 v = {(_Float16)l, (_Float16)h}; return WH(v); }
__device__ __forceinline__ u32 hpack2(int v) { return hpack2x(v, v); }
__device__ __forceinline__ int hlo(u32 x) { return (int)(float)HV(x).x; }
__device__ __forceinline__ int hhi(u32 x) { return (int)(float)HV(x).y; }
constexpr u32 H_NEGINF2 = 0xFC00FC00u, H_POSINF2 = 0x7C007C00u;

// the trace slab is written once and read back only along the path, long after it has left the caches
#if defined(PC_ABL_NOSTORE)          // timing experiment: the trace words are formed and dropped
#define SLAB_STORE(p, v) asm volatile("" :: "v"(v))
#elif !defined(PC_SLAB_TEMPORAL)
#define SLAB_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define SLAB_STORE(p, v) (*(p) = (v))
#endif
// K trace words of one lane (K = 4, or the 1..3 of a column's last group) to the block's slab: one buffer store
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x3 __attribute__((ext_vector_type(3)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
template <int K>
__device__ __forceinline__ void slab_store_group(const __amdgpu_buffer_rsrc_t rsrc, const u32 (&w)[4], const int lane, const int soff)
{
#if defined(PC_ABL_NOSTORE)
    asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
#else
#ifdef PC_SLAB_TEMPORAL
    constexpr int AUX = 0;
#else
    constexpr int AUX = 2;          // nt: written once, read back only along the path, long after it has left the caches
#endif
    if constexpr (K == 4) { const u32x4 v = {w[0], w[1], w[2], w[3]}; __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane * 16, soff, AUX); }
    else if constexpr (K == 3) { const u32x3 v = {w[0], w[1], w[2]}; __builtin_amdgcn_raw_buffer_store_b96(v, rsrc, lane * 12, soff, AUX); }
    else if constexpr (K == 2) { const u32x2 v = {w[0], w[1]}; __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, lane * 8, soff, AUX); }
    else __builtin_amdgcn_raw_buffer_store_b32(w[0], rsrc, lane * 4, soff, AUX);
#endif
}
#define PC_HMAX "v_pk_max_f16 "
#define PC_HADD "v_pk_add_f16 "
#define PC_HMAX3 "v_pk_maximum3_f16 "
#define PC_HFMA "v_pk_fma_f16 "
#define PC_BIT " neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
// byte accumulation of the previous row's four bits: an even row starts a byte (acc = 2*4 + b3)
#define PC_ACC3_EVEN PC_HADD "%[acc], %[pb3], %[eight]\n\t"
#define PC_ACC3_ODD PC_HFMA "%[acc], %[acc], %[two], %[pb3]\n\t"
#define PC_ACC2 PC_HFMA "%[acc], %[acc], %[two], %[pb2]\n\t"
#define PC_ACC1 PC_HFMA "%[acc], %[acc], %[two], %[pb1]\n\t"
#define PC_ACC0 PC_HFMA "%[acc], %[acc], %[two], %[pb0]\n\t"
// full row: ind(q = r+2) | chain(r) | bits(r) | acc(r-1)
#define PC_ROW16_FULL(ACC3)                                         \
    PC_HADD "%[b0q], %[tq], %[uq]" PC_BIT                           \
    ACC3                                                            \
    PC_HMAX "%[vs], %[vp], %[tu]\n\t"                               \
    PC_HADD "%[dq], %[dg], %[sq]\n\t"                               \
    PC_ACC2                                                         \
    PC_HMAX3 "%[mn], %[dr], %[hr], %[vs]\n\t"                       \
    PC_HMAX "%[uq], %[uq], %[tq]\n\t"                               \
    PC_ACC1                                                         \
    PC_HADD "%[tn], %[mn], %[oe]\n\t"                               \
    PC_HADD "%[b1], %[vs], %[vp]" PC_BIT                            \
    PC_ACC0                                                         \
    PC_HADD "%[b2], %[hr], %[vs]" PC_BIT                            \
    PC_HADD "%[b3], %[mn], %[dr] neg_lo:[0,1] neg_hi:[0,1] clamp"
// first row: nothing to accumulate yet
#define PC_ROW16_FIRST                                              \
    PC_HADD "%[b0q], %[tq], %[uq]" PC_BIT                           \
    PC_HMAX "%[vs], %[vp], %[tu]\n\t"                               \
    PC_HADD "%[dq], %[dg], %[sq]\n\t"                               \
    PC_HMAX "%[uq], %[uq], %[tq]\n\t"                               \
    PC_HMAX3 "%[mn], %[dr], %[hr], %[vs]\n\t"                       \
    PC_HADD "%[b1], %[vs], %[vp]" PC_BIT                            \
    PC_HADD "%[b2], %[hr], %[vs]" PC_BIT                            \
    PC_HADD "%[tn], %[mn], %[oe]\n\t"                               \
    PC_HADD "%[b3], %[mn], %[dr] neg_lo:[0,1] neg_hi:[0,1] clamp"
// last two rows: no row r+2
#define PC_ROW16_NOIND(ACC3)                                        \
    ACC3                                                            \
    PC_HMAX "%[vs], %[vp], %[tu]\n\t"                               \
    PC_ACC2                                                         \
    PC_HADD "%[b1], %[vs], %[vp]" PC_BIT                            \
    PC_HMAX3 "%[mn], %[dr], %[hr], %[vs]\n\t"                       \
    PC_ACC1                                                         \
    PC_HADD "%[b2], %[hr], %[vs]" PC_BIT                            \
    PC_HADD "%[tn], %[mn], %[oe]\n\t"                               \
    PC_ACC0                                                         \
    PC_HADD "%[b3], %[mn], %[dr] neg_lo:[0,1] neg_hi:[0,1] clamp"

// CHECK: debug build (PC_CHECK_RANGE=1, every row class): records the extremes of EVERY finite value the kernel
// forms in fp16 -- the column state T / U after each column, and inside the column each row's diagonal term d,
// vertical state V, cell maximum M and new T, the top-row term T~(0,j), the substitution-table terms, the tracked
// last-row term M(R,j) + R*eps the scout compares (`cand`, before its mask) and the packed running maxima `best2`,
// the warm-up columns' and the last-column re-run's values, the byte accumulators of the trace bits -- into err[4]
// (max) and err[5] (-min): the host-side range gate (pc_bounds.h f16_plan) asserted on the device; read with
// pc_debug_value_range.  (-inf only ever enters as the initial H / V of a column and as the scout's mask; it is
// absorbed by the first max and is not a "value formed".)
// SCORE: the same kernel as a score-only FIRST pass of the two-pass end scan (PC_MODE_TRACE over end windows, pc_api.cpp):
// every column runs the bare five-op recurrence of the warm-up loop below plus the packed scout, nothing is traced or stored,
// and the pair's end cell leaves as a score record (-2, J, I, 0, score, 0, 0, 0) -- what plan_kernel (end_records) turns into
// the window of the second, traced pass over the columns the path can occupy.
template <int R, bool CHECK = false, bool SCORE = false>
#ifndef PC_T16_WAVES_A
#define PC_T16_WAVES_A 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(R <= 30 ? PC_T16_WAVES_A : R <= 44 ? 3 : 2))) void trace16_kernel(ScanArgs a)
{
    static_assert(R >= 4 && R % 2 == 0, "row classes are even");
    constexpr int RP = (R + 3) & ~3;         // table row padded to whole b128 groups
    // dwords per table row.  Every lane reads ITS letter pair's row, four dwords at a time (ds_read_b128), so up to 25
    // different rows are read by one instruction: rows must start in different bank groups.  The stride in 16-byte slots is
    // kept ODD -- consecutive rows then walk through all eight slots of the 32 banks.  (RP + 4 alone made it 32 dwords for
    // the 28-row class of the ligation adapters: every row in the same four banks, each load serialised up to 25 deep.)
#ifdef PC_VARIANT_STRIDE4
    constexpr int STRIDE = RP + 4;
#else
    constexpr int STRIDE = ((RP / 4) % 2 == 0) ? RP + 4 : RP + 8;
#endif
    constexpr int NW = (R + 3) / 4;          // trace dwords per column per lane
    __shared__ uint16_t lut_lo[256], lut_hi[256];             // byte -> table row offset (dwords) of the lo / hi stream
    __shared__ __attribute__((aligned(16))) u32 s_tab[25 * STRIDE];
    // per-lane state that only the rare paths touch (a new maximum, a pair's last column) lives in LDS
    // during the column loop instead of in a dozen VGPRs: scout cells (score, I, J, tie) of both halves
    // and the forced end rows
    __shared__ int st_best[8][64], st_fr[2][64];
    uint2 *fin = (uint2 *)a.fin_scratch + (int64_t)blockIdx.x * R * 64;
    const int lane = threadIdx.x;
    for (int c = lane; c < 256; c += 64) {
        const int code = dna5_code(c);
        lut_lo[c] = (uint16_t)(code * 5 * STRIDE);
        lut_hi[c] = (uint16_t)(code * STRIDE);
    }
    const int eps = -a.gap_extend, CEN = a.f16_cen;
    const u32 OE2 = __builtin_amdgcn_readfirstlane(hpack2(a.gap_open + eps)), EPS2 = hpack2(eps), NEG2 = H_NEGINF2;
    const u32 TWO2 = 0x40004000u, EIGHT2 = 0x48004800u;
    u32 *slab = SCORE ? nullptr : a.slab + (int64_t)blockIdx.x * a.slab_stride;
#ifndef PC_SLAB_OLD
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, SCORE ? 0 : (int)(a.slab_stride * 4), 0x00020000);
#endif

    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const Tile tile = a.tiles[t];
        const int m_lo = a.ad_len[tile.adapter_lo];
        const int m_hi = a.ad_len[tile.adapter_hi];
        const u32 *codes_lo = a.ad_codes + (int64_t)tile.adapter_lo * pcb::MAX_ADAPTER;
        const u32 *codes_hi = a.ad_codes + (int64_t)tile.adapter_hi * pcb::MAX_ADAPTER;
        const int pad_lo = R - m_lo, pad_hi = R - m_hi;           // top padding rows of each half
        if (pad_lo < 0 || pad_hi < 0) {                           // host bug
            if (lane == 0) atomicAdd(a.err, 1u);
            continue;
        }
        int tab_max = 0, tab_min = 0;                             // CHECK: extremes of the table terms this lane wrote
        // ---- substitution table of this adapter pair: s_tab[(c_lo*5 + c_hi)][row] = packed
        // (sub_lo - open + eps | sub_hi - open + eps); a padding row scores 0, which keeps its
        // M = 0 and re-opens V exactly like the true row 0 (DESIGN.md "top padding")
        __syncthreads();
        for (int e = lane; e < 25 * RP; e += 64) {
            const int pair = e / RP, r = e - pair * RP;
            const int cl = pair / 5, ch = pair - cl * 5;
            int sl = 0, sh = 0;
            if (r < R) {
                const int il = r - pad_lo, ih = r - pad_hi;
                if (il >= 0) sl = ((int)codes_lo[il] == cl) ? a.match : a.mismatch;
                if (ih >= 0) sh = ((int)codes_hi[ih] == ch) ? a.match : a.mismatch;
            }
            s_tab[pair * STRIDE + r] = hpack2x(sl - a.gap_open + eps, sh - a.gap_open + eps);
            if constexpr (CHECK) {
                const int lo_ = sl < sh ? sl : sh, hi_ = sl > sh ? sl : sh;
                tab_max = tab_max > hi_ - a.gap_open + eps ? tab_max : hi_ - a.gap_open + eps;
                tab_min = tab_min < lo_ - a.gap_open + eps ? tab_min : lo_ - a.gap_open + eps;
            }
        }
        __syncthreads();

        // ---- this lane's two pairs -----------------------------------------------------
        // (a.perm: the second pass of the two-pass end scan takes the pairs of a segment in the order of their end columns --
        // slot s of the tile holds pair perm[s] of the same segment)
        const bool have_lo = lane < tile.count_lo, have_hi = lane < tile.count_hi;
        const int64_t p_lo = (a.perm && have_lo) ? a.perm[tile.out_lo + lane] : tile.out_lo + lane;
        const int64_t p_hi = (a.perm && have_hi) ? a.perm[tile.out_hi + lane] : tile.out_hi + lane;
        const int64_t wi_lo = a.win_by_out ? p_lo : tile.win_lo + lane;
        const int64_t wi_hi = a.win_by_out ? p_hi : tile.win_hi + lane;
        const bool one_stream = !a.win_by_out && tile.win_lo == tile.win_hi;
        const uint8_t *w_lo = a.arena + (have_lo ? a.win_off[wi_lo] : 0);
        const uint8_t *w_hi = a.arena + (have_hi ? a.win_off[wi_hi] : 0);
        const int n_lo = have_lo ? a.win_len[wi_lo] : 0;
        const int n_hi = have_hi ? a.win_len[wi_hi] : 0;
        const int c0_lo = (have_lo && a.col0) ? a.col0[p_lo] : 0;
        const int c0_hi = (have_hi && a.col0) ? a.col0[p_hi] : 0;
        {
            const int fl = (have_lo && a.force_row) ? a.force_row[p_lo] : -1;
            const int fh = (have_hi && a.force_row) ? a.force_row[p_hi] : -1;
            st_fr[0][lane] = fl; st_fr[1][lane] = fh;
            st_best[0][lane] = 0; st_best[1][lane] = m_lo; st_best[2][lane] = 0; st_best[3][lane] = 0;
            st_best[4][lane] = 0; st_best[5][lane] = m_hi; st_best[6][lane] = 0; st_best[7][lane] = 0;
        }

        // column 0 (see scan_kernel): M = 0, or the lower-bound state of an interior window start
        u32 T[R], U[R];
#pragma clang loop unroll(full)
        for (int r = 0; r < R; ++r) {
            const int vl = (c0_lo > 0 && r >= pad_lo) ? 2 * a.gap_open + (r - pad_lo) * a.init_extend : a.gap_open;
            const int vh = (c0_hi > 0 && r >= pad_hi) ? 2 * a.gap_open + (r - pad_hi) * a.init_extend : a.gap_open;
            T[r] = hpack2x(vl + (r + 2) * eps - CEN, vh + (r + 2) * eps - CEN);
            U[r] = NEG2;
        }
        u32 top = hpack2(a.gap_open + eps - CEN);                 // T~(0, j-1) entering column j
        // The scout of the last-row cells (columns 1..n-1) lives PACKED while the columns run: best2 = the running maxima of
        // the tracked term M(R,j) + R*eps, pos2 = their columns as two u16 (0: the corner (m, 0) every pair starts from),
        // bd2 / bg2 = the diagonal term d and max(H, V) of that cell (equal <=> SeqAn's tie flag; initially not).  A new
        // maximum is eight packed operations per column for both halves, branch-free.  (It used to be a branch taken whenever
        // ANY of a wave's 128 pairs improved -- with ~5 record columns per pair in 150 that is nearly every column -- into a
        // path of ~70 instructions around LDS-resident state: a fifth of the kernel's instructions for a 22-row tile.)
        u32 best2 = hpack2(R * eps), pos2 = 0u, bd2 = 0u, bg2 = 0x3C003C00u;

        int nmax = n_lo > n_hi ? n_lo : n_hi;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { const int o = __shfl_xor(nmax, s); nmax = o > nmax ? o : nmax; }
        if ((!SCORE && nmax > a.slab_cols) || nmax > a.f16_max_cols) {   // host sized the slab / chose this kernel from a wrong bound
            if (lane == 0) atomicAdd(a.err, 1u);
            nmax = 0;
        }
        int notrace_upto = 0;
        if (a.force_row && a.ad_window) {
            // traced columns before the end cell: the adapter's W + 2, or the pair's own (smaller) bound from plan_kernel
            int wl = a.ad_window[tile.adapter_lo] - a.ad_span[tile.adapter_lo] - 1 + 2;
            int wh = a.ad_window[tile.adapter_hi] - a.ad_span[tile.adapter_hi] - 1 + 2;
            if (a.trace_cols) {
                if (have_lo) { const int t = a.trace_cols[p_lo]; wl = t < wl ? t : wl; }
                if (have_hi) { const int t = a.trace_cols[p_hi]; wh = t < wh ? t : wh; }
            }
            int t0 = 1 << 30;
            if (have_lo && n_lo > 0) t0 = n_lo - wl;
            if (have_hi && n_hi > 0 && n_hi - wh < t0) t0 = n_hi - wh;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) { const int o = __shfl_xor(t0, s); t0 = o < t0 ? o : t0; }
            notrace_upto = __builtin_amdgcn_readfirstlane((t0 > 0 && t0 < nmax) ? t0 : 0);   // (a tile of empty windows has no t0)
        }
        // last-row cells are tracked in columns 1..n-1 of pairs that scout (no forced end cell):
        // +inf lets a half's candidate through, -inf blanks it
        auto track_limit = [&](int j) -> u32 {
            const bool tl = st_fr[0][lane] < 0 && j < n_lo, th = st_fr[1][lane] < 0 && j < n_hi;
            return (tl ? (H_POSINF2 & 0xFFFFu) : (H_NEGINF2 & 0xFFFFu)) | (th ? (H_POSINF2 & 0xFFFF0000u) : (H_NEGINF2 & 0xFFFF0000u));
        };
        u32 limit2 = track_limit(1);          // changes only in a column where some pair ends (any_fin below)
        auto scan_row = [&](int r, int j, u32 Tn, bool tie_l, bool tie_h, bool fin_lo, bool fin_hi, Best &b_lo, Best &b_hi,
                            int fr_lo, int fr_hi) {
            const int il = r - pad_lo + 1, ih = r - pad_hi + 1;
            const int off = a.gap_open + (r + 1 + j + 1) * eps - CEN;                  // T~ -> true M of this row
            const int cl = hlo(Tn) - off, ch = hhi(Tn) - off;
            if (fin_lo && il >= 1 && (fr_lo >= 0 ? (il == fr_lo) : (cl > b_lo.score))) { b_lo.score = cl; b_lo.I = il; b_lo.J = j; b_lo.tie = tie_l; }
            if (fin_hi && ih >= 1 && (fr_hi >= 0 ? (ih == fr_hi) : (ch > b_hi.score))) { b_hi.score = ch; b_hi.I = ih; b_hi.J = j; b_hi.tie = tie_h; }
        };
        auto load_best = [&](Best &b_lo, Best &b_hi) {
            b_lo = {st_best[0][lane], st_best[1][lane], st_best[2][lane], st_best[3][lane]};
            b_hi = {st_best[4][lane], st_best[5][lane], st_best[6][lane], st_best[7][lane]};
        };
        // Read bytes come 16 columns per load (q_lo / q_hi) and are handed on a dword at a time: every lane reads its
        // own window, so each load is an L2 request per lane whatever its width -- a dword per load asked L2 for
        // 32 x the bytes used (see pc_jit_source.h).  The next 16 columns are requested when the last dword of the
        // current ones has been taken, four columns before their first byte is needed.
        auto load_q = [&](const uint8_t *w, int n, int col) -> uint4 {  // the 16 columns from 0-based column col
            const int k = (col < n) ? col : (n > 0 ? ((n - 1) & ~15) : 0);
            uint4 v;
            __builtin_memcpy(&v, w + k, 16);                                   // any alignment: one global_load_dwordx4
            return v;
        };
        uint4 q_lo = load_q(w_lo, n_lo, 0), q_hi = one_stream ? make_uint4(0u, 0u, 0u, 0u) : load_q(w_hi, n_hi, 0);
        auto pick_dw = [&](const uint4 &q, int k) -> u32 { return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w; };
        u32 cur_lo = q_lo.x, cur_hi = q_hi.x;
        auto next_dword = [&](int j) {                                  // j % 4 == 0: the dword of 0-based columns j..j+3
            const int kq = (j >> 2) & 3;
            cur_lo = pick_dw(q_lo, kq);
            if (!one_stream) cur_hi = pick_dw(q_hi, kq);
            if (kq == 3) {
                q_lo = load_q(w_lo, n_lo, j + 4);
                if (!one_stream) q_hi = load_q(w_hi, n_hi, j + 4);
            }
        };
        auto row_of = [&](u32 bl, u32 bh) -> int {                      // table row (dword offset) of a column's byte pair
            return one_stream ? (int)lut_lo[bl] + (int)lut_hi[bl] : (int)lut_lo[bl] + (int)lut_hi[bh];
        };

        // End-aligned pass-2 windows (plan_kernel, end_align): a window that would start before its read's column 0
        // carries a lead-in of -col0 columns instead, so that every window of the tile ends in the same column and
        // the tile keeps its trace-free warm-up.  The lead-in is garbage: when a half reaches the read's column 0
        // its state is set to what column 0 is (M = 0 everywhere), in the drifted frame of that column.
        const int sh_lo = c0_lo < 0 ? -c0_lo : 0, sh_hi = c0_hi < 0 ? -c0_hi : 0;
        int shmax = sh_lo > sh_hi ? sh_lo : sh_hi;
#pragma unroll
        for (int s_ = 32; s_ >= 1; s_ >>= 1) { const int o = __shfl_xor(shmax, s_); shmax = o > shmax ? o : shmax; }
        shmax = __builtin_amdgcn_readfirstlane(shmax);
        auto reach_column0 = [&](int j) {
            const bool rl = sh_lo > 0 && j == sh_lo, rh = sh_hi > 0 && j == sh_hi;
            if (!__any(rl || rh)) return;
            const u32 ml = rl ? 0xFFFFu : 0u, mh = rh ? 0xFFFF0000u : 0u, keep = ~(ml | mh);
#pragma clang loop unroll(full)
            for (int r = 0; r < R; ++r) {
                const u32 init = hpack2(a.gap_open + (r + 2 + j) * eps - CEN);
                T[r] = (T[r] & keep) | (init & (ml | mh));
                U[r] = (U[r] & keep) | (NEG2 & (ml | mh));
            }
        };
        int trow = row_of(cur_lo & 0xFF, cur_hi & 0xFF);
        u32 vmax = H_NEGINF2, vmin = H_POSINF2;
        auto note = [&](u32 x) { vmax = hk_max(vmax, x); vmin = hk_min(vmin, x); };      // CHECK builds only
        // ---- warm-up columns of pass-2 windows: the traced path provably cannot reach them (pc_bounds.h),
        // every pair's end cell is forced and no window ends here, so they run the bare recurrence -- five
        // ops per two cells, no trace bits, no slab stores, no scout -- left to the compiler's scheduler
        int j_first = 1;
        for (; j_first <= notrace_upto; ++j_first) {
            const int trow_j = trow;
            cur_lo >>= 8; cur_hi >>= 8;
            if ((j_first & 3) == 0) next_dword(j_first);
            trow = row_of(cur_lo & 0xFF, cur_hi & 0xFF);
            const u32 topn = hk_add(top, EPS2);
            const uint4 *srow = (const uint4 *)(s_tab + trow_j);
            u32 dq = top, Tup = topn, Vp = NEG2;
#pragma clang loop unroll(full)
            for (int g = 0; g < RP / 4; ++g) {
                const uint4 v = srow[g];
                const u32 Sg[4] = {v.x, v.y, v.z, v.w};
#pragma clang loop unroll(full)
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * g + k;
                    if (r < R) {
                        const u32 Hs = hk_maximum(U[r], T[r]);
                        const u32 d = hk_add(dq, Sg[k]);
                        const u32 Vs = hk_maximum(Vp, Tup);
                        const u32 Tn = hk_add(hk_maximum(hk_maximum(d, Hs), Vs), OE2);
                        if constexpr (CHECK) { note(d); note(Vs); note(Tn); note(hk_maximum(hk_maximum(d, Hs), Vs)); vmax = hk_max(vmax, Hs); }
                        dq = T[r]; U[r] = Hs; T[r] = Tn; Tup = Tn; Vp = Vs;
                    }
                }
            }
            if constexpr (CHECK) note(topn);
            top = topn;
            if (j_first <= shmax) reach_column0(j_first);
        }
        for (int j = j_first; j <= nmax; ++j) {
            const int trow_j = trow;
            // next column's bytes / table row, one column ahead of their use
            cur_lo >>= 8; cur_hi >>= 8;
            if ((j & 3) == 0) next_dword(j);
            trow = row_of(cur_lo & 0xFF, cur_hi & 0xFF);

            const bool fin_lo = (j == n_lo), fin_hi = (j == n_hi);
            const bool any_fin = __any(fin_lo || fin_hi);
            if (any_fin) limit2 = track_limit(j);
            if (any_fin && (fin_lo || fin_hi)) {
#pragma clang loop unroll(full)
                for (int r = 0; r < R; ++r) fin[r * 64 + lane] = make_uint2(T[r], U[r]);
            }
            const u32 topn = hk_add(top, EPS2);                   // T~(0, j)
#ifdef PC_ABL_NOTABLE                 // timing experiment: one table row for every column (its loads hoist out of the loop)
            const uint4 *srow = (const uint4 *)(s_tab);
#else
            const uint4 *srow = (const uint4 *)(s_tab + trow_j);
#endif
            u32 d_last = 0, h_last = 0, v_last = 0;
            if constexpr (SCORE) {
                // ---- the column, score only: the bare recurrence (as in the warm-up columns above) -------------------
                u32 dq = top, Tup = topn, Vp = NEG2;
#pragma clang loop unroll(full)
                for (int g = 0; g < RP / 4; ++g) {
                    const uint4 v = srow[g];
                    const u32 Sg[4] = {v.x, v.y, v.z, v.w};
#pragma clang loop unroll(full)
                    for (int k = 0; k < 4; ++k) {
                        const int r = 4 * g + k;
                        if (r < R) {
                            const u32 Hs = hk_maximum(U[r], T[r]);
                            const u32 d = hk_add(dq, Sg[k]);
                            const u32 Vs = hk_maximum(Vp, Tup);
                            const u32 Tn = hk_add(hk_maximum(hk_maximum(d, Hs), Vs), OE2);
                            dq = T[r]; U[r] = Hs; T[r] = Tn; Tup = Tn; Vp = Vs;
                        }
                    }
                }
            } else {
#ifdef PC_SLAB_OLD
            u32 *trace_dst = slab + ((int64_t)((a.debug & 2) ? 0 : (j - 1)) * NW) * 64 + lane;
#else
            // this column's trace words leave in groups of four: ONE buffer_store_dwordx4 per 16 rows (address = the block's
            // buffer resource + a scalar column offset + a per-lane constant: no 64-bit address arithmetic per column, a
            // quarter of the store instructions and of the address traffic of a dword per 4 rows)
            const int col_soff = ((a.debug & 2) ? 0 : (j - 1)) * (NW * 256);
            u32 tw[4] = {0u, 0u, 0u, 0u};
#endif

            // ---- the column ------------------------------------------------------------------
            u32 S[RP];
            auto load_group = [&](int g) { const uint4 v = srow[g]; S[4 * g] = v.x; S[4 * g + 1] = v.y; S[4 * g + 2] = v.z; S[4 * g + 3] = v.w; };
            load_group(0);
            if (RP > 4) load_group(1);
            u32 dh[R], b0[R];                    // d and HOPEN of the rows in flight (a window of 3 is ever live)
            // independent halves of rows 0 and 1
            {
                asm volatile(PC_HADD "%[b00], %[t0], %[u0]" PC_BIT
                             PC_HADD "%[b01], %[t1], %[u1]" PC_BIT
                             PC_HADD "%[d0], %[top], %[s0]\n\t"
                             PC_HADD "%[d1], %[t0], %[s1]\n\t"
                             PC_HMAX "%[u0], %[u0], %[t0]\n\t"
                             PC_HMAX "%[u1], %[u1], %[t1]"
                             : [u0] "+v"(U[0]), [u1] "+v"(U[1]), [d0] "=&v"(dh[0]), [d1] "=&v"(dh[1]), [b00] "=&v"(b0[0]), [b01] "=&v"(b0[1])
                             : [t0] "v"(T[0]), [t1] "v"(T[1]), [top] "v"(top), [s0] "v"(S[0]), [s1] "v"(S[1]));
            }
            u32 Tup = topn, Vp = NEG2, acc = 0, accA = 0;
            u32 pb1 = 0, pb2 = 0, pb3 = 0;       // bits of the previous row (its HOPEN bit is b0[r-1])
#pragma clang loop unroll(full)
            for (int r = 0; r < R; ++r) {
                // table terms one group ahead of the row that needs them (row r+2 is fetched now)
                if (((r + 6) & 3) == 0 && (r + 6) < RP) load_group((r + 6) >> 2);
                u32 vs, mn, tn, b1, b2, b3;
                if (r + 2 < R) {
                    const int q = r + 2;
                    if (r == 0) {
                        asm volatile(PC_ROW16_FIRST
                                     : [uq] "+v"(U[q]), [vs] "=&v"(vs), [dq] "=&v"(dh[q]), [b0q] "=&v"(b0[q]), [mn] "=&v"(mn), [tn] "=&v"(tn),
                                       [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
                                     : [tq] "v"(T[q]), [vp] "v"(Vp), [tu] "v"(Tup), [dg] "v"(T[q - 1]), [sq] "v"(S[q]),
                                       [dr] "v"(dh[r]), [hr] "v"(U[r]), [oe] "s"(OE2));
                    } else if ((r - 1) & 1) {
                        asm volatile(PC_ROW16_FULL(PC_ACC3_ODD)
                                     : [uq] "+v"(U[q]), [vs] "=&v"(vs), [dq] "=&v"(dh[q]), [b0q] "=&v"(b0[q]), [mn] "=&v"(mn), [tn] "=&v"(tn),
                                       [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [acc] "+v"(acc)
                                     : [tq] "v"(T[q]), [vp] "v"(Vp), [tu] "v"(Tup), [dg] "v"(T[q - 1]), [sq] "v"(S[q]),
                                       [dr] "v"(dh[r]), [hr] "v"(U[r]), [oe] "s"(OE2), [two] "s"(TWO2),
                                       [pb0] "v"(b0[r - 1]), [pb1] "v"(pb1), [pb2] "v"(pb2), [pb3] "v"(pb3));
                    } else {
                        asm volatile(PC_ROW16_FULL(PC_ACC3_EVEN)
                                     : [uq] "+v"(U[q]), [vs] "=&v"(vs), [dq] "=&v"(dh[q]), [b0q] "=&v"(b0[q]), [mn] "=&v"(mn), [tn] "=&v"(tn),
                                       [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [acc] "=&v"(acc)
                                     : [tq] "v"(T[q]), [vp] "v"(Vp), [tu] "v"(Tup), [dg] "v"(T[q - 1]), [sq] "v"(S[q]),
                                       [dr] "v"(dh[r]), [hr] "v"(U[r]), [oe] "s"(OE2), [two] "s"(TWO2), [eight] "s"(EIGHT2),
                                       [pb0] "v"(b0[r - 1]), [pb1] "v"(pb1), [pb2] "v"(pb2), [pb3] "v"(pb3));
                    }
                } else if ((r - 1) & 1) {
                    asm volatile(PC_ROW16_NOIND(PC_ACC3_ODD)
                                 : [vs] "=&v"(vs), [mn] "=&v"(mn), [tn] "=&v"(tn), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [acc] "+v"(acc)
                                 : [vp] "v"(Vp), [tu] "v"(Tup), [dr] "v"(dh[r]), [hr] "v"(U[r]), [oe] "s"(OE2), [two] "s"(TWO2),
                                   [pb0] "v"(b0[r - 1]), [pb1] "v"(pb1), [pb2] "v"(pb2), [pb3] "v"(pb3));
                } else {
                    asm volatile(PC_ROW16_NOIND(PC_ACC3_EVEN)
                                 : [vs] "=&v"(vs), [mn] "=&v"(mn), [tn] "=&v"(tn), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [acc] "=&v"(acc)
                                 : [vp] "v"(Vp), [tu] "v"(Tup), [dr] "v"(dh[r]), [hr] "v"(U[r]), [oe] "s"(OE2), [two] "s"(TWO2), [eight] "s"(EIGHT2),
                                   [pb0] "v"(b0[r - 1]), [pb1] "v"(pb1), [pb2] "v"(pb2), [pb3] "v"(pb3));
                }
                // row r-1's byte half is complete now: every second row a byte, every fourth a slab word
                if (r >= 1) {
                    const int pr = r - 1;
                    if ((pr & 3) == 1) accA = acc;
                    if ((pr & 3) == 3) {
                        const u32 wd = __builtin_amdgcn_perm(accA, acc, 0x06020400u);
#ifdef PC_SLAB_OLD
                        SLAB_STORE(&trace_dst[(pr >> 2) * 64], wd);
#else
                        tw[(pr >> 2) & 3] = wd;
                        if (((pr >> 2) & 3) == 3) slab_store_group<4>(slab_rsrc, tw, lane, col_soff + (pr >> 4) * 1024);
#endif
                    }
                }
                if (r == R - 1) { d_last = dh[r]; h_last = U[r]; v_last = vs; }
                if constexpr (CHECK) {
                    // (a finished stream re-reads its last bytes and its state is never used: not noted)
                    if (j <= (n_lo > n_hi ? n_lo : n_hi)) { note(dh[r]); note(vs); note(mn); note(tn); vmax = hk_max(vmax, acc); }
                }
                T[r] = tn; Tup = tn; Vp = vs; pb1 = b1; pb2 = b2; pb3 = b3;
            }
            // the last row's bits (R is even, so this row completes a byte)
            asm volatile(PC_ACC3_ODD "s_nop 0\n\t" PC_ACC2 "s_nop 0\n\t" PC_ACC1 "s_nop 0\n\t"
                         PC_HFMA "%[acc], %[acc], %[two], %[pb0]"
                         : [acc] "+v"(acc)
                         : [two] "s"(TWO2), [pb0] "v"(b0[R - 1]), [pb1] "v"(pb1), [pb2] "v"(pb2), [pb3] "v"(pb3));
            {
                constexpr int pr = R - 1;
                const u32 wd = ((pr & 3) == 3) ? __builtin_amdgcn_perm(accA, acc, 0x06020400u)     // rows 4g..4g+3
                                               : __builtin_amdgcn_perm(acc, 0u, 0x0c060c04u);     // a last group of two rows
#ifdef PC_SLAB_OLD
                SLAB_STORE(&trace_dst[(pr >> 2) * 64], wd);
#else
                tw[(pr >> 2) & 3] = wd;
                slab_store_group<((NW - 1) & 3) + 1>(slab_rsrc, tw, lane, col_soff + ((NW - 1) >> 2) * 1024);
#endif
            }

            }   // (traced column)
            if constexpr (CHECK) {
                // only while a pair still runs: a finished stream re-reads its last bytes and its state is never used
                if (j <= (n_lo > n_hi ? n_lo : n_hi)) {
#pragma clang loop unroll(full)
                    for (int r = 0; r < R; ++r) { vmax = hk_max(vmax, hk_max(T[r], U[r])); vmin = hk_min(vmin, T[r]); }
                }
            }
            // ---- tracked cells ---------------------------------------------------------------
            const u32 cand = hk_min(hk_sub(T[R - 1], topn), limit2);     // M(R,j) + R*eps where tracked, else -inf
            if constexpr (CHECK) {
                if (j <= (n_lo > n_hi ? n_lo : n_hi)) { note(hk_sub(T[R - 1], topn)); note(topn); note(best2); }
            }
            {
                // strict '>' in visiting order (dp_scout.h:165-179): the maximum changes only where cand is larger, and an
                // earlier column keeps its place; -inf (untracked: limit2) never wins.  best2 is always finite.
                const u32 nb = hk_maximum(best2, cand);
                u32 chg, msk;
                asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(chg) : "v"(nb ^ best2), "s"(0x00010001u));      // 1 where a half has a new maximum
                asm volatile("v_pk_sub_u16 %0, 0, %1" : "=v"(msk) : "v"(chg));                                 // 0xFFFF there
                const u32 j2 = (u32)j * 0x00010001u;
                const u32 g = hk_max(h_last, v_last);
                pos2 = (j2 & msk) | (pos2 & ~msk);
                bd2 = (d_last & msk) | (bd2 & ~msk);
                bg2 = (g & msk) | (bg2 & ~msk);
                best2 = nb;
            }
            if (any_fin) {
                // a pair's last column: its scout leaves the packed form (every in-loop cell is a last-row cell: I = m) ...
                Best b_lo = {hlo(best2) - R * eps, m_lo, (int)(pos2 & 0xFFFFu), HV(bd2).x == HV(bg2).x ? 1 : 0};
                Best b_hi = {hhi(best2) - R * eps, m_hi, (int)(pos2 >> 16), HV(bd2).y == HV(bg2).y ? 1 : 0};
                const int fr_lo = st_fr[0][lane], fr_hi = st_fr[1][lane];
                // a pair's last column: rolled re-run from the saved previous column, tracked cells top
                // to bottom with strict '>' (dp_scout.h:165-179), plus every row's d == max(H,V) flag
                u32 dq = top, Tu2 = topn, Vp2 = NEG2;
#pragma unroll 1
                for (int r = 0; r < R; ++r) {
                    const uint2 old = (fin_lo || fin_hi) ? fin[r * 64 + lane] : make_uint2(0u, 0u);
                    const u32 sv = s_tab[trow_j + r];
                    const u32 d = hk_add(dq, sv);
                    const u32 Hs = hk_max(old.y, old.x);
                    const u32 Vs = hk_max(Vp2, Tu2);
                    const u32 g = hk_max(Hs, Vs);
                    const u32 Tn = hk_add(hk_max(d, g), OE2);
                    if constexpr (CHECK) {
                        if (fin_lo || fin_hi) { note(d); note(Vs); note(Tn); note(hk_max(d, g)); vmax = hk_max(vmax, Hs); }
                    }
                    dq = old.x; Tu2 = Tn; Vp2 = Vs;
                    scan_row(r, j, Tn, HV(d).x == HV(g).x, HV(d).y == HV(g).y, fin_lo, fin_hi, b_lo, b_hi, fr_lo, fr_hi);
                }
                // ... and is final: kept in LDS till the traceback (halves still running keep their packed state untouched)
                if (fin_lo) { st_best[0][lane] = b_lo.score; st_best[1][lane] = b_lo.I; st_best[2][lane] = b_lo.J; st_best[3][lane] = b_lo.tie; }
                if (fin_hi) { st_best[4][lane] = b_hi.score; st_best[5][lane] = b_hi.I; st_best[6][lane] = b_hi.J; st_best[7][lane] = b_hi.tie; }
            }
            top = topn;
            if (j <= shmax) reach_column0(j);
        }
        if constexpr (CHECK) {
            int hi = hlo(vmax) > hhi(vmax) ? hlo(vmax) : hhi(vmax), lo = hlo(vmin) < hhi(vmin) ? hlo(vmin) : hhi(vmin);
            hi = hi > tab_max ? hi : tab_max; lo = lo < tab_min ? lo : tab_min;
            if ((have_lo || have_hi) && nmax > 0) {
                atomicMax((int *)a.err + 4, hi); atomicMax((int *)a.err + 5, -lo);
                if (hi > pcb::kF16Limit || -lo > pcb::kF16Limit) atomicAdd(a.err, 1u);   // the on-device assertion
            }
        }
        Best b_lo, b_hi;
        load_best(b_lo, b_hi);
        if constexpr (SCORE) {
            // the end cell as a score record (what PC_MODE_SCORE leaves: plan_kernel's end_records)
            if (have_lo) { int4 *o = (int4 *)(a.out + p_lo * TRACE_OUT_INTS); o[0] = make_int4(-2, b_lo.J, b_lo.I, 0); o[1] = make_int4(b_lo.score, 0, 0, 0); }
            if (have_hi) { int4 *o = (int4 *)(a.out + p_hi * TRACE_OUT_INTS); o[0] = make_int4(-2, b_hi.J, b_hi.I, 0); o[1] = make_int4(b_hi.score, 0, 0, 0); }
            continue;
        }
        if (a.debug & 1) {
            if (have_lo) a.out[p_lo * TRACE_OUT_INTS + 4] = b_lo.score;
            if (have_hi) a.out[p_hi * TRACE_OUT_INTS + 4] = b_hi.score;
            continue;
        }
        if (a.walk_req) {        // the walks run in a launch of their own (walk_kernel): leave the end cells
            ((int4 *)a.walk_req)[((int64_t)t * 2 + 0) * 64 + lane] = make_int4(b_lo.score, b_lo.I, b_lo.J, b_lo.tie);
            ((int4 *)a.walk_req)[((int64_t)t * 2 + 1) * 64 + lane] = make_int4(b_hi.score, b_hi.I, b_hi.J, b_hi.tie);
            if (lane == 0) a.walk_req_tile[t] = notrace_upto;
            continue;
        }
#ifdef PC_SLAB_OLD
        constexpr bool kGrouped = false;
#else
        constexpr bool kGrouped = true;
#endif
        traceback_pairs<CHECK, kGrouped>(a, slab, R, NW, lane, b_lo, b_hi, pad_lo, pad_hi,
                               have_lo, have_hi, n_lo, n_hi, c0_lo, c0_hi, m_lo, m_hi, p_lo, p_hi, notrace_upto,
                               w_lo, w_hi, tile.adapter_lo, tile.adapter_hi);
    }
}
#undef PC_ROW16_FULL
#undef PC_ROW16_FIRST
#undef PC_ROW16_NOIND

// ---------------------------------------------------------------------------------------------
// The tracebacks of a traced launch as a launch of their own: one block (wave) per tile of that launch, tile t's slab at
// block t's place.  A walk is a chain of dependent loads from memory no cache holds; inside the scan kernel a wave sits
// through 40-60 of them with its 128 VGPRs of column state idle; here the waves are small (many per SIMD) and run beside the
// next launch's scan.  Same code (traceback_pairs), same records.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void walk_kernel(ScanArgs a, int rows)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const Tile tile = a.tiles[t];
    const int NW = (rows + 3) >> 2;
    const int m_lo = a.ad_len[tile.adapter_lo], m_hi = a.ad_len[tile.adapter_hi];
    const int pad_lo = rows - m_lo, pad_hi = rows - m_hi;
    const bool have_lo = lane < tile.count_lo, have_hi = lane < tile.count_hi;
    const int64_t p_lo = (a.perm && have_lo) ? a.perm[tile.out_lo + lane] : tile.out_lo + lane;
    const int64_t p_hi = (a.perm && have_hi) ? a.perm[tile.out_hi + lane] : tile.out_hi + lane;
    const int64_t wi_lo = a.win_by_out ? p_lo : tile.win_lo + lane;
    const int64_t wi_hi = a.win_by_out ? p_hi : tile.win_hi + lane;
    const int n_lo = have_lo ? a.win_len[wi_lo] : 0, n_hi = have_hi ? a.win_len[wi_hi] : 0;
    const int c0_lo = (have_lo && a.col0) ? a.col0[p_lo] : 0, c0_hi = (have_hi && a.col0) ? a.col0[p_hi] : 0;
    const int4 r_lo = ((const int4 *)a.walk_req)[((int64_t)t * 2 + 0) * 64 + lane], r_hi = ((const int4 *)a.walk_req)[((int64_t)t * 2 + 1) * 64 + lane];
    const Best b_lo = {r_lo.x, r_lo.y, r_lo.z, r_lo.w}, b_hi = {r_hi.x, r_hi.y, r_hi.z, r_hi.w};
    const u32 *slab = a.slab + (int64_t)t * a.slab_stride;
    traceback_pairs<false, true>(a, slab, rows, NW, lane, b_lo, b_hi, pad_lo, pad_hi, have_lo, have_hi, n_lo, n_hi, c0_lo, c0_hi, m_lo, m_hi,
                                 p_lo, p_hi, a.walk_req_tile[t]);
}

int launch_walk(const ScanArgs &a, int rows, int ntiles, void *stream)
{
    if (ntiles <= 0) return 0;
    hipLaunchKernelGGL(walk_kernel, dim3((unsigned)ntiles), dim3(64), 0, (hipStream_t)stream, a, rows);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// Planner between the two passes of a whole-read scan: bounded window ending at the max cell.
// ---------------------------------------------------------------------------------------------
__global__ void plan_kernel(PlanArgs a)
{
    const Tile tile = a.tiles[blockIdx.x];
    const int i = threadIdx.x & 63, hi = threadIdx.x >> 6;
    const bool have = i < (hi ? tile.count_hi : tile.count_lo);
    const int64_t slot = (hi ? tile.out_hi : tile.out_lo) + i;
    const int64_t p = (a.perm && have) ? a.perm[slot] : slot;    // pair (output index); with a.perm another one of the slot's segment
    const int64_t w = (hi ? tile.win_hi : tile.win_lo) + i + (p - slot);   // its window (windows and pairs of a segment run in step)
    __shared__ int tile_len;                                     // end-aligned tiles: the columns the tile's windows get
    if (threadIdx.x == 0) tile_len = 0;
    __syncthreads();
    int score = 0, I = 0, J = 0;
    if (have) {
        // merge the per-chunk maxima in the reference's visiting order (strict '>', earlier chunk wins)
        int nch = a.chunks > 1 ? a.chunks : 1;
        if (nch > 1 && a.chunk_len > 0) {             // the chunks that hold columns of this window (chunk 0 always)
            const int real = (a.win_len[w] + a.chunk_len - 1) / a.chunk_len;
            nch = real < 1 ? 1 : (real < nch ? real : nch);
        }
        const int nstride = a.chunks > 1 ? a.chunks : 1;
        if (a.end_records) {
            // PC_MODE_TRACE_AT: the caller's score record of this very pair, (-2, J, I, 0, score, ...)
            const int4 r0 = ((const int4 *)(a.end_records + p * TRACE_OUT_INTS))[0];
            score = a.end_records[p * TRACE_OUT_INTS + 4]; I = r0.z; J = r0.y;
            if (r0.x != -2 || J < 0 || J > a.win_len[w] || I < 0) {       // not a score record of this window: loud, never guessed
                atomicAdd(a.err, 1u);
                I = 0; J = 0;
            }
        } else {
            score = a.k1[(p * nstride) * 4 + 0]; I = a.k1[(p * nstride) * 4 + 1]; J = a.k1[(p * nstride) * 4 + 2];
            for (int c = 1; c < nch; ++c) {
                const int sc = a.k1[(p * nstride + c) * 4 + 0];
                if (sc > score) { score = sc; I = a.k1[(p * nstride + c) * 4 + 1]; J = a.k1[(p * nstride + c) * 4 + 2]; }
            }
        }
    }
    if (a.score_out) {       // score-only request: the end cell and its score are the whole answer
        if (have) {
            int4 *o = (int4 *)(a.score_out + p * TRACE_OUT_INTS);
            o[0] = make_int4(-2, J, I, 0);
            o[1] = make_int4(score, 0, 0, 0);
        }
        return;
    }
    int window = a.ad_window[hi ? tile.adapter_hi : tile.adapter_lo];
    if (a.window_cap > 0 && window > a.window_cap) window = a.window_cap;
    if (a.end_align) {
        // (a longer warm-up is still exact; the lead-in columns are garbage the kernel discards when it reaches
        // the read's column 0 -- they only have to be readable: not before the arena's first byte)
        const int wl = a.ad_window[tile.adapter_lo], wh = a.ad_window[tile.adapter_hi];
        window = wl > wh ? wl : wh;
        if (a.window_cap > 0 && window > a.window_cap) window = a.window_cap;
        // The tile's windows all get the length of the longest one any of its pairs needs -- a pair whose end cell lies J
        // columns into its read needs min(J, window): its read's column 0 is exact by itself -- not the full window: tiles
        // of pairs that end early (adapters at the start of their windows; callers hand such pairs over together) run that
        // many columns instead of `window` of which most would be lead-in.
        if (have && J > 0) atomicMax(&tile_len, J < window ? J : window);
    }
    __syncthreads();
    if (!have) return;
    int c0 = J - window;
    if (a.end_align && J > 0) {             // (J == 0: the end cell is the corner (m, 0) -- nothing to run, nothing to align)
        c0 = J - tile_len;
        const int64_t room = a.win_off[w];
        if (c0 < 0 && (int64_t)(-c0) > room) c0 = -(int)room;
    } else if (c0 < 0) c0 = 0;
    a.win_off2[p] = a.win_off[w] + c0;
    a.win_len2[p] = J - c0;
    a.col02[p] = c0;
    a.ntot2[p] = a.win_len[w];
    a.force_row2[p] = I;
    a.force_score2[p] = score;
    if (a.trace_cols2) {
        // Columns the traced path can touch left of J, from THIS pair's end cell: the path is an optimal one of score
        // `score` >= 0 that ends in adapter row I, so it has at most I diagonal steps earning at most `match` each, and every
        // gap character costs at least g = min(|open|, |extend|): read-gap columns <= (match I - score) / g, columns touched
        // <= I + that (pc_bounds.h derives the adapter-wide W the same way from I <= m, score >= 0).  + 2: a column's trace
        // bits depend on the column before it.  A walk that leaves the traced columns is flagged by the kernel (never expected).
        const int slack = a.match * I - score;
        a.trace_cols2[p] = I + (slack > 0 ? slack / a.gap_unit : 0) + 2;
    }
}

// ---------------------------------------------------------------------------------------------
// The real units of a chunked score pass: tile t has ceil(longest window / chunk_len) chunks that hold columns (at least
// one), the others would only be drawn -- one atomic each on ONE counter, 7 ns apiece -- to find nothing.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void unit_real_kernel(const Tile *tiles, int ntiles, const int32_t *win_len, int chunk_len, int32_t *real)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= ntiles) return;
    const Tile tile = tiles[t];
    int n = 0;
    if (lane < tile.count_lo) n = win_len[tile.win_lo + lane];
    if (lane < tile.count_hi) { const int h = win_len[tile.win_hi + lane]; n = h > n ? h : n; }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const int o = __shfl_xor(n, s); n = o > n ? o : n; }
    if (lane == 0) { const int r = (n + chunk_len - 1) / chunk_len; real[t] = r < 1 ? 1 : r; }
}

__global__ __launch_bounds__(1024) void unit_scan_kernel(const int32_t *real, int ntiles, int32_t *prefix)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (ntiles + 1023) / 1024, b = tid * per, e = b + per < ntiles ? b + per : ntiles;
    int sum = 0;
    for (int i = b; i < e; ++i) sum += real[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = acc; acc += v; } prefix[ntiles] = acc; }
    __syncthreads();
    int acc = part[tid];
    for (int i = b; i < e; ++i) { prefix[i] = acc; acc += real[i]; }
}

int launch_unit_prefix(const Tile *tiles, int ntiles, const int32_t *win_len, int chunk_len, int32_t *real, int32_t *prefix, void *stream)
{
    if (ntiles <= 0 || chunk_len <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(unit_real_kernel, dim3((unsigned)ntiles), dim3(64), 0, s, tiles, ntiles, win_len, chunk_len, real);
    hipLaunchKernelGGL(unit_scan_kernel, dim3(1), dim3(1024), 0, s, real, ntiles, prefix);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int trace_words_per_col(int rows) { return (rows + 3) / 4; }

template <bool TRACE>
static int launch_scan(const ScanArgs &a0, int rows, bool pad, int grid, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    ScanArgs a = a0;
    a.one2 = 0x00010001u; a.two2 = 0x00020002u; a.sixteen2 = 0x00100010u;
#define PC_EXACT(RR) case RR: hipLaunchKernelGGL((scan_kernel<RR, false, TRACE>), dim3(grid), dim3(64), 0, s, a); break;
#define PC_PADDED(RR) case RR: hipLaunchKernelGGL((scan_kernel<RR, true, TRACE>), dim3(grid), dim3(64), 0, s, a); break;
    if (rows == 0) {
        // generic: dynamic LDS = per-row constants + the DP column of 64 lanes
        const size_t lds = (size_t)a.gen_max_rows * 8 + (size_t)a.gen_max_rows * 64 * 8;
        if (hipFuncSetAttribute((const void *)scan_kernel<0, true, TRACE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return -2;
        hipLaunchKernelGGL((scan_kernel<0, true, TRACE>), dim3(grid), dim3(64), lds, s, a);
    } else if (!pad) {
        switch (rows) {
            PC_EXACT(22) PC_EXACT(24) PC_EXACT(28) PC_EXACT(32)
            default: return -1;
        }
    } else {
        switch (rows) {
            PC_PADDED(16) PC_PADDED(20) PC_PADDED(24) PC_PADDED(26) PC_PADDED(28) PC_PADDED(30) PC_PADDED(32) PC_PADDED(34)
            PC_PADDED(36) PC_PADDED(38) PC_PADDED(40) PC_PADDED(48) PC_PADDED(56) PC_PADDED(64) PC_PADDED(68) PC_PADDED(72)
            PC_PADDED(112) PC_PADDED(128)
            default: return -1;
        }
    }
#undef PC_EXACT
#undef PC_PADDED
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_trace(const ScanArgs &a, int rows, bool pad, int grid, void *stream) { return launch_scan<true>(a, rows, pad, grid, stream); }

bool trace16_has(int rows)
{
    for (int r : kTrace16Rows) if (r == rows) return true;
    return false;
}

int launch_trace16(const ScanArgs &a, int rows, int grid, void *stream, bool score_only)
{
    hipStream_t s = (hipStream_t)stream;
    if (score_only) {
#define PC_T16S(RR) case RR: hipLaunchKernelGGL((trace16_kernel<RR, false, true>), dim3(grid), dim3(64), 0, s, a); return hipGetLastError() == hipSuccess ? 0 : -2;
        switch (rows) {
            PC_T16S(16) PC_T16S(20) PC_T16S(22) PC_T16S(24) PC_T16S(26) PC_T16S(28) PC_T16S(30) PC_T16S(32) PC_T16S(34) PC_T16S(36)
            PC_T16S(38) PC_T16S(40) PC_T16S(48) PC_T16S(56) PC_T16S(64) PC_T16S(68) PC_T16S(72)
            default: return -1;
        }
#undef PC_T16S
    }
    if (a.debug & 4) {      // range-checking build, where instantiated
#define PC_T16C(RR) case RR: hipLaunchKernelGGL((trace16_kernel<RR, true>), dim3(grid), dim3(64), 0, s, a); return hipGetLastError() == hipSuccess ? 0 : -2;
        switch (rows) {
            PC_T16C(16) PC_T16C(20) PC_T16C(22) PC_T16C(24) PC_T16C(26) PC_T16C(28) PC_T16C(30) PC_T16C(32) PC_T16C(34) PC_T16C(36)
            PC_T16C(38) PC_T16C(40) PC_T16C(48) PC_T16C(56) PC_T16C(64) PC_T16C(68) PC_T16C(72)
            default: break;
        }
#undef PC_T16C
    }
#define PC_T16(RR) case RR: hipLaunchKernelGGL((trace16_kernel<RR>), dim3(grid), dim3(64), 0, s, a); break;
    switch (rows) {
        PC_T16(16) PC_T16(20) PC_T16(22) PC_T16(24) PC_T16(26) PC_T16(28) PC_T16(30) PC_T16(32) PC_T16(34) PC_T16(36)
        PC_T16(38) PC_T16(40) PC_T16(48) PC_T16(56) PC_T16(64) PC_T16(68) PC_T16(72)
        default: return -1;
    }
#undef PC_T16
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_score(const ScanArgs &a, int rows, bool pad, int grid, void *stream) { return launch_scan<false>(a, rows, pad, grid, stream); }

// ---------------------------------------------------------------------------------------------
// Pairs of a segment by end column (see BucketArgs).  Three small launches: per-(segment, bucket) counts, their prefix, the
// scatter.  A block takes one piece of one segment: LDS histogram, then one atomic per non-empty bucket.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int bucket_of(const int32_t *records, int64_t pair)
{
    const int2 v = *(const int2 *)(records + pair * TRACE_OUT_INTS);        // (flag, J)
    if (v.x != -2 || v.y < 0) return kBuckets - 1;                          // not a score record: among the longest
    const int b = v.y / kBucketWidth;
    return b < kBuckets ? b : kBuckets - 1;
}

__global__ __launch_bounds__(256) void bucket_count_kernel(BucketArgs a)
{
    __shared__ unsigned hist[kBuckets];
    const BucketBlock blk = a.blocks[blockIdx.x];
    if (threadIdx.x < kBuckets) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < blk.count; i += 256) atomicAdd(&hist[bucket_of(a.records, blk.first + i)], 1u);
    __syncthreads();
    if (threadIdx.x < kBuckets && hist[threadIdx.x]) atomicAdd(a.counts + (int64_t)blk.segment * kBuckets + threadIdx.x, hist[threadIdx.x]);
}

__global__ __launch_bounds__(256) void bucket_prefix_kernel(BucketArgs a)
{
    const int sgm = blockIdx.x * 256 + threadIdx.x;
    if (sgm >= a.nsegments) return;
    unsigned acc = 0;
    for (int b = 0; b < kBuckets; ++b) {        // counts -> first position of the bucket within its segment
        const unsigned c = a.counts[(int64_t)sgm * kBuckets + b];
        a.counts[(int64_t)sgm * kBuckets + b] = acc;
        acc += c;
    }
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(BucketArgs a)
{
    __shared__ unsigned hist[kBuckets], base[kBuckets];
    const BucketBlock blk = a.blocks[blockIdx.x];
    if (threadIdx.x < kBuckets) hist[threadIdx.x] = 0;
    __syncthreads();
    int mine[kBucketBlock / 256];
#pragma unroll
    for (int k = 0; k < kBucketBlock / 256; ++k) {
        const int i = threadIdx.x + 256 * k;
        mine[k] = i < blk.count ? bucket_of(a.records, blk.first + i) : -1;
        if (mine[k] >= 0) atomicAdd(&hist[mine[k]], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kBuckets) {
        const unsigned h = hist[threadIdx.x];
        base[threadIdx.x] = a.counts[(int64_t)blk.segment * kBuckets + threadIdx.x] +
                            (h ? atomicAdd(a.cursors + (int64_t)blk.segment * kBuckets + threadIdx.x, h) : 0u);
        hist[threadIdx.x] = 0;
    }
    __syncthreads();
    const int64_t seg0 = a.seg_first[blk.segment];
#pragma unroll
    for (int k = 0; k < kBucketBlock / 256; ++k) {
        if (mine[k] < 0) continue;
        const unsigned r = atomicAdd(&hist[mine[k]], 1u);
        a.perm[seg0 + base[mine[k]] + r] = blk.first + threadIdx.x + 256 * k;
    }
}

int launch_bucket_pairs(const BucketArgs &a, void *stream)
{
    if (a.nblocks <= 0 || a.nsegments <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const size_t bytes = (size_t)a.nsegments * kBuckets * 4;
    if (hipMemsetAsync(a.counts, 0, bytes, s) != hipSuccess || hipMemsetAsync(a.cursors, 0, bytes, s) != hipSuccess) return -2;
    hipLaunchKernelGGL(bucket_count_kernel, dim3((unsigned)a.nblocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bucket_prefix_kernel, dim3((unsigned)((a.nsegments + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3((unsigned)a.nblocks), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_plan(const PlanArgs &a, void *stream)
{
    if (a.ntiles <= 0) return 0;
    hipLaunchKernelGGL(plan_kernel, dim3((unsigned)a.ntiles), dim3(128), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace pck
