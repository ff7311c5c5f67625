// pc_bounds.h -- score-scheme preconditions and the exact window bound of the two-pass scan.
//
// The reference allocates a full (|read|+1) x (|adapter|+1) trace matrix
// (seqan/align/dp_algorithm_impl.h:1547-1569).  For whole-read ("middle") scans this library
// instead (1) runs a score-only forward pass that finds the reference's max cell (I,J), then
// (2) recomputes with trace only columns [J - window, J].  That is exact, not heuristic:
//
//  * trace span W.  The traced path is an optimal path with score = best >= 0 (the cell
//    (m,0)=0 is always tracked).  Every gap character costs at least g = min(|open|,|ext|),
//    at most m diagonals earn at most `match` each, so #read-gap columns <= match*m/g and the
//    path touches at most  W = m + floor(match*m/g)  read columns left of J.
//
//  * warm-up span SPAN.  Column c of the window DP is started from a state every entry of
//    which is the score of a real path (row-0 free start followed by a vertical gap), so all
//    window values are <= the true ones, and equal as soon as the true optimum for that cell
//    is reached by a path starting inside the window.  Any path spanning s read columns
//    scores <= match*m - g*(s - m); every cell state has a trivial in-column alternative
//    scoring >= -(2|open| + (m-1)|ext|).  Hence paths with
//        s > m + (match*m + 2|open| + (m-1)|ext|)/g
//    are never optimal and all three states of every cell are exact from column
//    c + SPAN on,  SPAN = m + floor((match*m + 2|open| + (m-1)|ext|)/g) + 1.
//
//  * trace bits of column k depend on columns k-1 and k, so  window = W + SPAN + 1.
//
// Preconditions (checked, not assumed): match > 0, match > mismatch, open < 0, ext < 0, and every
// DP value must fit the packed int16 lanes the kernels compute in.  open == ext is the
// reference's linear-gap dispatch (seqan/align/global_alignment_unbanded.h:217-220), see below.
#pragma once
#include <stdint.h>

namespace pcb {

constexpr int MAX_ADAPTER = 128;     // rows the kernels can keep in registers
constexpr int NEG16 = -16384;        // "-infinity" of the int16 lanes; never wins a max

struct Bounds {
    int W;        // columns the traced path can span left of J
    int SPAN;     // warm-up columns before values are exact
    int window;   // W + SPAN + 1
};

// gap_open == gap_extend selects the reference's LINEAR-gap recurrence (one matrix; ties prefer
// diagonal, then vertical; no _correctTraceValue).  It is the affine recurrence with extension
// made impossible: with H_ext = V_ext = "-infinity" every gap cell is an "open" from M, the
// M-origin decisions (d >= max(H,V); V >= H) are exactly the linear ones, and the traceback takes
// one gap cell per step.  The kernels therefore run the same code with kLinearExtend.
constexpr int kLinearExtend = -12000;

inline bool is_linear(int gap_open, int gap_extend) { return gap_open == gap_extend; }

inline bool scores_supported(int match, int mismatch, int gap_open, int gap_extend, int max_m)
{
    if (!(match > 0 && match > mismatch && gap_open < 0 && gap_extend < 0)) return false;
    const long D = (long)match - mismatch;
    if (6 * D > 16000) return false;                         // spaced base codes 0..5*D in u16
    // linear mode adds -12000 to live values once per cell: keep them within +-4000
    const long lim = is_linear(gap_open, gap_extend) ? 4000 : 8000;
    if ((long)match * max_m > lim) return false;             // upper range of M
    if (2L * -gap_open + (long)max_m * -gap_extend > lim) return false;   // lower range of M,H,V
    if (-mismatch > lim || -gap_open > lim || -gap_extend > lim) return false;
    return true;
}

// Packed-FP16 traced kernel (pc_kernels.hip, trace16_kernel): every DP value X of register row
// rho (1..R), jj columns into the window, is held as the fp16 number  X + (rho + jj [+1]) * eps - C,
// eps = -gap_extend (drifting coordinates: gap extensions cost nothing).  fp16 represents every
// integer of magnitude <= 2048 exactly and adds them exactly, so the kernel is bit-exact as long as
// every value ever formed stays within +-kF16Limit.  True values: M <= match*R; the lowest finite
// value is a T = M + open of a lower-bound start state or a diagonal off it,
//   low = 3*open + (R-1)*ext + min(mismatch, 0)   (conservative),
// C puts `low` at -kF16Limit, and the window may then have at most max_cols columns before the
// drift reaches +kF16Limit.  ok = the scheme and R leave at least 32 columns.
constexpr int kF16Limit = 2040;
struct F16Plan { bool ok; int cen; int max_cols; };
inline F16Plan f16_plan(int match, int mismatch, int gap_open, int gap_extend, int R)
{
    F16Plan p = {false, 0, 0};
    if (is_linear(gap_open, gap_extend)) return p;           // linear mode replaces gap_extend by -12000
    if (!scores_supported(match, mismatch, gap_open, gap_extend, R)) return p;
    const long eps = -(long)gap_extend;
    const long low = 3L * gap_open + (long)(R - 1) * gap_extend + (mismatch < 0 ? mismatch : 0);
    const long high = (long)match * R;
    if (-mismatch > 1000 || -gap_open > 1000 || match > 1000 || eps > 500) return p;
    // the substitution terms sub - open + eps of the kernel's table must be exact fp16 integers too
    if (match - gap_open + eps > kF16Limit || mismatch - gap_open + eps > kF16Limit || mismatch - gap_open + eps < -kF16Limit) return p;
    // the tracked last-row term  T~(R,j) - top~(j) = M(R,j) + R*eps  is formed in fp16 as well (the scout's packed
    // compare): it must be an exact integer too.  (Found by tools/fuzz_parity.py: match 29, a 64-base adapter copied
    // exactly into a 150-column window -- 1856 + 448 = 2304 is not an fp16 integer, the score came out one off and
    // the walk's score check flagged the pair.)
    if (high + (long)R * eps > kF16Limit) return p;
    const long cols = (2L * kF16Limit - (high - low)) / eps - R - 2;
    if (cols < 32) return p;
    p.ok = true;
    p.cen = (int)(low + kF16Limit);
    p.max_cols = (int)(cols > (1 << 20) ? (1 << 20) : cols);
    return p;
}

// Run-time specialised score kernel (pc_jit_source.h): the same drifting coordinates, renormalised (shifted back
// down) every `kren` columns, values X held as X + (rho + jj [+1]) * eps - cen.  f16: the packed-fp16 variant
// (5 ops per cell pair) is usable -- every value formed, the tracked term M + R*eps included, an exact fp16
// integer -- otherwise packed int16 (6 ops); ok = false: the scheme leaves no useful period (generic kernels).
struct SpecPlan { bool ok, f16; long kren, cen; };
inline SpecPlan spec_plan(int match, int mismatch, int gap_open, int gap_extend, int R, bool force_int16)
{
    SpecPlan p = {false, false, 0, 0};
    const long eps = -(long)gap_extend;
    const long low = 2L * gap_open + (long)(R - 1) * gap_extend < (long)gap_open + (long)(R - 1) * gap_extend + mismatch
                         ? 2L * gap_open + (long)(R - 1) * gap_extend
                         : (long)gap_open + (long)(R - 1) * gap_extend + mismatch;
    const long low2 = low < (long)gap_open ? low : (long)gap_open;
    const long high = (long)match * R;
    auto period = [&](long lim) -> long {
        const long k = (2 * lim - (high - low2) - (long)(R + 6) * eps) / eps;
        return k < 0 ? 0 : (k / 4 * 4 < (1L << 20) ? k / 4 * 4 : (1L << 20));
    };
    p.f16 = !force_int16 && period(kF16Limit) >= 64 && high + (long)R * eps <= kF16Limit && -mismatch <= 1000 && -gap_open <= 1000;
    const long lim = p.f16 ? kF16Limit : 32000;
    p.kren = period(lim);
    p.ok = p.kren >= 64;
    p.cen = low2 + lim;
    return p;
}

inline bool compute_bounds(int match, int mismatch, int gap_open, int gap_extend, int m, Bounds &b)
{
    if (!scores_supported(match, mismatch, gap_open, gap_extend, m > 0 ? m : 1)) return false;
    const int go = -gap_open, ge = -gap_extend;
    const int g = go < ge ? go : ge;
    b.W = m + (match * m) / g;
    b.SPAN = m + (match * m + 2 * go + (m - 1) * ge) / g + 1;
    b.window = b.W + b.SPAN + 1;
    return true;
}

}  // namespace pcb
