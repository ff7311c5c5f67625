// pc_walk.h -- traceback + digest of one alignment, written once for device and host.
//
// Replaces, for the 4-bit trace this library stores, what the reference does in
//   seqan/align/dp_algorithm_impl.h:1352-1369  (_correctTraceValue)
//   seqan/align/dp_traceback_impl.h:376-552    (_doTraceback / _computeTraceback, GapsLeft)
//   seqan/align/dp_traceback_adaptor.h:57-117  (segments -> two gapped rows)
//   porechop/src/alignment.cpp:6-111           (ScoredAlignment: overlap window, identities)
// but never materialises the gapped rows: the seven output fields are functions of a handful
// of counters that can be maintained while walking the path from its end to its start
// (derivation in DESIGN.md "digest without strings").
//
// Trace nibble of an interior cell (adapter row i>=1, read column j>=1):
//   bit0 HOPEN   : H[i][j] was opened from M[i][j-1]   (strict: extension < open)
//   bit1 VOPEN   : V[i][j] was opened from M[i-1][j]
//   bit2 FROMH   : max(H,V) came from H                (strict: V < H)
//   bit3 NOTDIAG : M[i][j] = max(H,V) > diagonal       (strict; tie -> diagonal)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PC_HD __host__ __device__ __forceinline__
#else
#define PC_HD inline
#endif

namespace pcw {

enum : int { NIB_HOPEN = 1, NIB_VOPEN = 2, NIB_FROMH = 4, NIB_NOTDIAG = 8 };

// the reference's trace byte values (seqan/align/dp_profile.h:142-156)
enum : int { T_NONE = 0, T_DIAG = 1, T_H = 2, T_V = 4, T_HOPEN = 8, T_VOPEN = 16, T_MAXH = 32, T_MAXV = 64 };

// Result record written per pair (8 x int32).
struct Digest {
    int32_t read_start, read_end, adapter_start, adapter_end, score;
    int32_t matches, aligned_len, full_len;
};

// The traceback as a resumable state machine: every call of consume() takes the trace nibble of
// ONE cell -- the cell (col,row) the state asked for -- and leaves in (col,row) the next cell it
// needs, or done.  That is the reference's _doTraceback loop (dp_traceback_impl.h:376-450) cut at
// its trace-matrix reads, so that a caller can fetch for several independent walks at once (the
// kernels run a lane's two pairs side by side: two load chains in flight instead of one).
//
// The gapped rows are never built, and almost nothing is counted per step.  With the alignment's
// columns written left to right as
//        H^a  V^b  P  H^c  V^d        (H = read base over '-', V = '-' over adapter base, D = both)
//   a = read bases before the path (col0 + its first column), b = adapter rows above its first cell
//   (a or b is 0: the path starts in row 0 or in column 0), P = the path, c = read bases after the end
//   cell, d = adapter rows below it (c or d is 0: the end cell is in the last row or the last column),
// ScoredAlignment's quantities (alignment.cpp:23-111) depend on P only through its length, its
// matches and its FIRST run seen from the end (type, length, type of what follows):
//   * `start` (first column where both rows have shown a base) is column a + b: after H^a the next
//     column has an adapter base (V^b is empty then, and a path leaving row 0 starts D or V), after
//     V^b the next has a read base.  readStart = a if b == 0 else 0, adapterStart = b if a == 0 else 0.
//   * `end` (same from the right): skip the one-sided columns at the right end -- V^d, or H^c, each
//     possibly continued by a leading run of the same kind in P -- the next column Y closes it.
//     Bases consumed from the right up to and including Y give readEnd / adapterEnd.
//   * aligned length = all columns - columns left of `start` - columns right of `end`;
//     full-adapter length = all columns - columns left of the first adapter base - right of the last.
//
// (I,J) is the end cell in LOCAL columns (J in [0, ncols]); col0 is the global column of local 0;
// n_total the whole read length; tie_fix: 0 none, 1 force "from V", 2 force "from H"
// (the _correctTraceValue outcome, decided by the kernel from d==max(H,V) at the end cell).
// err = 1 if the walk ran into the left edge of a window that does not start at the read's
// column 0 (impossible when the window obeys the bound in pc_bounds.h; reported loudly).
struct Walk {
    enum : int { DISPATCH = 0, VRUN = 1, HRUN = 2 };
    int col, row, mode, first, tie_fix, done, err;
    int cmin;                                     // local column of the read's column 0 (0, or the lead-in of an end-aligned window)
    int m, col0, n_total, score, I, J;
    int ndiag, nopen, nsteps;                     // diagonal steps, gap steps that OPENED their gap, all steps
    int matches_seen;                             // host cross-check only (consume())
    int cur_type, cur_len, changed;               // the run being walked, until it first changes
    int first_type, first_len, second_type;       // the first run seen from the end, and what follows it

    // cmin_ > 0: the window was given a lead-in of cmin_ columns before the read's column 0 (col0_ = -cmin_) so
    // that it ends where the other windows of its tile end; the path stops at local column cmin_
    PC_HD void start(int I_, int J_, int m_, int col0_, int n_total_, int score_, int tie_fix_, int cmin_ = 0) {
        col = J_; row = I_; I = I_; J = J_; mode = DISPATCH; first = 1; tie_fix = tie_fix_; err = 0; cmin = cmin_;
        m = m_; col0 = col0_; n_total = n_total_; score = score_;
        ndiag = 0; nopen = 0; nsteps = 0; matches_seen = 0; cur_type = 0; cur_len = 0; changed = 0; first_type = 0; first_len = 0; second_type = 0;
        done = !(col > cmin && row > 0);
    }
    // nib: trace nibble of cell (col,row); eq: read base (col-1) == adapter base (row-1)
    PC_HD void consume(int nib, bool eq) {
        const int before = ndiag;
        step(nib, 1);
        matches_seen += (ndiag != before && eq) ? 1 : 0;
    }
    // The same step with a predicate (active = 1/0) instead of a branch: the kernels advance a lane's
    // two walks in one loop and a finished walk just stops changing (flags are 0/1 ints throughout,
    // so that the device code is selects, not divergent branches).
    // Matches are not counted along the way (that would take the read base and the adapter base of
    // every diagonal cell): the path realises the end cell's score, so with D diagonal steps, O gap
    // steps that opened their gap and E that extended one,
    //     score = match * M + mismatch * (D - M) + O * gap_open + E * gap_extend
    // gives M exactly (finish()).  A gap step opened its gap iff the open bit of the cell it leaves is
    // set -- the bit the traceback itself follows -- which also covers schemes where two opens are
    // cheaper than open + extend, and the linear-gap mode (every gap step is an "open").
    PC_HD void step(int nib, int active) {
        int notdiag = (nib >> 3) & 1, fromh = (nib >> 2) & 1;
        const int vopen = (nib >> 1) & 1, hopen = nib & 1;
        // _correctTraceValue + "prefer the gap at the end" (dp_algorithm_impl.h:1352-1369,
        // dp_traceback_impl.h:456-483): both only redirect the end cell's own decision
        const int f1 = first & (tie_fix == 1 ? 1 : 0), f2 = first & (tie_fix == 2 ? 1 : 0);
        notdiag |= f1 | f2;
        fromh = (fromh & (f1 ^ 1)) | f2;
        first &= active ^ 1;
        // which step leaves this cell: inside a gap run the run's kind, otherwise M's origin
        const int disp = mode == DISPATCH ? 1 : 0;
        const int is_d = disp & (notdiag ^ 1);
        const int is_v = disp ? (notdiag & (fromh ^ 1)) : (mode == VRUN ? 1 : 0);
        const int t = is_d ? T_DIAG : (is_v ? T_V : T_H);
        ndiag += active & is_d;
        nopen += active & (is_d ^ 1) & (is_v ? vopen : hopen);
        nsteps += active;
        // the first run seen from the end
        const int same = (cur_type == 0 ? 1 : 0) | (t == cur_type ? 1 : 0);
        const int track = active & (changed ^ 1);
        const int ext = track & same, chg = track & (same ^ 1);
        first_type = chg ? cur_type : first_type;
        first_len = chg ? cur_len : first_len;
        second_type = chg ? t : second_type;
        changed |= chg;
        cur_type = ext ? t : cur_type;
        cur_len += ext;
        row -= active & (t != T_H ? 1 : 0);
        col -= active & (t != T_V ? 1 : 0);
        // a gap run goes on while the cell being left extended its gap (its open bit is clear) and the
        // run has not reached row / column 0; after its opening step the dispatch restarts (GapsLeft)
        const int go_v = (t == T_V ? 1 : 0) & (vopen ^ 1) & (row >= 1 ? 1 : 0);
        const int go_h = (t == T_H ? 1 : 0) & (hopen ^ 1) & (col >= cmin + 1 ? 1 : 0);
        const int nmode = go_v ? VRUN : (go_h ? HRUN : DISPATCH);
        mode = active ? nmode : mode;
        const int ndone = (nmode == DISPATCH ? 1 : 0) & ((col > cmin && row > 0) ? 0 : 1);
        done = active ? ndone : done;
    }
    // match, mismatch, gap_open, gap_extend: the scheme's real scores (linear mode: gap_extend = gap_open)
    PC_HD int finish(Digest &out, int match, int mismatch, int gap_open, int gap_extend) {
        const int gaps = nsteps - ndiag;
        const int num = score - mismatch * ndiag - nopen * gap_open - (gaps - nopen) * gap_extend;
        const int matches = num / (match - mismatch);
        if (num != matches * (match - mismatch) || matches < 0 || matches > ndiag) err = 1;   // the path does not realise the score
        return finish_with(out, matches);
    }
    // The digest with the matches COUNTED along the path (consume() with the bases): what the plain-coordinate kernel of
    // pc_slow.h uses -- valid for every scoring scheme, match == mismatch included, where the score equation above has no
    // unique solution.
    PC_HD int finish_counted(Digest &out) { return finish_with(out, matches_seen); }
    PC_HD int finish_with(Digest &out, int matches) {
        if (row > 0 && col == cmin && col0 + cmin > 0) err = 1;   // left the window: bound violated
        if (!changed) { first_type = cur_type; first_len = cur_len; second_type = 0; }
        const int a = col0 + col, b = row;                    // head: read bases / adapter rows before the path
        const int c = n_total - (col0 + J), d = m - I;        // tail
        const int total = a + b + nsteps + c + d;
        const int first_adapter_col = b > 0 ? 0 : a, first_read_col = a > 0 ? 0 : b;
        int e_off, rb_end, ab_end, la_off;
        if (d > 0) {                      // adapter rows hang past the read's end (c == 0, the path is not empty)
            int k = d, y = first_type;
            if (first_type == T_V) { k += first_len; y = second_type ? second_type : T_H; }
            e_off = k; ab_end = k + (y == T_DIAG ? 1 : 0); rb_end = 1; la_off = 0;
        } else if (c > 0) {               // read bases follow the alignment (end cell in the last row)
            int k = c, y = first_type ? first_type : T_V;
            if (first_type == T_H) { k += first_len; y = second_type ? second_type : T_V; }
            e_off = k; rb_end = k + (y == T_DIAG ? 1 : 0); ab_end = 1; la_off = k;
        } else if (first_type == T_H) {   // the end cell is the corner (m, n)
            const int y = second_type ? second_type : T_V;
            e_off = first_len; rb_end = first_len + (y == T_DIAG ? 1 : 0); ab_end = 1; la_off = first_len;
        } else if (first_type == T_V) {
            const int y = second_type ? second_type : T_H;
            e_off = first_len; ab_end = first_len + (y == T_DIAG ? 1 : 0); rb_end = 1; la_off = 0;
        } else {
            e_off = 0; rb_end = 1; ab_end = 1; la_off = 0;
        }
        out.read_start = first_adapter_col;
        out.adapter_start = first_read_col;
        out.read_end = n_total - rb_end;
        out.adapter_end = m - ab_end;
        out.score = score;
        out.matches = matches;
        out.aligned_len = total - (a + b) - e_off;
        out.full_len = total - first_adapter_col - la_off;
        return err;
    }
};

// TraceFn:   int nib(int local_col, int adapter_row)  for local_col>=1, adapter_row>=1
// MatchFn:   bool eq(int local_col, int adapter_row)  -> read base (col-1) == adapter base (row-1)
// One walk, start to finish (host tests; the kernels drive two Walk states themselves and never
// look at the bases: here the counted matches cross-check the score-derived ones).
template <typename TraceFn, typename MatchFn>
PC_HD int walk(TraceFn nib, MatchFn eq, int I, int J, int m, int col0, int n_total, int score,
               int tie_fix, int match, int mismatch, int gap_open, int gap_extend, Digest &out)
{
    Walk w;
    w.start(I, J, m, col0, n_total, score, tie_fix);
    while (!w.done) w.consume(nib(w.col, w.row), eq(w.col, w.row));
    int err = w.finish(out, match, mismatch, gap_open, gap_extend);
    if (out.matches != w.matches_seen) err = 1;      // the score-derived count against the bases themselves
    return err;
}

}  // namespace pcw
