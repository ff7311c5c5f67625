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

PC_HD int nib_to_byte(int nib) {
    int b = (nib & NIB_HOPEN) ? T_HOPEN : T_H;
    b |= (nib & NIB_VOPEN) ? T_VOPEN : T_V;
    if (nib & NIB_NOTDIAG) b |= (nib & NIB_FROMH) ? T_MAXH : T_MAXV;
    else b |= T_DIAG;
    return b;
}

// Result record written per pair (8 x int32).
struct Digest {
    int32_t read_start, read_end, adapter_start, adapter_end, score;
    int32_t matches, aligned_len, full_len;
};

// Counters maintained from the alignment's right end towards its left end.
struct Acc {
    int cnt;        // columns seen so far
    int lr_off, la_off;   // from-right offset of the right-most read / adapter base column
    int pr, pa;           // from-right offset of the left-most (so far) read / adapter base column
    int rb, ab;           // read / adapter bases seen so far
    int rb_end, ab_end;   // rb/ab including the column where both rows have "ended"
    int captured;
    int matches;
    PC_HD void init() {
        cnt = 0; lr_off = -1; la_off = -1; pr = -1; pa = -1; rb = 0; ab = 0;
        rb_end = 0; ab_end = 0; captured = 0; matches = 0;
    }
    // kind: T_DIAG (both bases), T_H (read base over '-'), T_V ('-' over adapter base)
    PC_HD void run(int kind, int len) {
        if (len <= 0) return;
        const int has_r = (kind != T_V), has_a = (kind != T_H);
        const int first = cnt, lastoff = cnt + len - 1;
        if (has_r) { if (lr_off < 0) lr_off = first; pr = lastoff; }
        if (has_a) { if (la_off < 0) la_off = first; pa = lastoff; }
        if (!captured && lr_off >= 0 && la_off >= 0) {
            captured = 1; rb_end = rb + has_r; ab_end = ab + has_a;
        }
        rb += has_r ? len : 0; ab += has_a ? len : 0; cnt += len;
    }
    // n, m: total read / adapter bases in the rows (= whole read length, adapter length)
    PC_HD void finish(int n, int m, int score, Digest &d) const {
        const int L1 = cnt - 1;
        const int end_off = lr_off > la_off ? lr_off : la_off;
        const int start_off = pr < pa ? pr : pa;
        d.read_start = L1 - pa;          // == index of first adapter-base column
        d.adapter_start = L1 - pr;       // == index of first read-base column
        d.read_end = n - rb_end;
        d.adapter_end = m - ab_end;
        d.score = score;
        d.matches = matches;
        d.aligned_len = start_off - end_off + 1;
        d.full_len = pa - la_off + 1;
    }
};

// The traceback as a resumable state machine: every call of consume() takes the trace nibble of
// ONE cell -- the cell (col,row) the state asked for -- and leaves in (col,row) the next cell it
// needs, or done.  That is the reference's _doTraceback loop (dp_traceback_impl.h:376-450) cut at
// its trace-matrix reads, so that a caller can fetch for several independent walks at once (the
// kernels run a lane's two pairs side by side: two load chains in flight instead of one).
//
// (I,J) is the end cell in LOCAL columns (J in [0, ncols]); col0 is the global column of local 0;
// n_total the whole read length; tie_fix: 0 none, 1 force "from V", 2 force "from H"
// (the _correctTraceValue outcome, decided by the kernel from d==max(H,V) at the end cell).
// err = 1 if the walk ran into the left edge of a window that does not start at the read's
// column 0 (impossible when the window obeys the bound in pc_bounds.h; reported loudly).
struct Walk {
    enum : int { DISPATCH = 0, VRUN = 1, HRUN = 2 };
    Acc acc;
    int col, row, mode, first, tie_fix, done, err;
    int m, col0, n_total, score;

    PC_HD void start(int I, int J, int m_, int col0_, int n_total_, int score_, int tie_fix_) {
        acc.init();
        col = J; row = I; mode = DISPATCH; first = 1; tie_fix = tie_fix_; err = 0;
        m = m_; col0 = col0_; n_total = n_total_; score = score_;
        // tail segments: adapter bases hanging past the read end / read bases after the alignment
        acc.run(T_V, m - row);
        acc.run(T_H, n_total - (col0 + col));
        done = !(col > 0 && row > 0);
    }
    PC_HD void next_cell() { mode = DISPATCH; done = !(col > 0 && row > 0); }
    // nib: trace nibble of cell (col,row); eq: read base (col-1) == adapter base (row-1)
    PC_HD void consume(int nib, bool eq) {
        int tv = nib_to_byte(nib);
        if (mode == DISPATCH) {
            if (first) {
                first = 0;
                if (tie_fix == 1)      tv = (tv & ~T_DIAG) | T_MAXV;
                else if (tie_fix == 2) tv = (tv & ~T_DIAG) | T_MAXH;
                if (tv & T_MAXV)       tv &= (T_V | T_VOPEN | T_MAXV);
                else if (tv & T_MAXH)  tv &= (T_H | T_HOPEN | T_MAXH);
            }
            if (tv & T_DIAG) {
                acc.matches += eq ? 1 : 0;
                acc.run(T_DIAG, 1);
                col--; row--;
                next_cell();
            } else if ((tv & T_MAXV) && (tv & T_V)) {
                // gap run: follow the extend bits of the cells being left, then the opening step
                acc.run(T_V, 1); row--;
                if (row >= 1) mode = VRUN;      // the run continues through cell (col,row): its bits decide
                else next_cell();
            } else if ((tv & T_MAXV) && (tv & T_VOPEN)) {
                acc.run(T_V, 1); row--;
                next_cell();
            } else if ((tv & T_MAXH) && (tv & T_H)) {
                acc.run(T_H, 1); col--;
                if (col >= 1) mode = HRUN;
                else next_cell();
            } else if ((tv & T_MAXH) && (tv & T_HOPEN)) {
                acc.run(T_H, 1); col--;
                next_cell();
            } else {
                err = 1; done = 1;
            }
        } else if (mode == VRUN) {
            // inside a vertical run: this cell's own extend/open bit says whether the run goes on
            if ((!(tv & T_VOPEN) || (tv & T_V)) && row != 1) { acc.run(T_V, 1); row--; }
            else { acc.run(T_V, 1); row--; next_cell(); }
        } else {
            if ((!(tv & T_HOPEN) || (tv & T_H)) && col != 1) { acc.run(T_H, 1); col--; }
            else { acc.run(T_H, 1); col--; next_cell(); }
        }
    }
    PC_HD int finish(Digest &out) {
        if (row > 0 && col == 0 && col0 > 0) err = 1;   // left the window: bound violated
        // head segments
        acc.run(T_V, row);
        acc.run(T_H, col0 + col);
        acc.finish(n_total, m, score, out);
        return err;
    }
};

// TraceFn:   int nib(int local_col, int adapter_row)  for local_col>=1, adapter_row>=1
// MatchFn:   bool eq(int local_col, int adapter_row)  -> read base (col-1) == adapter base (row-1)
// One walk, start to finish (host tests; the kernels drive two Walk states themselves).
template <typename TraceFn, typename MatchFn>
PC_HD int walk(TraceFn nib, MatchFn eq, int I, int J, int m, int col0, int n_total, int score,
               int tie_fix, Digest &out)
{
    Walk w;
    w.start(I, J, m, col0, n_total, score, tie_fix);
    while (!w.done) w.consume(nib(w.col, w.row), eq(w.col, w.row));
    return w.finish(out);
}

}  // namespace pcw
