// pc_slow.h -- one alignment in PLAIN int32 coordinates, for what the packed 16-bit kernels refuse: scoring schemes
// outside pcb::scores_supported (non-negative gap scores, match <= mismatch, magnitudes beyond the 16-bit lanes) and
// adapters longer than pcb::MAX_ADAPTER.  The reference accepts any four integers and any adapter
// (porechop/porechop.py:145,196-202, porechop/src/adapter_align.cpp:11-31); this is the path that keeps the drop-in
// boundary total.  Written once for device (pc_slow.hip: one LANE per pair, column state and trace in HBM, lane-
// interleaved) and host (tests/host/test_slow.cpp runs this very code against the oracle on unrestricted schemes).
//
// The recurrence, tie rules, scout order and _correctTraceValue are SURVEY.md 8a-2..a-4 read literally
// (seqan/align/dp_formula_affine.h:456-495, dp_formula_linear.h:150-182, dp_scout.h:165-179,
// dp_algorithm_impl.h:1352-1369); the trace is kept as the nibble pc_walk.h consumes and the traceback + digest ARE
// pc_walk.h (matches counted from the bases: finish_counted).  No drift, no bounds, no pruning: O(n m) cells, one byte
// of trace per cell.
#pragma once
#include <limits.h>
#include <stdint.h>

#include "pc_walk.h"

namespace pcs {

constexpr int NEG_INF = INT_MIN / 2;        // seqan/align/dp_cell.h:116-124

// Largest |score| and longest adapter this path takes; the caller also checks (n + m) * max|score| < 2^30 so that no
// sum leaves int32 (the reference's own arithmetic is int: beyond that it is undefined there too).
constexpr int MAX_ABS_SCORE = 1 << 20;
constexpr int MAX_ADAPTER = 4096;

PC_HD bool fits(int n, int m, int match, int mismatch, int gap_open, int gap_extend)
{
    auto ab = [](int x) { return x < 0 ? -(long long)x : (long long)x; };
    long long mx = ab(match);
    if (ab(mismatch) > mx) mx = ab(mismatch);
    if (ab(gap_open) > mx) mx = ab(gap_open);
    if (ab(gap_extend) > mx) mx = ab(gap_extend);
    return mx <= MAX_ABS_SCORE && m <= MAX_ADAPTER && ((long long)n + m + 2) * mx < (1ll << 30);
}

// Mem:  int &M(int i), int &H(int i)   for adapter rows i = 1..m (the column state)
//       uint8_t &T(int j, int i)       trace nibble of cell (column j = 1..n, row i = 1..m)
// rd(k) / ad(k): Dna5 code of the k-th (0-based) read / adapter base.
// -> 0, or 1 if the walk reported an inconsistency (never expected).  n or m == 0: the reference's failure record.
template <typename Mem, typename RdFn, typename AdFn>
PC_HD int align_pair(int n, int m, RdFn rd, AdFn ad, int match, int mismatch, int gap_open, int gap_extend, Mem mem,
                     pcw::Digest &out)
{
    if (n <= 0 || m <= 0) {                   // alignment.cpp:9-21: only field 0 and the score are defined
        out.read_start = -1; out.read_end = 0; out.adapter_start = -1; out.adapter_end = 0; out.score = INT_MIN;
        out.matches = 0; out.aligned_len = 0; out.full_len = 0;
        return 0;
    }
    const bool linear = gap_open == gap_extend;        // global_alignment_unbanded.h:217-220
    for (int i = 1; i <= m; ++i) { mem.M(i) = 0; mem.H(i) = NEG_INF; }
    int bestM = 0, bestH = NEG_INF, bestV = NEG_INF, bestI = m, bestJ = 0;
    for (int j = 1; j <= n; ++j) {
        int diag = 0, upM = 0, upV = NEG_INF;          // row 0: M = 0, V = -inf
        const int h = rd(j - 1);
        const bool last_col = j == n;
        for (int i = 1; i <= m; ++i) {
            const int Mi = mem.M(i);
            const int sub = (h == ad(i - 1)) ? match : mismatch;
            int S, Hs = NEG_INF, Vs = NEG_INF, nib;
            if (linear) {
                // strict '<' against vertical, then against horizontal: ties prefer diagonal, then vertical; every
                // gap step is a step of its own (both open bits set: pc_walk.h then never runs a gap)
                S = diag + sub; nib = pcw::NIB_HOPEN | pcw::NIB_VOPEN;
                int t = upM + gap_extend;
                if (S < t) { S = t; nib |= pcw::NIB_NOTDIAG; }
                t = Mi + gap_extend;
                if (S < t) { S = t; nib |= pcw::NIB_NOTDIAG | pcw::NIB_FROMH; }
            } else {
                nib = 0;
                Hs = mem.H(i) + gap_extend;
                int t = Mi + gap_open;
                if (Hs < t) { Hs = t; nib |= pcw::NIB_HOPEN; }
                Vs = upV + gap_extend;
                t = upM + gap_open;
                if (Vs < t) { Vs = t; nib |= pcw::NIB_VOPEN; }
                S = Vs;
                if (S < Hs) { S = Hs; nib |= pcw::NIB_FROMH; }
                const int d = diag + sub;
                if (S <= d) S = d; else nib |= pcw::NIB_NOTDIAG;
                mem.H(i) = Hs;
            }
            diag = Mi;
            mem.M(i) = S;
            upM = S; upV = Vs;
            mem.T(j, i) = (uint8_t)nib;
            if ((last_col || i == m) && S > bestM) { bestM = S; bestH = Hs; bestV = Vs; bestI = i; bestJ = j; }
        }
    }
    // _correctTraceValue (affine only; at (m, 0) H = V = -inf != M)
    int tie_fix = 0;
    if (!linear) tie_fix = (bestV == bestM) ? 1 : (bestH == bestM) ? 2 : 0;
    pcw::Walk w;
    w.start(bestI, bestJ, m, 0, n, bestM, tie_fix);
    while (!w.done) {
        const int c = w.col, r = w.row;
        w.consume(mem.T(c, r), rd(c - 1) == ad(r - 1));
    }
    return w.finish_counted(out);
}

}  // namespace pcs
