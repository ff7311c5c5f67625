// Audit of the exactness gates of the 16-bit kernels (pc_bounds.h f16_plan, spec_plan), host-compiled: for every
// scoring scheme of a grid and every row class, IF a gate admits the packed-fp16 kernel THEN every quantity that
// kernel forms -- bounded here independently, from the recurrence alone -- must be an integer fp16 holds exactly
// (|v| <= 2048).  The quantities, for R register rows, jj columns since the window start / the last renormalisation,
// eps = -gap_extend, values held as X + (rho + jj [+1]) * eps - C:
//     M in [open + (R-1)*ext, match*R]                (a row-0 start + a vertical gap is always available)
//     T = M + open,  H and V in [min T, max T],  d = M_diag + sub in [min M + mismatch, match*R]
//     the substitution terms sub - open + eps, the constants open + eps and R*eps,
//     the TRACKED last-row term  T~(R,j) - top~(j) = M + R*eps  (the scout's packed compare),
//     the column-0 state of an end-aligned window entered late: open + (rho + 1 + j) * eps - C.
// (The tracked term was missing from f16_plan until tools/fuzz_parity.py found match 29 x 64 rows.)
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "pc_bounds.h"
#include "pc_kernels.h"

static long bad = 0, admitted16 = 0, admitted_spec = 0, checked = 0;

static void need(bool ok, const char *what, int a, int x, int o, int e, int R, long v)
{
    if (!ok) {
        if (++bad <= 10) printf("VIOLATION %s: scheme (%d,%d,%d,%d) R=%d value %ld\n", what, a, x, o, e, R, v);
    }
}

int main()
{
    const long LIM = 2048;       // integers of magnitude <= 2048 are exact in fp16 (the gates use 2040)
    for (int a = 1; a <= 40; a += (a < 8 ? 1 : 3))
        for (int x = -40; x <= 2; x += (x > -8 ? 1 : 4))
            for (int o = -40; o <= -1; o += (o > -8 ? 1 : 3))
                for (int e = -40; e <= -1; e += (e > -8 ? 1 : 3)) {
                    if (x >= a) continue;
                    if (o == e) continue;                                   // linear-gap schemes never take these kernels
                    const long eps = -e;
                    for (int R : pck::kTrace16Rows) {
                        ++checked;
                        const long Mhi = (long)a * R, Mlo = (long)o + (long)(R - 1) * e;
                        const long Tlo = Mlo + o, dlo = Mlo + std::min(x, 0);
                        const long lo_true = std::min(Tlo, dlo), hi_true = Mhi;
                        // ---- traced fp16 kernel ------------------------------------------------------------
                        const pcb::F16Plan p = pcb::f16_plan(a, x, o, e, R);
                        if (p.ok) {
                            ++admitted16;
                            const long cols = p.max_cols, C = p.cen;
                            need(hi_true + (R + cols + 1) * eps - C <= LIM, "trace16 max state", a, x, o, e, R, hi_true + (R + cols + 1) * eps - C);
                            need(lo_true + eps - C >= -LIM, "trace16 min state", a, x, o, e, R, lo_true + eps - C);
                            need(Mhi + R * eps <= LIM, "trace16 tracked term", a, x, o, e, R, Mhi + R * eps);
                            need(std::abs((long)a - o + eps) <= LIM && std::abs((long)x - o + eps) <= LIM, "trace16 table term", a, x, o, e, R, a - o + eps);
                            need(std::abs((long)o + eps) <= LIM && R * eps <= LIM, "trace16 constants", a, x, o, e, R, o + eps);
                            need(std::abs((long)o + (R + 1 + cols) * eps - C) <= LIM, "trace16 late column-0 state", a, x, o, e, R, o + (R + 1 + cols) * eps - C);
                            need(cols >= 32, "trace16 useful window", a, x, o, e, R, cols);
                        }
                        // ---- specialised score kernel ------------------------------------------------------
                        const pcb::SpecPlan s = pcb::spec_plan(a, x, o, e, R, false);
                        if (s.ok && s.f16) {
                            ++admitted_spec;
                            // jj runs to kren + 3 (the state is shifted down at the first block start at or after kren)
                            const long jj = s.kren + 4, C = s.cen;
                            need(hi_true + (R + jj + 1) * eps - C <= LIM, "spec max state", a, x, o, e, R, hi_true + (R + jj + 1) * eps - C);
                            need(std::min(lo_true, (long)o) + eps - C >= -LIM, "spec min state", a, x, o, e, R, lo_true + eps - C);
                            need(Mhi + R * eps <= LIM, "spec tracked term", a, x, o, e, R, Mhi + R * eps);
                            need(std::abs((long)a - o + eps) <= LIM && std::abs((long)x - o + eps) <= LIM, "spec table term", a, x, o, e, R, a - o + eps);
                            // the renormalisation constant kren * eps exceeds 2048, but kren is a multiple of 4: a multiple of 4 below
                            // 8192 is an fp16 number, and subtracting it from an exact value gives the exact (representable) result
                            need(s.kren % 4 == 0 && s.kren * eps < 8192, "spec renormalisation step", a, x, o, e, R, s.kren * eps);
                            // after the shift the state is back in range: max before it minus the step
                            need(hi_true + (R + jj + 1) * eps - C - s.kren * eps >= -LIM, "spec state after renormalisation", a, x, o, e, R, 0);
                        }
                    }
                }
    printf("checked=%ld admitted_trace16=%ld admitted_spec=%ld bad=%ld\n", checked, admitted16, admitted_spec, bad);
    return bad ? 1 : 0;
}
