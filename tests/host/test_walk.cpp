// Host-side unit test of porechop_amd/csrc/pc_walk.h (the traceback+digest the HIP kernels run
// per lane) and of the two-pass window bound, against the oracle (oracle/pc_oracle.c).
// Test infrastructure only: builds a plain int32 DP here to produce the 4-bit trace the
// kernels would produce, then runs the SAME walk() code the device runs.
//
//   g++ -O2 -std=c++17 -I porechop_amd/csrc -I oracle tests/host/test_walk.cpp oracle/pc_oracle.c
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "pc_bounds.h"
#include "pc_walk.h"
extern "C" {
#include "pc_oracle.h"
}

static int dna5(unsigned char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
}

struct Dp {
    int n, m;
    std::vector<uint8_t> nib;   // (n+1)*(m+1)
    std::vector<uint8_t> tie;   // d == max(H,V)
    std::vector<int> Mlast_row, Mlast_col;
    int bestM, bestI, bestJ;
};

// window DP: columns c0+1..c0+w of read, interior init when c0>0; records trace/tie; if scout,
// finds the max like the reference does (only meaningful for c0==0,w==n).
static void run_dp(const std::string &rd, const std::string &ad, int c0, int w, int a, int b, int o, int e,
                   bool scout, Dp &dp)
{
    const int m = (int)ad.size();
    const int NEG = -(1 << 28);
    const bool linear = (o == e);
    const int e_real = e;
    if (linear) e = -(1 << 20);      // the kernels' way of running the linear recurrence (pc_bounds.h)
    dp.n = w; dp.m = m;
    dp.nib.assign((size_t)(w + 1) * (m + 1), 0);
    dp.tie.assign((size_t)(w + 1) * (m + 1), 0);
    std::vector<int> M(m + 1), H(m + 1, NEG), V(m + 1, NEG);
    for (int i = 0; i <= m; ++i) M[i] = 0;
    if (c0 > 0) for (int i = 1; i <= m; ++i) { M[i] = o + (i - 1) * e_real; V[i] = M[i]; }
    dp.bestM = 0; dp.bestI = m; dp.bestJ = 0;
    for (int j = 1; j <= w; ++j) {
        int diag = M[0], upM = 0, upV = NEG;
        const int h = dna5((unsigned char)rd[c0 + j - 1]);
        for (int i = 1; i <= m; ++i) {
            int nb = 0;
            int Hx = H[i] + e, t = M[i] + o, Hs = Hx;
            if (Hx < t) { Hs = t; nb |= pcw::NIB_HOPEN; }
            int Vx = upV + e; t = upM + o; int Vs = Vx;
            if (Vx < t) { Vs = t; nb |= pcw::NIB_VOPEN; }
            int g = Vs;
            if (Vs < Hs) { g = Hs; nb |= pcw::NIB_FROMH; }
            int d = diag + (h == dna5((unsigned char)ad[i - 1]) ? a : b);
            int S = d;
            if (d < g) { S = g; nb |= pcw::NIB_NOTDIAG; }
            dp.nib[(size_t)j * (m + 1) + i] = (uint8_t)nb;
            dp.tie[(size_t)j * (m + 1) + i] = (d == g);
            diag = M[i]; M[i] = S; H[i] = Hs; V[i] = Vs; upM = S; upV = Vs;
            if (scout && (j == w || i == m) && S > dp.bestM) { dp.bestM = S; dp.bestI = i; dp.bestJ = j; }
        }
    }
    dp.Mlast_col = M;
}

static bool g_linear = false;
static int tie_fix_of(const Dp &dp, int I, int J) {
    if (I <= 0 || J <= 0 || g_linear) return 0;
    const int nb = dp.nib[(size_t)J * (dp.m + 1) + I];
    const int tie = dp.tie[(size_t)J * (dp.m + 1) + I];
    if ((nb & pcw::NIB_NOTDIAG) || tie) return (nb & pcw::NIB_FROMH) ? 2 : 1;
    return 0;
}

static bool check(const std::string &rd, const std::string &ad, int a, int b, int o, int e, long &nwin)
{
    const int n = (int)rd.size(), m = (int)ad.size();
    g_linear = (o == e);
    pc_oracle_result R;
    if (pc_oracle_align_raw(rd.c_str(), n, ad.c_str(), m, a, b, o, e, &R) != 0) return true;
    Dp dp;
    run_dp(rd, ad, 0, n, a, b, o, e, true, dp);
    auto mk = [&](const Dp &D, int c0) {
        return std::make_pair(
            [&D](int col, int row) { return (int)D.nib[(size_t)col * (D.m + 1) + row]; },
            [&rd, &ad, c0](int col, int row) { return dna5((unsigned char)rd[c0 + col - 1]) == dna5((unsigned char)ad[row - 1]); });
    };
    pcw::Digest dg;
    auto f = mk(dp, 0);
    int err = pcw::walk(f.first, f.second, dp.bestI, dp.bestJ, m, 0, n, dp.bestM, tie_fix_of(dp, dp.bestI, dp.bestJ), a, b, o, e, dg);
    auto same = [&](const pcw::Digest &x) {
        return x.read_start == R.read_start && x.read_end == R.read_end && x.adapter_start == R.adapter_start &&
               x.adapter_end == R.adapter_end && x.score == R.score && x.matches == R.aligned_matches &&
               x.matches == R.full_matches && x.aligned_len == R.aligned_len && x.full_len == R.full_len;
    };
    if (err || !same(dg)) {
        printf("FULL MISMATCH err=%d rd=%s ad=%s scores=%d,%d,%d,%d\n got  %d,%d,%d,%d,%d m=%d al=%d fl=%d\n want %d,%d,%d,%d,%d m=%d/%d al=%d fl=%d\n",
               err, rd.c_str(), ad.c_str(), a, b, o, e, dg.read_start, dg.read_end, dg.adapter_start, dg.adapter_end, dg.score,
               dg.matches, dg.aligned_len, dg.full_len, R.read_start, R.read_end, R.adapter_start, R.adapter_end, R.score,
               R.aligned_matches, R.full_matches, R.aligned_len, R.full_len);
        return false;
    }
    // two-pass: recompute only the window the bound prescribes, interior init, forced end cell
    pcb::Bounds bd;
    if (!pcb::compute_bounds(a, b, o, e, m, bd)) return true;
    int c0 = dp.bestJ - bd.window;
    if (c0 > 0) {
        ++nwin;
        Dp wd;
        run_dp(rd, ad, c0, dp.bestJ - c0, a, b, o, e, false, wd);
        // the window's own value at the forced cell must equal the global best
        if (wd.Mlast_col[dp.bestI] != dp.bestM) { printf("WINDOW SCORE MISMATCH %d vs %d\n", wd.Mlast_col[dp.bestI], dp.bestM); return false; }
        auto fw = mk(wd, c0);
        pcw::Digest dw;
        err = pcw::walk(fw.first, fw.second, dp.bestI, dp.bestJ - c0, m, c0, n, dp.bestM,
                        tie_fix_of(wd, dp.bestI, dp.bestJ - c0), a, b, o, e, dw);
        if (err || !same(dw)) {
            printf("WINDOW MISMATCH err=%d c0=%d J=%d n=%d m=%d scores=%d,%d,%d,%d\n", err, c0, dp.bestJ, n, m, a, b, o, e);
            return false;
        }
    }
    return true;
}

int main(int argc, char **argv)
{
    const long cases = argc > 1 ? atol(argv[1]) : 20000;
    std::mt19937 rng(12345);
    const int schemes[][4] = {{3, -6, -5, -2}, {1, -1, -3, -1}, {5, -4, -10, -1}, {2, -3, -5, -2}, {3, -6, -2, -5}, {1, -5, -1, -3},
                              {3, -6, -5, -5}, {2, -3, -4, -4}, {1, -1, -1, -1}};
    const char *alpha[] = {"ACGT", "ACGT", "ACGTN", "AC", "ACGT-"};
    long bad = 0, nwin = 0;
    for (long it = 0; it < cases; ++it) {
        const int *sc = schemes[rng() % 9];
        const int nl[] = {1, 2, 5, 20, 50, 150, 150, 151, 300, 700, 1500};
        const int ml[] = {1, 3, 8, 22, 24, 28, 28, 33, 50, 63, 111};
        int n = nl[rng() % 11], m = ml[rng() % 11];
        const char *al = alpha[rng() % 5];
        const size_t alen = strlen(al);
        std::string rd(n, 'A'), ad(m, 'A');
        for (auto &c : rd) c = al[rng() % alen];
        for (auto &c : ad) c = "ACGT"[rng() % 4];
        if (rng() % 3 && n > 5) {   // implant a mutated copy
            std::string mut;
            for (char c : ad) {
                unsigned x = rng() % 100;
                if (x < 5) mut.push_back("ACGT"[rng() % 4]);
                else if (x < 9) {}
                else if (x < 13) { mut.push_back(c); mut.push_back("ACGT"[rng() % 4]); }
                else mut.push_back(c);
            }
            size_t pos = rng() % n;
            std::string j = rd.substr(0, pos) + mut + rd.substr(pos);
            rd = (rng() & 1) ? j.substr(0, n) : j.substr(j.size() - n);
        }
        if (!check(rd, ad, sc[0], sc[1], sc[2], sc[3], nwin)) { if (++bad > 5) break; }
    }
    printf("cases=%ld windows_checked=%ld bad=%ld\n", cases, nwin, bad);
    return bad ? 1 : 0;
}
