// Host-side test of porechop_amd/csrc/pc_slow.h -- the plain-int32 alignment the HIP library runs for scoring schemes
// and adapter lengths its packed 16-bit kernels refuse -- against the oracle (oracle/pc_oracle.c) on UNRESTRICTED
// integer schemes: positive and zero gap scores, match <= mismatch, gap_open == gap_extend (the reference's linear-gap
// dispatch), large magnitudes, adapters up to 300 bases.  The same align_pair() the device runs per lane.
//
//   g++ -O2 -std=c++17 -I porechop_amd/csrc -I oracle tests/host/test_slow.cpp oracle/pc_oracle.o
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "pc_slow.h"
extern "C" {
#include "pc_oracle.h"
}

static int dna5(unsigned char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
}

struct Mem {
    std::vector<int> *m, *h; std::vector<uint8_t> *t; int rows;
    int &M(int i) { return (*m)[i]; }
    int &H(int i) { return (*h)[i]; }
    uint8_t &T(int j, int i) { return (*t)[(size_t)j * rows + i]; }
};

int main(int argc, char **argv)
{
    const long cases = argc > 1 ? atol(argv[1]) : 20000;
    std::mt19937_64 rng(argc > 2 ? atol(argv[2]) : 12345);
    auto pick = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    long bad = 0, linear = 0, posgap = 0, failed = 0;
    const char *letters = "ACGTACGTACGTACGTNacgtU-X";
    for (long it = 0; it < cases; ++it) {
        int sc[4];
        const int mode = pick(0, 5);
        if (mode <= 1) for (int &x : sc) x = pick(-12, 12);
        else if (mode == 2) for (int &x : sc) x = pick(-100000, 100000);
        else if (mode == 3) { sc[0] = pick(1, 8); sc[1] = pick(-8, 0); sc[2] = pick(0, 6); sc[3] = pick(0, 6); }
        else if (mode == 4) for (int &x : sc) x = pick(-1, 1);
        else { sc[0] = pick(1, 6); sc[1] = pick(-8, -1); sc[2] = sc[3] = pick(-7, 2); }
        const int nn[] = {0, 1, 3, 10, 40, 150, 300}, mm[] = {0, 1, 2, 5, 22, 28, 60, 130, 300};
        const int n = nn[pick(0, 6)], m = mm[pick(0, 8)];
        std::string rd(n, 'A'), ad(m, 'A');
        for (char &c : rd) c = letters[pick(0, 23)];
        if (n >= m && m > 0 && pick(0, 9) < 6) {
            const int s = pick(0, n - m);
            for (int k = 0; k < m; ++k) ad[k] = pick(0, 99) < 15 ? "ACGTN"[pick(0, 4)] : rd[s + k];
        } else for (char &c : ad) c = "ACGT"[pick(0, 3)];
        if (!pcs::fits(n, m, sc[0], sc[1], sc[2], sc[3])) continue;
        pc_oracle_result R;
        if (pc_oracle_align_raw(rd.c_str(), n, ad.c_str(), m, sc[0], sc[1], sc[2], sc[3], &R) != 0) continue;
        std::vector<int> M(m + 1), H(m + 1);
        std::vector<uint8_t> T((size_t)(n + 1) * (m + 1));
        pcw::Digest d;
        const int err = pcs::align_pair(n, m, [&](int k) { return dna5((unsigned char)rd[k]); },
                                        [&](int k) { return dna5((unsigned char)ad[k]); }, sc[0], sc[1], sc[2], sc[3],
                                        Mem{&M, &H, &T, m + 1}, d);
        linear += sc[2] == sc[3]; posgap += sc[2] >= 0 || sc[3] >= 0; failed += R.failed;
        bool same;
        if (R.failed) same = d.read_start == -1 && (n == 0 || m == 0 ? d.score == R.score : true);
        else same = d.read_start == R.read_start && d.read_end == R.read_end && d.adapter_start == R.adapter_start &&
                    d.adapter_end == R.adapter_end && d.score == R.score && d.matches == R.aligned_matches &&
                    d.matches == R.full_matches && d.aligned_len == R.aligned_len && d.full_len == R.full_len;
        if (err || !same) {
            if (++bad <= 10)
                printf("MISMATCH err=%d rd=%s ad=%s scores=%d,%d,%d,%d\n got  %d,%d,%d,%d,%d m=%d al=%d fl=%d\n want %d,%d,%d,%d,%d m=%d/%d al=%d fl=%d failed=%d\n",
                       err, rd.c_str(), ad.c_str(), sc[0], sc[1], sc[2], sc[3], d.read_start, d.read_end, d.adapter_start, d.adapter_end,
                       d.score, d.matches, d.aligned_len, d.full_len, R.read_start, R.read_end, R.adapter_start, R.adapter_end, R.score,
                       R.aligned_matches, R.full_matches, R.aligned_len, R.full_len, R.failed);
        }
    }
    printf("cases=%ld bad=%ld linear=%ld nonnegative_gap=%ld reference_failed=%ld\n", cases, bad, linear, posgap, failed);
    return bad ? 1 : 0;
}
