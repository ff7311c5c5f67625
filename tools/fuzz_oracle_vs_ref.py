#!/usr/bin/env python3
"""The C restatement (oracle/pc_oracle.c) against the reference ITSELF (oracle/_ref/cpp_functions.so, compiled from
/root/reference by oracle/Makefile), far beyond what tests/test_oracle_vs_ref.py has time for: ANY four integers as the scoring
scheme (positive and zero gap scores, match <= mismatch, magnitudes up to 2^20 -- all of which the boundary accepts,
porechop.py:145,196-202), reads of 0 ... 600 bytes over odd alphabets (IUPAC codes, lower case, U, '-', digits, blanks),
adapters of 1 ... 140 bases, empty strings.  The oracle is what the GPU kernels are compared with: this pins it.
    python tools/fuzz_oracle_vs_ref.py [cases] [seed]"""
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.oracle import Oracle, Reference  # noqa: E402
from tests.pairgen import LINEAR_SCHEMES, SCHEMES, mutate, random_case  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o, ref = Oracle(), Reference()
ALPHABETS = ["ACGT", "ACGT", "ACGTN", "ACGTUacgtun", "ACGT-", "ACGTRYKMSWBDHVN", "AC", "A", "ACGT01 .*", "NNNNA", "-"]


def scheme():
    r = rng.random()
    if r < 0.35:
        return rng.choice(SCHEMES + LINEAR_SCHEMES)
    if r < 0.6:
        return (rng.randint(1, 30), -rng.randint(0, 40), -rng.randint(0, 40), -rng.randint(0, 40))
    if r < 0.85:
        return tuple(rng.randint(-12, 12) for _ in range(4))
    if r < 0.95:
        return tuple(rng.randint(-1000, 1000) for _ in range(4))
    return tuple(rng.choice([-(1 << 20), -70000, -1, 0, 1, 70000, 1 << 20]) for _ in range(4))


bad = 0
for k in range(cases):
    sc = scheme()
    if rng.random() < 0.5:
        rd, ad = random_case(rng)
    else:
        m = rng.choice([1, 2, 7, 22, 24, 28, 33, 64, 111, 128, 129, 140])
        ad = "".join(rng.choice(rng.choice(["ACGT", "ACGT", "ACGTN", "acgt", "ACGU"])) for _ in range(m))
        n = rng.choice([0, 1, 2, 10, 80, 150, 150, 151, 600])
        rd = "".join(rng.choice(rng.choice(ALPHABETS)) for _ in range(n))
        if n > m and rng.random() < 0.6:
            p = rng.randrange(n - m + 1)
            rd = rd[:p] + mutate(rng, ad, rate=rng.choice([0.0, 0.1, 0.3])) + rd[p + m:]
        if rng.random() < 0.02:
            ad = ""
    a, b = o.adapter_alignment(rd, ad, sc), ref.adapter_alignment(rd, ad, sc)
    if a != b:
        bad += 1
        if bad <= 20:
            print("BAD", sc, repr(rd[:80]), repr(ad[:40]), a, "|", b, flush=True)
print("cases=%d mismatches=%d" % (cases, bad))
