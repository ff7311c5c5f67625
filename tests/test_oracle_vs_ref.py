"""Live diff of the restatement against the reference itself (oracle/_ref/cpp_functions.so,
compiled from the reference's own sources by oracle/Makefile).  Skipped where neither the
prebuilt file nor /root/reference exists."""
import random

import pytest

from oracle.oracle import Reference
from tests.pairgen import LINEAR_SCHEMES, SCHEMES, random_case

pytestmark = pytest.mark.skipif(not Reference.available(), reason="no compiled reference here")


def test_oracle_equals_live_reference(oracle):
    ref = Reference()
    rng = random.Random(20260925)
    bad = []
    for _ in range(15000):
        sc = rng.choice(SCHEMES + LINEAR_SCHEMES)
        rd, ad = random_case(rng)
        a, b = oracle.adapter_alignment(rd, ad, sc), ref.adapter_alignment(rd, ad, sc)
        if a != b:
            bad.append((rd, ad, sc, a, b))
    assert not bad, bad[:5]


def test_oracle_equals_live_reference_long_reads(oracle):
    ref = Reference()
    rng = random.Random(7)
    for _ in range(60):
        rd, ad = random_case(rng, n=rng.choice([2000, 8000, 12000]), m=rng.choice([22, 28, 68, 111]))
        assert oracle.adapter_alignment(rd, ad) == ref.adapter_alignment(rd, ad)
