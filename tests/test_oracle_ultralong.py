"""The oracle pinned on ULTRA-LONG reads (>= 65 535 bases, up to 1 000 000): oracle/pc_oracle.c against the compiled
reference (oracle/_ref, porechop/src/adapter_align.cpp:11-31) on the cases tests/test_gpu_ultralong.py feeds the GPU.
No GPU needed."""
import pytest

from oracle.oracle import Oracle, Reference
from tests.longgen import MILLION, Y_BOTTOM, Y_TOP, cases, make_read, mutated

needs_ref = pytest.mark.skipif(not Reference.available(), reason="compiled reference unavailable")


@needs_ref
def test_oracle_equals_reference_beyond_65535_columns():
    ora, ref = Oracle(), Reference()
    cs = cases(lengths=(65535, 65536, 70000, 131073), chunk_cols=(32768, 65536))
    assert len(cs) > 100
    bad = [(lab, ora.adapter_alignment(rd, ad), ref.adapter_alignment(rd, ad)) for lab, rd, ad in cs
           if ora.adapter_alignment(rd, ad) != ref.adapter_alignment(rd, ad)]
    assert not bad, bad[:3]
    # the planted copies are found where they were put (the cases test what they claim to test)
    found = {lab: int(ora.adapter_alignment(rd, ad).split(",")[1]) for lab, rd, ad in cs if lab.startswith("col65536 n=70000")}
    assert found and all(abs(v - 65535) <= 1 for v in found.values()), found


@needs_ref
def test_oracle_equals_reference_on_a_million_bases():
    import numpy as np
    ora, ref = Oracle(), Reference()
    rng = np.random.default_rng(3)
    for ad, col in ((Y_TOP, 999_990), (Y_BOTTOM, 65_536), (Y_TOP, MILLION)):
        rd = make_read(MILLION, 11, [(col, mutated(rng, ad))], n_runs=[(500_000, 1000)], dash_runs=[(700_000, 5000)])
        got, want = ora.adapter_alignment(rd, ad), ref.adapter_alignment(rd, ad)
        assert got == want, (col, got, want)
        assert abs(int(got.split(",")[1]) - (col - 1)) <= 2
