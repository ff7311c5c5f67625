"""oracle/pc_oracle.c (our CPU restatement) against the committed goldens minted from the
compiled reference: every adapter_alignment call the reference CLI makes on its bundled
fixtures + 12 000 seeded synthetic cases over 6 scoring schemes."""
from tests.golden_io import comparable, load_synthetic


def test_oracle_matches_recorded_reference_calls(oracle, goldens):
    S, bad = goldens["strings"], []
    for ri, ai, sc, res in goldens["calls"]:
        got = oracle.adapter_alignment(S[ri], S[ai], tuple(sc))
        if comparable(got) != comparable(res):
            bad.append((len(S[ri]), S[ai], sc, res, got))
    assert not bad, bad[:5]


def test_oracle_matches_reference_on_synthetic(oracle):
    bad = []
    for rd, ad, sc, res in load_synthetic():
        got = oracle.adapter_alignment(rd, ad, sc)
        if comparable(got) != comparable(res):
            bad.append((rd, ad, sc, res, got))
    assert not bad, bad[:5]


def test_known_answer_vectors(oracle):
    # SURVEY.md section 8a "known-answer vectors from the compiled reference"
    kav = [
        ("ACGTACGTAC", "ACGT", "0,3,0,3,12,100.000000,100.000000"),
        ("TTTTACGTTTTT", "ACGT", "4,7,0,3,12,100.000000,100.000000"),
        ("ACGT", "TTACGTTT", "0,3,2,5,12,100.000000,50.000000"),
        ("GTTT", "ACGT", "0,1,2,3,6,100.000000,50.000000"),
        ("TTAC", "ACGT", "2,3,0,1,6,100.000000,50.000000"),
        ("NNNNNNNN", "ACGT", "0,0,4,3,0,-nan,0.000000"),
        ("ACNNGT", "ACNNGT", "0,5,0,5,18,100.000000,100.000000"),
        ("AC--GT", "ACGT", "0,5,0,3,5,66.666667,66.666667"),
        ("ACXXGT", "ACNNGT", "0,5,0,5,18,100.000000,100.000000"),
        ("A", "C", "0,0,1,0,0,-nan,0.000000"),
        ("A", "ACGT", "0,0,0,0,3,100.000000,25.000000"),
        ("TTTTACGAACGTTTTT", "ACGTACGT", "4,11,0,7,15,87.500000,87.500000"),
        ("TTTTACGTTACGTTTTT", "ACGTACGT", "4,12,0,7,19,88.888889,88.888889"),
        ("AAAAAAAAAA", "CCCC", "0,0,4,3,0,-nan,0.000000"),
    ]
    for rd, ad, want in kav:
        assert oracle.adapter_alignment(rd, ad) == want, (rd, ad)


def test_empty_inputs_report_failure(oracle):
    assert oracle.adapter_alignment("", "ACGT").split(",")[0] == "-1"
    assert oracle.adapter_alignment("ACGT", "").split(",")[0] == "-1"
