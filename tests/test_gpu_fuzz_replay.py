"""A 150-case slice of the differential-fuzz corpus, replayed through porechop_amd.runner over the HIP library (VERDICT r5,
task 4).  tests/golden/fuzz_cases.json holds, per case, ONE seed (tests/fuzzcase.py regenerates the input -- plain / gzip in
three layouts / FASTA / an Albacore-style directory, odd reads among them --, the output mode and the options from it) and the
md5 of every file the UNCHANGED reference CLI wrote for it in the build container (tools/diff_fuzz.py --emit; the full
1 020-case campaign on the MI355X is profiles/r06_replay_fuzz.txt).  Half of the cases take the middle scan behind the exact
prefilter, half behind the score bound; a third run as a stream of small blocks."""
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = os.path.join(REPO, "tests", "golden", "fuzz_cases.json")


def _cases():
    with open(CASES) as f:
        return json.load(f)


def test_the_recipes_regenerate_the_recorded_inputs(tmp_path):
    """(no GPU) one seed -> the same input bytes, mode and options as when the reference ran over them."""
    from porechop_amd import io as pio
    from tests.fuzzcase import content_md5, make_case
    cases = _cases()
    assert len(cases) == 150
    for k, c in enumerate(cases[::6]):
        case = make_case(c["cseed"], str(tmp_path / ("c%d" % k)), sized_gzip=pio.gzip_file)
        assert case["mode"] == c["mode"] and case["argv"] == c["argv"] and case["prefilter"] == c["prefilter"] and case["blocks"] == c["blocks"]
        assert content_md5(case["input"]) == c["content_md5"], c["cseed"]
    kinds = {(c["prefilter"], bool(c["blocks"])) for c in cases}
    assert len(kinds) == 4                                   # both proofs, whole and streamed
    assert sum(1 for c in cases if c["input"].endswith(".gz")) >= 20 and sum(1 for c in cases if c["input"] == "indir") >= 5


@pytest.mark.gpu
def test_fuzz_corpus_slice_on_the_gpu_equals_the_reference_cli():
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from replay_fuzz import replay
    n, bad, routes = replay(_cases())
    assert n == 150 and bad == 0, (bad, routes)
