"""tests/ref_cli.py (the harness bench.py and the GPU drop-in tests drive the reference's own CLI with) on the CPU: the staged,
unchanged porechop.porechop.main() over its own compiled cpp_functions.so reproduces the recorded output of the reference
CLI, with its phases timed -- and the GPU backend's product entry point (dropin.GpuBackend.align_product) orders its pairs
read-major, checked here with a stand-in aligner that answers from the oracle (no GPU)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.ref_cli import DEFAULT_STAGE, PHASES, staged

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not staged(), reason="no staged reference (make -C oracle ref where /root/reference exists)")
def test_harness_runs_the_staged_reference_cli(goldens, tmp_path):
    out = str(tmp_path / "out.fastq")
    report = str(tmp_path / "report.json")
    res = subprocess.run([sys.executable, os.path.join(REPO, "tests", "ref_cli.py"), "--report", report, "--", "-i",
                          os.path.join(DEFAULT_STAGE, "test", "test_one_adapter_set.fastq"), "-o", out, "-v", "0", "--threads", "2"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == goldens["runs"]["one_default"]["output_md5"]
    rep = json.load(open(report))
    assert set(PHASES) <= set(rep["phase_s"]) and rep["main_s"] > 0
    assert os.path.realpath(rep["cpp_functions_so"]).startswith(os.path.realpath(DEFAULT_STAGE))      # the reference's own library
    assert "dropin" not in rep


def test_gpu_backend_product_order_with_a_stand_in_aligner(oracle):
    from porechop_amd import dropin

    class StubAligner:
        def __init__(self, adapters, scores):
            self.adapters, self.scores = list(adapters), scores

        def align_host(self, arena, off, ln, idx):
            recs = []
            for o, n, a in zip(off, ln, idx):
                r = oracle.align_raw(bytes(arena[o:o + n]), self.adapters[a], self.scores)
                recs.append([r.read_start, r.read_end, r.adapter_start, r.adapter_end, r.score, r.aligned_matches, r.aligned_len, r.full_len]
                            if not r.failed else [-1, 0, -1, 0, -2147483648, 0, 0, 0])
            return np.array(recs, dtype=np.int32)

    formatted = []

    def fake_format(recs):
        out = []
        for r in recs:
            if r[0] == -1:
                out.append("FAILED")
            else:
                out.append("%d,%d,%d,%d,%d,%f,%f" % (r[0], r[1], r[2], r[3], r[4], 100.0 * r[5] / r[6] if r[6] else float("nan"),
                                                      100.0 * r[5] / r[7] if r[7] else float("nan")))
        formatted.append(len(out))
        return out

    be = dropin.GpuBackend()
    be._aligner = lambda adapters, scores: StubAligner(adapters, scores)
    import porechop_amd.batch as batch
    real = batch.format_results
    batch.format_results = fake_format
    try:
        reads = ["TTTTACGTTTTT", "ACGTACGTAC", "GGGGGGGG", "TTAC"]
        ads = ["ACGT", "GGGG", "TTTT"]
        got = be.align_product(reads, ads, (3, -6, -5, -2))
        pairs = [(r, a) for r in reads for a in ads]
        assert len(got) == len(pairs) == formatted[-1]
        for (r, a), s in zip(pairs, got):
            want = oracle.adapter_alignment(r, a, (3, -6, -5, -2))
            assert s.split(",")[:5] == want.split(",")[:5], (r, a, s, want)
        # and through the memo: every (read, adapter) key answers with its own pair's string
        st = dropin._State(be)
        st.prefetch_product(reads + reads[:2], ads, (3, -6, -5, -2))
        assert len(st.memo) == len(pairs)
        for r, a in pairs:
            assert st.lookup(r, a, [3, -6, -5, -2]).split(",")[:5] == oracle.adapter_alignment(r, a, (3, -6, -5, -2)).split(",")[:5]
        assert st.misses == 0 and st.hits == len(pairs)
    finally:
        batch.format_results = real
