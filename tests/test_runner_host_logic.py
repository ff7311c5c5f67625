"""Host logic of the end-to-end runner (set rules, trims, barcode calls, splits, naming, formats,
writer) on the seeded synthetic inputs: every output file must have the content the unchanged
reference CLI produced (tests/golden/runner_goldens.json).  Alignments come from the oracle through
tests/cpu_aligner.py -- the GPU run of the same cases is tests/test_gpu_runner.py."""
from tests.cpu_aligner import OracleAligner
from tests.runner_cases import GPU_ONLY, load_cases, run_case


def test_runner_cases_match_reference_cli(oracle, tmp_path):
    cases = load_cases()
    assert len(cases) >= 40
    datasets = {}
    for name, case in sorted(cases.items()):
        if name in GPU_ONLY:
            continue
        got = run_case(name, case, str(tmp_path), datasets, make_aligner=lambda sc: OracleAligner(oracle, sc))
        assert got == case["outputs"], (name, got, case["outputs"])


def test_usage_errors():
    import pytest
    from porechop_amd import runner
    with pytest.raises(runner.UsageError):
        runner.run("/nonexistent/reads.fastq", output="/tmp/x.fastq", aligner=object())
    with pytest.raises(runner.UsageError):
        runner.run(__file__, output="/tmp/x.fastq", barcode_dir="/tmp/y", aligner=object())
    with pytest.raises(runner.UsageError):
        runner.run(__file__, output="/tmp/x.fastq", options=runner.Options(untrimmed=True), aligner=object())
