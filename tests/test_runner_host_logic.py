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


def test_stdout_output(tmp_path):
    """No -o and no -b: the reads go to stdout (porechop.py:706-711), written by the C++ writer through
    the process's stdout; a '.gz' format is left alone there and, not being 'fasta', prints FASTQ."""
    import hashlib
    import os
    import subprocess
    import sys
    from tests import readgen
    cases = load_cases()
    inp = readgen.build_dataset("native", str(tmp_path / "datasets"))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r)
from oracle.oracle import Oracle
from tests.cpu_aligner import OracleAligner
from porechop_amd import runner
opts = runner.Options(format=sys.argv[2])
runner.run(sys.argv[1], options=opts, aligner=OracleAligner(Oracle(), opts.scoring_scheme))
''' % repo
    for fmt, golden in (("auto", "native_default"), ("fasta", "native_to_fasta"), ("fasta.gz", "native_default")):
        res = subprocess.run([sys.executable, "-c", code, inp, fmt], capture_output=True, timeout=600, cwd=repo)
        assert res.returncode == 0, res.stderr[-2000:]
        want = list(cases[golden]["outputs"].values())[0]
        assert hashlib.md5(res.stdout).hexdigest() == want, (fmt, len(res.stdout))


def test_read_blocks_do_not_change_the_outputs(oracle, tmp_path, monkeypatch):
    """Phases B and C run over blocks of reads (bounded scratch); with blocks of 7 reads instead of one
    block the output files must still be the reference CLI's (a native-barcoding run with chimeras, a
    plain run with middle splits)."""
    from porechop_amd import runner
    monkeypatch.setattr(runner, "MIN_READ_BLOCK", 7)
    monkeypatch.setattr(runner, "READ_BLOCK_PAIRS", 1)
    cases = load_cases()
    datasets = {}
    done = 0
    for name, case in sorted(cases.items()):
        if name in GPU_ONLY or not (name.startswith("native_") or name.startswith("chimera")):
            continue
        got = run_case(name, case, str(tmp_path), datasets, make_aligner=lambda sc: OracleAligner(oracle, sc))
        assert got == case["outputs"], (name, got, case["outputs"])
        done += 1
        if done >= 6:
            break
    assert done >= 3


def test_panel_follows_the_reference_module_when_importable():
    """load_panel() reads porechop.adapters.ADAPTERS at run time when the reference is importable, and the
    recorded copy (panel.json) otherwise; the two must be the same table (where /root/reference exists)."""
    import os
    import sys
    import pytest
    from porechop_amd import panel
    recorded = panel.load_panel(prefer_reference=False)
    assert len(recorded) == 119
    if not os.path.isdir("/root/reference/porechop"):
        pytest.skip("reference checkout not present")
    sys.path.insert(0, "/root/reference")
    try:
        for m in [k for k in sys.modules if k == "porechop" or k.startswith("porechop.")]:
            del sys.modules[m]
        live = panel.load_panel()
    finally:
        sys.path.remove("/root/reference")
        for m in [k for k in sys.modules if k == "porechop" or k.startswith("porechop.")]:
            del sys.modules[m]
    assert [(s.name, s.start, s.end) for s in live] == [(s.name, s.start, s.end) for s in recorded]
