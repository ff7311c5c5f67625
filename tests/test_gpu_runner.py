"""The end-to-end runner on the GPU: same seeded inputs, same goldens from the reference CLI
(tests/golden/runner_goldens.json), alignments from the HIP library through the C ABI."""
import pytest

from tests.runner_cases import load_cases, run_case

pytestmark = pytest.mark.gpu


def test_runner_cases_match_reference_cli_on_gpu(tmp_path):
    cases = load_cases()
    datasets = {}
    for name, case in sorted(cases.items()):
        got = run_case(name, case, str(tmp_path), datasets, device="cuda")
        assert got == case["outputs"], (name, got, case["outputs"])


def test_command_line_entry(tmp_path):
    """`python -m porechop_amd` with the reference's option names: two golden cases through the CLI."""
    import os
    import subprocess
    import sys
    from tests import readgen
    cases = load_cases()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("native_split_sizes", "native_bins_strict"):
        case = cases[name]
        inp = readgen.build_dataset(case["dataset"], str(tmp_path / "datasets"))
        target = str(tmp_path / name / ("bins" if case["mode"] == "b" else case["mode"][2:]))
        os.makedirs(os.path.dirname(target), exist_ok=True)
        argv = [sys.executable, "-m", "porechop_amd", "-i", inp] + (["-b", target] if case["mode"] == "b" else ["-o", target]) + case["argv"]
        res = subprocess.run(argv, cwd=repo, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        assert "adapter sets:" in res.stdout
        assert readgen.output_md5s(target) == case["outputs"], name


def test_streamed_run_on_gpu(tmp_path, monkeypatch):
    """The streamed path (blocks of a few kB here; 256 MB in production) with the HIP library behind it: the
    reference CLI's recorded outputs again, from many blocks."""
    from porechop_amd import runner
    monkeypatch.setenv("PC_STREAM_BLOCK_BYTES", "6000")
    blocks = []
    real = runner.ReadSet.segment
    monkeypatch.setattr(runner.ReadSet, "segment", staticmethod(lambda p, b, t: (blocks.append(b), real(p, b, t))[1]))
    cases = load_cases()
    datasets = {}
    for name in ("native_check20", "native_check0"):
        got = run_case(name, cases[name], str(tmp_path), datasets, device="cuda")
        assert got == cases[name]["outputs"], (name, got)
    assert len(blocks) > 10
