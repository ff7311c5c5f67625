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
