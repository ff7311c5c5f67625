"""Ultra-long reads (> 65 535 bases) through the HOST logic above the C ABI -- Pipeline.phase_c's mask-and-realign rounds
and runner.run() file to file -- over the oracle-backed stand-in (no GPU): tests/test_gpu_ultralong.py runs the same two
checks over the HIP library."""
from tests.cpu_aligner import OracleAligner
from tests.longgen import Y_BOTTOM, Y_TOP
from tests.test_gpu_ultralong import check_phase_c, check_runner


def test_phase_c_over_ultralong_reads_host_logic(oracle):
    from porechop_amd.pipeline import AdapterSet, Pipeline, ScanParams
    p = ScanParams()
    pl = Pipeline([AdapterSet("SQK-NSK007", ("SQK-NSK007_Y_Top", Y_TOP), ("SQK-NSK007_Y_Bottom", Y_BOTTOM))], p,
                  aligner=OracleAligner(oracle, p.scores))
    check_phase_c(pl, "cpu")


def test_runner_over_ultralong_reads_host_logic(oracle, tmp_path):
    check_runner(tmp_path, aligner=OracleAligner(oracle, (3, -6, -5, -2)))
