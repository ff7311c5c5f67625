"""The end-to-end batch runner (porechop_amd/runner.py: loading, set rules, trims, barcode calls,
splits, writing) against the REFERENCE CLI's own output files on the reference's own fixtures.

The goldens are the md5s tests/golden/make_golden.py recorded while running the unchanged reference
(`runs` in tests/golden/ref_calls.json.gz).  The alignments here come from the oracle through the
Aligner-shaped stand-in of tests/cpu_aligner.py -- this test is about the HOST logic; the same
runner on the GPU is tests/test_gpu_runner.py.  Needs the fixture files of /root/reference/test, so
it runs in the build container only."""
import hashlib
import os

import pytest

from tests.cpu_aligner import OracleAligner
from tests.golden_io import load_ref_calls

REF_TEST = "/root/reference/test"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_TEST), reason="reference fixtures not present")


def md5_of_outputs(path):
    h = hashlib.md5()
    if os.path.isdir(path):
        for fn in sorted(os.listdir(path)):
            h.update(fn.encode())
            with open(os.path.join(path, fn), "rb") as f:
                h.update(f.read())
    else:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def options_from_argv(tail):
    from porechop_amd.runner import Options
    o = Options()
    it = iter(tail)
    barcode = False
    for t in it:
        if t == "--threads":
            next(it)
        elif t == "--end_size":
            o.end_size = int(next(it))
        elif t == "--middle_threshold":
            o.middle_threshold = float(next(it))
        elif t == "--no_split":
            o.no_split = True
        elif t == "--scoring_scheme":
            o.scoring_scheme = tuple(int(x) for x in next(it).split(","))
        elif t == "-b":
            next(it)
            barcode = True
        else:
            raise AssertionError("unhandled option " + t)
    return o, barcode


def test_runner_reproduces_reference_outputs(oracle, tmp_path):
    from porechop_amd import runner
    runs = load_ref_calls()["runs"]
    assert len(runs) >= 15
    for name, info in runs.items():
        opts, barcode = options_from_argv(info["argv_tail"])
        inp = os.path.join(REF_TEST, info["fixture"])
        al = OracleAligner(oracle, opts.scoring_scheme)
        if barcode:
            target = str(tmp_path / ("out_" + name))
            res = runner.run(inp, barcode_dir=target, options=opts, aligner=al)
        else:
            target = str(tmp_path / ("out_" + name + ".fastq"))
            res = runner.run(inp, output=target, options=opts, aligner=al)
        assert md5_of_outputs(target) == info["output_md5"], (name, res.matching_sets, res.files)
