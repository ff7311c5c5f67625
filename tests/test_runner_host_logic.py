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
    runs = (("auto", "native_default"), ("fasta", "native_to_fasta"), ("fasta.gz", "native_default"))
    procs = [subprocess.Popen([sys.executable, "-c", code, inp, fmt], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=repo)
             for fmt, _ in runs]                                       # (side by side: three interpreters start at once)
    for (fmt, golden), pr in zip(runs, procs):
        out, err = pr.communicate(timeout=600)
        assert pr.returncode == 0, err[-2000:]
        want = list(cases[golden]["outputs"].values())[0]
        assert hashlib.md5(out).hexdigest() == want, (fmt, len(out))


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


def test_panel_default_is_the_recorded_one_and_agrees_with_the_reference_module():
    """load_panel() uses the recorded panel.json whatever is importable as `porechop` (the tested, benchmarked
    panel); reading porechop.adapters.ADAPTERS is an explicit opt-in, and where /root/reference exists the two
    must be the same table."""
    import os
    import sys
    import pytest
    from porechop_amd import panel
    recorded = panel.load_panel()
    assert len(recorded) == 119 and panel.PANEL_SOURCE == "recorded"
    if not os.path.isdir("/root/reference/porechop"):
        pytest.skip("reference checkout not present")
    sys.path.insert(0, "/root/reference")
    try:
        for m in [k for k in sys.modules if k == "porechop" or k.startswith("porechop.")]:
            del sys.modules[m]
        assert [(s.name, s.start, s.end) for s in panel.load_panel()] == [(s.name, s.start, s.end) for s in recorded]
        assert panel.PANEL_SOURCE == "recorded"          # importable or not, the default does not change
        live = panel.load_panel(prefer_reference=True)
        assert panel.PANEL_SOURCE == "reference module"
    finally:
        sys.path.remove("/root/reference")
        for m in [k for k in sys.modules if k == "porechop" or k.startswith("porechop.")]:
            del sys.modules[m]
    assert [(s.name, s.start, s.end) for s in live] == [(s.name, s.start, s.end) for s in recorded]


def test_streamed_runs_write_the_same_files(oracle, tmp_path, monkeypatch):
    """A plain FASTQ input larger than two stream blocks is run as a stream of blocks (runner.run_streamed: parse
    block k+1 | scan block k | write block k-1).  With blocks of a few kB: (1) the recorded runs of the reference CLI
    are still reproduced byte for byte, and (2) with a small --check_reads -- so that the first block does not
    swallow the file -- streamed and whole-file runs of the same options write identical files (bins, splits,
    numbering, FASTA / gzip output, --untrimmed, --discard_unassigned)."""
    import hashlib
    import os
    from porechop_amd import runner
    from tests import readgen
    cases = load_cases()
    datasets = {}
    seen_blocks = []
    real_segment = runner.ReadSet.segment

    def counting_segment(path, begin, target):
        rs, nxt = real_segment(path, begin, target)
        seen_blocks.append((begin, nxt))
        return rs, nxt
    monkeypatch.setattr(runner.ReadSet, "segment", staticmethod(counting_segment))
    monkeypatch.setenv("PC_STREAM_BLOCK_BYTES", "6000")
    # (1) goldens
    done = 0
    for name in ("native_check20", "native_check0", "native_default", "native_bins", "ligation_default", "edge_default"):
        case = cases[name]
        got = run_case(name, case, str(tmp_path), datasets, make_aligner=lambda sc: OracleAligner(oracle, sc))
        assert got == case["outputs"], (name, got, case["outputs"])
        done += 1
    assert done == 6 and len(seen_blocks) > 20          # the small --check_reads cases really ran as many blocks

    # (2) streamed == whole-file, options that depend on per-read results only
    def md5s(path):
        return readgen.output_md5s(path)
    inp = datasets[cases["native_bins"]["dataset"]]
    variants = [("o", "out.fastq", {}), ("o", "out.fasta", {}), ("o", "out.fastq.gz", {}), ("b", "bins", {}),
                ("b", "bins_u", {"untrimmed": True}), ("b", "bins_d", {"discard_unassigned": True, "require_two_barcodes": True}),
                ("o", "split.fastq", {"min_split_read_size": 100, "middle_threshold": 80.0})]
    for k, (mode, fname, extra) in enumerate(variants):
        outs = []
        for streamed in (True, False):
            monkeypatch.setenv("PC_STREAM_BLOCK_BYTES", "6000" if streamed else str(1 << 40))
            opts = runner.Options(check_reads=15, **extra)
            work = tmp_path / ("v%d_%d" % (k, streamed))
            work.mkdir()
            target = str(work / fname)
            n0 = len(seen_blocks)
            if mode == "b":
                runner.run(inp, barcode_dir=target, options=opts, aligner=OracleAligner(oracle, opts.scoring_scheme))
            else:
                runner.run(inp, output=target, options=opts, aligner=OracleAligner(oracle, opts.scoring_scheme))
            assert (len(seen_blocks) - n0 > 3) == streamed
            outs.append(md5s(target))
        assert outs[0] == outs[1] and outs[0], (fname, outs)


def test_streamed_run_that_fails_midway_leaves_no_partial_output(tmp_path, oracle):
    """A later block that cannot be parsed ends a streamed run with the loader's error -- and without the files the
    earlier blocks had already written (a whole-file run would not have written anything)."""
    import pytest
    from porechop_amd import runner
    from tests.cpu_aligner import OracleAligner
    import random
    rng = random.Random(3)
    recs = []
    for i in range(40):
        s = "".join(rng.choice("ACGT") for _ in range(300))
        recs.append("@r%d\n%s\n+\n%s\n" % (i, s, "5" * 300))
    text = "".join(recs[:30]) + "@broken\nACGT\n" + "".join(recs[30:])        # a record without its '+' and quality lines
    inp = tmp_path / "in.fastq"
    inp.write_text(text)
    out = tmp_path / "out.fastq"
    opts = runner.Options()
    opts.check_reads = 5              # the first block only has to hold these: the broken record lies in a later block
    with pytest.raises(ValueError):
        runner.run_streamed(str(inp), str(out), None, opts, aligner=OracleAligner(oracle, tuple(opts.scoring_scheme)), block_bytes=2500)
    assert not out.exists()
