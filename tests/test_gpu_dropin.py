"""The drop-in boundary exercised the way a Porechop user would: the reference's OWN, UNCHANGED Python (staged by
`make -C oracle ref` into the git-ignored oracle/_ref/porechop_ref, which travels to the GPU box) over the real HIP library.

  mode B  tests/ref_cli.py --dropin: `porechop_amd.dropin.install(pp); pp.main()` -- every recorded run of
          tests/golden/ref_calls.json.gz (the reference CLI's own outputs on its own fixtures): output md5 equal, ZERO memo
          misses, as many lookups as the reference made calls (porechop/cpp_function_wrappers.py:42-63 replaced by the memo,
          nanopore_read.py:17,476-491 unchanged).
  mode A  only `porechop/cpp_functions.so` swapped for libporechop_amd.so, nothing in Python patched: porechop-runner.py on
          test_one_adapter_set.fastq gives the reference's md5 (215ec381ab055447df3fc14fa9bc1a73), every call
          through `adapterAlignment` / `freeCString` of include/porechop_amd.h part 1.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import pytest

from tests.ref_cli import DEFAULT_STAGE, staged

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not staged(), reason="no staged reference (make -C oracle ref)")]

RUNS = ["one_default", "one_threads8", "one_end50", "one_end100", "one_end200", "one_mid96", "one_mid97", "one_nosplit",
        "one_scheme", "two_default", "barcodes_default", "choose1", "choose2", "albacore", "albacore_mid85"]


def md5_of(path):
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


def porechop_argv(info, out):
    """The command line tests/golden/make_golden.py recorded the run with -> (argv after the program name, output path)."""
    argv = ["-i", os.path.join(DEFAULT_STAGE, "test", info["fixture"]), "-v", "0"]
    tail = [out if t == "BARCODE_DIR" else t for t in info["argv_tail"]]
    if "-b" in tail:
        target = out
    else:
        target = out + ".fastq"
        argv += ["-o", target]
    argv += tail
    if "--threads" not in tail:
        argv += ["--threads", "1"]
    return argv, target


@pytest.mark.parametrize("run", RUNS)
def test_unchanged_reference_main_over_the_gpu_dropin(goldens, tmp_path, run):
    info = goldens["runs"][run]
    argv, target = porechop_argv(info, str(tmp_path / "out"))
    report = str(tmp_path / "report.json")
    res = subprocess.run([sys.executable, os.path.join(REPO, "tests", "ref_cli.py"), "--dropin", "--report", report, "--"] + argv,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    assert md5_of(target) == info["output_md5"]
    with open(report) as f:
        rep = json.load(f)
    assert rep["dropin"]["misses"] == 0, rep
    assert rep["dropin"]["hits"] == info["calls"], (rep, info["calls"])


def test_mode_a_only_the_shared_object_swapped(goldens, tmp_path):
    import porechop_amd
    stage = tmp_path / "stage"
    shutil.copytree(DEFAULT_STAGE, stage, ignore=shutil.ignore_patterns("test", "__pycache__", "cpp_functions.so"))
    shutil.copy(porechop_amd.LIB_PATH, stage / "porechop" / "cpp_functions.so")
    out = str(tmp_path / "out.fastq")
    env = dict(os.environ)
    env["PC_MODE_A_MAPS"] = str(tmp_path / "maps.txt")
    # porechop-runner.py itself, unchanged; a sitecustomize-free way to see which library it mapped: run it under -c
    code = ("import runpy, sys, os; sys.argv = %r; "
            "\ntry:\n    runpy.run_path(%r, run_name='__main__')\nfinally:\n"
            "    open(os.environ['PC_MODE_A_MAPS'], 'w').write(open('/proc/self/maps').read())\n"
            % (["porechop-runner.py", "-i", os.path.join(DEFAULT_STAGE, "test", "test_one_adapter_set.fastq"), "-o", out,
                "-v", "0", "--threads", "4"], str(stage / "porechop-runner.py")))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=str(stage))
    assert res.returncode == 0, res.stderr[-3000:]
    assert md5_of(out) == goldens["runs"]["one_default"]["output_md5"] == "215ec381ab055447df3fc14fa9bc1a73"
    maps = open(env["PC_MODE_A_MAPS"]).read()
    assert str(stage / "porechop" / "cpp_functions.so") in maps and "libamdhip64" in maps
    assert "oracle/_ref" not in maps
