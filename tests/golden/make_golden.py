#!/usr/bin/env python3
"""Generate the committed golden vectors from the REFERENCE ITSELF.

Run in the build container (needs /root/reference and g++):

    python tests/golden/make_golden.py

What it does
  1. ``make -C oracle ref``  -> oracle/_ref/cpp_functions.so from the reference's own sources.
  2. Copies the reference's *Python* package to a scratch dir outside the repo, drops the
     compiled .so next to it (that is where porechop/cpp_function_wrappers.py:21-25 looks),
     and imports it from there.  Nothing from the reference is copied into this repository.
  3. Runs the reference CLI (porechop.porechop.main) over its bundled fixtures with the
     option sets its own tests use, with ``porechop.nanopore_read.adapter_alignment``
     wrapped by a recorder: every (read window, adapter, scoring scheme) -> 7-field string
     that crosses the C ABI is captured -> tests/golden/ref_calls.json.gz
     (+ md5 of each run's output file -> the end-to-end goldens).
  4. Runs the compiled reference over the seeded synthetic cases of tests/pairgen.py
     -> tests/golden/ref_synthetic.json.gz   (inputs are regenerated from the seed at test
     time; a sha1 of the inputs is stored to detect generator drift).
  5. Records the adapter panel's sequences (start/end of the 119 sets, read from the
     unchanged porechop/adapters.py at run time) -> tests/golden/panel.json
     so GPU-box tests and bench.py can use the real panel without /root/reference.
"""
import gzip
import hashlib
import io
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile
from contextlib import redirect_stdout, redirect_stderr

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)

from tests.pairgen import LINEAR_SCHEMES, SCHEMES, case_stream  # noqa: E402


def stage_reference(tmp):
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "ref"])
    dst = os.path.join(tmp, "ref")
    os.makedirs(dst)
    shutil.copytree(os.path.join(REFERENCE, "porechop"), os.path.join(dst, "porechop"),
                    ignore=shutil.ignore_patterns("include", "src", "*.so", "__pycache__"))
    shutil.copy(os.path.join(REPO, "oracle", "_ref", "cpp_functions.so"),
                os.path.join(dst, "porechop", "cpp_functions.so"))
    return dst


RUNS = [
    # (name, fixture (relative to reference test/), argv tail)
    ("one_default", "test_one_adapter_set.fastq", []),
    ("one_threads8", "test_one_adapter_set.fastq", ["--threads", "8"]),
    ("one_end50", "test_one_adapter_set.fastq", ["--end_size", "50"]),
    ("one_end100", "test_one_adapter_set.fastq", ["--end_size", "100"]),
    ("one_end200", "test_one_adapter_set.fastq", ["--end_size", "200"]),
    ("one_mid96", "test_one_adapter_set.fastq", ["--middle_threshold", "96"]),
    ("one_mid97", "test_one_adapter_set.fastq", ["--middle_threshold", "97"]),
    ("one_nosplit", "test_one_adapter_set.fastq", ["--no_split"]),
    ("one_scheme", "test_one_adapter_set.fastq", ["--scoring_scheme", "2,-3,-5,-2"]),
    ("two_default", "test_two_adapter_sets.fastq", []),
    ("barcodes_default", "test_barcodes.fastq", ["-b", "BARCODE_DIR"]),
    ("choose1", "test_choose_barcodes_1.fasta", ["-b", "BARCODE_DIR"]),
    ("choose2", "test_choose_barcodes_2.fasta", ["-b", "BARCODE_DIR"]),
    ("albacore", "test_albacore_directory", ["-b", "BARCODE_DIR"]),
    ("albacore_mid85", "test_albacore_directory", ["-b", "BARCODE_DIR", "--middle_threshold", "85"]),
]


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


def record_reference_runs(refdir, tmp):
    sys.path.insert(0, refdir)
    import porechop.nanopore_read as nr          # the reference's module, unchanged
    import porechop.porechop as pp
    import porechop.adapters as adapters_mod

    strings, str_idx = [], {}
    calls, call_idx = [], {}
    runs = {}
    real = nr.adapter_alignment
    current = {"name": None, "n": 0}

    def intern(s):
        i = str_idx.get(s)
        if i is None:
            i = len(strings)
            str_idx[s] = i
            strings.append(s)
        return i

    def recorder(read_seq, adapter_seq, scores):
        out = real(read_seq, adapter_seq, scores)
        key = (read_seq, adapter_seq, tuple(scores))
        current["n"] += 1
        if key not in call_idx:
            call_idx[key] = len(calls)
            calls.append([intern(read_seq), intern(adapter_seq), list(scores), out])
        return out

    nr.adapter_alignment = recorder
    for name, fixture, tail in RUNS:
        # the panel objects carry state (best scores) across runs: reset like a fresh process
        for a in adapters_mod.ADAPTERS:
            a.best_start_score, a.best_end_score = 0.0, 0.0
        inp = os.path.join(REFERENCE, "test", fixture)
        outdir = os.path.join(tmp, "out_" + name)
        argv = ["porechop", "-i", inp, "-v", "0"]
        tail = [outdir if t == "BARCODE_DIR" else t for t in tail]
        if "-b" in tail:
            out_target = outdir
        else:
            out_target = outdir + ".fastq"
            argv += ["-o", out_target]
        argv += tail
        if "--threads" not in tail:
            argv += ["--threads", "1"]
        current["name"], current["n"] = name, 0
        sys.argv = argv
        buf = io.StringIO()
        with redirect_stdout(buf), redirect_stderr(buf):
            pp.main()
        runs[name] = {"argv_tail": tail if "-b" not in tail else [t if t != outdir else "BARCODE_DIR" for t in tail],
                      "fixture": fixture, "calls": current["n"],
                      "output_md5": md5_of_outputs(out_target)}
        print("  run %-18s calls=%6d md5=%s" % (name, current["n"], runs[name]["output_md5"]))
    nr.adapter_alignment = real

    panel = []
    for a in adapters_mod.ADAPTERS:
        panel.append({"name": a.name,
                      "start": list(a.start_sequence) if a.start_sequence else None,
                      "end": list(a.end_sequence) if a.end_sequence else None})
    return strings, calls, runs, panel


def runner_goldens(refdir, tmp, only=None):
    """Whole-run goldens on the seeded synthetic inputs of tests/readgen.py: the unchanged reference
    CLI's output files (md5 of their content) for every case in readgen.RUNNER_CASES."""
    from tests import readgen
    import porechop.porechop as pp
    import porechop.adapters as adapters_mod
    out = {}
    built = {}
    cwd = os.getcwd()
    for name, dataset, mode, extra in readgen.RUNNER_CASES:
        if only is not None and name not in only:
            continue
        if dataset not in built:
            built[dataset] = readgen.build_dataset(dataset, os.path.join(tmp, "datasets"))
        inp = built[dataset]
        for a in adapters_mod.ADAPTERS:
            a.best_start_score, a.best_end_score = 0.0, 0.0
        work = os.path.join(tmp, "run_" + name)
        os.makedirs(work)
        if mode == "b":
            target = os.path.join(work, "bins")
            argv = ["porechop", "-i", inp, "-b", target, "-v", "0", "--threads", "1"] + extra
        else:
            target = os.path.join(work, mode[2:])
            argv = ["porechop", "-i", inp, "-o", target, "-v", "0", "--threads", "1"] + extra
        sys.argv = argv
        os.chdir(work)                       # the reference writes a TEMP_<pid> file into the cwd for .gz outputs
        buf = io.StringIO()
        try:
            with redirect_stdout(buf), redirect_stderr(buf):
                pp.main()
            outputs = readgen.output_md5s(target)
            error = None
        except SystemExit as e:
            outputs, error = {}, str(e)
        finally:
            os.chdir(cwd)
        out[name] = {"dataset": dataset, "mode": mode, "argv": extra, "input_sha1": readgen.dataset_sha1(inp),
                     "outputs": outputs, "exit": error}
        print("  case %-28s files=%d %s" % (name, len(outputs), error or ""))
    return out


def synthetic(ref_so):
    import ctypes
    lib = ctypes.CDLL(ref_so)
    lib.adapterAlignment.argtypes = [ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_int] * 4
    lib.adapterAlignment.restype = ctypes.c_void_p
    lib.freeCString.argtypes = [ctypes.c_void_p]
    sets = []
    for seed, count, schemes, tag in [(101, 6000, SCHEMES, "affine"), (102, 6000, SCHEMES, "affine"),
                                      (103, 4000, LINEAR_SCHEMES, "linear")]:
        rng = random.Random(seed * 7 + 1)
        h = hashlib.sha1()
        outs = []
        for rd, ad in case_stream(seed, count):
            sc = rng.choice(schemes)
            h.update(("%s|%s|%r\n" % (rd, ad, sc)).encode())
            p = lib.adapterAlignment(rd.encode(), ad.encode(), *sc)
            outs.append(ctypes.cast(p, ctypes.c_char_p).value.decode())
            lib.freeCString(p)
        sets.append({"seed": seed, "count": count, "scheme_seed": seed * 7 + 1, "schemes": tag,
                     "inputs_sha1": h.hexdigest(), "results": outs})
    return sets


def main():
    tmp = tempfile.mkdtemp(prefix="pc_golden_")
    try:
        refdir = stage_reference(tmp)
        if len(sys.argv) > 2 and sys.argv[1] == "--runner-cases":
            # mint only the named whole-run cases and merge them into runner_goldens.json
            sys.path.insert(0, refdir)
            path = os.path.join(HERE, "runner_goldens.json")
            with open(path) as f:
                doc = json.load(f)
            doc["cases"].update(runner_goldens(refdir, tmp, only=set(sys.argv[2:])))
            with open(path, "w") as f:
                json.dump(doc, f, indent=1, sort_keys=True)
            return
        print("recording reference runs ...")
        strings, calls, runs, panel = record_reference_runs(refdir, tmp)
        meta = {"generator": "tests/golden/make_golden.py", "reference": "rrwick/Porechop v0.2.4 "
                "compiled by oracle/Makefile (g++ -std=c++14 -O3 -DNDEBUG)",
                "n_strings": len(strings), "n_calls": len(calls)}
        with gzip.open(os.path.join(HERE, "ref_calls.json.gz"), "wt", compresslevel=9) as f:
            json.dump({"meta": meta, "runs": runs, "strings": strings, "calls": calls}, f)
        # the adapter panel (names + sequences: data, not code) -- once for the tests, once as the
        # package's own table (porechop_amd/panel.py)
        for dst in (os.path.join(HERE, "panel.json"), os.path.join(REPO, "porechop_amd", "panel.json")):
            with open(dst, "w") as f:
                json.dump(panel, f, indent=0)
        print("unique calls: %d, unique strings: %d" % (len(calls), len(strings)))
        print("whole-run goldens on synthetic inputs ...")
        with open(os.path.join(HERE, "runner_goldens.json"), "w") as f:
            json.dump({"meta": meta, "cases": runner_goldens(refdir, tmp)}, f, indent=1, sort_keys=True)
        print("synthetic cases ...")
        sets = synthetic(os.path.join(REPO, "oracle", "_ref", "cpp_functions.so"))
        with gzip.open(os.path.join(HERE, "ref_synthetic.json.gz"), "wt", compresslevel=9) as f:
            json.dump({"meta": meta, "sets": sets}, f)
        print("done")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
