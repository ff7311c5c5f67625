#!/usr/bin/env python3
"""Differential fuzz of the end-to-end runner's HOST logic against the unchanged reference CLI
(build container only: needs /root/reference and the compiled reference, oracle/_ref).  Random small
inputs and random option combinations; the runner's alignments come from the oracle through the
test stand-in (tests/cpu_aligner.py), so this exercises loading, set rules, trims, barcode calls,
splits, naming and writing -- not the kernels.      python tools/diff_fuzz.py [cases] [seed]"""
import io
import os
import random
import shutil
import sys
import tempfile
from contextlib import redirect_stderr, redirect_stdout

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests import readgen  # noqa: E402
from tests.cpu_aligner import OracleAligner  # noqa: E402
from tests.golden.make_golden import stage_reference  # noqa: E402
from tests.runner_cases import options_from_argv  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from porechop_amd import runner  # noqa: E402

# --emit DIR: besides comparing, keep every case's input and the reference's output md5s under DIR
# (cases.json), so that tools/replay_fuzz.py can run the same cases on a GPU box without the reference
emit = None
if "--emit" in sys.argv:
    i = sys.argv.index("--emit")
    emit = sys.argv[i + 1]
    del sys.argv[i:i + 2]
    os.makedirs(emit, exist_ok=True)
emitted = []
# --dropin: instead of the runner, the UNCHANGED reference Python is run a second time with porechop_amd.dropin installed over
# an oracle backend (mode B of INTEGRATION.md): same files, and not one alignment asked for that the batching had not foreseen
dropin_mode = "--dropin" in sys.argv
if dropin_mode:
    sys.argv.remove("--dropin")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp(prefix="pc_fuzz_")
refdir = stage_reference(tmp)
sys.path.insert(0, refdir)
import porechop.porechop as pp  # noqa: E402
import porechop.adapters as adapters_mod  # noqa: E402
oracle = Oracle()


from tests.fuzzcase import make_case, content_md5  # noqa: E402
from porechop_amd import io as pio  # noqa: E402

bad = 0
for k in range(cases):
    cseed = rng.randint(1, 10 ** 9)
    work = os.path.join(tmp, "case%d" % k)
    case = make_case(cseed, work, sized_gzip=pio.gzip_file)
    inp, mode, extra, kind = case["input"], case["mode"], case["argv"], case["kind"]
    # ---- reference
    for a in adapters_mod.ADAPTERS:
        a.best_start_score, a.best_end_score = 0.0, 0.0
    rwork = os.path.join(work, "ref"); os.makedirs(rwork)
    rtarget = os.path.join(rwork, "bins" if mode == "b" else mode[2:])
    sys.argv = ["porechop", "-i", inp, "-v", "0", "--threads", "1"] + (["-b", rtarget] if mode == "b" else ["-o", rtarget]) + extra
    cwd = os.getcwd(); os.chdir(rwork)
    try:
        with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
            pp.main()
        want, wexit = readgen.output_md5s(rtarget) if os.path.exists(rtarget) else {}, None
    except SystemExit as e:
        want, wexit = {}, str(e)
    except Exception:                                     # the reference dies with a traceback (an empty file in a directory, say)
        want, wexit = {}, "traceback"
    finally:
        os.chdir(cwd)
    # ---- the reference again, over the drop-in
    misses = None
    if dropin_mode:
        import importlib
        import porechop_amd.dropin as dropin
        from tests.test_dropin_reference import OracleBackend, OracleProductBackend
        nr = importlib.import_module("porechop.nanopore_read")
        saved = (pp.find_matching_adapter_sets, pp.find_adapters_at_read_ends, pp.find_adapters_in_read_middles, nr.adapter_alignment)
        for a in adapters_mod.ADAPTERS:
            a.best_start_score, a.best_end_score = 0.0, 0.0
        dropin.install(pp, rng.choice([OracleBackend, OracleProductBackend])(oracle))
        gwork = os.path.join(work, "got"); os.makedirs(gwork)
        gtarget = os.path.join(gwork, "bins" if mode == "b" else mode[2:])
        sys.argv = ["porechop", "-i", inp, "-v", "0", "--threads", "1"] + (["-b", gtarget] if mode == "b" else ["-o", gtarget]) + extra
        cwd = os.getcwd(); os.chdir(gwork)
        try:
            with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
                pp.main()
            got, gexit = readgen.output_md5s(gtarget) if os.path.exists(gtarget) else {}, None
        except SystemExit as e:
            got, gexit = {}, str(e)
        except Exception:
            got, gexit = {}, "traceback"
        finally:
            os.chdir(cwd)
            pp.find_matching_adapter_sets, pp.find_adapters_at_read_ends, pp.find_adapters_in_read_middles, nr.adapter_alignment = saved
        misses = dropin.stats().get("misses")
        ok = (got == want) and (gexit == wexit) and (misses == 0 or gexit is not None)
        bad += not ok
        print("%s case %2d %-8s %-12s %-14s misses %s %s%s" % ("ok " if ok else "BAD", k, kind, os.path.basename(inp), mode, misses, " ".join(extra), "" if ok else
              "\n     want %r %r\n     got  %r %r" % (wexit, want, gexit, got)), flush=True)
        continue
    # ---- runner
    opts = options_from_argv(extra)
    gtarget = os.path.join(work, "got", "bins" if mode == "b" else mode[2:])
    os.makedirs(os.path.dirname(gtarget))
    stand_in = OracleAligner(oracle, opts.scoring_scheme)
    stand_in.fast_prefilter = case["prefilter"]
    os.environ.pop("PC_STREAM_BLOCK_BYTES", None)
    if case["blocks"]:
        os.environ["PC_STREAM_BLOCK_BYTES"] = case["blocks"]
    try:
        runner.run(inp, barcode_dir=gtarget if mode == "b" else None, output=None if mode == "b" else gtarget,
                   options=opts, aligner=stand_in)
        got, gexit = readgen.output_md5s(gtarget) if os.path.exists(gtarget) else {}, None
    except runner.UsageError as e:
        got, gexit = {}, str(e)
    except ValueError as e:                               # (what the loader raises for files it cannot read)
        got, gexit = {}, "error: " + str(e)
    if wexit == "traceback" and gexit is not None:
        gexit = "traceback"                                # the reference has no message to compare: failing is what counts
    if emit:
        # the RECIPE (one seed regenerates input, mode and options: tests/fuzzcase.py) and what the reference made of it
        emitted.append({"cseed": cseed, "content_md5": content_md5(inp), "input": os.path.relpath(inp, work), "mode": mode, "argv": extra,
                        "prefilter": case["prefilter"], "blocks": case["blocks"], "kind": kind, "outputs": want, "exit": wexit,
                        "agreed_here": bool((got == want) and (gexit == wexit))})
    ok = (got == want) and (gexit == wexit)
    bad += not ok
    print("%s case %2d %-8s %-12s %-14s %s %s%s" % ("ok " if ok else "BAD", k, kind, os.path.basename(inp), mode, ("prefilter" if stand_in.fast_prefilter else "bound    ") + (" blocks=" + os.environ["PC_STREAM_BLOCK_BYTES"] if "PC_STREAM_BLOCK_BYTES" in os.environ else ""), " ".join(extra), "" if ok else "\n     want %r %r\n     got  %r %r" % (wexit, want, gexit, got)), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
if emit:
    import json
    with open(os.path.join(emit, "cases.json"), "w") as f:
        json.dump(emitted, f)
print("cases=%d mismatches=%d" % (cases, bad))
