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


def random_options(barcodes):
    o = []
    def maybe(p, *args):
        if rng.random() < p:
            o.extend(args)
    maybe(0.3, "--end_size", str(rng.choice([30, 80, 120, 150, 200])))
    maybe(0.3, "--min_trim_size", str(rng.choice([0, 2, 4, 10])))
    maybe(0.3, "--extra_end_trim", str(rng.choice([0, 1, 2, 7])))
    maybe(0.3, "--end_threshold", str(rng.choice([60, 75, 90])))
    maybe(0.3, "--middle_threshold", str(rng.choice([75, 85, 90, 97])))
    maybe(0.2, "--adapter_threshold", str(rng.choice([80, 90, 97])))
    maybe(0.2, "--check_reads", str(rng.choice([5, 40, 10000])))
    maybe(0.2, "--min_split_read_size", str(rng.choice([1, 200, 1000, 3000])))
    maybe(0.2, "--extra_middle_trim_good_side", str(rng.choice([0, 10, 50])))
    maybe(0.2, "--extra_middle_trim_bad_side", str(rng.choice([0, 100, 300])))
    maybe(0.15, "--scoring_scheme", rng.choice(["3,-6,-5,-2", "2,-3,-5,-2", "3,-6,-2,-5", "4,-5,-6,-6"]))
    maybe(0.15, "--no_split")
    maybe(0.15, "--discard_middle")
    maybe(0.25, "--format", rng.choice(["fasta", "fastq", "fastq.gz", "auto"]))
    if barcodes:
        maybe(0.3, "--require_two_barcodes")
        maybe(0.3, "--barcode_threshold", str(rng.choice([60, 75, 85])))
        maybe(0.3, "--barcode_diff", str(rng.choice([0, 5, 15])))
        maybe(0.2, "--untrimmed")
        maybe(0.2, "--discard_unassigned")
    return o


bad = 0
for k in range(cases):
    kind = rng.choice(["native", "native", "rapid", "ligation", "edge"])
    seed = rng.randint(1, 10 ** 6)
    nreads = rng.choice([25, 60])
    reads = {"native": lambda: readgen.native_reads(seed, nreads, barcodes=tuple(rng.sample(range(1, 13), 3))),
             "rapid": lambda: readgen.rapid_reads(seed, nreads), "ligation": lambda: readgen.ligation_reads(seed, nreads),
             "edge": lambda: None}[kind]()
    work = os.path.join(tmp, "case%d" % k)
    os.makedirs(work)
    if reads is not None and rng.random() < 0.3:
        # odd reads among the ordinary ones: RNA (more U than T: aligned as T, written back with EVERY T as U,
        # nanopore_read.py:23-35,97-147), a few U's only, lower case, runs of N / '-', qualities shorter than the sequence,
        # names with tabs and repeated blanks, an empty read
        odd = []
        for name, seq, qual in reads:
            r = rng.random()
            if r < 0.08:
                seq = seq.replace("T", "U")
            elif r < 0.12:
                seq = "".join(("U" if c == "T" and rng.random() < 0.3 else c) for c in seq)
            elif r < 0.18:
                seq = seq.lower()
            elif r < 0.22 and len(seq) > 300:
                p0 = rng.randrange(len(seq) - 100)
                seq = seq[:p0] + rng.choice("N-n") * rng.randrange(1, 90) + seq[p0 + 60:]
                qual = (qual * 2)[:len(seq)]
            elif r < 0.25:
                qual = qual[:rng.randrange(len(qual) + 1)]
            elif r < 0.28:
                name = name + "\tx  y " + name
            elif r < 0.29:
                seq, qual = "", ""
            odd.append((name, seq, qual))
        reads = odd
    if reads is None:
        inp = readgen.build_dataset("edge", work)
    elif rng.random() < 0.15:
        # a directory the way Albacore / Guppy lay them out (porechop.py:232-259): fastq files found recursively, in path
        # order, the check reads spread over the files, the basecaller's barcode taken from a /barcodeNN/ or /unclassified/
        # path component; some files gzip-ed, a file that is not a fastq in between, an upper-case extension
        import gzip
        inp = os.path.join(work, "indir")
        subs = rng.sample(["", "pass/barcode01", "pass/barcode02", "pass/barcode11", "fail/unclassified", "x/barcode07/y", "misc"], rng.randrange(1, 5))
        nfiles = rng.randrange(1, 7)
        cuts = sorted(rng.randrange(len(reads) + 1) for _ in range(nfiles - 1))
        for j, (a, b) in enumerate(zip([0] + cuts, cuts + [len(reads)])):
            d = os.path.join(inp, rng.choice(subs))
            os.makedirs(d, exist_ok=True)
            text = readgen.fastq_text(reads[a:b]).encode()
            if not text and rng.random() < 0.7:
                continue                                               # (an empty .fastq file ends the reference with an error: rarely)
            ext = rng.choice([".fastq", ".fastq", ".fastq.gz", ".FASTQ"])
            with open(os.path.join(d, "part%d%s" % (rng.randrange(1000), ext)), "wb") as f:
                f.write(gzip.compress(text, 1) if ext.endswith(".gz") else text)
        os.makedirs(os.path.join(inp, "misc"), exist_ok=True)
        open(os.path.join(inp, "misc", "notes.txt"), "w").write("not reads\n")
    else:
        as_fasta = rng.random() < 0.2
        inp = os.path.join(work, "in.fasta" if as_fasta else "in.fastq")
        with open(inp, "w") as f:
            f.write(readgen.fasta_text(reads) if as_fasta else readgen.fastq_text(reads))
        # a third of the inputs gzip-ed, in the layouts the readers tell apart (the reference decides by magic bytes,
        # porechop/misc.py:60-81): one member, several members with zero padding between two of them, sized members
        layout = rng.choice(["", "", "one", "members", "sized"])
        if layout:
            import gzip
            text = open(inp, "rb").read()
            os.remove(inp)
            inp += ".gz"
            if layout == "one":
                blob = gzip.compress(text, rng.choice([1, 6, 9]))
            elif layout == "sized":
                from porechop_amd import io as pio
                open(inp[:-3], "wb").write(text)
                pio.gzip_file(inp[:-3], inp)
                os.remove(inp[:-3])
                blob = None
            else:
                marks = sorted({0, len(text)} | {m for m in (text.find(b"\n>" if as_fasta else b"\n@", rng.randrange(max(1, len(text)))) + 1 for _ in range(3)) if m > 0})
                blob = b""
                for a, b in zip(marks, marks[1:]):
                    blob += gzip.compress(text[a:b], 1) + (b"\0" * rng.choice([0, 0, 13, 600]))
            if blob is not None:
                open(inp, "wb").write(blob)
    barcodes = kind in ("native", "rapid", "edge") and rng.random() < 0.5
    extra = random_options(barcodes)
    if not barcodes and "--untrimmed" in extra:
        extra.remove("--untrimmed")
    mode = "b" if barcodes else "o:" + rng.choice(["out.fastq", "out.fasta", "out.fastq.gz", "out.txt"])
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
    # half of the runs take the middle scan behind the exact PREFILTER, the GPU library's default route (Pipeline.phase_c(prefilter
    # =True) -> _prefiltered_scan: survivors, set grouping, sparse records, the masked rounds), the stand-in deciding "within
    # max_edits" by the oracle's plain edit-distance DP; the other half behind the score bound
    stand_in.fast_prefilter = rng.random() < 0.5
    # a third of the runs as a STREAM of small blocks (run_streamed: plain and gzip FASTQ files above two blocks; parse k+1 ||
    # scan k || write k-1, phase A on the first block, gzip output appended member by member)
    os.environ.pop("PC_STREAM_BLOCK_BYTES", None)
    if rng.random() < 0.33:
        os.environ["PC_STREAM_BLOCK_BYTES"] = str(rng.choice([3000, 20000, 150000]))
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
        keep = os.path.join(emit, "case%d_%s" % (k, os.path.basename(inp)))
        if os.path.isdir(inp):
            shutil.copytree(inp, keep)
        else:
            shutil.copy(inp, keep)
        emitted.append({"input": os.path.basename(keep), "mode": mode, "argv": extra, "outputs": want, "exit": wexit})
    ok = (got == want) and (gexit == wexit)
    bad += not ok
    print("%s case %2d %-8s %-12s %-14s %s %s%s" % ("ok " if ok else "BAD", k, kind, os.path.basename(inp), mode, ("prefilter" if stand_in.fast_prefilter else "bound    ") + (" blocks=" + os.environ["PC_STREAM_BLOCK_BYTES"] if "PC_STREAM_BLOCK_BYTES" in os.environ else ""), " ".join(extra), "" if ok else "\n     want %r %r\n     got  %r %r" % (wexit, want, gexit, got)), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
if emit:
    import json
    with open(os.path.join(emit, "cases.json"), "w") as f:
        json.dump(emitted, f)
print("cases=%d mismatches=%d" % (cases, bad))
