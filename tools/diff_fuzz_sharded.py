#!/usr/bin/env python3
"""tools/diff_fuzz.py with the runner side run by W processes over gloo (CPU; the oracle stands in for the GPUs): the reference
CLI against porechop_amd.runner.run under torch.distributed with 2 ... 5 ranks, on random reads and options.  Inputs: one plain
FASTQ file or one gzip file of sized members (the sharded route: every rank parses / inflates only its own share and writes its
own span of the shared output files), plain FASTA (cut at '>' lines since round 6) and one gzip member (cut at member starts
since round 6: a single member is all rank 0's).  The multi-GPU path cannot be run on hardware from here; this is its
functional evidence (round 6: profiles/r06_diff_fuzz_sharded.txt).
    python tools/diff_fuzz_sharded.py [cases] [seed] [world] [--long]      (--long: reads of 100 kb and more among them)"""
import io
import os
import random
import shutil
import socket
import sys
import tempfile
from contextlib import redirect_stderr, redirect_stdout

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker(rank, world, port, tasks, results):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from porechop_amd import runner
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import options_from_argv
    oracle = Oracle()
    took = []
    orig = runner.run_sharded

    def spy(*a, **kw):
        r = orig(*a, **kw)
        took.append(r is not None)
        return r
    runner.run_sharded = spy
    while True:
        task = tasks.get()
        if task is None:
            break
        inp, mode, extra, gtarget, fast = task
        opts = options_from_argv(extra)
        stand_in = OracleAligner(oracle, opts.scoring_scheme)
        stand_in.fast_prefilter = fast
        del took[:]
        try:
            res = runner.run(inp, barcode_dir=gtarget if mode == "b" else None, output=None if mode == "b" else gtarget,
                             options=opts, aligner=stand_in)
            out = (None, len(res.start_trim) if res.start_trim is not None else -1, res.n_reads, bool(took and took[0]))
        except runner.UsageError as e:
            out = (str(e), -1, -1, bool(took and took[0]))
        except ValueError as e:
            out = ("error: " + str(e), -1, -1, False)
        dist.barrier()
        results.put((rank, out))
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    from tests import readgen
    from tests.golden.make_golden import stage_reference
    long_lines = "--long" in sys.argv
    if long_lines:
        sys.argv.remove("--long")
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    tmp = tempfile.mkdtemp(prefix="pc_fuzz_sh_")
    refdir = stage_reference(tmp)
    sys.path.insert(0, refdir)
    import porechop.porechop as pp
    import porechop.adapters as adapters_mod
    from porechop_amd import io as pio
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    tasks = [ctx.Queue() for _ in range(world)]
    results = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, tasks[r], results)) for r in range(world)]
    for p in procs:
        p.start()

    def maybe(o, p, *args):
        if rng.random() < p:
            o.extend(args)
    bad = sharded_runs = 0
    for k in range(cases):
        kind = rng.choice(["native", "native", "rapid", "ligation"])
        seed, nreads = rng.randint(1, 10 ** 6), rng.choice([7, 25, 60])
        reads = {"native": lambda: readgen.native_reads(seed, nreads, barcodes=tuple(rng.sample(range(1, 13), 3))),
                 "rapid": lambda: readgen.rapid_reads(seed, nreads), "ligation": lambda: readgen.ligation_reads(seed, nreads)}[kind]()
        if rng.random() < 0.2:                                       # very uneven shares: a few long reads among short ones
            reads = [(n, s_ * (6 if i % 9 == 0 else 1), q * (6 if i % 9 == 0 else 1)) for i, (n, s_, q) in enumerate(reads)]
        if long_lines and rng.random() < 0.5:                        # lines that span several 64 KB gzip members
            reads = [(n, s_ * (45 if i % 11 == 3 else 1), q * (45 if i % 11 == 3 else 1)) for i, (n, s_, q) in enumerate(reads)]
        work = os.path.join(tmp, "case%d" % k)
        os.makedirs(work)
        layout = rng.choice(["plain", "plain", "sized", "sized", "one", "fasta"])
        inp = os.path.join(work, "in.fasta" if layout == "fasta" else "in.fastq")
        with open(inp, "w") as f:
            f.write(readgen.fasta_text(reads) if layout == "fasta" else readgen.fastq_text(reads))
        if layout == "sized":
            pio.gzip_file(inp, inp + ".gz"); os.remove(inp); inp += ".gz"
        elif layout == "one":
            import gzip
            data = open(inp, "rb").read(); os.remove(inp); inp += ".gz"
            open(inp, "wb").write(gzip.compress(data, 1))
        barcodes = kind in ("native", "rapid") and rng.random() < 0.5
        extra = []
        maybe(extra, 0.3, "--end_size", str(rng.choice([80, 150, 200])))
        maybe(extra, 0.3, "--extra_end_trim", str(rng.choice([0, 2, 7])))
        maybe(extra, 0.3, "--middle_threshold", str(rng.choice([75, 85, 97])))
        maybe(extra, 0.3, "--check_reads", str(rng.choice([3, 10, 40, 10000])))
        maybe(extra, 0.2, "--min_split_read_size", str(rng.choice([1, 1000])))
        maybe(extra, 0.15, "--no_split")
        maybe(extra, 0.15, "--discard_middle")
        maybe(extra, 0.25, "--format", rng.choice(["fasta", "fastq", "fastq.gz", "auto"]))
        if barcodes:
            maybe(extra, 0.3, "--require_two_barcodes")
            maybe(extra, 0.3, "--barcode_diff", str(rng.choice([0, 5, 15])))
            maybe(extra, 0.2, "--untrimmed")
            maybe(extra, 0.2, "--discard_unassigned")
        mode = "b" if barcodes else "o:" + rng.choice(["out.fastq", "out.fasta", "out.fastq.gz", "out.txt"])
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
        finally:
            os.chdir(cwd)
        gtarget = os.path.join(work, "got", "bins" if mode == "b" else mode[2:])
        os.makedirs(os.path.dirname(gtarget))
        fast = rng.random() < 0.5
        for q in tasks:
            q.put((inp, mode, extra, gtarget, fast))
        outs = dict(results.get(timeout=900) for _ in range(world))
        got = readgen.output_md5s(gtarget) if os.path.exists(gtarget) else {}
        gexit = outs[0][0]
        took_sharded = all(o[3] for o in outs.values())
        ok = got == want and gexit == wexit and all(o[0] == gexit for o in outs.values())
        if took_sharded and gexit is None:
            sharded_runs += 1
            total = outs[0][2]
            ok = ok and sum(o[1] for o in outs.values()) == total            # the ranks' shares add up: nobody held every read twice
        if layout in ("plain", "sized") and gexit is None and not took_sharded:
            ok = False                                                        # these layouts must take the sharded route
        bad += not ok
        print("%s case %2d %-8s %-6s %-14s %s %s%s" % ("ok " if ok else "BAD", k, kind, layout, mode, "sharded " if took_sharded else "gathered",
              " ".join(extra), "" if ok else "\n     want %r %r\n     got  %r %r %r" % (wexit, want, gexit, got, outs)), flush=True)
    for q in tasks:
        q.put(None)
    for p in procs:
        p.join(timeout=60)
    shutil.rmtree(tmp, ignore_errors=True)
    print("cases=%d world=%d mismatches=%d (runs that took the sharded route: %d)" % (cases, world, bad, sharded_runs))


if __name__ == "__main__":
    main()
