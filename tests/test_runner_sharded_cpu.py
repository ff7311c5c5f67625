"""World-size-2 run of the runner over ONE shared plain FASTQ file (gloo, CPU; the oracle stands in for the GPU through
tests/cpu_aligner.py): neither rank loads the whole file -- each parses the records that start in its half of the bytes --
and each writes its own span of the shared output files (runner.run_sharded).  The files must be the single-process ones,
i.e. the reference CLI's (tests/golden/runner_goldens.json): one file, barcode bins, FASTA and gzip outputs, a run in
which nothing is found, and a three-rank run whose ranks get very uneven shares."""
import os
import socket

import pytest
import torch.multiprocessing as mp

CASES = {2: ["native_default", "native_bins_gz", "native_to_fasta", "edge_bins", "nothing_found", "native_bins_untrimmed", "ligation_default"],
         3: ["native_gz_out", "native_bins", "edge_default"]}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, workdir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from porechop_amd import runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import load_cases, options_from_argv
    oracle = Oracle()
    cases = load_cases()
    out, shares = {}, {}
    seen_sharded = []
    orig = runner.run_sharded

    def spy(*a, **kw):
        r = orig(*a, **kw)
        seen_sharded.append(r is not None)
        return r
    runner.run_sharded = spy
    built = {}
    for name in CASES[world]:
        case = cases[name]
        if rank == 0 and case["dataset"] not in built:            # ONE copy of the input, shared by the ranks
            built[case["dataset"]] = readgen.build_dataset(case["dataset"], os.path.join(workdir, "datasets"))
            assert readgen.dataset_sha1(built[case["dataset"]]) == case["input_sha1"]
        box = [built.get(case["dataset"])]
        dist.broadcast_object_list(box, src=0)
        inp = box[0]
        built[case["dataset"]] = inp
        opts = options_from_argv(case["argv"])
        work = os.path.join(workdir, "run_" + name)
        if rank == 0:
            os.makedirs(work)
        dist.barrier()
        target = os.path.join(work, "bins" if case["mode"] == "b" else case["mode"][2:])
        kw = {"options": opts, "aligner": OracleAligner(oracle, opts.scoring_scheme)}
        try:
            res = runner.run(inp, barcode_dir=target, **kw) if case["mode"] == "b" else runner.run(inp, output=target, **kw)
        except runner.UsageError as e:       # the reference ends such runs with sys.exit(message): same message on every rank
            assert case["exit"] == str(e), (name, case["exit"], str(e))
            seen_sharded.append(True)
            res = None
        dist.barrier()
        out[name] = readgen.output_md5s(target) if (rank == 0 and os.path.exists(target)) else {}
        shares[name] = (len(res.start_trim), res.n_reads, dict(res.files)) if res is not None else None
    q.put((rank, out, shares, seen_sharded))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_runs_write_the_reference_files(tmp_path, world):
    from tests.runner_cases import load_cases
    cases = load_cases()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, out, shares, seen = q.get(timeout=900)
        got[rank] = (out, shares, seen)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name in CASES[world]:
        assert got[0][0][name] == cases[name]["outputs"], (name, got[0][0][name])
        if cases[name]["exit"] is not None:
            continue
        total = got[0][1][name][1]
        mine = [got[r][1][name][0] for r in range(world)]
        assert sum(mine) == total and max(mine) < total, (name, mine, total)          # no rank held every read
        assert all(got[r][1][name][2] == got[0][1][name][2] for r in range(world))   # every rank reports the same file statistics
    assert all(all(got[r][2]) and len(got[r][2]) == len(CASES[world]) for r in range(world))  # every run took the sharded route


GZ_CASES = ["native_default", "native_bins", "edge_default", "native_gz_out"]


def _gz_worker(rank, world, port, workdir, q):
    """The same runs over a gzip input of SIZED members (this package's own output layout; bgzip): the ranks cut the
    INFLATED bytes and each inflates only its members (runner.run_sharded -> pc_gz_sized_find_record / pc_readset_load_gz_range)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from porechop_amd import runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import load_cases, options_from_argv
    oracle = Oracle()
    cases = load_cases()
    seen_sharded, out, shares = [], {}, {}
    orig = runner.run_sharded

    def spy(*a, **kw):
        r = orig(*a, **kw)
        seen_sharded.append(r is not None)
        return r
    runner.run_sharded = spy
    for name in GZ_CASES:
        case = cases[name]
        opts = options_from_argv(case["argv"])
        inp = os.path.join(workdir, "gz_inputs", case["dataset"] + ".fastq.gz")
        work = os.path.join(workdir, "gzrun_" + name)
        if rank == 0:
            os.makedirs(work)
        dist.barrier()
        target = os.path.join(work, "bins" if case["mode"] == "b" else case["mode"][2:])
        kw = {"options": opts, "aligner": OracleAligner(oracle, opts.scoring_scheme)}
        res = runner.run(inp, barcode_dir=target, **kw) if case["mode"] == "b" else runner.run(inp, output=target, **kw)
        dist.barrier()
        out[name] = readgen.output_md5s(target) if (rank == 0 and os.path.exists(target)) else {}
        shares[name] = (len(res.start_trim), res.n_reads)
    q.put((rank, out, shares, seen_sharded))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_runs_over_a_gzip_input_of_sized_members(tmp_path, world):
    from oracle.oracle import Oracle
    from porechop_amd import io as pio, runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import load_cases, options_from_argv
    cases = load_cases()
    oracle = Oracle()
    os.makedirs(tmp_path / "gz_inputs")
    want = {}
    for name in GZ_CASES:                      # the single-process run over the same .gz file is the expectation
        case = cases[name]
        plain = readgen.build_dataset(case["dataset"], str(tmp_path / "datasets"))
        if os.path.isdir(plain) or not open(plain, "rb").read(1) == b"@":
            pytest.skip("dataset %s is not one FASTQ file" % case["dataset"])
        inp = str(tmp_path / "gz_inputs" / (case["dataset"] + ".fastq.gz"))
        if not os.path.exists(inp):
            pio.gzip_file(plain, inp)
        opts = options_from_argv(case["argv"])
        target = str(tmp_path / ("single_" + name) / ("bins" if case["mode"] == "b" else case["mode"][2:]))
        os.makedirs(os.path.dirname(target))
        kw = {"options": opts, "aligner": OracleAligner(oracle, opts.scoring_scheme)}
        runner.run(inp, barcode_dir=target, **kw) if case["mode"] == "b" else runner.run(inp, output=target, **kw)
        want[name] = readgen.output_md5s(target)
        if case["mode"] != "b":
            assert want[name] == case["outputs"], name          # ... and, to one file, the reference CLI's on the plain input
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gz_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, out, shares, seen = q.get(timeout=900)
        got[rank] = (out, shares, seen)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name in GZ_CASES:
        assert got[0][0][name] == want[name] and want[name], (name, got[0][0][name], want[name])
        total = got[0][1][name][1]
        mine = [got[r][1][name][0] for r in range(world)]
        assert sum(mine) == total and max(mine) < total, (name, mine, total)          # no rank held every read
    assert all(all(got[r][2]) and len(got[r][2]) == len(GZ_CASES) for r in range(world))      # every run took the sharded route


# ---- FASTA input over the ranks (cut where a line begins with '>': pc_fastq_find_record / pc_readset_load_segment) ----------
FASTA_RUNS = [("o:out.fasta", []), ("b", []), ("o:out.fastq", ["--format", "fastq"]), ("o:out.fasta.gz", ["--discard_middle"])]


def _fasta_input(path):
    """Multi-line FASTA with a few odd records (an empty name, whose bases go to the next record; blank and padded lines)."""
    import random
    from tests import readgen
    rng = random.Random(11)
    reads = readgen.native_reads(77, 60, barcodes=(2, 5, 9))
    with open(path, "w") as f:
        for k, (name, seq, _) in enumerate(reads):
            f.write(">" + ("" if k in (7, 31) else name) + "\n")
            width = rng.choice([60, 80, 10 ** 9])
            for i in range(0, len(seq), width):
                f.write(("  " if rng.random() < 0.02 else "") + seq[i:i + width] + "\n")
            if rng.random() < 0.1:
                f.write("\n")


def _fasta_worker(rank, world, port, workdir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from porechop_amd import runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import options_from_argv
    oracle = Oracle()
    seen, out, shares = [], {}, {}
    orig = runner.run_sharded

    def spy(*a, **kw):
        r = orig(*a, **kw)
        seen.append(r is not None)
        return r
    runner.run_sharded = spy
    inp = os.path.join(workdir, "in.fasta")
    for k, (mode, argv) in enumerate(FASTA_RUNS):
        opts = options_from_argv(argv)
        work = os.path.join(workdir, "fa_run%d" % k)
        if rank == 0:
            os.makedirs(work)
        dist.barrier()
        target = os.path.join(work, "bins" if mode == "b" else mode[2:])
        kw = {"options": opts, "aligner": OracleAligner(oracle, opts.scoring_scheme)}
        res = runner.run(inp, barcode_dir=target, **kw) if mode == "b" else runner.run(inp, output=target, **kw)
        dist.barrier()
        out[k] = readgen.output_md5s(target) if rank == 0 else {}
        shares[k] = (len(res.start_trim), res.n_reads, res.read_type)
    q.put((rank, out, shares, seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fasta_input_equals_the_single_process_run(tmp_path, world):
    from oracle.oracle import Oracle
    from porechop_amd import runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import options_from_argv
    inp = str(tmp_path / "in.fasta")
    _fasta_input(inp)
    oracle = Oracle()
    want = {}
    for k, (mode, argv) in enumerate(FASTA_RUNS):          # the single-process files (the whole-file loader)
        opts = options_from_argv(argv)
        target = str(tmp_path / ("single%d" % k) / ("bins" if mode == "b" else mode[2:]))
        os.makedirs(os.path.dirname(target))
        kw = {"options": opts, "aligner": OracleAligner(oracle, opts.scoring_scheme)}
        res = runner.run(inp, barcode_dir=target, **kw) if mode == "b" else runner.run(inp, output=target, **kw)
        assert res.read_type == "FASTA" and res.n_reads == 58        # (two headers without a name: their bases join the next record)
        want[k] = readgen.output_md5s(target)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fasta_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, out, shares, seen = q.get(timeout=900)
        got[rank] = (out, shares, seen)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in range(len(FASTA_RUNS)):
        assert got[0][0][k] == want[k], (k, got[0][0][k], want[k])
        mine = [got[r][1][k][0] for r in range(world)]
        assert sum(mine) == 58 and max(mine) < 58, mine                    # no rank held every read
        assert all(got[r][1][k][1:] == (58, "FASTA") for r in range(world))
    assert all(all(got[r][2]) and len(got[r][2]) == len(FASTA_RUNS) for r in range(world))   # every run took the sharded route


def test_streamed_fasta_input_equals_the_whole_file_run(tmp_path, monkeypatch):
    """A plain FASTA file as a stream of small blocks (runner.run_streamed): the same files as the whole-file run."""
    from oracle.oracle import Oracle
    from porechop_amd import runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import options_from_argv
    import gzip
    from porechop_amd import io as pio
    plain = str(tmp_path / "in.fasta")
    _fasta_input(plain)
    one = str(tmp_path / "one.fasta.gz")
    with open(one, "wb") as f:
        f.write(gzip.compress(open(plain, "rb").read(), 6))
    sized = str(tmp_path / "sized.fasta.gz")
    pio.gzip_file(plain, sized)
    oracle = Oracle()
    took = []
    orig = runner.run_streamed

    def spy(*a, **kw):
        r = orig(*a, **kw)
        took.append(r is not None)
        return r
    monkeypatch.setattr(runner, "run_streamed", spy)
    for k, (mode, argv) in enumerate(FASTA_RUNS):
        opts = options_from_argv(argv + ["--check_reads", "20"])
        files = {}
        # (the gzip forms of the same file -- one member, sized members -- stream too: pc_gzstream_next cuts FASTA like the plain file)
        for form, inp in (("plain", plain), ("one", one), ("sized", sized)):
            for blocks in ((None, "5000", "40000") if form == "plain" else ("5000",)):
                if blocks:
                    monkeypatch.setenv("PC_STREAM_BLOCK_BYTES", blocks)
                else:
                    monkeypatch.delenv("PC_STREAM_BLOCK_BYTES", raising=False)
                target = str(tmp_path / ("s%d_%s_%s" % (k, form, blocks)) / ("bins" if mode == "b" else mode[2:]))
                os.makedirs(os.path.dirname(target))
                kw = {"options": opts, "aligner": OracleAligner(oracle, opts.scoring_scheme)}
                res = runner.run(inp, barcode_dir=target, **kw) if mode == "b" else runner.run(inp, output=target, **kw)
                assert res.read_type == "FASTA" and res.n_reads == 58
                # (bins of a gzip-ed input are written gzip-ed -- porechop.py:640-651: same contents under a `.gz` name)
                files[(form, blocks)] = {(n[:-3] if n.endswith(".gz") else n): h for n, h in readgen.output_md5s(target).items()}
        want = {(n[:-3] if n.endswith(".gz") else n): h for n, h in files[("plain", None)].items()}
        assert all(v == want for v in files.values()), (k, files)
    assert took.count(True) >= 4 * len(FASTA_RUNS)                    # the small-block runs really streamed, the gzip ones among them


# ---- an ordinary multi-member gzip input (`cat *.fastq.gz`, no sizes) over the ranks: cut at member starts ------------------
CATGZ_RUNS = [("o:out.fastq", []), ("b", []), ("o:out.fastq.gz", ["--no_split"])]


def _catgz_input(path):
    """120 reads as 17 gzip members of whole records, zero padding between some of them (what concatenated files look like)."""
    import gzip
    import random
    from tests import readgen
    rng = random.Random(5)
    reads = readgen.native_reads(91, 120, barcodes=(3, 7, 11))
    cuts = sorted(rng.sample(range(1, 120), 16))
    blob = b""
    for a, b in zip([0] + cuts, cuts + [120]):
        blob += gzip.compress(readgen.fastq_text(reads[a:b]).encode(), rng.choice([1, 6, 9])) + b"\0" * rng.choice([0, 0, 19, 512])
    with open(path, "wb") as f:
        f.write(blob)


def _catgz_worker(rank, world, port, workdir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from porechop_amd import runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import options_from_argv
    oracle = Oracle()
    seen, out, shares = [], {}, {}
    orig = runner.run_sharded

    def spy(*a, **kw):
        r = orig(*a, **kw)
        seen.append(r is not None)
        return r
    runner.run_sharded = spy
    inp = os.path.join(workdir, "cat.fastq.gz")
    for k, (mode, argv) in enumerate(CATGZ_RUNS):
        opts = options_from_argv(argv)
        work = os.path.join(workdir, "cg_run%d" % k)
        if rank == 0:
            os.makedirs(work)
        dist.barrier()
        target = os.path.join(work, "bins" if mode == "b" else mode[2:])
        kw = {"options": opts, "aligner": OracleAligner(oracle, opts.scoring_scheme)}
        res = runner.run(inp, barcode_dir=target, **kw) if mode == "b" else runner.run(inp, output=target, **kw)
        dist.barrier()
        out[k] = readgen.output_md5s(target) if rank == 0 else {}
        shares[k] = (len(res.start_trim), res.n_reads)
    q.put((rank, out, shares, seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_concatenated_gzip_input_equals_the_single_process_run(tmp_path, world):
    from oracle.oracle import Oracle
    from porechop_amd import runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import options_from_argv
    inp = str(tmp_path / "cat.fastq.gz")
    _catgz_input(inp)
    oracle = Oracle()
    want = {}
    for k, (mode, argv) in enumerate(CATGZ_RUNS):
        opts = options_from_argv(argv)
        target = str(tmp_path / ("single%d" % k) / ("bins" if mode == "b" else mode[2:]))
        os.makedirs(os.path.dirname(target))
        kw = {"options": opts, "aligner": OracleAligner(oracle, opts.scoring_scheme)}
        res = runner.run(inp, barcode_dir=target, **kw) if mode == "b" else runner.run(inp, output=target, **kw)
        assert res.n_reads == 120
        want[k] = readgen.output_md5s(target)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_catgz_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, out, shares, seen = q.get(timeout=900)
        got[rank] = (out, shares, seen)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in range(len(CATGZ_RUNS)):
        assert got[0][0][k] == want[k], (k, got[0][0][k], want[k])
        mine = [got[r][1][k][0] for r in range(world)]
        assert sum(mine) == 120 and max(mine) < 120, mine                   # no rank inflated every member
    assert all(all(got[r][2]) and len(got[r][2]) == len(CATGZ_RUNS) for r in range(world))   # every run took the sharded route
