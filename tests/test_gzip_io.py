"""gzip in and out at the speed of the rest (pc_gz.h / pc_io.cpp; replaces Python's gzip module on the way in,
porechop/misc.py:60-81,151-168, and `pigz -p <threads>` on the way out, porechop/porechop.py:640-651,685-729).  No GPU.

 * the compressor's two layouts -- sized members (BGZF) and ONE pigz-style member -- inflate to the input with Python's
   gzip module and with the `gzip` program, with libdeflate and with zlib behind them;
 * the whole-file loader reads all of them (sized members in parallel) into the same read set as the plain file;
 * the streamed reader hands over exactly the blocks pc_readset_load_segment cuts the plain file into, whatever the
   layout, keeps the check reads in its first block, and refuses what the whole-file loader must handle;
 * runner.run() .fastq.gz -> .fastq.gz through the streamed route equals the plain route's output, byte for byte after
   gunzip, and a sharded write with PC_IO_MMAP=1 no longer truncates another rank's span."""
import gzip
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from porechop_amd import io as pio

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_fastq(path, n, seed=1, max_len=3000):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for i in range(n):
            L = int(rng.integers(1, max_len))
            seq = "".join(np.array(list("ACGTN"))[rng.choice(5, L, p=[.245, .245, .245, .245, .02])])
            q = "".join(chr(33 + int(x)) for x in rng.integers(0, 40, L))
            f.write("@r%d desc=%d\n%s\n+\n%s\n" % (i, L, seq, q))
    return path


def md5(b):
    return hashlib.md5(b).hexdigest()


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("gz")
    plain = write_fastq(str(d / "reads.fastq"), 6000)
    out = {"plain": plain, "dir": str(d)}
    pio.gzip_file(plain, str(d / "sized.fastq.gz"))
    pio.gzip_file(plain, str(d / "single.fastq.gz"), single_member=True)
    subprocess.run("gzip -1 -c %s > %s" % (plain, d / "cli.fastq.gz"), shell=True, check=True)
    # two members the ordinary way (cat a.gz b.gz), no size subfields
    data = open(plain, "rb").read()
    cut = data.index(b"\n@r3000 ") + 1
    with open(d / "concat.fastq.gz", "wb") as f:
        f.write(gzip.compress(data[:cut], 1))
        f.write(gzip.compress(data[cut:], 1))
    # forty members the ordinary way (what `cat run/*.fastq.gz` makes of a run's small files): no size subfields, found by
    # guessing (MemberSpeculator) and inflated ahead by worker threads; a few members are empty, one is followed by zero padding
    marks = [0] + [data.index(b"\n@r%d " % (k * 150)) + 1 for k in range(1, 40)] + [len(data)]
    with open(d / "cat40.fastq.gz", "wb") as f:
        for k in range(40):
            f.write(gzip.compress(data[marks[k]:marks[k + 1]], 1 + k % 6))
            if k % 13 == 5:
                f.write(gzip.compress(b""))
        f.write(b"\0" * 37)
    out.update(sized=str(d / "sized.fastq.gz"), single=str(d / "single.fastq.gz"), cli=str(d / "cli.fastq.gz"), concat=str(d / "concat.fastq.gz"),
               cat40=str(d / "cat40.fastq.gz"))
    return out


def test_compressor_layouts_inflate_to_the_input(files):
    want = md5(open(files["plain"], "rb").read())
    for k in ("sized", "single"):
        assert md5(gzip.open(files[k], "rb").read()) == want, k
        assert md5(subprocess.run(["gzip", "-dc", files[k]], capture_output=True, check=True).stdout) == want, k
    raw = open(files["sized"], "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC"
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))       # the BGZF end marker
    # every level, and zlib behind the same interface (a fresh process: the choice is made once)
    for lvl in (1, 9):
        dst = os.path.join(files["dir"], "lvl%d.gz" % lvl)
        pio.gzip_file(files["plain"], dst, level=lvl)
        assert md5(gzip.open(dst, "rb").read()) == want
    code = ("import sys, gzip, hashlib; sys.path.insert(0, %r); from porechop_amd import io as pio; "
            "pio.gzip_file(%r, %r); pio.gzip_file(%r, %r, single_member=True); "
            "print(hashlib.md5(gzip.open(%r, 'rb').read()).hexdigest(), hashlib.md5(gzip.open(%r, 'rb').read()).hexdigest(), pio.ReadSet(%r).count)"
            % (REPO, files["plain"], files["dir"] + "/z1.gz", files["plain"], files["dir"] + "/z2.gz", files["dir"] + "/z1.gz",
               files["dir"] + "/z2.gz", files["dir"] + "/z1.gz"))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PC_NO_LIBDEFLATE="1"))
    assert res.stdout.split() == [want, want, "6000"], res.stdout + res.stderr
    # empty input
    empty = os.path.join(files["dir"], "empty")
    open(empty, "wb").close()
    for single in (False, True):
        pio.gzip_file(empty, empty + ".gz", single_member=single)
        assert gzip.open(empty + ".gz", "rb").read() == b""


def test_whole_file_loader_reads_every_layout(files):
    ref = pio.ReadSet(files["plain"])
    want = (ref.count, md5(ref.arena.tobytes()), md5(ref.lengths.tobytes()), ref.name(17), ref.quals(5999))
    for k in ("sized", "single", "cli", "concat", "cat40"):
        rs = pio.ReadSet(files[k])
        assert (rs.count, md5(rs.arena.tobytes()), md5(rs.lengths.tobytes()), rs.name(17), rs.quals(5999)) == want, k
        rs.close()
    # a damaged sized member falls back to the stream reader, which reports the damage like any gzip error
    raw = bytearray(open(files["sized"], "rb").read())
    raw[len(raw) // 2] ^= 0xFF
    bad = os.path.join(files["dir"], "bad.fastq.gz")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        pio.ReadSet(bad)


def test_stream_blocks_equal_the_plain_files_segments(files):
    size = os.path.getsize(files["plain"])
    for block in (300_000, 2_000_000, 1 << 30):
        want, pos = [], 0
        while pos < size:
            rs, nxt = pio.ReadSet.segment(files["plain"], pos, block)
            want.append((rs.count, md5(rs.arena.tobytes()), rs.name(0), rs.quals(rs.count - 1)))
            rs.close()
            pos = nxt
        for k in ("sized", "single", "cli", "concat", "cat40"):
            st = pio.GzStream(files[k])
            got = []
            while True:
                rs = st.next(block)
                assert rs is not False
                if rs is None:
                    break
                got.append((rs.count, md5(rs.arena.tobytes()), rs.name(0), rs.quals(rs.count - 1)))
                rs.close()
            st.close()
            assert got == want, (k, block, len(got), len(want))
    # the first block holds the check reads however small the target
    st = pio.GzStream(files["sized"])
    rs = st.next(10_000, min_reads=2500)
    assert rs.count >= 2500
    st.close()
    # closing a stream whose producer is still ahead does not hang
    st = pio.GzStream(files["cli"])
    st.next(1000).close()
    st.close()


def test_stream_refuses_what_the_whole_file_loader_must_handle(files):
    d = files["dir"]
    fasta = os.path.join(d, "x.fasta")
    open(fasta, "w").write(">a\nACGT\n>b\nGGCC\n")
    pio.gzip_file(fasta, fasta + ".gz")
    st = pio.GzStream(fasta + ".gz")             # (round 6: gzip-ed FASTA streams too, cut where a line begins with '>')
    rs = st.next(1000)
    assert rs is not False and rs is not None and not rs.is_fastq and rs.count == 2 and rs.seq(1) == "GGCC"
    assert st.next(1000) is None
    st.close()
    assert pio.ReadSet(fasta + ".gz").count == 2
    other = os.path.join(d, "x.txt")
    open(other, "w").write("neither FASTA nor FASTQ\n" * 10)
    pio.gzip_file(other, other + ".gz")
    st = pio.GzStream(other + ".gz")
    assert st.next(1000) is False
    st.close()
    irregular = os.path.join(d, "irr.fastq")
    open(irregular, "w").write("@a\nACGT\n+\nIIII\n\n@b\nAC\n+\nII\n" * 50)
    pio.gzip_file(irregular, irregular + ".gz", single_member=True)
    st = pio.GzStream(irregular + ".gz")
    assert st.next(64) is False
    st.close()
    with pytest.raises(ValueError):
        pio.GzStream(files["plain"])          # not gzip
    truncated = os.path.join(d, "trunc.fastq.gz")
    open(truncated, "wb").write(open(files["cli"], "rb").read()[:-4000])
    st = pio.GzStream(truncated)
    seen = []
    while True:
        rs = st.next(1 << 30)
        seen.append(rs)
        if rs is None or rs is False:
            break
    assert seen[-1] is False
    st.close()


def test_runner_gz_to_gz_streamed_equals_plain_route(oracle, tmp_path, monkeypatch):
    from porechop_amd import runner
    from tests import readgen
    from tests.cpu_aligner import OracleAligner
    inp = readgen.build_dataset("native", str(tmp_path / "datasets"))
    for layout, single in (("sized", False), ("single", True)):
        pio.gzip_file(inp, str(tmp_path / (layout + ".fastq.gz")), single_member=single)
    mk = lambda: OracleAligner(oracle, (3, -6, -5, -2))
    runner.run(inp, output=str(tmp_path / "plain.fastq"), aligner=mk())
    want = md5(open(tmp_path / "plain.fastq", "rb").read())
    runner.run(inp, barcode_dir=str(tmp_path / "bins_plain"), aligner=mk())
    want_bins = {f: md5(open(tmp_path / "bins_plain" / f, "rb").read()) for f in os.listdir(tmp_path / "bins_plain")}
    monkeypatch.setenv("PC_STREAM_BLOCK_BYTES", "5000")
    streamed = []
    real = runner.run_streamed
    monkeypatch.setattr(runner, "run_streamed", lambda *a, **k: (streamed.append(1), real(*a, **k))[1])
    for layout in ("sized", "single"):
        out = tmp_path / (layout + "_out.fastq.gz")
        res = runner.run(str(tmp_path / (layout + ".fastq.gz")), output=str(out), aligner=mk())
        assert md5(gzip.open(out, "rb").read()) == want, layout
        assert open(out, "rb").read()[12:14] == b"BC" and str(out) in res.files
        bins = tmp_path / ("bins_" + layout)
        runner.run(str(tmp_path / (layout + ".fastq.gz")), barcode_dir=str(bins), aligner=mk())
        # (gz input and -b: the reference writes gz bins, porechop.py:627-631)
        got = {f[:-3]: md5(gzip.open(bins / f, "rb").read()) for f in os.listdir(bins)}
        assert all(f.endswith(".gz") for f in os.listdir(bins)) and got == want_bins, layout
    assert len(streamed) == 4
    # plain in, gz out, whole-file route; and nothing to write still leaves a valid (empty) gzip file
    monkeypatch.setenv("PC_STREAM_BLOCK_BYTES", str(1 << 30))
    runner.run(inp, output=str(tmp_path / "whole.fastq.gz"), aligner=mk())
    assert md5(gzip.open(tmp_path / "whole.fastq.gz", "rb").read()) == want
    empty = tmp_path / "none.fastq"
    empty.write_text("@a\nACGT\n+\nIIII\n")
    opts = runner.Options(min_split_read_size=1000, discard_middle=False)
    runner.run(str(empty), output=str(tmp_path / "e.fastq.gz"), options=opts, aligner=mk())
    assert gzip.open(tmp_path / "e.fastq.gz", "rb").read() in (b"", b"@a\nACGT\n+\nIIII\n")


def test_shared_write_with_mmap_keeps_other_spans(tmp_path, monkeypatch):
    """with PC_IO_MMAP=1 the shared writer's ftruncate cut a higher rank's span off."""
    plain = write_fastq(str(tmp_path / "r.fastq"), 4000, seed=3, max_len=6000)
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from porechop_amd import io as pio
rs = pio.ReadSet(%r)
n = rs.count
pr = np.arange(n, dtype=np.int64); ps = np.zeros(n, dtype=np.int32); pn = rs.lengths.copy(); num = np.zeros(n, dtype=np.int32)
half = n // 2
sizes_a = rs.write_sizes(pr[:half], ps[:half], pn[:half], num[:half], np.zeros(half, dtype=np.int32), 1, True)
open(%r, "wb").close()
# the SECOND half first, at its offset; then the first half at 0
rs.write_shared(pr[half:], ps[half:], pn[half:], num[half:], np.zeros(n - half, dtype=np.int32), [%r], True, np.array([sizes_a[0]], dtype=np.int64))
rs.write_shared(pr[:half], ps[:half], pn[:half], num[:half], np.zeros(half, dtype=np.int32), [%r], True, np.zeros(1, dtype=np.int64))
''' % (REPO, plain, str(tmp_path / "o.fastq"), str(tmp_path / "o.fastq"), str(tmp_path / "o.fastq"))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PC_IO_MMAP="1"))
    assert res.returncode == 0, res.stderr
    assert md5(open(tmp_path / "o.fastq", "rb").read()) == md5(open(plain, "rb").read())


def test_many_gz_files_are_read_side_by_side_and_parsed_in_order(files, tmp_path):
    """A Guppy / Albacore style directory (porechop.py:216-268): many .fastq.gz files -- inflated one thread each, parsed in
    file order; a broken file in the middle reports what it would report alone, and the files before it do not matter."""
    data = open(files["plain"], "rb").read()
    marks = [0] + [data.index(b"\n@r%d " % (k * 200)) + 1 for k in range(1, 30)] + [len(data)]
    paths = []
    for k in range(30):
        p = tmp_path / ("part%02d.fastq%s" % (k, ".gz" if k % 3 else ""))
        blob = data[marks[k]:marks[k + 1]]
        p.write_bytes(gzip.compress(blob, 1) if k % 3 else blob)
        paths.append(str(p))
    ref = pio.ReadSet(files["plain"])
    rs = pio.ReadSet(paths)
    assert rs.count == ref.count and md5(rs.arena.tobytes()) == md5(ref.arena.tobytes())
    assert rs.name(4321) == ref.name(4321) and rs.quals(5999) == ref.quals(5999)
    assert list(np.unique(rs.file_index)) == list(range(30)) and int(rs.file_index[-1]) == 29
    rs.close()
    bad = tmp_path / "part13.fastq.gz"
    bad.write_bytes(b"\x1f\x8b\x08" + b"garbage" * 10)
    with pytest.raises(ValueError) as one:
        pio.ReadSet(str(bad))
    with pytest.raises(ValueError) as many:
        pio.ReadSet(paths)
    assert str(one.value) == str(many.value)


def test_guessed_members_larger_than_the_cap_take_the_serial_path(files):
    """PC_GZ_SPEC_CAP_MB=1 (a fresh process: read once): the forty members of cat40 inflate to about 0.45 MB each -- with a cap
    of 1 MB all are taken from the workers; the two members of concat (9 MB each) exceed it and stream through one core's
    zlib.  Both ways the blocks are the plain file's; PC_GZ_NO_SPECULATION=1 gives the same again."""
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from porechop_amd import io as pio\n"
            "for path in sys.argv[1:]:\n"
            "    st = pio.GzStream(path); h = hashlib.md5(); n = 0\n"
            "    while True:\n"
            "        rs = st.next(700000)\n"
            "        assert rs is not False\n"
            "        if rs is None: break\n"
            "        h.update(rs.arena.tobytes()[:-64]); n += rs.count; rs.close()\n"
            "    st.close(); print(n, h.hexdigest())\n" % REPO)
    outs = []
    for env in ({}, {"PC_GZ_SPEC_CAP_MB": "1"}, {"PC_GZ_NO_SPECULATION": "1"}, {"PC_IO_THREADS": "2"}):
        r = subprocess.run([sys.executable, "-c", code, files["cat40"], files["concat"], files["single"]], capture_output=True, text=True,
                           env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split())
    assert all(o == outs[0] for o in outs), outs
    ref = pio.ReadSet(files["plain"])
    assert outs[0][0] == str(ref.count) and outs[0][1] == md5(ref.arena.tobytes()[:-64])
    assert outs[0][:2] == outs[0][2:4] == outs[0][4:6]


def test_whole_file_route_skips_zero_padding_and_refuses_a_truncated_stream(tmp_path):
    """porechop/misc.py:60-81 reads .gz through Python's gzip module: zero padding between (and after) members is skipped,
    a stream that ends inside a member raises.  zlib's gzread -- the whole-file route before tests/host/fuzz_io.cpp -- stopped
    at the padding without a word and returned a truncated stream's bytes as if they were the file."""
    plain = write_fastq(str(tmp_path / "in.fastq"), 900, seed=3)
    text = open(plain, "rb").read()
    cut = text.index(b"\n@r450 ") + 1
    padded = str(tmp_path / "padded.fastq.gz")
    open(padded, "wb").write(gzip.compress(text[:cut]) + b"\0" * 512 + gzip.compress(text[cut:]) + b"\0" * 100)
    with gzip.open(padded, "rb") as f:                        # what the reference's reader makes of it
        assert f.read() == text
    ref = pio.ReadSet(plain)
    want = (ref.count, md5(ref.arena.tobytes()), md5(ref.lengths.tobytes()), ref.name(449), ref.quals(899))
    rs = pio.ReadSet(padded)
    assert ref.count == 900 and (rs.count, md5(rs.arena.tobytes()), md5(rs.lengths.tobytes()), rs.name(449), rs.quals(899)) == want
    st, n = pio.GzStream(padded), 0
    while True:
        blk = st.next(1 << 16)
        assert blk is not False
        if blk is None:
            break
        n += blk.count
    st.close()
    assert n == 900

    whole = gzip.compress(text)
    short = str(tmp_path / "short.fastq.gz")
    open(short, "wb").write(whole[:len(whole) * 2 // 3])
    with pytest.raises(EOFError):
        with gzip.open(short, "rb") as f:
            f.read()
    with pytest.raises(ValueError) as e:
        pio.ReadSet(short)
    assert "gzip stream error" in str(e.value)
    junk = str(tmp_path / "junk.fastq.gz")
    open(junk, "wb").write(whole + b"not a gzip member")
    with pytest.raises(gzip.BadGzipFile):
        with gzip.open(junk, "rb") as f:
            f.read()
    with pytest.raises(ValueError):
        pio.ReadSet(junk)


def test_one_big_member_through_the_watched_one_shot_route(files):
    """pc_io.cpp oneshot_member: ONE libdeflate call per big member on its own thread, its output handed over while it appears.
    Forced on for every member here (PC_GZ_ONESHOT_MIN_MB=0; the default takes members of 32 MB and more), with room and with
    a room the member outgrows after a hand-over (zlib restarts and discards it): the blocks equal the plain file's."""
    code = r"""
import hashlib, sys
sys.path.insert(0, %r)
from porechop_amd import io as pio
for path in sys.argv[1:]:
    st, h, n = pio.GzStream(path), hashlib.md5(), 0
    while True:
        rs = st.next(3_000_000)
        assert rs is not False
        if rs is None:
            break
        h.update(rs.arena.tobytes()[:int(rs.lengths.sum())]); h.update(rs.name(rs.count - 1).encode()); h.update(rs.quals(0).encode()); n += rs.count
        rs.close()
    st.close()
    print("GOT", n, h.hexdigest())
""" % REPO
    ref = pio.ReadSet(files["plain"])
    size = os.path.getsize(files["plain"])
    want_n = ref.count
    outs = {}
    for tag, extra in (("zlib", {"PC_GZ_NO_ONESHOT": "1"}), ("oneshot", {"PC_GZ_ONESHOT_MIN_MB": "0", "PC_GZ_VERBOSE": "1"}),
                       ("outgrown", {"PC_GZ_ONESHOT_MIN_MB": "0", "PC_GZ_VERBOSE": "1", "PC_GZ_ONESHOT_ROOM_KB": str(size // 1024 * 3 // 4)})):
        res = subprocess.run([sys.executable, "-c", code, files["single"], files["cli"], files["concat"]], capture_output=True, text=True,
                             env=dict(os.environ, PC_GZ_NO_SPECULATION="1", **extra), timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs[tag] = [l for l in res.stdout.splitlines() if l.startswith("GOT")]
        assert len(outs[tag]) == 3 and all(int(l.split()[1]) == want_n for l in outs[tag]), (tag, outs[tag])
        if tag == "oneshot":
            assert res.stderr.count("result 1") >= 4, res.stderr[-1500:]          # single, cli, and the two members of concat
        if tag == "outgrown":
            assert "result 2" in res.stderr, res.stderr[-1500:]
    assert outs["zlib"] == outs["oneshot"] == outs["outgrown"]


def test_members_dense_with_the_gzip_magic_do_not_hang_the_reader(tmp_path):
    """40 `cat`-ed members, half of them STORED (their bytes appear verbatim), whose quality lines hold
    `1f 8b 08 00` every nine bytes.  The reader guesses member starts from those bytes; the guesses it has already passed used
    to fill its window, the scan never reached the member the consumer asked for, and `ReadSet(path)` hung with every thread
    asleep.  Must load every read (in a subprocess with a timeout: a regression is a hang, not an exception)."""
    import gzip
    import io
    import random
    import subprocess
    import sys
    rng = random.Random(7)
    recs = []
    n = 12000
    for i in range(n):
        L = rng.randint(50, 200)
        q = bytearray(rng.choice(b"!#5:I") for _ in range(L))
        for k in range(0, L - 4, 9):
            q[k:k + 4] = b"\x1f\x8b\x08\x00"
        recs.append(b"@r%d\n" % i + "".join(rng.choice("ACGT") for _ in range(L)).encode() + b"\n+\n" + bytes(q) + b"\n")
    parts = []
    for m in range(40):
        b = io.BytesIO()
        with gzip.GzipFile(fileobj=b, mode="wb", compresslevel=0 if m % 2 == 0 else 6, mtime=0) as f:
            f.write(b"".join(recs[m * (n // 40):(m + 1) * (n // 40)]))
        parts.append(b.getvalue())
    path = tmp_path / "dense.fastq.gz"
    path.write_bytes(b"".join(parts))
    code = ("import sys; sys.path.insert(0, %r); from porechop_amd.io import ReadSet; rs = ReadSet(%r); "
            "print(rs.count, rs.seq(0)[:5], rs.name(rs.count - 1))" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(path)))
    for env in ({}, {"PC_GZ_NO_SPECULATION": "1"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.split()[0] == str(n) and r.stdout.split()[2] == "r%d" % (n - 1), r.stdout
