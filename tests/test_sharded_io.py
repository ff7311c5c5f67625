"""The host pieces of a sharded run (include/porechop_amd.h, last block): pc_fastq_find_record cuts a plain FASTQ file at
the record boundaries pc_readset_load_segment would choose (quality lines that start with '@' included), the segments of any
number of ranks concatenate to the whole file's reads, pc_readset_write_sizes predicts exactly what pc_readset_write puts
into each file, and spans written by pc_readset_write_shared at the exchanged positions rebuild the single writer's files."""
import os
import random

import numpy as np

from porechop_amd.io import ReadSet, fastq_record_start


def _fastq(path, n, rng):
    with open(path, "w") as f:
        for i in range(n):
            L = rng.choice([1, 30, 200, 1500])
            seq = "".join(rng.choice("ACGT") for _ in range(L))
            qual = "".join(rng.choice("@+5I") for _ in range(L))          # '@' and '+' at line starts on purpose
            f.write("@read%d some description\n%s\n+\n%s\n" % (i, seq, qual))


def test_segments_of_any_rank_count_concatenate_to_the_file(tmp_path):
    rng = random.Random(3)
    p = str(tmp_path / "reads.fastq")
    _fastq(p, 700, rng)
    size = os.path.getsize(p)
    whole = ReadSet(p)
    names = [whole.name(i) for i in range(whole.count)]
    for world in (1, 2, 3, 7, 64):
        cuts = [0] + [fastq_record_start(p, size * r // world) for r in range(1, world)] + [size]
        assert all(c is not None for c in cuts) and cuts == sorted(cuts)
        got = []
        for r in range(world):
            if cuts[r + 1] > cuts[r]:
                rs, nxt = ReadSet.segment(p, cuts[r], cuts[r + 1] - cuts[r])
                assert rs is not None and nxt == cuts[r + 1]
                got += [(rs.name(i), rs.seq(i), rs.quals(i)) for i in range(rs.count)]
                rs.close()
        assert [g[0] for g in got] == names
        assert got == [(whole.name(i), whole.seq(i), whole.quals(i)) for i in range(whole.count)]
    assert fastq_record_start(p, size + 5) == size
    assert fastq_record_start(str(tmp_path / "reads.fastq"), 1) == len("@read0 some description\n") + 2 * (whole.lengths[0] + 1) + 2
    whole.close()
    # a plain FASTA file is cut where a line begins with '>' (round 6); anything else is not streamable: the caller falls back
    fa = str(tmp_path / "reads.fasta")
    with open(fa, "w") as f:
        f.write(">a\nACGT\n>b\nAC\nGT\n>c\nT\n")
    assert fastq_record_start(fa, 0) == 0 and fastq_record_start(fa, 3) == 8 and fastq_record_start(fa, 9) == 17
    assert fastq_record_start(fa, 18) == 22 and fastq_record_start(fa, 99) == 22          # no header left: the file's size
    got = []
    pos = 0
    while pos < 22:
        rs, nxt = ReadSet.segment(fa, pos, 5)
        assert rs is not None and not rs.is_fastq and nxt > pos
        got += [(rs.name(i), rs.seq(i)) for i in range(rs.count)]
        pos = nxt
    assert got == [("a", "ACGT"), ("b", "ACGT"), ("c", "T")]
    other = str(tmp_path / "notes.txt")
    with open(other, "w") as f:
        f.write("not reads\n")
    assert fastq_record_start(other, 3) is None


def test_sizes_and_shared_spans_rebuild_the_single_writers_files(tmp_path):
    rng = random.Random(5)
    p = str(tmp_path / "reads.fastq")
    _fastq(p, 300, rng)
    rs = ReadSet(p)
    n = rs.count
    pr = np.array(sorted(rng.sample(range(n), 200) + rng.sample(range(n), 40)), dtype=np.int64)      # some reads in two pieces
    ps = np.array([rng.randint(0, int(rs.lengths[r]) // 2) for r in pr], dtype=np.int32)
    pn = np.array([rng.randint(0, int(rs.lengths[r]) - int(s)) for r, s in zip(pr, ps)], dtype=np.int32)
    num = np.array([rng.choice([0, 0, 1, 2]) for _ in pr], dtype=np.int32)
    pf = np.array([rng.randrange(3) for _ in pr], dtype=np.int32)
    for fastq in (True, False):
        ext = "fq" if fastq else "fa"
        single = [str(tmp_path / ("single%d.%s" % (k, ext))) for k in range(3)]
        rs.write(pr, ps, pn, num, pf, single, fastq)
        sizes = rs.write_sizes(pr, ps, pn, num, pf, 3, fastq)
        assert [os.path.getsize(x) for x in single] == sizes.tolist()
        # three "ranks" hold consecutive thirds of the pieces and write their spans in arbitrary order
        shared = [str(tmp_path / ("shared%d.%s" % (k, ext))) for k in range(3)]
        for x in shared:
            open(x, "wb").close()
        parts = np.array_split(np.arange(pr.size), 3)
        part_sizes = [rs.write_sizes(pr[i], ps[i], pn[i], num[i], pf[i], 3, fastq) for i in parts]
        for rank in (2, 0, 1):
            i = parts[rank]
            pos = np.ascontiguousarray(sum(part_sizes[:rank], np.zeros(3, dtype=np.int64)))
            rs.write_shared(pr[i], ps[i], pn[i], num[i], pf[i], shared, fastq, pos)
        for a, b in zip(single, shared):
            assert open(a, "rb").read() == open(b, "rb").read()
    rs.close()


def test_sized_member_gzip_is_cut_like_the_plain_file(tmp_path):
    """pc_gz_sized_find_record / pc_readset_load_gz_range: a gzip file of sized members addressed by positions in its
    INFLATED bytes gives every rank the cut and the reads the plain file gives it -- long lines that span many 64 KB
    members included -- and any other gzip file says "not such a file" (the caller falls back)."""
    import gzip
    from porechop_amd import io as pio
    rng = random.Random(11)
    p = str(tmp_path / "reads.fastq")
    with open(p, "w") as f:
        for i in range(500):
            L = rng.choice([1, 30, 200, 1500, 1500, 70000, 200000] if i % 50 == 7 else [1, 30, 200, 1500])
            seq = "".join(rng.choice("ACGT") for _ in range(L))
            qual = "".join(rng.choice("@+5I") for _ in range(L))
            f.write("@read%d some description\n%s\n+\n%s\n" % (i, seq, qual))
    size = os.path.getsize(p)
    z = str(tmp_path / "reads.fastq.gz")
    pio.gzip_file(p, z)
    assert pio.gz_sized_size(z) == size
    whole = ReadSet(p)
    want = [(whole.name(i), whole.seq(i), whole.quals(i)) for i in range(whole.count)]
    for pos in [0, 1, 2, size - 1, size, size + 9] + [rng.randrange(size) for _ in range(60)]:
        assert pio.gz_sized_record_start(z, pos) == fastq_record_start(p, pos), pos
    for world in (1, 2, 3, 7, 64):
        cuts = [0] + [pio.gz_sized_record_start(z, size * r // world) for r in range(1, world)] + [size]
        assert cuts == [0] + [fastq_record_start(p, size * r // world) for r in range(1, world)] + [size]
        got = []
        for r in range(world):
            rs = ReadSet.gz_range(z, cuts[r], cuts[r + 1])
            assert rs is not None
            got += [(rs.name(i), rs.seq(i), rs.quals(i)) for i in range(rs.count)]
            rs.close()
        assert got == want
    whole.close()
    # any other layout: not addressable without inflating everything
    text = open(p, "rb").read()
    plainz = str(tmp_path / "one.fastq.gz")
    open(plainz, "wb").write(gzip.compress(text[:200000], 1))
    assert pio.gz_sized_size(plainz) is None and pio.gz_sized_record_start(plainz, 5) is None and ReadSet.gz_range(plainz, 0, 10) is None
    assert pio.gz_sized_size(p) is None
    # a damaged member inside a rank's range is an error for that rank, not a silent loss
    raw = bytearray(open(z, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = str(tmp_path / "bad.fastq.gz")
    open(bad, "wb").write(bytes(raw))
    if pio.gz_sized_size(bad) == size:            # (the flip may hit a header: then the file is not "sized" any more)
        assert ReadSet.gz_range(bad, 0, size) is None
