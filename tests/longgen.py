"""Seeded ULTRA-LONG whole-read cases (reads of 65 535 bases and more) for the parity tests.

The reference takes any ``char*`` (porechop/src/adapter_align.cpp:11-27) and nanopore reads of 100 kb - 1 Mb are
routine; on the GPU path the columns beyond 65 535 are a code path of their own (the specialised score kernel keeps
the running maximum's column as packed u16 up to 65 000 columns and as plain ints beyond, pc_jit_source.h).  The cases
plant adapter copies where that matters: before and after column 65 535, across it, in the read's last columns, at a
given column (the caller passes chunk boundaries), twice (the second copy is what mask-and-realign finds in its second
round, nanopore_read.py:210-243), with runs of 'N' and of '-' (masked bases) around.
"""
import numpy as np

Y_TOP = "AATGTACTTCGTTCAGTTACGTATTGCT"        # SQK-NSK007 Y_Top, 28 bases (porechop/adapters.py:77-79)
Y_BOTTOM = "GCAATACGTAACTGAACGAAGT"           # SQK-NSK007 Y_Bottom, 22 bases
LENGTHS = (65535, 65536, 65537, 70000, 131073)
MILLION = 1_000_000


def long_adapter(m=111, seed=7):
    """A 111-base adapter (the length of the reference's generated full barcode adapters, porechop.py:410-436)."""
    rng = np.random.default_rng(seed)
    return "".join("ACGT"[i] for i in rng.integers(0, 4, m))


def mutated(rng, seq, subs=1, dels=1, ins=1):
    s = list(seq)
    for _ in range(subs):
        k = int(rng.integers(0, len(s)))
        s[k] = "ACGT"[("ACGT".index(s[k]) + 1 + int(rng.integers(0, 3))) % 4]
    for _ in range(dels):
        del s[int(rng.integers(1, len(s) - 1))]
    for _ in range(ins):
        s.insert(int(rng.integers(1, len(s) - 1)), "ACGT"[int(rng.integers(0, 4))])
    return "".join(s)


def make_read(n, seed, plants=(), n_runs=(), dash_runs=()):
    """n random bases; plants: (end_column, sequence) -- the copy's last base lands on 1-based column end_column;
    n_runs / dash_runs: (start, length) of 'N' / '-' runs (0-based).  -> str"""
    rng = np.random.default_rng(seed)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    for s, ln in n_runs:
        a[max(0, s):min(n, s + ln)] = ord("N")
    for s, ln in dash_runs:
        a[max(0, s):min(n, s + ln)] = ord("-")
    for end_col, seq in plants:
        b = np.frombuffer(seq.encode(), dtype=np.uint8)
        e = min(n, end_col)
        s = e - len(b)
        if s < 0:
            b = b[-s:]
            s = 0
        a[s:e] = b[:e - s]
    return a.tobytes().decode()


def cases(lengths=LENGTHS, adapters=None, seed=2025, chunk_cols=()):
    """-> list of (label, read, adapter).  For every length and adapter: no copy at all; an exact copy near the start;
    a mutated copy ending at columns 65 534 / 65 535 / 65 536 / 65 537 (where the read is long enough) and straddling
    65 535; an exact copy in the read's last columns and one cut off by the read's end; copies at the given chunk
    boundaries; two copies (a strong one late, a weaker one early) with N and '-' runs beside them."""
    adapters = adapters or [Y_BOTTOM, Y_TOP, long_adapter()]
    rng = np.random.default_rng(seed)
    out = []
    k = 0
    for n in lengths:
        for ad in adapters:
            m = len(ad)
            mut = mutated(rng, ad)
            k += 1
            out.append(("none n=%d m=%d" % (n, m), make_read(n, seed + k), ad))
            out.append(("start n=%d m=%d" % (n, m), make_read(n, seed + k, [(m + 40, ad)]), ad))
            for col in (65534, 65535, 65536, 65537, 65535 + m // 2):
                if col <= n:
                    out.append(("col%d n=%d m=%d" % (col, n, m), make_read(n, seed + k, [(col, mut)]), ad))
            out.append(("last n=%d m=%d" % (n, m), make_read(n, seed + k, [(n, ad)]), ad))
            out.append(("cut n=%d m=%d" % (n, m), make_read(n, seed + k, [(n + m // 3, ad)]), ad))
            for col in chunk_cols:
                if m < col <= n:
                    out.append(("chunk%d n=%d m=%d" % (col, n, m), make_read(n, seed + k, [(col + m // 2, mut)]), ad))
            late = min(n - 100, max(66000, n - 1000))
            out.append(("two n=%d m=%d" % (n, m),
                        make_read(n, seed + k, [(late, ad), (30000, mut)], n_runs=[(late + 5, 300), (29000, 50)],
                                  dash_runs=[(late - m - 400, 200), (64000, 2000)]), ad))
    return out
