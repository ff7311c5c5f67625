"""GPU parity tests: every call goes through the C ABI (include/porechop_amd.h) and is compared
bit-exactly -- ints and the printed identity decimals -- with
  * the committed goldens minted from the compiled reference (tests/golden/), and
  * the CPU oracle (oracle/pc_oracle.c) on seeded inputs.
"""
import os
import random

import numpy as np
import pytest

from tests.golden_io import comparable, load_synthetic
from tests.pairgen import LINEAR_SCHEMES, SCHEMES, random_case

pytestmark = pytest.mark.gpu

MAX_GPU_ADAPTER = 128


@pytest.fixture(scope="module")
def pa():
    import porechop_amd
    return porechop_amd


def run_batch(pa, cases, scores, mode=0):
    """cases: list of (read, adapter) strings -> list of 7-field strings via the batch C ABI."""
    ads, idx = [], {}
    pairs = []
    for rd, ad in cases:
        if ad not in idx:
            idx[ad] = len(ads)
            ads.append(ad)
        pairs.append((rd, idx[ad]))
    al = pa.Aligner(ads, scores)
    recs = al.align_pairs(pairs, mode)
    al.close()
    return [pa.format_result(r) for r in recs]


def test_known_answer_vectors_per_call(pa):
    # SURVEY.md section 8a, through the reference-shaped per-call symbol
    kav = [
        ("ACGTACGTAC", "ACGT", "0,3,0,3,12,100.000000,100.000000"),
        ("TTTTACGTTTTT", "ACGT", "4,7,0,3,12,100.000000,100.000000"),
        ("ACGT", "TTACGTTT", "0,3,2,5,12,100.000000,50.000000"),
        ("GTTT", "ACGT", "0,1,2,3,6,100.000000,50.000000"),
        ("TTAC", "ACGT", "2,3,0,1,6,100.000000,50.000000"),
        ("NNNNNNNN", "ACGT", "0,0,4,3,0,-nan,0.000000"),
        ("ACNNGT", "ACNNGT", "0,5,0,5,18,100.000000,100.000000"),
        ("AC--GT", "ACGT", "0,5,0,3,5,66.666667,66.666667"),
        ("ACXXGT", "ACNNGT", "0,5,0,5,18,100.000000,100.000000"),
        ("A", "C", "0,0,1,0,0,-nan,0.000000"),
        ("A", "ACGT", "0,0,0,0,3,100.000000,25.000000"),
        ("TTTTACGAACGTTTTT", "ACGTACGT", "4,11,0,7,15,87.500000,87.500000"),
        ("TTTTACGTTACGTTTTT", "ACGTACGT", "4,12,0,7,19,88.888889,88.888889"),
        ("AAAAAAAAAA", "CCCC", "0,0,4,3,0,-nan,0.000000"),
    ]
    for rd, ad, want in kav:
        assert pa.adapter_alignment(rd, ad, [3, -6, -5, -2]) == want, (rd, ad)
    assert pa.adapter_alignment("", "ACGT", [3, -6, -5, -2]).split(",")[0] == "-1"
    assert pa.adapter_alignment("ACGT", "", [3, -6, -5, -2]).split(",")[0] == "-1"


def test_recorded_reference_calls(pa, goldens):
    """All 25 680 distinct adapter_alignment calls the reference CLI makes on its bundled
    fixtures (end windows, whole-read middle scans incl. masked reads, 2 scoring schemes)."""
    S = goldens["strings"]
    by_scheme = {}
    for ri, ai, sc, res in goldens["calls"]:
        by_scheme.setdefault(tuple(sc), []).append((S[ri], S[ai], res))
    total = 0
    for sc, items in by_scheme.items():
        got = run_batch(pa, [(r, a) for r, a, _ in items], sc)
        bad = [(len(r), a, want, g) for (r, a, want), g in zip(items, got) if comparable(g) != comparable(want)]
        assert not bad, (sc, len(bad), bad[:5])
        total += len(items)
    assert total == len(goldens["calls"])


def test_synthetic_goldens_all_schemes(pa):
    by_scheme = {}
    for rd, ad, sc, res in load_synthetic():
        by_scheme.setdefault(tuple(sc), []).append((rd, ad, res))
    for sc, items in by_scheme.items():
        got = run_batch(pa, [(r, a) for r, a, _ in items], sc)
        bad = [(r, a, want, g) for (r, a, want), g in zip(items, got) if comparable(g) != comparable(want)]
        assert not bad, (sc, len(bad), bad[:5])


def test_random_end_windows_vs_oracle(pa, oracle):
    rng = random.Random(424242)
    for sc in SCHEMES:
        cases = [random_case(rng) for _ in range(6000)]
        got = run_batch(pa, cases, sc)
        bad = []
        for (rd, ad), g in zip(cases, got):
            want = oracle.adapter_alignment(rd, ad, sc)
            if comparable(g) != comparable(want):
                bad.append((rd, ad, want, g))
        assert not bad, (sc, len(bad), bad[:3])


def test_whole_read_two_pass_vs_oracle(pa, oracle):
    """Middle-scan shape: whole reads (2-12 kb) x short adapters; the GPU runs the score-only pass
    + bounded traced window, the oracle the full matrix."""
    rng = random.Random(99)
    cases = []
    for _ in range(300):
        n = rng.choice([700, 2000, 5000, 8000, 12000])
        m = rng.choice([22, 24, 28, 28, 33, 50])
        cases.append(random_case(rng, n=n, m=m, alphabet=rng.choice(["ACGT", "ACGT", "ACGT-", "ACGTN"])))
    got = run_batch(pa, cases, (3, -6, -5, -2))
    bad = [(len(rd), ad, g) for (rd, ad), g in zip(cases, got) if g != oracle.adapter_alignment(rd, ad)]
    assert not bad, (len(bad), bad[:3])
    # large gap-extension penalties: the register kernels' drifting coordinates are renormalised every
    # few hundred columns (eps = 40 -> every ~470), or cannot drift at all (eps = 140 -> LDS-state kernel)
    for sc in ((5, -4, -10, -40), (4, -7, -10, -140)):
        got = run_batch(pa, cases[:120], sc)
        bad = [(len(rd), ad, g) for (rd, ad), g in zip(cases[:120], got) if g != oracle.adapter_alignment(rd, ad, sc)]
        assert not bad, (sc, len(bad), bad[:3])


def test_modes_agree(pa):
    """The same pairs through the one-pass traced kernel and through the two-pass scan."""
    rng = random.Random(5)
    cases = [random_case(rng, n=rng.choice([300, 900, 1500]), m=rng.choice([22, 24, 28, 40])) for _ in range(400)]
    a = run_batch(pa, cases, (3, -6, -5, -2), mode=pa.MODE_TRACE)
    b = run_batch(pa, cases, (3, -6, -5, -2), mode=pa.MODE_TWO_PASS)
    assert a == b


def test_ragged_and_unaligned_windows(pa, oracle):
    """Windows at every byte alignment, lengths 1..160, adapters 1..56, sharing one arena."""
    rng = random.Random(17)
    arena = "".join(rng.choice("ACGT") for _ in range(5000))
    ads = ["".join(rng.choice("ACGT") for _ in range(m)) for m in (1, 2, 7, 16, 22, 23, 24, 28, 31, 32, 33, 40, 50, 56)]
    al = pa.Aligner(ads)
    offs, lens, idx = [], [], []
    for k in range(4000):
        n = rng.randint(1, 160)
        o = rng.randint(0, len(arena) - n)
        offs.append(o); lens.append(n); idx.append(rng.randrange(len(ads)))
    recs = al.align_host(arena.encode(), offs, lens, idx)
    al.close()
    for o, n, i, r in zip(offs, lens, idx, recs):
        assert pa.format_result(r) == oracle.adapter_alignment(arena[o:o + n], ads[i]), (o, n, ads[i])


def test_linear_gap_schemes_vs_oracle(pa, oracle):
    """gap_open == gap_extend: the reference's linear-gap dispatch (global_alignment_unbanded.h:217-220),
    end windows and whole reads (two-pass)."""
    rng = random.Random(606)
    for sc in LINEAR_SCHEMES:
        cases = [random_case(rng) for _ in range(3000)]
        cases += [random_case(rng, n=rng.choice([900, 4000]), m=rng.choice([22, 28, 33])) for _ in range(40)]
        got = run_batch(pa, cases, sc)
        bad = [(rd, ad, g) for (rd, ad), g in zip(cases, got)
               if comparable(g) != comparable(oracle.adapter_alignment(rd, ad, sc))]
        assert not bad, (sc, len(bad), bad[:3])


def test_any_scheme_and_any_adapter_length_is_computed(pa, oracle):
    """The reference takes any four integers and any adapter (porechop.py:145,196-202): what the packed 16-bit kernels
    refuse -- non-negative gap scores, match <= mismatch, magnitudes beyond 16 bits, adapters above 128 bases -- runs the
    plain-int32 kernel (csrc/pc_slow.hip) and gives the oracle's strings (the oracle is pinned on the compiled reference
    for such schemes too: tests/test_oracle_vs_ref.py).  End windows and whole reads, the batch API and the per-call
    symbol."""
    rng = random.Random(4242)
    schemes = [(3, -6, 5, -2), (3, 4, -5, -2), (3000, -6, -5, -2), (0, 0, 0, 0), (1, -1, 0, 0), (2, -3, 1, 2), (-2, 3, -1, -4),
               (5, -4, 3, 3), (100000, -70000, -90000, -30000), (3, 3, -5, -2), (1, 1, 1, 1), (4, -5, -3, 0)]
    for sc in schemes:
        cases = [random_case(rng) for _ in range(400)]
        cases += [random_case(rng, n=rng.choice([900, 2500]), m=rng.choice([22, 28, 33])) for _ in range(12)]
        cases += [("", "ACGT"), ("ACGT", "")]
        got = run_batch(pa, cases, sc)
        bad = [(rd, ad, g, oracle.adapter_alignment(rd, ad, sc)) for (rd, ad), g in zip(cases, got)
               if comparable(g) != comparable(oracle.adapter_alignment(rd, ad, sc))]
        assert not bad, (sc, len(bad), bad[:3])
        assert comparable(pa.adapter_alignment("TTTTACGTTTTT", "ACGT", list(sc))) == comparable(oracle.adapter_alignment("TTTTACGTTTTT", "ACGT", sc))
    # long adapters under the default scheme, mixed with short ones in one batch (the short ones keep the packed kernels)
    sc = (3, -6, -5, -2)
    cases = []
    for _ in range(150):
        rd, ad = random_case(rng, n=rng.choice([150, 600, 1500]), m=rng.choice([129, 200, 333, 700]))
        cases.append((rd, ad))
    cases += [random_case(rng) for _ in range(300)]
    rng.shuffle(cases)
    got = run_batch(pa, cases, sc)
    bad = [(len(rd), len(ad), g) for (rd, ad), g in zip(cases, got) if comparable(g) != comparable(oracle.adapter_alignment(rd, ad, sc))]
    assert not bad, (len(bad), bad[:3])


def test_scores_beyond_the_32_bit_safe_range_fail_loudly(pa):
    with pytest.raises(RuntimeError):
        pa.Aligner(["ACGT"], scores=(3, -6, -5, -(1 << 21)))
    with pytest.raises(RuntimeError):
        pa.Aligner(["A" * 5000])                             # beyond PC_MAX_ADAPTER_ANY
    with pytest.raises(RuntimeError):
        pa.adapter_alignment("ACGT", "ACGT", [1 << 22, -6, -5, -2])


def test_specialised_kernel_matches_generic(pa, oracle):
    """The run-time specialised (hiprtc) score pass, forced on for a small job through
    PC_JIT_MIN_CELLS in a fresh process, against the oracle: dual-adapter one-stream tiles, padding
    rows in the shorter half, chunked windows, odd row counts; ragged tiles (per-stream masks) and
    equal-length tiles (block-resolved maxima); a scheme whose drifting coordinates are
    renormalised every ~200 columns, and one with gap_extend more negative than gap_open."""
    import os
    import re
    import subprocess
    import sys
    code = r'''
import random, sys
sys.path.insert(0, ".")
import numpy as np, torch
import porechop_amd
from oracle.oracle import Oracle
from tests.pairgen import random_case
rng = random.Random(77)
o = Oracle()
ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "CTTCGTTCAGTTACGTATTGCTGGCGTCTGCTT", "ACGTNACGTTAGC",
       "GGTTGTTTCTGTTGGTGCTGATATTGCTGGCGTCTGCTT", "AAGCAGACGCCAGCAATATCAGCACCAACAGAAACAAC",
       # full native-barcode adapters (68 / 63 bases) and a 111-base one: one wave per SIMD, rows parked in AGPRs
       "AATGTACTTCGTTCAGTTACGTATTGCTAAGGTTAACACAAAGACACCGACAACTTTCTTCAGCACCT",
       "AGGTGCTGAAGAAAGTTGTCGGTGTCTTTGTGTTAACCTTAGCAATACGTAACTGAACGAAGT",
       "AATGTACTTCGTTCAGTTACGGCTTGGGTGTTTAACCAAGAAAGTTGTCGGTGTCTTTGTGGTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"]
for scores, lengths in [((3, -6, -5, -2), [900, 2500, 6000]), ((3, -6, -5, -2), [5003]),
                        ((20, -30, -25, -12), [3000]), ((3, -6, -2, -5), [1800, 1801]), ((20, -30, -25, -12), [700, 2100])]:
    reads = [random_case(rng, n=rng.choice(lengths), m=28)[0] for _ in range(150)]
    for i in range(0, 150, 3):                      # implant copies of each adapter
        a = ads[(i // 3) % 9]; p = rng.randint(0, len(reads[i]) - 120)
        reads[i] = reads[i][:p] + a + reads[i][p + len(a):]
    al = porechop_amd.Aligner(ads, scores=scores)
    arena = torch.from_numpy(np.frombuffer(("".join(reads)).encode() + b"N" * 64, dtype=np.uint8).copy()).cuda()
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    woff, wlen = torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda()
    n = len(reads)
    for (a, b) in [(0, 1), (2, 3), (1, -1), (4, 5), (6, 7), (8, -1)]:
        out = torch.zeros((n * (2 if b >= 0 else 1), 8), dtype=torch.int32, device="cuda")
        al.scan_device(arena, woff, wlen, [a], [0, n], int(lens.max()), out, porechop_amd.MODE_TWO_PASS, job_adapter_b=[b])
        al.sync()
        rec = out.cpu().numpy()
        for i, r in enumerate(reads):
            assert porechop_amd.format_result(rec[i]) == o.adapter_alignment(r, ads[a], scores), (scores, a, i)
            if b >= 0:
                assert porechop_amd.format_result(rec[n + i]) == o.adapter_alignment(r, ads[b], scores), (scores, b, i)
print("SPEC_OK")
'''
    for int16 in ("0", "1"):      # packed-fp16 (5 ops, v_pk_maximum3_f16) and packed-int16 (6 ops) variants
        env = dict(os.environ, PC_JIT_MIN_CELLS="1", PC_JIT_INT16=int16, PC_JIT_VERBOSE="1")
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert "SPEC_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
        assert "hiprtc" not in res.stderr and "no specialised kernel" not in res.stderr, res.stderr[-2000:]
        built = re.findall(r"specialised kernel R=(\d+) K=\d+ f16=(\d) kren=(\d+)", res.stderr)
        assert len(built) == 18, res.stderr[-2000:]                  # 6 adapter pairs (22..111 rows) x 3 schemes
        if int16 == "1":
            assert all(f == "0" for _, f, _ in built)
        else:
            # fp16 wherever every value fits its exact-integer range (the long adapters under the
            # 20/-30/-25/-12 scheme do not: they take the int16 variant)
            assert all(f == "1" for r, f, _ in built if int(r) <= 40) and sum(f == "1" for _, f, _ in built) >= 16
        if int16 == "0":
            assert min(int(k) for _, _, k in built) < 300            # the renormalisation path ran


def test_specialised_kernel_random_schemes(pa, oracle):
    """Drifting-coordinate constants (centre, renormalisation period, fp16 vs int16 choice) are
    derived from the scoring scheme: random valid schemes, whole reads, specialised kernels forced
    on, against the oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import random, sys
sys.path.insert(0, ".")
import numpy as np, torch
import porechop_amd
from oracle.oracle import Oracle
from tests.pairgen import random_case
rng = random.Random(4242)
o = Oracle()
ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "CTTCGTTCAGTTACGTATTGCTGGCGTCTGCTT"]
done = 0
while done < 10:
    match = rng.randint(1, 40); mismatch = -rng.randint(0, 60); go = -rng.randint(1, 60); ge = -rng.randint(1, 60)
    if go == ge or match <= mismatch:
        continue
    scores = (match, mismatch, go, ge)
    try:
        al = porechop_amd.Aligner(ads, scores=scores)
    except RuntimeError:
        continue                                   # outside the exact path: refused loudly, fine
    done += 1
    reads = [random_case(rng, n=rng.choice([1500, 4000]), m=28)[0] for _ in range(96)]
    for i in range(0, 96, 2):
        a = ads[(i // 2) % 3]; p = rng.randint(0, len(reads[i]) - 60)
        reads[i] = reads[i][:p] + a + reads[i][p + len(a):]
    arena = torch.from_numpy(np.frombuffer(("".join(reads)).encode() + b"N" * 64, dtype=np.uint8).copy()).cuda()
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    woff, wlen = torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda()
    n = len(reads)
    for (a, b) in [(0, 1), (2, -1)]:
        out = torch.zeros((n * (2 if b >= 0 else 1), 8), dtype=torch.int32, device="cuda")
        al.scan_device(arena, woff, wlen, [a], [0, n], int(lens.max()), out, porechop_amd.MODE_TWO_PASS, job_adapter_b=[b])
        al.sync()
        rec = out.cpu().numpy()
        for i, r in enumerate(reads):
            assert porechop_amd.format_result(rec[i]) == o.adapter_alignment(r, ads[a], scores), (scores, a, i)
            if b >= 0:
                assert porechop_amd.format_result(rec[n + i]) == o.adapter_alignment(r, ads[b], scores), (scores, b, i)
    lo, hi = al.debug_value_range()
    al.close()
    print("SCHEME", scores, "RANGE", lo, hi)
print("SPEC_OK")
'''
    # PC_JIT_CHECK_RANGE=1: the kernels are built with the on-device range assertion (every T / U held must
    # stay within what fp16 / int16 represent exactly, else the launch fails); al.sync() above would raise
    env = dict(os.environ, PC_JIT_MIN_CELLS="1", PC_JIT_VERBOSE="1", PC_JIT_CHECK_RANGE="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "SPEC_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
    assert "hiprtc" not in res.stderr, res.stderr[-2000:]
    assert res.stderr.count("specialised kernel R=") >= 6, res.stderr[-2000:]      # most schemes do specialise
    ranges = [l.split("RANGE")[1].split() for l in res.stdout.splitlines() if l.startswith("SCHEME")]
    assert len(ranges) == 10 and any(int(hi) > 500 for _, hi in ranges)           # the recorder really ran
    assert all(-32000 <= int(lo) and int(hi) <= 32000 for lo, hi in ranges), ranges


def test_fp16_and_int16_traced_kernels_agree(pa, oracle):
    """The traced scan has two implementations of the same recurrence: packed fp16 (trace16_kernel,
    the default wherever pc_bounds.h f16_plan says every value stays exact) and packed int16
    (scan_kernel<R,PAD,true>, PC_DISABLE_F16=1).  The same batch -- end windows of every length and
    alphabet against adapters of 1..72 bases under seven schemes, ragged dual-adapter tiles,
    pass-2 windows of whole reads -- through both, each in a fresh process: identical digests, and
    the fp16 run equal to the oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, random, sys
sys.path.insert(0, ".")
import numpy as np, torch
import porechop_amd
from oracle.oracle import Oracle
from tests.golden_io import comparable
from tests.pairgen import SCHEMES, random_case
check = sys.argv[1] == "check"
o = Oracle()
rng = random.Random(20260925)
h = hashlib.sha1()
for sc in SCHEMES + [(20, -30, -25, -12)]:
    cases = [random_case(rng, m=rng.choice([1, 7, 16, 22, 24, 24, 28, 28, 30, 33, 40, 47, 56, 63, 68, 72])) for _ in range(2500)]
    cases += [random_case(rng, n=rng.choice([160, 200, 412, 700, 3000]), m=rng.choice([22, 24, 28, 33, 50, 68])) for _ in range(60)]
    ads, idx, pairs = [], {}, []
    for rd, ad in cases:
        if ad not in idx:
            idx[ad] = len(ads); ads.append(ad)
        pairs.append((rd, idx[ad]))
    al = porechop_amd.Aligner(ads, sc)
    recs = al.align_pairs(pairs)
    al.close()
    h.update(recs.tobytes())
    if check:
        for (rd, ad), r in zip(cases, recs):
            want = o.adapter_alignment(rd, ad, sc)
            assert comparable(porechop_amd.format_result(r)) == comparable(want), (sc, rd, ad, want, porechop_amd.format_result(r))
# dual-adapter one-stream tiles with ragged windows (device API), the phase-B shape
ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "GGTTGTTTCTGTTGGTGCTGATATTGCTGGCGTCTGCTT", "AAGAAAGTTGTCGGTGTCTTTGTG"]
reads = [random_case(rng, n=rng.choice([3, 40, 149, 150, 150, 150]), m=28)[0] for _ in range(1000)]
for i in range(0, 1000, 2):
    a = ads[(i // 2) % 4]
    reads[i] = (a + reads[i])[:150] if i % 4 == 0 else (reads[i] + a)[-150:]
arena = torch.from_numpy(np.frombuffer(("".join(reads)).encode() + b"N" * 64, dtype=np.uint8).copy()).cuda()
lens = np.array([len(r) for r in reads], dtype=np.int32)
offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
woff, wlen = torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda()
n = len(reads)
al = porechop_amd.Aligner(ads)
for (a, b) in [(0, 1), (2, 3), (1, -1)]:
    out = torch.zeros((n * (2 if b >= 0 else 1), 8), dtype=torch.int32, device="cuda")
    al.scan_device(arena, woff, wlen, [a], [0, n], 150, out, porechop_amd.MODE_TRACE, job_adapter_b=[b])
    al.sync()
    rec = out.cpu().numpy()
    h.update(rec.tobytes())
    if check:
        for i, r in enumerate(reads):
            assert porechop_amd.format_result(rec[i]) == o.adapter_alignment(r, ads[a]), (a, i)
            if b >= 0:
                assert porechop_amd.format_result(rec[n + i]) == o.adapter_alignment(r, ads[b]), (b, i)
print("DIGEST", h.hexdigest(), porechop_amd.load_library().pc_trace_ops_x100(al._ctx))
'''
    outs = []
    # PC_CHECK_RANGE=1: row classes 24/28/30/40 run the build with the on-device assertion |value| <= 2040
    for extra, arg in (({"PC_CHECK_RANGE": "1"}, "check"), ({"PC_DISABLE_F16": "1"}, "nocheck")):
        res = subprocess.run([sys.executable, "-c", code, arg], capture_output=True, text=True, timeout=1200,
                             env=dict(os.environ, **extra), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        lines = [l for l in res.stdout.splitlines() if l.startswith("DIGEST")]
        assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-3000:]
        outs.append(lines[0].split())
    assert outs[0][1] == outs[1][1]
    assert (outs[0][2], outs[1][2]) == ("1325", "2100")          # the two runs really took different kernels


def test_large_match_score_long_adapter_tracked_term_stays_exact(pa, oracle):
    """Regression (found by tools/fuzz_parity.py 36 5, block 10): scheme (29, -19, -14, -7), a 64-base adapter
    copied exactly -- or nearly -- into 149..151-column windows.  The traced fp16 kernel's tracked last-row term
    M + R*eps reaches 1856 + 448 = 2304, which fp16 does not hold exactly; the host gate (pc_bounds.h f16_plan)
    must send such a scheme / row class to the int16 kernel.  Every record against the oracle."""
    import random
    rng = random.Random(29)
    scheme = (29, -19, -14, -7)
    ad = "".join(rng.choice("ACGT") for _ in range(64))
    others = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (24, 38, 22, 5)]
    pairs = []
    for k in range(3000):
        n = rng.choice([149, 150, 150, 151, 100])
        w = [rng.choice("ACGT") for _ in range(n)]
        copy = list(ad)
        for _ in range(rng.choice([0, 0, 1, 3])):                    # exact copies and lightly mutated ones
            copy[rng.randrange(64)] = rng.choice("ACGT")
        p = rng.randrange(0, max(1, n - 64))
        w[p:p + 64] = copy[: max(0, n - p)]
        pairs.append(("".join(w[:n]), 0))
        pairs.append(("".join(w[:n]), 1 + k % 4))
    al = pa.Aligner([ad] + others, scores=scheme)
    recs = al.align_pairs(pairs)
    al.close()
    ads = [ad] + others
    n_full = 0
    for (rd, ai), rec in zip(pairs, recs):
        want = oracle.adapter_alignment(rd, ads[ai], scheme)
        assert pa.format_result(rec) == want, (len(rd), ai, pa.format_result(rec), want)
        n_full += int(rec[4] >= 29 * 60)
    assert n_full > 500                                              # the near-perfect 64-base hits were there


def test_leaving_the_packed_kernels_is_said_on_stderr():
    """`--scoring_scheme 3,-6,-5,0` (a zero gap-extension score: the reference's CLI takes it, porechop.py:145,196-202) and an
    adapter above 128 bases run the plain-int32 kernel -- a hundred times slower per cell, no prunings: the library says so,
    once; PC_QUIET=1 silences it."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import porechop_amd\n"
            "for k in range(2):\n"
            "    al = porechop_amd.Aligner(['AATGTACTTCGTTCAGTTACGTATTGCT'], (3, -6, -5, 0)); al.align_pairs([('ACGT' * 40, 0)]); al.close()\n"
            "al = porechop_amd.Aligner(['ACGT' * 50, 'ACGTACGTACGTAAAC']); al.align_pairs([('ACGT' * 40, 0), ('ACGT' * 40, 1)]); al.close()\n" % repo)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count("scoring scheme 3,-6,-5,0 is outside the packed 16-bit kernels' exact range") == 1, r.stderr[-2000:]
    assert r.stderr.count("an adapter of 200 bases is longer than the 128") == 1, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=repo, env=dict(os.environ, PC_QUIET="1"))
    assert r.returncode == 0 and "plain-int32" not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["PC_TWO_PASS_ENDS", "PC_SPLIT_WALK"])
def test_measured_and_shelved_variants_stay_exact(switch):
    """The two-pass end scan and the tracebacks-as-their-own-launch variant (built, measured slower, left
    behind an environment switch that is read once per process) must keep giving the reference's records: the recorded
    reference calls and the synthetic goldens through each of them, in a process of its own."""
    import subprocess
    import sys
    env = dict(os.environ)
    env[switch] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                        "recorded or synthetic or random_end_windows or modes_agree or ragged", "--deselect", os.path.abspath(__file__) + "::test_measured_and_shelved_variants_stay_exact"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, r.stdout[-500:]
