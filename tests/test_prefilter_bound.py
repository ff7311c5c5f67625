"""The exactness argument of the bit-parallel prefilter (porechop_amd/csrc/pc_prefilter.hip), pinned on the
REFERENCE's own outputs.

Claim: an alignment whose full-adapter identity is F has at most  (full span columns - matches)  non-matching
columns inside the adapter's span; they are unit-cost edits, so the smallest edit distance between the whole
adapter and any substring of the read (oracle/pc_oracle.c pc_oracle_min_edits) is at most that, and at most
pc_prefilter_max_edits(m, F).  Checked for every call of the compiled reference recorded in tests/golden/
(25 680 calls made by the reference's CLI over its own fixtures, whole reads and masked reads included, plus
12 000 synthetic ones).  So: min_edits > max_edits(m, threshold)  =>  the reference's alignment is not a hit."""
import math

import numpy as np
import torch

from tests import readgen
from tests.cpu_aligner import OracleAligner
from tests.golden_io import load_ref_calls, load_synthetic


def max_edits(m, threshold):
    return OracleAligner(None).max_edits(m, threshold)


def full_identity_fields(result):
    f = result.split(",")
    if f[0] == "-1":
        return None
    return float(f[6])


def check_case(oracle, read, adapter, result):
    full = full_identity_fields(result)
    if full is None or not adapter or not read:
        return 0
    m = len(adapter)
    d = oracle.min_edits(read.upper(), adapter)
    assert 0 <= d <= m
    # for every threshold the reference's alignment reaches, the prefilter's bound must let the pair through
    for thr in (full, math.floor(full), 90.0, 85.0, 75.0):
        if full >= thr and thr > 0:
            assert d <= max_edits(m, thr), (read[:60], adapter, result, d, thr, max_edits(m, thr))
    return 1


def test_bound_holds_on_every_recorded_reference_call(oracle):
    g = load_ref_calls()
    strings = g["strings"]
    n = 0
    for ri, ai, _scheme, res in g["calls"]:
        n += check_case(oracle, strings[ri], strings[ai], res)
    assert n > 20000


def test_bound_holds_on_the_synthetic_goldens(oracle):
    n = 0
    for rd, ad, _sc, res in load_synthetic():
        n += check_case(oracle, rd, ad, res)
    assert n > 8000


def test_max_edits_values():
    # 90 %: one edit per nine matches; the rounding of the identity to six decimals never loses a hit
    assert [max_edits(m, 90.0) for m in (9, 18, 22, 24, 27, 28, 33, 36)] == [1, 2, 2, 2, 3, 3, 3, 4]
    assert max_edits(24, 100.0) == 0 and max_edits(24, 0.0) == 24 and max_edits(0, 90.0) == -1
    for m in range(1, 130):
        for t in (50.0, 75.0, 85.0, 90.0, 95.0, 99.0):
            k = max_edits(m, t)
            # k is the largest e with  M / (M + e) >= t/100 (after %f rounding) for some M <= m
            ok = [e for e in range(0, m + 1) if round(100.0 * m / (m + e), 6) >= t]
            assert k == max(ok), (m, t, k, ok[-1])


def test_library_restates_the_same_bound():
    import porechop_amd
    lib = porechop_amd.load_library()
    for m in range(0, 130):
        for t in (0.0, 50.0, 75.0, 85.0, 90.0, 95.0, 99.0, 100.0):
            assert lib.pc_prefilter_max_edits(m, t) == max_edits(m, t), (m, t)


def test_prefiltered_middle_scan_finds_the_same_hits(oracle):
    """phase_c(prefilter=True) with the oracle stand-in (its prefilter is the plain DP of the contract): hits,
    order, rounds and alignment counts equal the full scan's; few pairs reach the DP."""
    from porechop_amd.panel import load_panel
    from porechop_amd.pipeline import DeviceReads, Pipeline, ScanParams
    rr = readgen.ligation_reads(31, 60) + readgen.native_reads(11, 40)
    seqs = [r[1].upper().replace("U", "T") for r in rr]
    arena = np.frombuffer(("".join(seqs)).encode() + b"N" * 64, dtype=np.uint8).copy()
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    reads = DeviceReads(torch.from_numpy(arena), torch.from_numpy(offs), torch.from_numpy(lens))
    panel = load_panel()
    for thr in (90.0, 80.0):
        p = ScanParams(middle_threshold=thr)
        pl = Pipeline(panel, p, aligner=OracleAligner(oracle, p.scores))
        matching = [i for i, s in enumerate(panel) if s.name in ("SQK-NSK007", "Barcode 1 (reverse)", "Barcode 2 (reverse)", "Rapid")]
        st, et = pl.phase_b(reads, matching)
        h0 = pl.phase_c(reads, st, et, matching)
        h1 = pl.phase_c(reads, st, et, matching, prefilter=True)
        assert h0.read.numel() >= 5
        for f in ("read", "adapter", "start", "end", "identity"):
            assert torch.equal(getattr(h0, f), getattr(h1, f)), f
        assert (h0.rounds, h0.alignments) == (h1.rounds, h1.alignments)
        assert pl.stats["pairs_middle_scanned_after_prefilter"] < 0.5 * pl.stats["pairs_middle_prefiltered"]
