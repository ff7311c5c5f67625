"""the pruned adapter-set search (score-only pass, traceback only where the score
can still mean an identity >= --adapter_threshold) must find exactly the sets the full search finds,
with the same best identities for them; entries of sets below the threshold may only get smaller.
Host logic with the oracle stand-in here; the same check on the GPU in tests/test_gpu_pipeline.py."""
import numpy as np
import torch

from tests import readgen
from tests.cpu_aligner import OracleAligner


def check_pruned_equals_full(pl, reads, check):
    bs0, be0 = pl.phase_a(reads, check)
    bs1, be1 = pl.phase_a(reads, check, prune=True)
    m0, m1 = pl.matching_sets(bs0, be0), pl.matching_sets(bs1, be1)
    assert m0 == m1 and len(m0) >= 1
    thr = pl.p.adapter_threshold
    for a, b in ((bs0, bs1), (be0, be1)):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        reach = a >= thr
        assert np.array_equal(a[reach], b[reach])
        assert (b <= a).all()
    return m0


def test_pruned_search_finds_the_same_sets(oracle):
    from porechop_amd.panel import load_panel
    from porechop_amd.pipeline import DeviceReads, Pipeline, ScanParams
    rr = readgen.native_reads(11, 120) + readgen.rapid_reads(21, 60)
    seqs = [r[1].upper().replace("U", "T") for r in rr]
    arena = np.frombuffer(("".join(seqs)).encode() + b"N" * 64, dtype=np.uint8).copy()
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    reads = DeviceReads(torch.from_numpy(arena), torch.from_numpy(offs), torch.from_numpy(lens))
    for scores in ((3, -6, -5, -2), (2, -3, -5, -2)):
        p = ScanParams(scores=scores)
        pl = Pipeline(load_panel(), p, aligner=OracleAligner(oracle, scores))
        assert pl.presence_score_bound(24) is not None
        names = [pl.sets[i].name for i in check_pruned_equals_full(pl, reads, torch.arange(len(seqs)))]
        assert "SQK-NSK007" in names and any(n.startswith("Barcode") for n in names)
        assert pl.stats["pairs_end_traced_after_pruning"] < 0.2 * pl.stats["pairs_end"]
    # a scheme whose mismatches are too cheap... gives a bound only if t*match > P*(1-t)
    pl = Pipeline(load_panel(), ScanParams(scores=(1, -20, -30, -30), adapter_threshold=90.0), aligner=OracleAligner(oracle, (1, -20, -30, -30)))
    assert pl.presence_score_bound(24) is None


def test_proven_middle_scan_finds_the_same_hits(oracle):
    """phase_c(prove=True): score-only pass + traceback of the pairs whose score can still mean an
    identity >= --middle_threshold; hits, order and alignment counts equal the full scan's."""
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
        matching = [i for i, s in enumerate(panel) if s.name in ("SQK-NSK007", "Barcode 1 (reverse)", "Barcode 2 (reverse)")]
        st, et = pl.phase_b(reads, matching)
        h0 = pl.phase_c(reads, st, et, matching)
        h1 = pl.phase_c(reads, st, et, matching, prove=True)
        assert h0.read.numel() >= 5
        for f in ("read", "adapter", "start", "end", "identity"):
            assert torch.equal(getattr(h0, f), getattr(h1, f)), f
        assert (h0.rounds, h0.alignments) == (h1.rounds, h1.alignments)
        assert pl.stats["pairs_middle_traced_after_proof"] < 0.3 * len(seqs) * len(pl.middle_adapters)
