"""The bounds behind the exact pruning of phase B, checked against the oracle on the host.

The pruned phase B (porechop_amd/csrc/pc_select.hip; Pipeline._phase_b_bounds is the same arithmetic in torch, and
tests/test_gpu_phase_b_pruning.py checks the two against each other for every pair of its batches) traces an end-window
alignment only if the END CELL (I adapter bases, Jc window columns consumed) and score S of the score-only pass leave it
a chance to change a trim or a barcode call.  Here, without a GPU: for thousands of window / adapter pairs and several
scoring schemes the oracle gives the reference's alignment AND its end cell; what that alignment contributes to
nanopore_read.py:166-208 (its trim) and :399-466 (its full identity) must never exceed the bounds formed from
(I, Jc, S) alone."""
import random

import numpy as np
import torch

from tests.cpu_aligner import OracleAligner
from tests.pairgen import mutate


def windows_and_adapters(rng, n, adapters):
    """end-window-like cases: random windows of 1..150 bases with (mutated, truncated, shifted) adapter copies at
    every kind of position -- at the outer edge, at the inner edge, in the middle, hanging over either end"""
    out = []
    for k in range(n):
        ln = rng.choice([1, 2, 5, 17, 40, 100, 149, 150, 150, 150, 150])
        w = "".join(rng.choice("ACGT" if k % 11 else "ACGTN") for _ in range(ln))
        ad = rng.choice(adapters)
        if rng.random() < 0.8:
            c = mutate(rng, ad, rng.choice([0.0, 0.0, 0.05, 0.15, 0.3]))
            if rng.random() < 0.4:
                c = c[rng.randint(0, max(0, len(c) - 1)):] if rng.random() < 0.5 else c[:rng.randint(1, max(1, len(c)))]
            pos = rng.randint(-len(c) + 1, ln - 1)
            if pos < 0:
                c, pos = c[-pos:], 0
            w = (w[:pos] + c + w[pos + len(c):])[:ln]
        out.append((w, ad))
    return out


def test_no_alignment_exceeds_the_bounds_formed_from_its_end_cell(oracle):
    from porechop_amd.panel import load_panel
    from porechop_amd.pipeline import Pipeline, ScanParams, _identities
    rng = random.Random(20240)
    panel = load_panel()
    adapters = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "AAGAAAGTTGTCGGTGTCTTTGTG", "CACAAAGACACCGACAACTTTCTT",
                "GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA", "ACGTA", "TTTTTTTTCCTGTACTTCGTTCAGTTACGTATTGCT"]
    cases = windows_and_adapters(rng, 2500, adapters)
    total = contributing = tight = pruned = 0
    for scores in ((3, -6, -5, -2), (2, -3, -5, -2), (5, -4, -10, -1), (3, -6, -2, -5), (1, -1, -4, -3)):
        p = ScanParams(scores=scores)
        pl = Pipeline(panel, p, aligner=OracleAligner(oracle, scores))
        for side in (0, 1):
            # one "job" per adapter (that is what _phase_b_bounds works on): [J, R] tables of score records / full records
            by_ad = {}
            for w, ad in cases:
                by_ad.setdefault(ad, []).append(w)
            R = min(len(v) for v in by_ad.values())
            J = len(by_ad)
            score_rec = torch.zeros((J, R, 8), dtype=torch.int32)
            full_rec = torch.zeros((J, R, 8), dtype=torch.int32)
            wlen = torch.zeros((J, R), dtype=torch.int32)
            jobs, where = [], []
            for j, (ad, ws) in enumerate(by_ad.items()):
                if ad not in pl.seq_index:
                    pl.seq_index[ad] = len(pl.seqs)
                    pl.seqs.append(ad)
                jobs.append((pl.seq_index[ad], None, None))
                where.append((side, 0))
                for r, w in enumerate(ws[:R]):
                    res = oracle.align_raw(w, ad, scores)
                    wlen[j, r] = len(w)
                    score_rec[j, r] = torch.tensor([-2, res.end_j, res.end_i, 0, res.score, 0, 0, 0], dtype=torch.int32)
                    if res.failed:
                        full_rec[j, r, 0] = -1
                    else:
                        full_rec[j, r] = torch.tensor([res.read_start, res.read_end, res.adapter_start, res.adapter_end, res.score,
                                                       res.aligned_matches, res.aligned_len, res.full_len], dtype=torch.int32)
            # (the bounds take one window length per read; here every pair has its own window: job by job)
            for j in range(J):
                ub, ub_full = pl._phase_b_bounds(score_rec[j:j + 1], jobs[j:j + 1], where[j:j + 1], wlen[j], wlen[j])
                rec = full_rec[j:j + 1]
                full, partial = _identities(rec)
                ok = rec[..., 0] != -1
                rs = rec[..., 0].to(torch.int64)
                re = rec[..., 1].to(torch.int64) + 1
                good = ok & (partial > p.end_threshold) & ((re - rs) >= p.min_trim_size)
                if side == 0:
                    val = torch.where(good & (re != p.end_size), re + p.extra_end_trim, torch.zeros_like(re))
                else:
                    # nanopore_read.py:200-204 measures the end trim on a window of end_size bases; a shorter window
                    # (read shorter than end_size) still reports end_size - read_start
                    val = torch.where(good & (rs != 0), p.end_size - rs + p.extra_end_trim, torch.zeros_like(re))
                bad = torch.nonzero(val > ub)
                assert bad.shape[0] == 0, (scores, side, jobs[j], score_rec[j, bad[0, 1]].tolist(), rec[0, bad[0, 1]].tolist(),
                                           int(val[0, bad[0, 1]]), int(ub[0, bad[0, 1]]))
                fullv = torch.where(ok, torch.nan_to_num(full, nan=0.0), torch.zeros_like(full))
                badf = torch.nonzero(fullv > ub_full + 1e-6)
                assert badf.shape[0] == 0, (scores, side, jobs[j], score_rec[j, badf[0, 1]].tolist(), rec[0, badf[0, 1]].tolist())
                total += R
                contributing += int((val > 0).sum())
                tight += int(((val > 0) & (val == ub)).sum())
                pruned += int((ub == 0).sum())
    # the test is only worth something if the cases exercise the bounds: alignments that do trim, bounds that are met
    # with equality, and pairs the bounds rule out
    print("phase B bounds on the host: %d pairs, %d with a trim of their own, %d of them at their bound, %d ruled out" %
          (total, contributing, tight, pruned))
    assert total > 20000 and contributing > 0.1 * total and tight > 0.2 * contributing and pruned > 0.1 * total
