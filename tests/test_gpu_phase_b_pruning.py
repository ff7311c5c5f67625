"""Exact pruning of phase B (Pipeline._phase_b_pruned_records): the score pass's end cell bounds what an end-window
alignment can contribute to a read's trims and barcode call; only pairs whose bound can matter are traced.
Checked on the device for EVERY pair of every batch: the bounds against the full traced records, and the pruned phase B's
trims and calls against the unpruned one's."""
import random

import numpy as np
import pytest
import torch

from tests.golden_io import load_panel

pytestmark = pytest.mark.gpu


def panel_sets():
    from porechop_amd.pipeline import AdapterSet
    return [AdapterSet(a["name"], tuple(a["start"]) if a["start"] else None, tuple(a["end"]) if a["end"] else None)
            for a in load_panel()]


def check_bounds_and_results(pl, reads, matching, bins, opts):
    from porechop_amd.batch import MODE_SCORE, MODE_TRACE
    from porechop_amd.pipeline import _identities
    p = pl.p
    jobs, where = pl._phase_b_jobs(reads, matching)
    so, sl = pl._end_windows(reads, None, "start")
    eo, el = pl._end_windows(reads, None, "end")
    full_rec = torch.stack(pl._scan_jobs(reads.arena, jobs, MODE_TRACE, p.end_size))
    pl.aligner.sync()
    # the pruned run, with the device's bounds and its untouched score records handed out
    pl.stats["pairs_end"] = 0
    pl.stats["pairs_end_traced_after_pruning"] = 0
    pl.debug_bounds = {}
    if bins is not None:
        b = pl.phase_b_demux(reads, matching, bins, opts.barcode_threshold, opts.barcode_diff, opts.require_two_barcodes, prune=True)
    else:
        b = pl.phase_b(reads, matching, prune=True)
    pl.aligner.sync()
    probe, pl.debug_bounds = pl.debug_bounds, None
    score = torch.stack([probe["score_records"][o:o + reads.n] for o in probe["rec_off"]])
    ub, ub_full = probe["ub_trim"].to(torch.int64), probe["ub_full"]
    # the same bounds in torch (Pipeline._phase_b_bounds: the readable statement of them)
    ub_t, ub_full_t = pl._phase_b_bounds(score, jobs, where, sl, el)
    assert torch.equal(ub, ub_t), "device and torch bounds differ"
    assert float((ub_full - ub_full_t).abs().max()) < 1e-9
    full, partial = _identities(full_rec)
    ok = full_rec[..., 0] != -1
    rs = full_rec[..., 0].to(torch.int64)
    re = full_rec[..., 1].to(torch.int64) + 1
    is_end = torch.tensor([w[0] for w in where], dtype=torch.bool, device=ub.device)[:, None]
    good = ok & (partial > p.end_threshold) & ((re - rs) >= p.min_trim_size)
    val_s = torch.where(good & (re != p.end_size), re + p.extra_end_trim, torch.zeros_like(re))
    val_e = torch.where(good & (rs != 0), p.end_size - rs + p.extra_end_trim, torch.zeros_like(re))
    val = torch.where(is_end, val_e, val_s)
    bad = torch.nonzero(val > ub)
    assert bad.shape[0] == 0, ("a trim above its bound", bad[:5].tolist(), val[bad[:5, 0], bad[:5, 1]].tolist(), ub[bad[:5, 0], bad[:5, 1]].tolist(),
                               score[bad[:5, 0], bad[:5, 1]].tolist(), full_rec[bad[:5, 0], bad[:5, 1]].tolist())
    fullv = torch.where(ok, torch.nan_to_num(full, nan=0.0), torch.zeros_like(full))
    badf = torch.nonzero(fullv > ub_full + 1e-6)          # (the identity is rounded to six decimals, possibly upwards)
    assert badf.shape[0] == 0, ("a full identity above its bound", int(badf.shape[0]),
                                [(bj, br, float(fullv[bj, br]), float(ub_full[bj, br]), score[bj, br].tolist(), full_rec[bj, br].tolist())
                                 for bj, br in badf[:4].tolist()])
    # results
    if bins is not None:
        a = pl.phase_b_demux(reads, matching, bins, opts.barcode_threshold, opts.barcode_diff, opts.require_two_barcodes)
        assert np.array_equal(a[2], b[2])
    else:
        a = pl.phase_b(reads, matching)
    pl.aligner.sync()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    print("phase B pruning: %d pairs, %d traced (%.1f %%), %d with a trim of their own" % (
        int(ub.numel()), pl.stats["pairs_end_traced_after_pruning"], 100.0 * pl.stats["pairs_end_traced_after_pruning"] / max(1, int(ub.numel())),
        int((val > 0).sum())))
    return int(ub.numel()), pl.stats["pairs_end_traced_after_pruning"], int((val > 0).sum())


def test_bounds_hold_for_every_pair_and_results_are_identical():
    from porechop_amd import panel as rules
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.runner import Options, barcode_bins
    from porechop_amd.synth import make_reads, make_ragged_reads, reads_from_strings
    from tests import readgen
    from tests.pairgen import mutate
    opts = Options()
    fw = [a for a in load_panel() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
    for scores in ((3, -6, -5, -2), (2, -3, -5, -2), (5, -4, -10, -1), (3, -6, -2, -5)):
        p = ScanParams(scores=scores)
        pl = Pipeline(panel_sets(), p)
        # (1) the configs[2]/[4] shape: barcoded 8-kb reads, ~98 matching sets, barcode calls
        reads = make_reads(6000, 8000, seed=12, start_frac=0.9, end_frac=0.5, chimera_frac=0.0,
                           barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
        matching = [i for i, s in enumerate(pl.sets) if s.name == "SQK-NSK007" or (s.name.startswith("Barcode ") and "(forward)" in s.name)]
        bc = [i for i in matching if rules.is_barcode(pl.sets[i])]
        names, bins = barcode_bins(pl, bc)
        n, traced, mattering = check_bounds_and_results(pl, reads, matching, bins, opts)
        if scores == (3, -6, -5, -2):
            assert traced < 0.25 * n, (traced, n)
        # (2) short and ragged reads (windows shorter than end_size, reads shorter than the adapters), no barcodes
        reads = make_ragged_reads(20000, mean_len=160, sigma=0.9, min_len=1, seed=5)
        matching = [i for i, s in enumerate(pl.sets) if s.name in ("SQK-NSK007", "Rapid", "SQK-MAP006", "PCR adapters 1", "cDNA SSP")]
        check_bounds_and_results(pl, reads, matching, None, opts)
        # (3) hand-made hard cases: adapter copies at every offset of both windows, overlapping the inner edge,
        # truncated, with N's; reads of 1..200 bases
        rng = random.Random(7)
        y_top, y_bot = "AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"
        seqs = []
        for k in range(4000):
            ln = rng.choice([1, 3, 10, 30, 100, 149, 150, 151, 152, 200, 400])
            r = "".join(rng.choice("ACGT" if k % 9 else "ACGTN") for _ in range(ln))
            ad = mutate(rng, rng.choice([y_top, y_bot, fw[k % 96]["start"][1], fw[k % 96]["end"][1]]), rng.choice([0.0, 0.05, 0.15]))
            if rng.random() < 0.5:
                ad = ad[rng.randint(0, len(ad) - 1):] if rng.random() < 0.5 else ad[:rng.randint(1, len(ad))]
            pos = rng.randint(0, max(0, ln - 1))
            r = (r[:pos] + ad + r[pos + len(ad):])[:max(ln, 1)] if rng.random() < 0.8 else r
            seqs.append(r)
        reads, _ = reads_from_strings(seqs)
        matching = [i for i, s in enumerate(pl.sets) if s.name == "SQK-NSK007" or (s.name.startswith("Barcode ") and "(forward)" in s.name)][:40]
        bc = [i for i in matching if rules.is_barcode(pl.sets[i])]
        names, bins = barcode_bins(pl, bc)
        check_bounds_and_results(pl, reads, matching, bins, opts)
        pl.close()
