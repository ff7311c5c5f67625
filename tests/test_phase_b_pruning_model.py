"""The two-round scheme behind the exact pruning of phase B, replayed on the host.

porechop_amd/csrc/pc_select.hip decides on the device which end-window alignments of a barcoded batch are traced; the GPU
test (tests/test_gpu_phase_b_pruning.py) shows that its results equal the unpruned phase B's.  This test states the SCHEME
itself in numpy / torch on the host and checks it against the oracle, without a GPU: with every alignment of a batch known
(the oracle's record, end cell and score), reveal to the reduction only the alignments the two rounds would trace --
  round 1: per read and side, the two best-scoring pairs whose trim bound is positive and the two best-scoring barcode pairs;
  round 2: pairs whose trim bound exceeds the trim so far, barcode pairs that could come within --barcode_diff of
           max(best revealed, --barcode_threshold)
-- and everything else as "no alignment": trims (nanopore_read.py:166-208) and barcode calls (nanopore_read.py:399-466)
must come out exactly as with every alignment revealed, whatever the thresholds."""
import random

import numpy as np
import torch

from tests.cpu_aligner import OracleAligner
from tests.pairgen import mutate

END = 150


def make_reads(rng, n, y_top, y_bot, bcs):
    """barcoded reads of every awkward kind: clean, heavily mutated (near-ties between barcodes), different barcodes at
    the two ends, a second barcode deeper in the window, truncated adapters at the very edge, short reads whose two end
    windows overlap, reads without anything"""
    out = []
    for k in range(n):
        kind = k % 8
        ln = rng.choice([60, 140, 151, 300, 600]) if kind == 6 else rng.randint(400, 900)
        body = "".join(rng.choice("ACGT") for _ in range(ln))
        if kind == 7:
            out.append(body)
            continue
        b = rng.randrange(len(bcs))
        b2 = rng.randrange(len(bcs)) if kind in (2, 3) else b
        rate = rng.choice([0.0, 0.05, 0.12]) if kind != 1 else rng.choice([0.2, 0.3, 0.4])
        start = mutate(rng, y_top, rate) + mutate(rng, bcs[b][0], rate)
        end = mutate(rng, bcs[b2][1], rate) + mutate(rng, y_bot, rate)
        if kind == 3:                                      # a second, different barcode a little deeper
            start = start + "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 30))) + mutate(rng, bcs[(b + 1) % len(bcs)][0], 0.05)
        if kind == 4:                                      # truncated at the outer edge
            start = start[rng.randint(1, 25):]
            end = end[:len(end) - rng.randint(1, 20)]
        if kind == 5:                                      # only one end carries anything
            end = ""
        s = (start + body + end)
        out.append(s[:max(1, len(s))])
    return out


def reductions(p, rec, sides, bins, job_of, thr, diff, two):
    """trims and calls from [J, R, 8] records (a record with field 0 == -1 is "no alignment")"""
    from porechop_amd.pipeline import _identities, call_barcodes
    full, partial = _identities(rec)
    ok = rec[..., 0] != -1
    rs = rec[..., 0].to(torch.int64)
    re = rec[..., 1].to(torch.int64) + 1
    good = ok & (partial > p.end_threshold) & ((re - rs) >= p.min_trim_size)
    is_end = torch.tensor(sides, dtype=torch.bool)[:, None]
    val = torch.where(is_end, torch.where(good & (rs != 0), END - rs + p.extra_end_trim, torch.zeros_like(re)),
                      torch.where(good & (re != END), re + p.extra_end_trim, torch.zeros_like(re)))
    R = rec.shape[1]
    zero = torch.zeros((1, R), dtype=torch.int64)
    st = torch.cat([val[~is_end[:, 0]], zero]).amax(dim=0)
    et = torch.cat([val[is_end[:, 0]], zero]).amax(dim=0)
    fulls = torch.where(ok, torch.nan_to_num(full, nan=0.0), torch.zeros_like(full))
    zeros = torch.zeros(R, dtype=torch.float64)
    S = torch.stack([fulls[job_of[(b[0], 0)]] if b[0] is not None else zeros for b in bins], dim=1)
    E = torch.stack([fulls[job_of[(b[1], 1)]] if b[1] is not None else zeros for b in bins], dim=1)
    return st, et, call_barcodes(len(bins), S, E, thr, diff, two), fulls


def test_two_round_selection_reproduces_every_trim_and_call(oracle):
    from porechop_amd import panel as rules
    from porechop_amd.panel import load_panel
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.runner import barcode_bins
    rng = random.Random(77)
    panel = load_panel()
    by_name = {s.name: s for s in panel}
    y = by_name["SQK-NSK007"]
    bc_sets = [by_name["Barcode %d (forward)" % k] for k in range(1, 17)]
    bcs = [(s.start[1], s.end[1]) for s in bc_sets]
    seqs = make_reads(rng, 320, y.start[1], y.end[1], bcs)
    R = len(seqs)
    checked = changed_by_round2 = traced_total = 0
    for scores in ((3, -6, -5, -2), (2, -3, -5, -2)):
        p = ScanParams(scores=scores)
        pl = Pipeline(panel, p, aligner=OracleAligner(oracle, scores))
        matching = [i for i, s in enumerate(pl.sets) if s.name == "SQK-NSK007" or s in bc_sets]
        bc_idx = [i for i in matching if rules.is_barcode(pl.sets[i])]
        names, bins = barcode_bins(pl, bc_idx)
        jobs, where = [], []
        for si in matching:
            s = pl.sets[si]
            jobs.append((pl.seq_index[s.start[1]], None, None)); where.append((0, si))
            jobs.append((pl.seq_index[s.end[1]], None, None)); where.append((1, si))
        J = len(jobs)
        sides = [w[0] for w in where]
        job_of = {(si, side): k for k, (side, si) in enumerate(where)}
        full = torch.zeros((J, R, 8), dtype=torch.int32)
        score = torch.zeros((J, R, 8), dtype=torch.int32)
        wl = torch.tensor([min(len(s), END) for s in seqs], dtype=torch.int32)
        for j, (job, (side, si)) in enumerate(zip(jobs, where)):
            ad = pl.seqs[job[0]]
            for r, s in enumerate(seqs):
                w = s[-END:] if side else s[:END]
                res = oracle.align_raw(w, ad, scores)
                score[j, r] = torch.tensor([-2, res.end_j, res.end_i, 0, res.score, 0, 0, 0], dtype=torch.int32)
                if res.failed:
                    full[j, r, 0] = -1
                else:
                    full[j, r] = torch.tensor([res.read_start, res.read_end, res.adapter_start, res.adapter_end, res.score,
                                               res.aligned_matches, res.aligned_len, res.full_len], dtype=torch.int32)
        ub, ub_full = pl._phase_b_bounds(score, jobs, where, wl, wl)
        S = score[..., 4].to(torch.int64)
        is_end = torch.tensor(sides, dtype=torch.bool)
        calls_j = torch.tensor([w[1] in bc_idx for w in where], dtype=torch.bool)
        mj = torch.tensor([len(pl.seqs[j[0]]) for j in jobs], dtype=torch.float64)[:, None]
        Pc = float(max(-scores[1], -scores[2], -scores[3], 0))
        none = torch.tensor([-1, 0, -1, 0, -2147483648, 0, 0, 0], dtype=torch.int32)

        def top2(rows, eligible):
            pick = torch.zeros((J, R), dtype=torch.bool)
            idx = torch.nonzero(rows).flatten()
            if idx.numel():
                sub = torch.where(eligible[idx] & (S[idx] >= 0), S[idx], torch.full_like(S[idx], -1))
                top = torch.topk(sub, min(2, int(idx.numel())), dim=0)
                sel = torch.zeros_like(sub, dtype=torch.bool)
                sel.scatter_(0, top.indices, top.values >= 0)
                pick[idx] = sel
            return pick

        for thr, diff, two in ((75.0, 5.0, False), (75.0, 5.0, True), (60.0, 1.0, False), (90.0, 10.0, False), (70.0, 0.0, True)):
            want = reductions(p, full, sides, bins, job_of, thr, diff, two)
            # round 1
            need1 = torch.zeros((J, R), dtype=torch.bool)
            for side in (False, True):
                need1 |= top2(is_end == side, ub > 0)
                need1 |= top2((is_end == side) & calls_j, torch.ones((J, R), dtype=torch.bool))
            shown = none.repeat(J, R, 1)
            shown[need1] = full[need1]
            st1, et1, calls1, fulls1 = reductions(p, shown, sides, bins, job_of, thr, diff, two)
            best = torch.stack([torch.cat([fulls1[calls_j & ~is_end], torch.zeros((1, R), dtype=torch.float64)]).amax(dim=0),
                                torch.cat([fulls1[calls_j & is_end], torch.zeros((1, R), dtype=torch.float64)]).amax(dim=0)])
            # round 2
            so_far = torch.where(is_end[:, None], et1[None, :], st1[None, :])
            level = torch.clamp(best, min=thr) - diff
            lvl = torch.where(is_end[:, None], level[1][None, :], level[0][None, :]) - 1e-6
            smin = torch.floor(mj * ((lvl / 100.0) * (scores[0] + Pc) - Pc) - 1e-9)
            need2 = ~need1 & ((ub > so_far) | (calls_j[:, None] & (ub_full >= lvl) & (S.to(torch.float64) >= smin)))
            shown[need2] = full[need2]
            got = reductions(p, shown, sides, bins, job_of, thr, diff, two)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (scores, thr, diff, two, "trims")
            bad = np.nonzero(got[2] != want[2])[0]
            assert bad.size == 0, (scores, thr, diff, two, "calls", bad[:5].tolist(), got[2][bad[:5]].tolist(), want[2][bad[:5]].tolist())
            checked += R
            changed_by_round2 += int((torch.ne(st1, want[0]) | torch.ne(et1, want[1])).sum()) + int((calls1 != want[2]).sum())
            traced_total += int(need1.sum()) + int(need2.sum())
        assert float(need1.sum() + need2.sum()) < 0.6 * J * R
    # the cases must make round 2 matter (else the test would pass with round 1 alone) and prune for real
    print("two-round scheme on the host: %d read evaluations, round 2 changed %d of them; %.1f %% of the pairs revealed" %
          (checked, changed_by_round2, 100.0 * traced_total / (checked * J)))
    assert checked >= 3000 and changed_by_round2 > 0
