"""PC_MODE_TRACE_AT (include/porechop_amd.h): the traced record of pairs whose end cell is known from their PC_MODE_SCORE
record -- only the columns the path can occupy are traced.  Must equal PC_MODE_TRACE's record (and the oracle's) for every
pair: end windows of 1..150 bases in ragged tiles, the first window of the arena (no room for a lead-in), windows longer than
the bound's window, several row classes and scoring schemes (packed-fp16 and packed-int16 traced kernels)."""
import random

import numpy as np
import pytest
import torch

from tests.pairgen import mutate

pytestmark = pytest.mark.gpu


def _batch(rng, ads, n, lens):
    reads, aidx = [], []
    for i in range(n):
        L = rng.choice(lens)
        a = rng.randrange(len(ads))
        body = [rng.choice("ACGT") for _ in range(L)]
        if rng.random() < 0.8 and L >= 4:
            cp = mutate(rng, ads[a], rng.choice([0.0, 0.05, 0.15, 0.3]))
            if rng.random() < 0.3 and cp:
                k = rng.randint(0, len(cp) - 1)
                cp = cp[k:] if rng.random() < 0.5 else cp[:len(cp) - k]
            pos = rng.choice([0, 0, max(0, L - len(cp)), rng.randint(0, L - 1)])
            for k, ch in enumerate(cp[:L - pos]):
                body[pos + k] = ch
        if rng.random() < 0.05:
            body[rng.randrange(L)] = "N"
        reads.append("".join(body))
        aidx.append(a)
    return reads, aidx


@pytest.mark.parametrize("scores", [(3, -6, -5, -2), (2, -3, -5, -2), (20, -30, -25, -12), (3, -6, -5, -5), (3, -6, -2, -5)])
def test_trace_at_equals_trace_and_the_oracle(oracle, scores):
    import porechop_amd
    from porechop_amd.batch import MODE_SCORE, MODE_TRACE, MODE_TRACE_AT
    rng = random.Random(hash(scores) & 0xFFFF)
    dev = torch.device("cuda")
    ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "AAGAAAGTTGTCGGTGTCTTTGTG", "CTTCGTTCAGTTACGTATTGCTGGCGTCTGCTT",
           "ACGTAC", "GGTTGTTTCTGTTGGTGCTGATATTGCTGGCGTCTGCTTGGGTGTTTAACC"]
    al = porechop_amd.Aligner(ads, scores)
    try:
        for max_len, lens in ((150, [1, 2, 7, 40, 90, 149, 150, 150, 150]), (150, [150]), (640, [30, 200, 333, 640])):
            reads, aidx = _batch(rng, ads, 3000, lens)
            order = sorted(range(len(reads)), key=lambda i: aidx[i])           # one job per adapter
            reads = [reads[i] for i in order]
            aidx = [aidx[i] for i in order]
            text = "".join(reads).encode()
            arena = torch.from_numpy(np.concatenate([np.frombuffer(text, dtype=np.uint8), np.full(64, ord("N"), np.uint8)])).to(dev)
            ln = np.array([len(r) for r in reads], dtype=np.int32)
            off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
            d_off, d_len = torch.from_numpy(off).to(dev), torch.from_numpy(ln).to(dev)
            jobs = sorted(set(aidx))
            starts = np.array([aidx.index(a) for a in jobs] + [len(aidx)], dtype=np.int64)
            n = len(reads)
            out_t = torch.empty((n, 8), dtype=torch.int32, device=dev)
            out_s = torch.empty((n, 8), dtype=torch.int32, device=dev)
            al.scan_device(arena, d_off, d_len, np.array(jobs, dtype=np.int32), starts, max_len, out_t, MODE_TRACE)
            al.scan_device(arena, d_off, d_len, np.array(jobs, dtype=np.int32), starts, max_len, out_s, MODE_SCORE)
            al.sync()
            assert bool((out_s[:, 0] == -2).all())
            out_a = out_s.clone()
            al.scan_device(arena, d_off, d_len, np.array(jobs, dtype=np.int32), starts, max_len, out_a, MODE_TRACE_AT)
            al.sync()
            t, a_ = out_t.cpu().numpy(), out_a.cpu().numpy()
            differing = np.nonzero((t != a_).any(axis=1))[0]
            assert differing.size == 0, (scores, max_len, differing[:5], t[differing[:3]], a_[differing[:3]])
            for i in rng.sample(range(n), 300):
                assert porechop_amd.format_result(a_[i]) == oracle.adapter_alignment(reads[i], ads[aidx[i]], scores), (reads[i], ads[aidx[i]])
        # a record that is not a score record of its window must be refused loudly, never guessed
        bad = out_t.clone()
        al.scan_device(arena, d_off, d_len, np.array(jobs, dtype=np.int32), starts, max_len, bad, MODE_TRACE_AT)
        with pytest.raises(RuntimeError):
            al.sync()
    finally:
        al.close()
