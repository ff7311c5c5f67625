"""The batched GPU phases A/B/C against the reference's sequential per-read logic driven by the
oracle: same matching sets, same trim amounts, same middle hits in the same order."""
import random

import pytest

from tests import ref_pipeline
from tests.golden_io import load_panel
from tests.pairgen import mutate, synthetic_read

pytestmark = pytest.mark.gpu

Y_TOP = "AATGTACTTCGTTCAGTTACGTATTGCT"
Y_BOTTOM = "GCAATACGTAACTGAACGAAGT"


def panel_sets(pl):
    from porechop_amd.pipeline import AdapterSet
    return [AdapterSet(a["name"], tuple(a["start"]) if a["start"] else None, tuple(a["end"]) if a["end"] else None)
            for a in load_panel()]


def make_reads(rng, n):
    reads = []
    for i in range(n):
        ln = rng.choice([60, 140, 400, 1500, 3000, 8000])
        chim = None
        if i % 7 == 0 and ln >= 1500:
            chim = Y_BOTTOM + Y_TOP
        r = synthetic_read(rng, ln, Y_TOP if rng.random() < 0.8 else None, Y_BOTTOM if rng.random() < 0.6 else None, chim)
        if i % 11 == 0 and ln >= 3000:       # two more middle copies -> several mask rounds
            p1, p2 = rng.randint(200, ln // 2), rng.randint(ln // 2, ln - 200)
            r = r[:p1] + mutate(rng, Y_TOP, 0.04) + r[p1:p2] + mutate(rng, Y_BOTTOM, 0.0) + r[p2:]
        if i % 13 == 0:
            r = r.lower()
        reads.append(r)
    return reads


def test_phases_match_sequential_reference_logic(oracle):
    import torch
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import reads_from_strings

    rng = random.Random(2024)
    raw = make_reads(rng, 96)
    p = ScanParams()
    pl = Pipeline(panel_sets(None), p)
    dreads, norm = reads_from_strings(raw)

    # phase A over all reads (check_reads > n)
    bs, be = pl.phase_a(dreads)
    matching = pl.matching_sets(bs, be)
    want_bs, want_be = ref_pipeline.phase_a(oracle.adapter_alignment, norm, pl.sets, p)
    assert [round(x, 6) for x in bs.cpu().tolist()] == [round(x, 6) for x in want_bs]
    assert [round(x, 6) for x in be.cpu().tolist()] == [round(x, 6) for x in want_be]
    want_matching = [i for i, s in enumerate(pl.sets) if "(full sequence)" not in s.name
                     and max(want_bs[i], want_be[i]) >= p.adapter_threshold]
    assert matching == want_matching
    assert [pl.sets[i].name for i in matching] == ["SQK-NSK007"]

    # phase B
    st, et = pl.phase_b(dreads, matching)
    st, et = st.cpu().tolist(), et.cpu().tolist()
    for r, seq in enumerate(norm):
        assert (st[r], et[r]) == ref_pipeline.phase_b(oracle.adapter_alignment, seq, pl.sets, matching, p), r

    # phase C
    hits = pl.phase_c(dreads, torch.tensor(st, dtype=torch.int32, device="cuda"),
                      torch.tensor(et, dtype=torch.int32, device="cuda"), matching)
    pl.aligner.sync()
    got = {}
    for r, a, s, e, idn in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(),
                               hits.end.cpu().tolist(), hits.identity.cpu().tolist()):
        got.setdefault(r, []).append((a, s, e, round(idn, 6)))
    n_hits = 0
    calls = [0]

    def counting(*a):
        calls[0] += 1
        return oracle.adapter_alignment(*a)

    for r, seq in enumerate(norm):
        want = ref_pipeline.phase_c(counting if ref_pipeline.trimmed(seq, st[r], et[r]) else oracle.adapter_alignment,
                                    seq, st[r], et[r], pl.middle_adapters, p)
        want = [(a, s, e, round(f, 6)) for a, s, e, f in want]
        # the reference visits adapters in order and hits of one adapter in discovery order
        assert got.get(r, []) == want, (r, got.get(r), want)
        n_hits += len(want)
    assert n_hits >= 10 and hits.rounds >= 2
    # the alignments consumed are exactly the reference's sequence of calls (the rest were speculative)
    assert hits.alignments == calls[0]
    pl.close()


def test_pruned_phase_a_on_gpu():
    """SURVEY.md 8f-4 on the device: score-only pass (PC_MODE_SCORE) + traceback of the candidates
    only; same matching sets and best identities as the full search, 10 000 check reads."""
    import time
    import torch
    from porechop_amd.panel import load_panel
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_reads
    from tests.test_phase_a_pruning import check_pruned_equals_full
    pl = Pipeline(load_panel(), ScanParams())
    reads = make_reads(20_000, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
    check = torch.arange(10_000, device="cuda")
    names = [pl.sets[i].name for i in check_pruned_equals_full(pl, reads, check)]
    assert "SQK-NSK007" in names
    assert pl.stats["pairs_end_traced_after_pruning"] < 0.05 * pl.stats["pairs_end"]
    for prune in (False, True):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            pl.phase_a(reads, check, prune=prune)
        torch.cuda.synchronize()
        print("phase A, prune=%s: %.2f ms" % (prune, (time.perf_counter() - t) / 3 * 1e3))
    pl.close()


def test_proven_middle_scan_on_gpu():
    """phase_c(prove=True) on the device, 200 k x 8 kb reads with 1 % chimeras: identical hits."""
    import torch
    from porechop_amd.panel import load_panel
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_reads
    p = ScanParams()
    pl = Pipeline(load_panel(), p)
    reads = make_reads(200_000, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
    bs, be = pl.phase_a(reads, torch.arange(10_000, device="cuda"))
    matching = pl.matching_sets(bs, be)
    st, et = pl.phase_b(reads, matching)
    h0 = pl.phase_c(reads, st, et, matching)
    h1 = pl.phase_c(reads, st, et, matching, prove=True)
    pl.aligner.sync()
    assert h0.read.numel() > 1000
    for f in ("read", "adapter", "start", "end", "identity"):
        assert torch.equal(getattr(h0, f), getattr(h1, f)), f
    assert (h0.rounds, h0.alignments) == (h1.rounds, h1.alignments)
    assert pl.stats["pairs_middle_traced_after_proof"] < 0.05 * 200_000 * len(pl.middle_adapters)
    pl.close()


def test_ragged_lengths_and_demux_reduce_vs_reference_logic(oracle):
    """A realistic length distribution (log-normal, 20 bp .. several kb, reads shorter than the end
    windows included): whole-read scans run length-sorted and un-permuted (Pipeline._scan_jobs),
    phase B is reduced by the library kernel (pc_phase_b_reduce) -- trims, and barcode calls under both
    calling rules -- and everything must equal the reference's sequential per-read logic on the oracle."""
    import torch
    from porechop_amd.pipeline import DeviceReads, Pipeline, ScanParams
    from porechop_amd.runner import barcode_bins
    from porechop_amd import panel as rules
    from porechop_amd.synth import make_ragged_reads

    p = ScanParams()
    pl = Pipeline(panel_sets(None), p)
    assert pl.native_reduce
    reads = make_ragged_reads(640, mean_len=1500, sigma=0.8, min_len=20, seed=12, start_frac=0.8, end_frac=0.6, chimera_frac=0.1,
                              pool=128)
    n = reads.n
    assert int(reads.length.min()) < 150 and int(reads.length.max()) > 4000
    host = reads.arena.cpu().numpy().tobytes().decode()
    offs, lens = reads.off.cpu().tolist(), reads.length.cpu().tolist()
    seqs = [host[o:o + l] for o, l in zip(offs, lens)]
    bs, be = pl.phase_a(reads)
    matching = pl.matching_sets(bs, be)
    assert "SQK-NSK007" in [pl.sets[i].name for i in matching]
    st, et = pl.phase_b(reads, matching)
    hits = pl.phase_c(reads, st, et, matching)
    pl.aligner.sync()
    stl, etl = st.cpu().tolist(), et.cpu().tolist()
    got = {}
    for r, a, s, e in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(), hits.end.cpu().tolist()):
        got.setdefault(r, []).append((a, s, e))
    nh = 0
    for r, seq in enumerate(seqs):
        assert (stl[r], etl[r]) == ref_pipeline.phase_b(oracle.adapter_alignment, seq, pl.sets, matching, p), r
        want = [(a, s, e) for a, s, e, _ in ref_pipeline.phase_c(oracle.adapter_alignment, seq, stl[r], etl[r], pl.middle_adapters, p)]
        assert got.get(r, []) == want, r
        nh += len(want)
    assert nh >= 20

    # demultiplexing reduce: plant forward barcodes on a subset, call with both rule sets
    panel = load_panel()
    fw = [i for i, s in enumerate(pl.sets) if s.name.startswith("Barcode ") and "(forward)" in s.name][:12]
    rng = random.Random(3)
    seqs2 = []
    for i in range(320):
        body = "".join(rng.choice("ACGT") for _ in range(rng.choice([30, 100, 400, 900])))
        k = rng.randrange(len(fw))
        k2 = k if rng.random() < 0.7 else rng.randrange(len(fw))
        s5 = mutate(rng, pl.sets[fw[k]].start[1], rng.choice([0.0, 0.1, 0.2])) if rng.random() < 0.8 else ""
        s3 = mutate(rng, pl.sets[fw[k2]].end[1], rng.choice([0.0, 0.1, 0.2])) if rng.random() < 0.8 else ""
        seqs2.append(s5 + body + s3)
    from porechop_amd.synth import reads_from_strings
    dreads, norm = reads_from_strings(seqs2)
    names, bins = barcode_bins(pl, fw)
    for two in (False, True):
        st2, et2, calls = pl.phase_b_demux(dreads, fw, bins, 75.0, 5.0, two)
        pl.aligner.sync()
        s2l, e2l = st2.cpu().tolist(), et2.cpu().tolist()
        n_called = 0
        for r, seq in enumerate(norm):
            ws, we, ss, es = ref_pipeline.phase_b_barcodes(oracle.adapter_alignment, seq, pl.sets, fw, p, "forward")
            want = ref_pipeline.determine_barcode(ss, es, 75.0, 5.0, two)
            g = names[calls[r]] if calls[r] >= 0 else "none"
            assert (s2l[r], e2l[r], g) == (ws, we, want), (two, r)
            n_called += g != "none"
        assert n_called >= 40
    pl.close()


def test_copy_windows_packs_and_pads():
    """pc_copy_windows: windows of any length and alignment copied back to back, the gap up to the next
    copy's start filled with the pad byte (the maskable copies of reads with middle hits)."""
    import torch
    from porechop_amd.batch import Aligner
    al = Aligner(["ACGT"])
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    arena = torch.randint(65, 91, (200_000,), dtype=torch.uint8, device="cuda", generator=g)
    n = 500
    ln = torch.randint(1, 3000, (n,), device="cuda", generator=g).to(torch.int32)
    ln[7], ln[8] = 1, 70_000                       # one byte; longer than a workgroup's stride many times over
    off = torch.randint(0, 200_000 - 70_000, (n,), device="cuda", generator=g).to(torch.int64)
    stride = (ln.to(torch.int64) + 23) // 16 * 16
    ends = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    ends[1:] = torch.cumsum(stride, 0)
    dst = torch.full((int(ends[-1]) + 64,), 7, dtype=torch.uint8, device="cuda")
    al.copy_windows(arena, off, ln, dst, ends, ord("N"))
    al.sync()
    a, d = arena.cpu().numpy(), dst.cpu().numpy()
    for i, (o, l, s, e) in enumerate(zip(off.tolist(), ln.tolist(), ends[:-1].tolist(), ends[1:].tolist())):
        assert (d[s:s + l] == a[o:o + l]).all(), i
        assert (d[s + l:e] == ord("N")).all(), i
    assert (d[int(ends[-1]):] == 7).all()          # nothing written past the last copy
    al.close()
