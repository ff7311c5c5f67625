"""The exact bit-parallel prefilter (pc_prefilter_device) on the MI355X, through the C ABI, against the oracle's
plain dynamic programme of the same contract (oracle/pc_oracle.c pc_oracle_min_edits), and the prefiltered
middle scan against the full one."""
import random

import numpy as np
import pytest

from tests.pairgen import mutate

pytestmark = pytest.mark.gpu


def plant(rng, read, adapter, edits_rate):
    """A (possibly truncated) mutated copy of the adapter somewhere in the read, ends included."""
    mut = mutate(rng, adapter, edits_rate)
    where = rng.random()
    if where < 0.15:                                   # overhanging the read's start
        k = rng.randint(0, max(0, len(mut) // 3))
        return mut[k:] + read[len(mut) - k:]
    if where < 0.30:                                   # overhanging its end
        k = rng.randint(0, max(0, len(mut) // 3))
        return read[:max(0, len(read) - (len(mut) - k))] + mut[:len(mut) - k]
    pos = rng.randint(0, max(0, len(read) - 1))
    return (read[:pos] + mut + read[pos + len(mut):])[:max(len(read), 1)]


def make_cases(seed, n, lengths, adapters, alphabet="ACGT"):
    rng = random.Random(seed)
    reads = []
    for i in range(n):
        ln = rng.choice(lengths)
        r = "".join(rng.choice(alphabet) for _ in range(ln))
        if ln and rng.random() < 0.7:
            r = plant(rng, r, rng.choice(adapters), rng.choice([0.0, 0.04, 0.08, 0.12, 0.2]))
        if ln > 40 and rng.random() < 0.1:
            p = rng.randint(0, ln - 20)
            r = r[:p] + "-" * rng.randint(1, 19) + r[p + 19:]      # a masked stretch, as phase C makes them
            r = r[:ln]
        reads.append(r)
    return reads


def run_and_check(oracle, reads, adapters, edits, scores=(3, -6, -5, -2), hint=0):
    import porechop_amd
    from porechop_amd.synth import reads_from_strings
    dr, norm = reads_from_strings(reads)
    al = porechop_amd.Aligner(adapters, scores)
    al.set_length_hint(hint)
    got = al.prefilter(dr.arena, dr.off, dr.length, max(1, int(dr.length.max())), list(range(len(adapters))), edits).cpu().numpy()
    al.sync()
    al.close()
    arena = dr.arena.cpu().numpy()
    offs, lens = dr.off.cpu().numpy(), dr.length.cpu().numpy()
    exact = sound = 0
    for j, (ad, k) in enumerate(zip(adapters, edits)):
        d = oracle.min_edits_many(arena, offs, lens, ad)
        want = (d <= (k if k >= 0 else 1 << 30)) & (lens > 0)
        missed = want & ~got[j]
        assert not missed.any(), ("prefilter dropped a pair within the bound", ad, k, np.nonzero(missed)[0][:5], d[missed][:5])
        sound += int(want.sum())
        if len(ad) <= 32:
            wrong = got[j] & ~want
            assert not wrong.any(), ("prefilter kept a pair beyond the bound", ad, k, np.nonzero(wrong)[0][:5], d[wrong][:5])
            exact += int(lens.shape[0])
    return exact, sound


def test_prefilter_equals_the_plain_dp_for_adapters_up_to_32_bases(oracle):
    rng = random.Random(5)
    adapters = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "A", "ACG", "ACGTACGTAC",
                "".join(rng.choice("ACGT") for _ in range(24)), "".join(rng.choice("ACGT") for _ in range(31)),
                "".join(rng.choice("ACGT") for _ in range(32)), "ACGTNNACGTTTGACCAGTNAC", "".join(rng.choice("ACGT") for _ in range(17))]
    reads = make_cases(11, 3000, [0, 1, 2, 7, 15, 16, 17, 31, 33, 100, 150, 151, 600, 1000, 2500], adapters) + \
        make_cases(12, 300, [100, 150, 400], adapters, alphabet="ACGTN") + \
        make_cases(13, 200, [64, 300], adapters, alphabet="acgtuUXN-")
    for thr in (90.0, 80.0, 70.0):
        edits = [max(0, int(len(a) * (100 - thr) / thr)) for a in adapters]
        exact, sound = run_and_check(oracle, reads, adapters, edits)
        assert exact == len(reads) * len(adapters) and sound > 1000
    # edge bounds: 0 edits, a bound as large as the adapter (everything passes), "do not filter"
    run_and_check(oracle, reads[:600], adapters, [0] * len(adapters))
    run_and_check(oracle, reads[:600], adapters, [len(a) for a in adapters])
    run_and_check(oracle, reads[:600], adapters, [-1] * len(adapters))


def test_prefilter_chunked_long_reads_and_ragged_lengths(oracle):
    """Few long reads: every read is cut into 512-column chunks (occurrences across chunk boundaries), lengths
    differ inside a wave (the masked tail path), with and without the length hint."""
    adapters = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "CAGCACCTGGTTAACCTTAGCAAT"]
    rng = random.Random(3)
    reads = []
    for i in range(220):
        ln = rng.choice([511, 512, 513, 1023, 1030, 3000, 8000, 8000, 20000])
        r = "".join(rng.choice("ACGT") for _ in range(ln))
        for _ in range(rng.randint(0, 3)):
            ad = rng.choice(adapters)
            mut = mutate(rng, ad, rng.choice([0.0, 0.05, 0.1]))
            # aim at chunk boundaries (multiples of 512 and of the uniform chunk lengths the library may choose)
            pos = max(0, min(ln - 1, rng.choice([512, 1024, 1536, 2000, 2048, 4000, 4096]) * rng.randint(0, 4) + rng.randint(-30, 10)))
            r = (r[:pos] + mut + r[pos + len(mut):])[:ln]
        reads.append(r)
    for hint in (0, 3000):
        exact, sound = run_and_check(oracle, reads, adapters, [3, 2, 2], hint=hint)
        assert sound > 100


def test_prefilter_is_sound_for_long_adapters_and_many_groups(oracle):
    """Adapters above 32 bases are cut into pieces (pigeonhole: a superset); a 40-adapter list runs as five groups of
    eight pieces per lane."""
    from tests.golden_io import load_panel
    panel = load_panel()
    adapters = [a["start"][1] for a in panel if a["name"].startswith("Barcode ") and "(forward)" in a["name"]][:34]
    adapters += ["AATGTACTTCGTTCAGTTACGTATTGCTAAGGTTAA" + adapters[0] + "CAGCACCT",
                 "AATGTACTTCGTTCAGTTACGGCTTGGGTGTTTAACC" + adapters[1] + "GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA",
                 "ACTTGCCTGTCGCTCTATCTTCTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT"[:50], "GGTTGTTTCTGTTGGTGCTGATATTGCTGGG",
                 "AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"]
    reads = make_cases(21, 1500, [150, 150, 400, 1200, 3000], adapters)
    edits = [max(0, int(len(a) * 10 / 90)) for a in adapters]
    exact, sound = run_and_check(oracle, reads, adapters, edits)
    assert exact == len(reads) * sum(1 for a in adapters if len(a) <= 32) and sound > 500


def test_prefiltered_middle_scan_equals_full_scan_on_gpu():
    """phase_c(prefilter=True) on the device, 200 k x 8 kb reads with 1 % chimeras: identical hits, rounds and
    alignment counts; about 1 % of the pairs reach the DP."""
    import torch
    from porechop_amd.panel import load_panel
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_reads, make_ragged_reads
    p = ScanParams()
    pl = Pipeline(load_panel(), p)
    for ragged in (False, True):
        if ragged:
            reads = make_ragged_reads(60_000, mean_len=6000, sigma=0.6, min_len=20, seed=5, chimera_frac=0.02)
        else:
            reads = make_reads(200_000, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
        bs, be = pl.phase_a(reads, torch.arange(10_000, device="cuda"))
        matching = pl.matching_sets(bs, be)
        st, et = pl.phase_b(reads, matching)
        pl.stats["pairs_middle_prefiltered"] = pl.stats["pairs_middle_scanned_after_prefilter"] = 0
        h0 = pl.phase_c(reads, st, et, matching)
        h1 = pl.phase_c(reads, st, et, matching, prefilter=True)
        pl.aligner.sync()
        assert h0.read.numel() > 500
        for f in ("read", "adapter", "start", "end", "identity"):
            assert torch.equal(getattr(h0, f), getattr(h1, f)), (ragged, f)
        assert (h0.rounds, h0.alignments) == (h1.rounds, h1.alignments)
        assert pl.stats["pairs_middle_scanned_after_prefilter"] < 0.06 * pl.stats["pairs_middle_prefiltered"]
    pl.close()


def test_prefiltered_middle_scan_with_a_barcode_panel_vs_reference_logic(oracle):
    """Reads carrying native barcodes and chimeric junctions, ~30 matching sets: phase C with the prefilter against
    the reference's sequential per-read logic driven by the oracle (tests/ref_pipeline.py)."""
    import torch
    from porechop_amd.panel import load_panel
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import reads_from_strings
    from tests import readgen, ref_pipeline
    panel = load_panel()
    rr = readgen.native_reads(7, 150) + readgen.ligation_reads(9, 100)
    seqs = [r[1] for r in rr]
    p = ScanParams()
    pl = Pipeline(panel, p)
    reads, norm = reads_from_strings(seqs)
    matching = [i for i, s in enumerate(panel) if s.name == "SQK-NSK007" or (s.name.startswith("Barcode ") and "(reverse)" in s.name)]
    st, et = pl.phase_b(reads, matching)
    hits = pl.phase_c(reads, st, et, matching, prefilter=True)
    pl.aligner.sync()
    got = {}
    for r, a, s, e in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(), hits.end.cpu().tolist()):
        got.setdefault(r, []).append((a, s, e))
    stl, etl = st.cpu().tolist(), et.cpu().tolist()
    n_hits = 0
    for r, seq in enumerate(norm):
        want = [(a, s, e) for a, s, e, _ in ref_pipeline.phase_c(oracle.adapter_alignment, seq, stl[r], etl[r], pl.middle_adapters, p)]
        assert got.get(r, []) == want, (r, got.get(r), want)
        n_hits += len(want)
    assert n_hits > 20
    pl.close()


_CHILD = r"""
import random, sys
sys.path.insert(0, %r)
from oracle.oracle import Oracle
from tests.test_gpu_prefilter import make_cases, run_and_check
rng = random.Random(5)
adapters = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "ACGTACGTAC", "".join(rng.choice("ACGT") for _ in range(24)),
            "".join(rng.choice("ACGT") for _ in range(32)), "ACGTNNACGTTTGACCAGTNAC", "AAAAAAAAAAAAAAAAAAAAAAAA"]
reads = make_cases(31, 1500, [0, 5, 17, 150, 151, 600, 2500], adapters) + ["A" * 3000, "ACGT" * 500]
edits = [3, 2, 1, 2, 3, 2, 2]
exact, sound = run_and_check(Oracle(), reads, adapters, edits)
print("CHILD_OK", exact, sound)
"""


@pytest.mark.parametrize("env", [{}, {"PC_PF_NO_SEEDS": "1"}, {"PC_PF_SEED_CAP": "64"}, {"PC_PF_MULTI_Q": "1"}, {"PC_PF_SINGLE_Q": "1"}])
def test_seed_stage_exhaustive_kernel_and_overflow_fallback_agree_with_the_plain_dp(env):
    """The same batch -- seeds of two lengths, an adapter without seeds (N inside), a low-complexity adapter against
    poly-A reads -- through the seed stage (one seed length for all pieces or a bitmap per length, as the expected candidate rate decides; PC_PF_SINGLE_Q=1 / PC_PF_MULTI_Q=1 force either), through
    the exhaustive kernel alone (PC_PF_NO_SEEDS=1) and through the overflow fallback (a 64-entry candidate list): each
    equals the oracle's plain DP."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CHILD % repo], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600, cwd=repo)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    if "PC_PF_SEED_CAP" in env:
        assert "filtered by the exhaustive kernel" in r.stderr
