"""Reads at 2 bits per base across PCIe, the device half (pc_unpack_device) and the whole path behind it: the unpacked arena
equals the host's canonical bytes (+ the 'N' padding), and phases A + B + C over reads that were uploaded packed give the
results of the same reads uploaded one byte per base -- on reads that hold 'N', '-', lower case and 'U' as well."""
import random

import numpy as np
import pytest
import torch

from porechop_amd.io import pack_reads, unpack_reads_host

pytestmark = pytest.mark.gpu
ALPHABET = b"ACGTacgtUuNn-RYKMSWXZ*"


@pytest.fixture(scope="module")
def aligner():
    import porechop_amd
    al = porechop_amd.Aligner(["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"])
    yield al
    al.close()


def test_unpack_device_equals_the_host_inverse(aligner):
    rng = np.random.default_rng(3)
    dev = torch.device("cuda")
    for n in [0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1000, 4097, 1 << 20, (1 << 24) + 5]:
        for p_exc in (0.0, 0.02, 0.5):
            w = np.array([1 - p_exc] * 4 + [p_exc * 4 / (len(ALPHABET) - 4)] * (len(ALPHABET) - 4))
            arr = np.frombuffer(ALPHABET, dtype=np.uint8)[rng.choice(len(ALPHABET), size=n, p=w / w.sum())]
            pk, exc = pack_reads(arr)
            for pad in (0, 16, 64):
                arena = torch.full((n + pad + 32,), 0xEE, dtype=torch.uint8, device=dev)
                out = aligner.unpack_device(torch.from_numpy(pk).to(dev), n, torch.from_numpy(exc).to(dev) if exc.size else None,
                                            arena=arena, pad=pad)
                aligner.sync()
                got = out.cpu().numpy()
                assert np.array_equal(got[:n], unpack_reads_host(pk, n, exc)), (n, p_exc)
                assert np.all(got[n:n + pad] == ord("N")) and np.all(got[n + pad:] == 0xEE), (n, pad)


def test_known_answers_through_the_packed_route(aligner, oracle):
    """SURVEY.md 8a's known-answer reads with non-ACGT letters: packed on the host, unpacked on the device, aligned there;
    the strings equal the oracle's on the ORIGINAL reads."""
    import porechop_amd
    cases = [("NNNNNNNN", "ACGT"), ("ACNNGT", "ACNNGT"), ("AC--GT", "ACGT"), ("ACXXGT", "ACNNGT"), ("ttttacgttttt", "ACGT"),
             ("TTTTACGUUUUU", "ACGT"), ("acgtacgtac", "ACGT"), ("A", "C"), ("TTTTACGAACGTTTTT", "ACGTACGT")]
    dev = torch.device("cuda")
    al = porechop_amd.Aligner(sorted({a for _, a in cases}))
    try:
        ads = sorted({a for _, a in cases})
        text = "".join(r for r, _ in cases)
        arr = np.frombuffer(text.encode(), dtype=np.uint8)
        pk, exc = pack_reads(arr)
        arena = al.unpack_device(torch.from_numpy(pk).to(dev), arr.size, torch.from_numpy(exc).to(dev) if exc.size else None)
        al.sync()
        host = arena.cpu().numpy()
        offs = np.cumsum([0] + [len(r) for r, _ in cases])[:-1]
        recs = al.align_host(host, offs, [len(r) for r, _ in cases], [ads.index(a) for _, a in cases])
        for (r, a), rec in zip(cases, recs):
            assert porechop_amd.format_result(rec) == oracle.adapter_alignment(r, a), (r, a)
    finally:
        al.close()


def test_pipeline_over_packed_upload_equals_byte_upload():
    import porechop_amd
    from porechop_amd.pipeline import AdapterSet, DeviceReads, Pipeline, ScanParams
    from tests.pairgen import synthetic_read
    rng = random.Random(5)
    y_top, y_bottom = "AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"
    reads = []
    for i in range(600):
        r = synthetic_read(rng, rng.choice([300, 900, 2500]), y_top if rng.random() < 0.9 else None,
                           y_bottom if rng.random() < 0.5 else None, (y_bottom + y_top) if i % 7 == 0 else None)
        r = list(r)
        for _ in range(rng.choice([0, 0, 1, 4, 40])):          # sprinkle letters the packing has to list as exceptions
            r[rng.randrange(len(r))] = rng.choice("NnX-RYacgtUu")
        reads.append("".join(r))
    text = "".join(reads).encode()
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
    dev = torch.device("cuda")
    sets = [AdapterSet("SQK-NSK007", ("SQK-NSK007_Y_Top", y_top), ("SQK-NSK007_Y_Bottom", y_bottom))]
    pl = Pipeline(sets, ScanParams(), device=dev)
    try:
        arr = np.frombuffer(text, dtype=np.uint8)
        as_bytes = DeviceReads(torch.from_numpy(np.concatenate([arr, np.full(64, ord("N"), np.uint8)])).to(dev),
                               torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev))
        pk, exc = pack_reads(arr)
        assert exc.size > 0
        as_packed = DeviceReads.from_packed(pl.aligner, torch.from_numpy(pk).to(dev), arr.size, torch.from_numpy(exc).to(dev),
                                            torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev))
        out = []
        for rd in (as_bytes, as_packed):
            bs, be = pl.phase_a(rd, torch.arange(rd.n, device=dev))
            st, et = pl.phase_b(rd, [0])
            hits = pl.phase_c(rd, st, et, [0])
            out.append((bs.cpu(), be.cpu(), st.cpu(), et.cpu(), hits.read.cpu(), hits.adapter.cpu(), hits.start.cpu(), hits.end.cpu()))
        for a, b in zip(*out):
            assert torch.equal(a, b)
        assert out[0][4].numel() > 20
    finally:
        pl.close()


def test_readset_to_device_packed(aligner, tmp_path):
    """ReadSet.to_device(packed=True): ingest -> pc_pack_reads -> upload -> pc_unpack_device gives the canonical bytes of the
    arena ReadSet.to_device() uploads as it is (same offsets and lengths)."""
    from porechop_amd.io import ReadSet
    rng = random.Random(9)
    p = tmp_path / "reads.fastq"
    with open(p, "w") as f:
        for i in range(300):
            L = rng.choice([1, 40, 333, 2000])
            seq = "".join(rng.choice("ACGTACGTACGTNacgtu-") for _ in range(L))
            f.write("@r%d\n%s\n+\n%s\n" % (i, seq, "I" * L))
    rs = ReadSet(str(p))
    try:
        plain = rs.to_device("cuda")
        packed = rs.to_device("cuda", packed=True, aligner=aligner)
        aligner.sync()
        assert torch.equal(plain.off, packed.off) and torch.equal(plain.length, packed.length)
        nb = int(rs.offsets[-1] + rs.lengths[-1])
        a = plain.arena[:nb].cpu().numpy()
        want = np.full(nb, ord("N"), dtype=np.uint8)
        for src, dst in ((b"Aa", "A"), (b"Cc", "C"), (b"Gg", "G"), (b"TtUu", "T")):
            want[np.isin(a, np.frombuffer(src, dtype=np.uint8))] = ord(dst)
        assert np.array_equal(packed.arena[:nb].cpu().numpy(), want)
        assert bool((packed.arena[nb:nb + 16] == ord("N")).all())
    finally:
        rs.close()


# ---- reads that STAY at 2 bits per base (task 4): the prefilter over the plane, windows unpacked on demand ----
def _packed_plane(arr, dev):
    pk, exc = pack_reads(arr)
    plane = torch.zeros(pk.size + 64, dtype=torch.uint8, device=dev)
    plane[:pk.size] = torch.from_numpy(pk).to(dev)
    return plane, (torch.from_numpy(exc).to(dev) if exc.size else None), exc


def test_unpack_windows_equals_slices_of_the_full_unpack(aligner):
    rng = np.random.default_rng(8)
    dev = torch.device("cuda")
    n = 200_000
    w = np.array([0.245] * 4 + [0.02 / (len(ALPHABET) - 4)] * (len(ALPHABET) - 4))
    arr = np.frombuffer(ALPHABET, dtype=np.uint8)[rng.choice(len(ALPHABET), size=n, p=w / w.sum())]
    plane, d_exc, exc = _packed_plane(arr, dev)
    full = unpack_reads_host(pack_reads(arr)[0], n, exc)
    # ascending, non-overlapping windows of every length and alignment, the last one ending at the last base
    starts, lens, pos = [], [], 0
    while pos < n - 5000:
        ln = int(rng.choice([0, 1, 3, 15, 16, 17, 150, 151, 1000, 4097]))
        starts.append(pos); lens.append(ln)
        pos += ln + int(rng.integers(0, 40))
    starts.append(n - 777); lens.append(777)
    so = torch.tensor(starts, dtype=torch.int64, device=dev)
    ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    stride = (ln.to(torch.int64) + 23) // 16 * 16
    do = torch.zeros(len(starts) + 1, dtype=torch.int64, device=dev)
    do[1:] = torch.cumsum(stride, 0)
    dst = torch.full((int(do[-1]) + 16,), 0xEE, dtype=torch.uint8, device=dev)
    aligner.unpack_windows(plane, d_exc, so, ln, dst, do, pad=ord("-"))
    aligner.sync()
    got, doh = dst.cpu().numpy(), do.cpu().numpy()
    for i, (s, l) in enumerate(zip(starts, lens)):
        assert np.array_equal(got[doh[i]:doh[i] + l], full[s:s + l]), i
        assert np.all(got[doh[i] + l:doh[i + 1]] == ord("-")), i
    assert np.all(got[doh[-1]:] == 0xEE)


def test_prefilter_over_the_plane_equals_the_byte_route(oracle):
    """pc_prefilter_packed vs pc_prefilter_device: the same mask on reads made of A/C/G/T; on reads that hold other letters
    a SUPERSET (non-bases are scanned as 'A'), and never a pair within the bound dropped (the plain DP of the oracle)."""
    import porechop_amd
    from tests.test_gpu_prefilter import make_cases
    rng = random.Random(5)
    adapters = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "".join(rng.choice("ACGT") for _ in range(24)),
                "".join(rng.choice("ACGT") for _ in range(33)), "".join(rng.choice("ACGT") for _ in range(30)),
                "".join(rng.choice("ACGT") for _ in range(38))]
    dev = torch.device("cuda")
    for seed, alphabet, lengths, hint in ((1, "ACGT", [0, 1, 5, 16, 17, 100, 150, 600, 2500], 0), (2, "ACGTN-", [40, 150, 1000, 9000], 0),
                                          (3, "ACGT", [100, 3000, 70000], 3000)):
        reads = [r.upper() for r in make_cases(seed, 1500 if max(lengths) < 20000 else 300, lengths, adapters, alphabet=alphabet)]
        text = "".join(reads).encode()
        arr = np.frombuffer(text, dtype=np.uint8)
        lens = np.array([len(r) for r in reads], dtype=np.int32)
        offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
        d_off, d_len = torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev)
        plane, d_exc, exc = _packed_plane(arr, dev)
        arena = torch.from_numpy(np.concatenate([arr, np.full(64, ord("N"), np.uint8)])).to(dev)
        al = porechop_amd.Aligner(adapters)
        try:
            al.set_length_hint(hint)
            # (80 %: parts shorter than six bases -- no seeds, so not a list for the packed route)
            assert al.prefilter_mask_packed(plane, d_off, d_len, int(lens.max()), list(range(len(adapters))),
                                            [al.max_edits(len(a), 80.0) for a in adapters]) is None
            for thr in (90.0, 95.0):
                ks = [al.max_edits(len(a), thr) for a in adapters]
                ids = list(range(len(adapters)))
                m_bytes = al.prefilter_mask(arena, d_off, d_len, int(lens.max()), ids, ks)
                m_plane = al.prefilter_mask_packed(plane, d_off, d_len, int(lens.max()), ids, ks)
                al.sync()
                assert m_plane is not None
                a, b = m_bytes.cpu().numpy(), m_plane.cpu().numpy()
                if alphabet == "ACGT":
                    assert np.array_equal(a, b), (seed, thr, int((a != b).sum()))
                else:
                    assert np.all((a & ~b) == 0), (seed, thr)                      # nothing the byte route keeps is dropped
                    assert int((b & ~a != 0).sum()) < 0.2 * len(reads)              # and not much is added
                bits = (b[:, 0][None, :] >> np.arange(len(adapters))[:, None]) & 1
                for j, (ad, k) in enumerate(zip(adapters, ks)):
                    d = oracle.min_edits_many(arr, offs, lens, ad) if len(ad) <= 32 else None
                    if d is not None:
                        assert not ((d <= k) & (lens > 0) & (bits[j] == 0)).any(), (seed, thr, ad)
        finally:
            al.close()
    # an adapter with a letter other than A/C/G/T, or one the seed stage cannot cover: the packed route says no
    al = porechop_amd.Aligner(["ACGTNNACGTTTGACCAGTNAC", "ACGTACGTAC"])
    try:
        z64, z32 = torch.zeros(1, dtype=torch.int64, device=dev), torch.full((1,), 100, dtype=torch.int32, device=dev)
        assert al.prefilter_mask_packed(torch.zeros(256, dtype=torch.uint8, device=dev), z64, z32, 100, [0], [2]) is None
        assert al.prefilter_mask_packed(torch.zeros(256, dtype=torch.uint8, device=dev), z64, z32, 100, [1], [1]) is None
    finally:
        al.close()


def _packed_vs_bytes(env_cap=None):
    import porechop_amd
    from porechop_amd.pipeline import DeviceReads, Pipeline, ScanParams
    from porechop_amd.panel import load_panel
    from tests.pairgen import synthetic_read
    rng = random.Random(9)
    y_top, y_bottom = "AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"
    reads = []
    for i in range(3000):
        r = synthetic_read(rng, rng.choice([40, 300, 900, 2500, 9000]), y_top if rng.random() < 0.9 else None,
                           y_bottom if rng.random() < 0.5 else None, (y_bottom + y_top) if i % 9 == 0 else None)
        r = list(r)
        for _ in range(rng.choice([0, 0, 1, 4, 40])):
            r[rng.randrange(len(r))] = rng.choice("NnX-RYacgtUu")
        reads.append("".join(r))
    text = "".join(reads).encode()
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
    dev = torch.device("cuda")
    pl = Pipeline(load_panel(), ScanParams(), device=dev)
    try:
        arr = np.frombuffer(text, dtype=np.uint8)
        as_bytes = DeviceReads(torch.from_numpy(np.concatenate([arr, np.full(64, ord("N"), np.uint8)])).to(dev),
                               torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev))
        pk, exc = pack_reads(arr)
        as_packed = DeviceReads.packed_only(pl.aligner, torch.from_numpy(pk).to(dev), arr.size, torch.from_numpy(exc).to(dev),
                                            torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev))
        assert as_packed.arena is None
        out = []
        for rd in (as_bytes, as_packed):
            bs, be = pl.phase_a(rd, torch.arange(min(rd.n, 1000), device=dev))
            matching = pl.matching_sets(bs, be)
            st, et = pl.phase_b(rd, matching)
            hits = pl.phase_c(rd, st, et, matching, prefilter=True)
            pl.aligner.sync()
            out.append((bs.cpu(), be.cpu(), st.cpu(), et.cpu(), hits.read.cpu(), hits.adapter.cpu(), hits.start.cpu(), hits.end.cpu()))
        return out, as_packed, dict(pl.stats), int(arr.size)
    finally:
        pl.close()


def test_pipeline_over_reads_that_stay_packed_equals_the_byte_route():
    out, as_packed, stats, nbases = _packed_vs_bytes()
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert out[0][4].numel() > 100
    assert as_packed.arena is None and "packed_route_refused" not in stats
    assert 0 < stats["bases_unpacked_after_prefilter"] < 0.5 * nbases           # only the survivors became bytes


def test_packed_route_with_an_overflowing_candidate_list_excludes_nothing():
    """PC_PF_SEED_CAP=64: the seed scan over the plane finds more candidates than its list holds; there is no exhaustive
    kernel over the plane, so the batch is handed to the DP whole (mask all ones) -- same results, only slower."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from tests.test_gpu_packed import _packed_vs_bytes\n"
            "out, rd, stats, n = _packed_vs_bytes()\n"
            "assert all(torch.equal(a, b) for a, b in zip(*out)) and out[0][4].numel() > 100\n"
            "print('CHILD_OK', stats.get('bases_unpacked_after_prefilter'), n)\n" % repo)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PC_PF_SEED_CAP="64"), capture_output=True, text=True, timeout=900, cwd=repo)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    unpacked, n = r.stdout.split("CHILD_OK")[1].split()[:2]
    assert int(unpacked) >= 0.8 * int(n)                # every (trimmed) read survived "the prefilter"


def test_packed_prefilter_with_a_bitmap_per_seed_length_equals_the_byte_route():
    """A barcode panel's worth of adapters keeps its seed lengths apart (the candidate rate decides, pc_api.cpp): the scan over
    the plane then probes one bitmap per length, like the byte route's -- same mask on reads made of A/C/G/T."""
    import porechop_amd
    from porechop_amd.panel import load_panel
    from tests.test_gpu_prefilter import make_cases
    seqs = []
    for s in load_panel():
        for side in (s.start, s.end):
            if side is not None and side[1] not in seqs and set(side[1]) <= set("ACGT"):
                seqs.append(side[1])
    seqs = seqs[:200]
    assert len(seqs) > 150
    reads = [r.upper() for r in make_cases(7, 1200, [150, 1000, 6000], seqs[:60], alphabet="ACGT")]
    arr = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
    dev = torch.device("cuda")
    d_off, d_len = torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev)
    plane, d_exc, exc = _packed_plane(arr, dev)
    arena = torch.from_numpy(np.concatenate([arr, np.full(64, ord("N"), np.uint8)])).to(dev)
    al = porechop_amd.Aligner(seqs)
    try:
        ks = [al.max_edits(len(a), 90.0) for a in seqs]
        ids = list(range(len(seqs)))
        m_bytes = al.prefilter_mask(arena, d_off, d_len, int(lens.max()), ids, ks)
        m_plane = al.prefilter_mask_packed(plane, d_off, d_len, int(lens.max()), ids, ks)
        al.sync()
        if m_plane is not None:                      # (None: a sequence of the panel the seed stage cannot cover at 90 %)
            assert torch.equal(m_bytes, m_plane)
            assert int((m_plane != 0).any(dim=1).sum()) > 300
    finally:
        al.close()
