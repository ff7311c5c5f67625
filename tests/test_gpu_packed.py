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
