"""Reads at 2 bits per base for the trip over PCIe (pc_pack_reads / pc_unpack_device, include/porechop_amd.h): the host half.
The packing must be alignment-neutral: SeqAn's Dna5 conversion (alphabet_residue_tabs.h:113-140: ACGT and U in either
case, everything else N) of the unpacked byte equals that of the original byte -- checked here against the oracle on the
ORIGINAL strings, incl. 'N', '-', lower case, 'U' and IUPAC letters."""
import random

import numpy as np

from porechop_amd.io import pack_reads, unpack_reads_host

ALPHABET = b"ACGTacgtUuNn-RYKMSWXZ*"


def canonical(arr):
    out = np.full(arr.shape, ord("N"), dtype=np.uint8)
    for src, dst in ((b"Aa", "A"), (b"Cc", "C"), (b"Gg", "G"), (b"TtUu", "T")):
        out[np.isin(arr, np.frombuffer(src, dtype=np.uint8))] = ord(dst)
    return out


def test_pack_then_unpack_gives_the_canonical_bytes_and_sorted_exceptions():
    rng = np.random.default_rng(7)
    for n in list(range(0, 70)) + [127, 128, 129, 4095, 4096, 4097, 262144 * 5 + 3, 3_000_001]:
        for weights in (None, [30] * 4 + [1] * (len(ALPHABET) - 4)):
            p = None if weights is None else np.array(weights, dtype=np.float64) / sum(weights)
            arr = np.frombuffer(ALPHABET, dtype=np.uint8)[rng.choice(len(ALPHABET), size=n, p=p)]
            pk, exc = pack_reads(arr)
            assert pk.size == (n + 15) // 16 * 4
            want = canonical(arr)
            assert np.array_equal(exc, np.nonzero(want == ord("N"))[0])
            assert np.array_equal(unpack_reads_host(pk, n, exc), want)


def test_prefix_of_a_buffer_and_preallocated_output():
    arr = np.frombuffer(b"ACGTNACGTTTGA-CC" * 5, dtype=np.uint8)
    out = np.full(64, 0xEE, dtype=np.uint8)
    pk, exc = pack_reads(arr, nbases=37, out=out)
    assert pk.size == 12 and pk.base is out or np.shares_memory(pk, out)
    assert np.all(out[12:] == 0xEE)
    assert np.array_equal(unpack_reads_host(pk, 37, exc), canonical(arr[:37]))


def test_alignments_of_unpacked_reads_equal_those_of_the_original_reads(oracle):
    rng = random.Random(11)
    schemes = [(3, -6, -5, -2), (2, -3, -5, -2), (5, -4, -8, -6)]
    for k in range(400):
        n = rng.choice([1, 5, 28, 60, 150, 151, 400])
        read = "".join(rng.choice("ACGT" * 6 + "acgtUuNn-RY") for _ in range(n))
        m = rng.randint(4, 40)
        start = rng.randint(0, max(0, n - m))
        adapter = "".join(rng.choice("ACGT" * 8 + "N") for _ in range(m)) if rng.random() < 0.3 else \
            canonical(np.frombuffer(read[start:start + m].encode(), dtype=np.uint8)).tobytes().decode() or "ACGT"
        arr = np.frombuffer(read.encode(), dtype=np.uint8)
        pk, exc = pack_reads(arr)
        back = unpack_reads_host(pk, n, exc).tobytes().decode()
        sc = schemes[k % len(schemes)]
        assert oracle.adapter_alignment(back, adapter, sc) == oracle.adapter_alignment(read, adapter, sc), (read, adapter)
