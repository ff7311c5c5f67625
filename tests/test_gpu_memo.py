"""The drop-in boundary with the prefetch memo, on the GPU: pc_prefetch() aligns a batch, later
per-call adapterAlignment() calls with the same arguments are memo hits and return the reference's
strings; the GPU-backed porechop_amd.dropin backend agrees with the oracle."""
import ctypes
import random

import numpy as np
import pytest

from tests.pairgen import random_case

pytestmark = pytest.mark.gpu


def test_prefetch_then_per_call_hits(oracle):
    import porechop_amd
    lib = porechop_amd.load_library()
    lib.pc_memo_clear()
    rng = random.Random(31)
    ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "ACGTACGTAGGCATTAGC"]
    reads = [random_case(rng, n=rng.choice([150, 150, 3000]), m=28)[0] for _ in range(300)]
    arena = "".join(reads).encode()
    offs = np.cumsum([0] + [len(r) for r in reads[:-1]]).astype(np.int64)
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    aidx = np.array([i % 3 for i in range(len(reads))], dtype=np.int32)
    arr = (ctypes.c_char_p * 3)(*[a.encode() for a in ads])
    rc = lib.pc_prefetch(arena, len(arena), offs.ctypes.data, lens.ctypes.data, arr, aidx.ctypes.data, len(reads), 3, -6, -5, -2)
    assert rc == 0
    h, m, e = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    lib.pc_memo_stats(ctypes.byref(h), ctypes.byref(m), ctypes.byref(e))
    assert e.value == len(set(zip(reads, aidx.tolist())))
    for r, i in zip(reads, aidx.tolist()):
        assert porechop_amd.adapter_alignment(r, ads[i], [3, -6, -5, -2]) == oracle.adapter_alignment(r, ads[i])
    lib.pc_memo_stats(ctypes.byref(h), ctypes.byref(m), ctypes.byref(e))
    assert h.value == len(reads) and m.value == 0
    # an unprefetched pair is a miss served by a single-pair launch -- still exact
    assert porechop_amd.adapter_alignment("TTTTACGTTTTT", "ACGT", [3, -6, -5, -2]) == "4,7,0,3,12,100.000000,100.000000"
    lib.pc_memo_stats(ctypes.byref(h), ctypes.byref(m), ctypes.byref(e))
    assert m.value == 1
    lib.pc_memo_clear()


def test_gpu_backend_of_dropin_matches_oracle(oracle):
    from porechop_amd.dropin import GpuBackend
    rng = random.Random(8)
    pairs = [random_case(rng) for _ in range(2000)]
    got = GpuBackend().align(pairs, (3, -6, -5, -2))
    for (rd, ad), g in zip(pairs, got):
        want = oracle.adapter_alignment(rd, ad)
        assert g == want or (g.split(",")[0] == "-1" and want.split(",")[0] == "-1")
