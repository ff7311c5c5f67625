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


def test_per_call_symbol_from_sixteen_threads(oracle):
    """Porechop calls adapter_alignment from a multiprocessing.dummy pool of --threads Python threads
    (porechop.py:309-322, 496-509, 579-591); ctypes drops the GIL for the call.  Sixteen threads, a mix of
    memo hits (half of the pairs are prefetched) and misses (single-pair launches): every string is the
    reference's, nothing deadlocks, and hits do not wait behind the launches of the misses."""
    import time
    from multiprocessing.dummy import Pool as ThreadPool
    import porechop_amd
    lib = porechop_amd.load_library()
    lib.pc_memo_clear()
    rng = random.Random(77)
    ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"]
    reads = [random_case(rng, n=150, m=28)[0] for _ in range(1200)]
    half = reads[:600]
    arena = "".join(half).encode()
    offs = np.cumsum([0] + [len(r) for r in half[:-1]]).astype(np.int64)
    lens = np.array([len(r) for r in half], dtype=np.int32)
    aidx = np.zeros(len(half), dtype=np.int32)
    arr = (ctypes.c_char_p * 2)(*[a.encode() for a in ads])
    assert lib.pc_prefetch(arena, len(arena), offs.ctypes.data, lens.ctypes.data, arr, aidx.ctypes.data, len(half), 3, -6, -5, -2) == 0
    jobs = [(r, ads[0]) for r in reads] + [(r, ads[1]) for r in reads[:300]]
    rng.shuffle(jobs)

    def one(job):
        return porechop_amd.adapter_alignment(job[0], job[1], [3, -6, -5, -2])

    t0 = time.time()
    with ThreadPool(16) as pool:
        got = pool.map(one, jobs, chunksize=8)
    assert time.time() - t0 < 120
    for (rd, ad), g in zip(jobs, got):
        assert g == oracle.adapter_alignment(rd, ad), (rd, ad)
    h, m, e = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    lib.pc_memo_stats(ctypes.byref(h), ctypes.byref(m), ctypes.byref(e))
    assert h.value >= 600 and h.value + m.value == len(jobs)
    lib.pc_memo_clear()


def test_memo_is_bounded_and_verified():
    """PC_MEMO_MAX_ENTRIES bounds the memo (an epoch clear when the bound is reached); entries carry
    their lengths and an independent digest, checked on every hit."""
    import os
    import subprocess
    import sys
    code = r'''
import ctypes, sys
sys.path.insert(0, ".")
import porechop_amd
lib = porechop_amd.load_library()
for i in range(40):
    rd = "ACGT" * 5 + format(i, "06b").replace("0", "A").replace("1", "C")
    assert porechop_amd.adapter_alignment(rd, "ACGTACGT", [3, -6, -5, -2]).split(",")[4] != ""
h, m, e = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
lib.pc_memo_stats(ctypes.byref(h), ctypes.byref(m), ctypes.byref(e))
print("ENTRIES", e.value, m.value)
'''
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PC_MEMO_MAX_ENTRIES="16"),
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    line = [l for l in res.stdout.splitlines() if l.startswith("ENTRIES")]
    assert line, res.stdout[-1000:] + res.stderr[-2000:]
    entries, misses = int(line[0].split()[1]), int(line[0].split()[2])
    assert misses == 40 and 0 < entries <= 16
