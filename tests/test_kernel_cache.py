"""The on-disk cache of specialised score kernels (csrc/pc_jit.cpp) and the ahead-of-time build of the static panel's
kernels (porechop_amd/aot.py).  hiprtc compiles for gfx950 without a GPU, so the build half runs anywhere; the load
half needs the device."""
import ctypes
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_precompile_writes_a_kernel_once(tmp_path):
    import porechop_amd
    lib = porechop_amd.load_library()
    d = str(tmp_path / "kc").encode()
    dd = d.decode()
    a, b = b"ACGTTGCAAGGCTTAACGTAGCATCGA", b"TTGACCATGCAAGTCAGT"
    assert lib.pc_jit_precompile(a, b, 3, -6, -5, -2, d) == 0
    files = os.listdir(dd)
    assert len(files) == 1 and files[0].endswith(".pcjk") and os.path.getsize(os.path.join(dd, files[0])) > 4096
    assert lib.pc_jit_precompile(a, b, 3, -6, -5, -2, d) == 1            # already there
    assert lib.pc_jit_precompile(a, b"", 3, -6, -5, -2, d) == 0           # the same adapter alone is another kernel
    assert lib.pc_jit_precompile(a, b, 2, -3, -5, -2, d) == 0            # so is another scheme
    assert len(os.listdir(dd)) == 3
    assert lib.pc_jit_precompile(a, b, 3, -6, -5, -5, d) < 0             # linear-gap schemes have no specialised kernel
    assert lib.pc_jit_precompile(b"", b"", 3, -6, -5, -2, d) < 0
    # a truncated file is ignored (recompiled), never loaded
    victim = os.path.join(dd, files[0])
    with open(victim, "r+b") as f:
        f.truncate(1000)
    assert lib.pc_jit_precompile(a, b, 3, -6, -5, -2, d) == 0


def test_panel_kernels_are_listed_canonically():
    from porechop_amd import aot
    from porechop_amd.panel import load_panel
    panel = load_panel()
    pairs = aot.panel_kernel_pairs(panel)
    assert ("AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT") in pairs       # SQK-NSK007: longer sequence first
    assert all(b is None or len(a) >= len(b) for a, b in pairs)
    assert len(pairs) >= 119 and len(set(pairs)) == len(pairs)
    # what Pipeline._scan_jobs pairs for one set is what aot lists for it
    from porechop_amd.pipeline import AdapterSet
    s = AdapterSet("x", ("s", "ACGTACGTAC"), ("e", "ACGTACGTACGGTT"))
    assert aot.canonical_pair(s.start[1], s.end[1]) == ("ACGTACGTACGGTT", "ACGTACGTAC")


def test_build_ships_the_panel_kernels():
    """__graft_entry__.build() leaves one kernel per panel set in porechop_amd/kernel_cache/ (skipped where the build
    has not run)."""
    from porechop_amd import aot
    if not os.path.isdir(aot.CACHE_DIR):
        pytest.skip("kernel cache not built")
    assert len([f for f in os.listdir(aot.CACHE_DIR) if f.endswith(".pcjk")]) >= 100


_CHILD = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
import torch
import porechop_amd
from porechop_amd.batch import MODE_SCORE
from porechop_amd.synth import make_reads
reads = make_reads(4096, 4000, seed=9, start_frac=0.5, end_frac=0.5)
al = porechop_amd.Aligner(["ACGTTGCAAGGCTTAACGTAGCATCGA", "TTGACCATGCAAGTCAGT"])
import numpy as np
out = torch.empty((2 * reads.n, 8), dtype=torch.int32, device="cuda")
for _ in range(2):
    al.scan_device(reads.arena, reads.off, reads.length, np.array([0], dtype=np.int32), np.array([0, reads.n], dtype=np.int64), 4000, out,
                   MODE_SCORE, job_adapter_b=np.array([1], dtype=np.int32))
    al.sync()
c, d = ctypes.c_int64(), ctypes.c_int64()
al.lib.pc_jit_stats(ctypes.byref(c), ctypes.byref(d))
print("STATS", c.value, d.value, int(out[:, 4].to(torch.int64).sum()))
"""


@pytest.mark.gpu
def test_second_process_loads_instead_of_compiling(tmp_path):
    """A pair that is not in the in-tree cache: the first process compiles it (hiprtc) and leaves it in the user cache
    directory, the second process loads it from there -- zero compiles -- and computes the same scores."""
    env = dict(os.environ, PC_JIT_CACHE_DIR=str(tmp_path / "user_cache"), PC_JIT_MIN_CELLS="0")
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", _CHILD % REPO], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([int(x) for x in [l for l in r.stdout.splitlines() if l.startswith("STATS")][-1].split()[1:]])
    assert outs[0][0] == 1 and outs[0][1] == 0, outs          # compiled once, nothing on disk yet
    assert outs[1][0] == 0 and outs[1][1] == 1, outs          # second process: from the cache on disk
    assert outs[0][2] == outs[1][2]
